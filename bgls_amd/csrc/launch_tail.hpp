// Launchers of k_sumtree.hip (declared apart from launch.hpp, which every unit includes: only engine.hip needs these).
#pragma once
#include "dev_common.hpp"

namespace bgls {
namespace kl {

// The tree above a key sum's main pass in ONE launch (k_sumtree.hip): cnt Jacobian partial sums (G2) -> their sum, as affine
// wire bytes (d_bytes != nullptr) and / or as one Jacobian record (d_jac != nullptr).  store: cnt Jacobian records of scratch,
// tickets: cnt words, zero on entry and zero again on exit.
template <class C> void sum_tree(hipStream_t st, const void* in, size_t cnt, void* store, uint32_t* tickets, uint8_t* d_bytes, void* d_jac);

}  // namespace kl
}  // namespace bgls
