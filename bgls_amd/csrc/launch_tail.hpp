// Launchers of k_sumtree.hip and k_g1x.hip (declared apart from launch.hpp, which every unit includes: only engine.hip needs these).
#pragma once
#include "dev_common.hpp"

namespace bgls {
namespace kl {

// The tree above a key sum's main pass in ONE launch (k_sumtree.hip): cnt Jacobian partial sums (G2) -> their sum, as affine
// wire bytes (d_bytes != nullptr) and / or as one Jacobian record (d_jac != nullptr).  store: sum_tree_store_bytes(cnt) of scratch,
// tickets: cnt words, zero on entry and zero again on exit.
template <class C> void sum_tree(hipStream_t st, const void* in, size_t cnt, void* store, uint32_t* tickets, uint8_t* d_bytes, void* d_jac);
template <class C> size_t sum_tree_store_bytes(size_t cnt);

// ---- k_finalx.hip: the final exponentiation with its three result words {verdict, this stage's flags, *flags_in} written side by side
// (res3: 12 bytes), so that ONE copy brings them to the host and no flag word has to be cleared beforehand
template <class C> void finalx_res(hipStream_t st, const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out, uint32_t* res3, const uint32_t* flags_in);

// ---- k_g1x.hip: scalar multiplications on G1, one point per lane on the carry-free limbs (rx_jac1.hpp)
template <class C> void scale_aff_g1x(hipStream_t st, const Aff<F1<C>>* g1_pts, const uint8_t* scalars, size_t n, uint8_t* out);
template <class C> void scale_g1x(hipStream_t st, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n, uint8_t* out, uint32_t* flags, int sbytes);

// ---- k_millerlatx.hip: the narrow passes of the reduce stage on the two-wave 36-lane product (finalx.hpp)
template <class C> void reduce_fx(hipStream_t st, const Fp2<C>* in, size_t count, int R, Fp2<C>* out);
// the verification epilogue's two chains as separate launches (1: the signature pair, 2: rest^h and the product of the two)
template <class C> void cofactor_epiloguex_part(hipStream_t st, int part, const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines, Fp2<C>* tmp, uint8_t* out);

}  // namespace kl
}  // namespace bgls
