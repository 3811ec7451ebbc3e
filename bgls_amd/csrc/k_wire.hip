// BLS12-381 compressed wire formats over a batch (wire.hpp, second half: the ebfull/pairing layout the reference names as its
// target, curves/bls12_381.go:54-62,115-123; UnmarshalG1 / UnmarshalG2 on 48 / 96 bytes, :242-264).  One point per thread:
// a decoder is one (G1) or two (G2) square-root exponentiations, a Legendre symbol and an inversion, then Check() -- the
// subgroup test the reference applies to every unmarshalled point (G1: [r]P = infinity; G2: the endomorphism criterion).
// Its own translation unit: k_points.hip is the slowest unit of the build already.
#include "dev_common.hpp"
#include "wire.hpp"
#include "launch.hpp"
#include "../../include/bgls_hip.h"

using namespace bgls;

template <int GROUP>
__global__ void __launch_bounds__(64) k_decompress_bls(const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
  typedef BLS381 C;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int CB = GROUP == BGLS_G1 ? 48 : 96, UB = 2 * CB;
  bool good;
  if constexpr (GROUP == BGLS_G1) {
    Aff<F1<C>> p;
    good = g1_decompress_zc<C>(p, in + i * CB);
    good = good && g1_in_subgroup<C>(p);                   // UnmarshalG1: !result.Check() -> nil, false
    if (good) g1_to_bytes<C>(out + i * UB, p);
  } else {
    Aff<F2<C>> p;
    good = g2_decompress_zc<C>(p, in + i * CB);
    good = good && g2_in_subgroup<C>(p);
    if (good) g2_to_bytes<C>(out + i * UB, p);
  }
  if (!good)
    for (int k = 0; k < UB; ++k) out[i * UB + k] = 0;
  ok[i] = good ? 1 : 0;
}

template <int GROUP>
__global__ void __launch_bounds__(64) k_compress_bls(const uint8_t* in, size_t n, uint8_t* out, uint32_t* flags) {
  typedef BLS381 C;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int CB = GROUP == BGLS_G1 ? 48 : 96, UB = 2 * CB;
  if constexpr (GROUP == BGLS_G1) {
    Aff<F1<C>> p;
    bool good = g1_from_bytes<C>(p, in + i * UB) && aff_on_curve<F1<C>>(p);
    if (!good) atomicOr(flags, FLAG_ENC);
    g1_compress_zc<C>(out + i * CB, p);
  } else {
    Aff<F2<C>> p;
    bool good = g2_from_bytes<C>(p, in + i * UB) && aff_on_curve<F2<C>>(p);
    if (!good) atomicOr(flags, FLAG_ENC);
    g2_compress_zc<C>(out + i * CB, p);
  }
}

namespace bgls {
namespace kl {

void compress_bls(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint32_t* flags) {
  if (group == BGLS_G1) k_compress_bls<BGLS_G1><<<nblk(n, 64), 64, 0, st>>>(in, n, out, flags);
  else k_compress_bls<BGLS_G2><<<nblk(n, 64), 64, 0, st>>>(in, n, out, flags);
}
void decompress_bls(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
  if (group == BGLS_G1) k_decompress_bls<BGLS_G1><<<nblk(n, 64), 64, 0, st>>>(in, n, out, ok);
  else k_decompress_bls<BGLS_G2><<<nblk(n, 64), 64, 0, st>>>(in, n, out, ok);
}

}  // namespace kl
}  // namespace bgls
