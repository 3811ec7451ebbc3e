// Scalar multiplications on G1 at the seam, one point per lane on the carry-free limbs (rx_jac1.hpp):
//   k_scale_aff_g1x   Sign over a batch: sigs[i] = sk_i * H(m_i) (bgls/bgls.go:46-56; the hash points arrive as resident
//                     Montgomery affine points), or sk_i * g1 when pts == nullptr
//   k_scale_g1x       ScalePoints / Point.Mul on G1 (curves/curve.go:190-214; wire bytes in, sign bytes: 1 = negate first,
//                     2 = nil factor -> Copy())
// Same windows, same group law and therefore the same points as k_scale_aff / k_scale of k_points.hip (which keep serving
// G2, alt-bn128's G1 -- no gain there: ten 28-bit limbs against eight 32-bit ones -- and, with BGLS_LEGACY bit 32, BLS12-381's):
// only the field arithmetic under the chain changed.  BLS12-381 at 2^18 points: Sign 55 -> 43 ms, ScalePoints 36 -> 29 ms.
#include "dev_common.hpp"
#include "rx_jac1.hpp"
#include "points_inl.hpp"
#include "launch_tail.hpp"
#include "../../include/bgls_hip.h"

using namespace bgls;

template <class C>
__device__ __forceinline__ void scalar_words(const uint8_t* s, int sbytes, u32 (&k)[8], int& nbits) {
  const int nw = sbytes / 4;
  int top = -1;
  for (int j = 0; j < 8; ++j) {
    k[j] = 0;
    if (j < nw) {
      const uint8_t* q = s + 4 * (nw - 1 - j);
      k[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
  }
  for (int j = 7; j >= 0 && top < 0; --j)
    if (k[j]) top = j * 32 + (31 - __clz(k[j]));
  nbits = top + 1;
}

template <class C>
__global__ void __launch_bounds__(64) k_scale_aff_g1x(const Aff<F1<C>>* pts, const uint8_t* scalars, size_t n, uint8_t* out) {
  typedef F1<C> F;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p;
  if (pts) p = pts[i];
  else p = Aff<F>{fp_load<C>(C::G1X), fp_load<C>(C::G1Y), false};
  u32 k[8];
  int nbits;
  scalar_words<C>(scalars + i * 32, 32, k, nbits);
  const Jac1<C> r = jac1_mul_w4<C>(aff1_from_mont<C>(p), k, nbits);
  aff_to_bytes<F>(out + i * 2 * C::FP_BYTES, jac_to_aff<F>(jac1_to_mont<C>(r)));
}

template <class C>
__global__ void __launch_bounds__(64) k_scale_g1x(const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n, uint8_t* out,
                                                  uint32_t* flags, int sbytes) {
  typedef F1<C> F;
  constexpr int PT = 2 * C::FP_BYTES;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p;
  bool ok = aff_from_bytes<F>(p, pts + i * PT);
  ok = ok && aff_on_curve<F>(p);
  if (!ok) atomicOr(flags, FLAG_ENC);
  const uint8_t sg = signs ? signs[i] : 0;
  if (sg == 2) {                                              // nil factor: Copy()
    aff_to_bytes<F>(out + i * PT, p);
    return;
  }
  u32 k[8];
  int nbits;
  scalar_words<C>(scalars + i * (size_t)sbytes, sbytes, k, nbits);
  if (sg == 1) p = aff_neg<F>(p);
  const Jac1<C> r = jac1_mul_w4<C>(aff1_from_mont<C>(p), k, nbits);
  aff_to_bytes<F>(out + i * PT, jac_to_aff<F>(jac1_to_mont<C>(r)));
}

namespace bgls {
namespace kl {

template <class C>
void scale_aff_g1x(hipStream_t st, const Aff<F1<C>>* g1_pts, const uint8_t* scalars, size_t n, uint8_t* out) {
  k_scale_aff_g1x<C><<<nblk(n, 64), 64, 0, st>>>(g1_pts, scalars, n, out);
}
template <class C>
void scale_g1x(hipStream_t st, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n, uint8_t* out, uint32_t* flags, int sbytes) {
  k_scale_g1x<C><<<nblk(n, 64), 64, 0, st>>>(pts, scalars, signs, n, out, flags, sbytes);
}
template void scale_aff_g1x<BN254>(hipStream_t, const Aff<F1<BN254>>*, const uint8_t*, size_t, uint8_t*);
template void scale_aff_g1x<BLS381>(hipStream_t, const Aff<F1<BLS381>>*, const uint8_t*, size_t, uint8_t*);
template void scale_g1x<BN254>(hipStream_t, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint32_t*, int);
template void scale_g1x<BLS381>(hipStream_t, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint32_t*, int);

}  // namespace kl
}  // namespace bgls
