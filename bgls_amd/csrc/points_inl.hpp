// Inline point helpers shared by the point kernels (k_points.hip) and the bucket-method weighted sums (k_msm.hip):
// wire-format parsing per field type, the in-place Jacobian formulas (no calls, no private stack in a loop) and a
// lane shuffle of a whole Jacobian point.
#pragma once
#include "dev_common.hpp"
#include "wire.hpp"

namespace bgls {

template <class F>
__device__ __forceinline__ bool aff_from_bytes(Aff<F>& p, const uint8_t* b);
template <>
__device__ __forceinline__ bool aff_from_bytes<F1<BN254>>(Aff<F1<BN254>>& p, const uint8_t* b) { return g1_from_bytes<BN254>(p, b); }
template <>
__device__ __forceinline__ bool aff_from_bytes<F1<BLS381>>(Aff<F1<BLS381>>& p, const uint8_t* b) { return g1_from_bytes<BLS381>(p, b); }
template <>
__device__ __forceinline__ bool aff_from_bytes<F2<BN254>>(Aff<F2<BN254>>& p, const uint8_t* b) { return g2_from_bytes<BN254>(p, b); }
template <>
__device__ __forceinline__ bool aff_from_bytes<F2<BLS381>>(Aff<F2<BLS381>>& p, const uint8_t* b) { return g2_from_bytes<BLS381>(p, b); }

template <class F>
__device__ __forceinline__ void aff_to_bytes(uint8_t* b, const Aff<F>& p);
template <>
__device__ __forceinline__ void aff_to_bytes<F1<BN254>>(uint8_t* b, const Aff<F1<BN254>>& p) { g1_to_bytes<BN254>(b, p); }
template <>
__device__ __forceinline__ void aff_to_bytes<F1<BLS381>>(uint8_t* b, const Aff<F1<BLS381>>& p) { g1_to_bytes<BLS381>(b, p); }
template <>
__device__ __forceinline__ void aff_to_bytes<F2<BN254>>(uint8_t* b, const Aff<F2<BN254>>& p) { g2_to_bytes<BN254>(b, p); }
template <>
__device__ __forceinline__ void aff_to_bytes<F2<BLS381>>(uint8_t* b, const Aff<F2<BLS381>>& p) { g2_to_bytes<BLS381>(b, p); }

// ---- AggregatePoints main pass (curves/curve.go:73-121; BASELINE config 4: 2^20 G2 keys) -------------------------------
// Thread t walks points t, t + T, t + 2T, ... (neighbouring lanes read neighbouring points) and keeps a Jacobian
// running sum; the mixed addition is expanded in place (no calls, no private stack in the loop) and the grid is sized to
// two waves per SIMD, which is what hides the multiplier's dependent-carry latency.  The per-thread sums are then folded
// 64 at a time with lane shuffles (k_sum_wave).
// Exceptional inputs keep the result exact: infinity on either side is a select, equal x (P = +-Q, e.g. a key listed
// twice) takes the out-of-line general addition.
template <class F>
__device__ __forceinline__ Jac<F> jac_dbl_inl(const Jac<F>& p) {       // dbl-2009-l, products expanded in place
  typedef typename F::T T;
  const T A = F::sqr_inl(p.X);
  const T B = F::sqr_inl(p.Y);
  const T Cc = F::sqr_inl(B);
  const T D = F::dbl(F::sub(F::sub(F::sqr_inl(F::add(p.X, B)), A), Cc));
  const T E = F::add(F::dbl(A), A);
  Jac<F> r;
  r.X = F::sub(F::sqr_inl(E), F::dbl(D));
  r.Y = F::sub(F::mul_inl(E, F::sub(D, r.X)), F::dbl(F::dbl(F::dbl(Cc))));
  r.Z = F::dbl(F::mul_inl(p.Y, p.Z));
  return r;
}

template <class F>
__device__ __forceinline__ Jac<F> jac_madd_inl(const Jac<F>& p, const Aff<F>& q) {
  typedef typename F::T T;
  const T Z1Z1 = F::sqr_inl(p.Z);
  const T U2 = F::mul_inl(q.x, Z1Z1);
  const T S2 = F::mul_inl(F::mul_inl(q.y, p.Z), Z1Z1);
  const T H = F::sub(U2, p.X);
  if (__builtin_expect(F::is_zero(H) && !jac_is_inf<F>(p) && !q.inf, 0)) {     // same x: P = Q (double) or P = -Q (infinity)
    if (F::is_zero(F::sub(S2, p.Y))) return jac_dbl_inl<F>(p);
    return jac_inf<F>();
  }
  const T rr = F::dbl(F::sub(S2, p.Y));
  const T HH = F::sqr_inl(H);
  const T I = F::dbl(F::dbl(HH));
  const T J = F::mul_inl(H, I);
  const T V = F::mul_inl(p.X, I);
  Jac<F> r;
  r.X = F::sub(F::sub(F::sqr_inl(rr), J), F::dbl(V));
  r.Y = F::sub(F::mul_inl(rr, F::sub(V, r.X)), F::dbl(F::mul_inl(p.Y, J)));
  r.Z = F::sub(F::sub(F::sqr_inl(F::add(p.Z, H)), Z1Z1), HH);
  const bool pinf = jac_is_inf<F>(p);
  r.X = F::select(q.inf, p.X, F::select(pinf, q.x, r.X));
  r.Y = F::select(q.inf, p.Y, F::select(pinf, q.y, r.Y));
  r.Z = F::select(q.inf, p.Z, F::select(pinf, F::one(), r.Z));
  return r;
}

// general Jacobian addition (add-2007-bl), products expanded in place; exceptional cases exact
template <class F>
__device__ __forceinline__ Jac<F> jac_add_inl(const Jac<F>& p, const Jac<F>& q) {
  typedef typename F::T T;
  const T Z1Z1 = F::sqr_inl(p.Z);
  const T Z2Z2 = F::sqr_inl(q.Z);
  const T U1 = F::mul_inl(p.X, Z2Z2);
  const T U2 = F::mul_inl(q.X, Z1Z1);
  const T S1 = F::mul_inl(F::mul_inl(p.Y, q.Z), Z2Z2);
  const T S2 = F::mul_inl(F::mul_inl(q.Y, p.Z), Z1Z1);
  const T H = F::sub(U2, U1);
  const bool pinf = jac_is_inf<F>(p), qinf = jac_is_inf<F>(q);
  if (__builtin_expect(F::is_zero(H) && !pinf && !qinf, 0)) {
    if (F::is_zero(F::sub(S2, S1))) return jac_dbl_inl<F>(p);
    return jac_inf<F>();
  }
  const T rr = F::dbl(F::sub(S2, S1));
  const T I = F::sqr_inl(F::dbl(H));
  const T J = F::mul_inl(H, I);
  const T V = F::mul_inl(U1, I);
  Jac<F> r;
  r.X = F::sub(F::sub(F::sqr_inl(rr), J), F::dbl(V));
  r.Y = F::sub(F::mul_inl(rr, F::sub(V, r.X)), F::dbl(F::mul_inl(S1, J)));
  r.Z = F::mul_inl(F::sub(F::sub(F::sqr_inl(F::add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
  r.X = F::select(qinf, p.X, F::select(pinf, q.X, r.X));
  r.Y = F::select(qinf, p.Y, F::select(pinf, q.Y, r.Y));
  r.Z = F::select(qinf, p.Z, F::select(pinf, q.Z, r.Z));
  return r;
}

template <class F>
__device__ __forceinline__ Jac<F> jac_shfl_down(const Jac<F>& a, int off) {
  Jac<F> r;
  constexpr int ND = sizeof(Jac<F>) / 4;
  const u32* src = reinterpret_cast<const u32*>(&a);
  u32* dst = reinterpret_cast<u32*>(&r);
#pragma unroll
  for (int k = 0; k < ND; ++k) dst[k] = __shfl_down(src[k], off);
  return r;
}

// big-endian field element through 16-byte loads (the wire formats are multiples of 16 bytes; p must be 16-byte aligned)
template <class C>
__device__ __forceinline__ Fp<C> fp_from_be16(const uint8_t* p) {
  Fp<C> r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) {
    const uint4 v = q[k];                      // bytes 16k .. 16k+15 = limbs L-1-4k .. L-4-4k, most significant first
    r.v[C::L - 1 - 4 * k] = __builtin_bswap32(v.x);
    r.v[C::L - 2 - 4 * k] = __builtin_bswap32(v.y);
    r.v[C::L - 3 - 4 * k] = __builtin_bswap32(v.z);
    r.v[C::L - 4 - 4 * k] = __builtin_bswap32(v.w);
  }
  return r;
}
template <class F, class C>
__device__ __forceinline__ bool aff_from_bytes16(Aff<F>& p, const uint8_t* b) {
  constexpr int N = C::FP_BYTES;
  if constexpr (F::NFP == 1) {
    const Fp<C> x = fp_from_be16<C>(b), y = fp_from_be16<C>(b + N);
    const bool ok = !fp_geq_p<C>(x) && !fp_geq_p<C>(y);
    p.inf = fp_is_zero<C>(x) && fp_is_zero<C>(y);
    p.x = fp_mul_inl<C>(x, fp_load<C>(C::R2));
    p.y = fp_mul_inl<C>(y, fp_load<C>(C::R2));
    return ok;
  } else {
    const Fp<C> xi = fp_from_be16<C>(b), xr = fp_from_be16<C>(b + N), yi = fp_from_be16<C>(b + 2 * N), yr = fp_from_be16<C>(b + 3 * N);
    const bool ok = !fp_geq_p<C>(xi) && !fp_geq_p<C>(xr) && !fp_geq_p<C>(yi) && !fp_geq_p<C>(yr);
    p.inf = fp_is_zero<C>(xi) && fp_is_zero<C>(xr) && fp_is_zero<C>(yi) && fp_is_zero<C>(yr);
    const Fp<C> r2 = fp_load<C>(C::R2);
    p.x = {fp_mul_inl<C>(xr, r2), fp_mul_inl<C>(xi, r2)};
    p.y = {fp_mul_inl<C>(yr, r2), fp_mul_inl<C>(yi, r2)};
    return ok;
  }
}
template <class F>
__device__ __forceinline__ bool aff_on_curve_inl(const Aff<F>& a) {
  if (a.inf) return true;
  return F::eq(F::sqr_inl(a.y), F::add(F::mul_inl(F::sqr_inl(a.x), a.x), F::curve_b()));
}

}  // namespace bgls
