// Latency-oriented final exponentiation: ONE Fp12 value, 36 lanes.
//
// The final exponentiation happens once per verification (bgls/bgls.go:115-118 compares the
// product of ALL pairings with 1; the reference pays it n+1 times inside Pair, curves/curve.go:
// 132-134), so it is a serial chain of ~300 Fp12 multiplications that no batch dimension hides.
// Here each Fp12 product is spread over 36 lanes: lane 6j+t computes term t of output
// coefficient j (one Karatsuba Fp2 product in double width), the six partial sums of a
// coefficient meet in LDS, lanes t=0 / t=1 reduce the real / imaginary part.  Values live in LDS
// "slots" (6 coefficients x {plain, xi-multiplied}), so the exponentiation is a small register
// machine over slot numbers.  Exponent exactly (p^12-1)/r, same chains as pairing.hpp.
#pragma once
#include "coop.hpp"

namespace bgls {

template <class C>
struct FE {
  static constexpr int L = C::L, W = 2 * C::L, S2 = 2 * C::L;
  static constexpr int SLOT = 12 * S2;
  static constexpr int NSLOT = 16;
  static constexpr int SCR = NSLOT * SLOT;            // 36 lanes x 3 wide products
  static constexpr int LDS_DW = SCR + 36 * 3 * W;
  static constexpr int LDS_BYTES = LDS_DW * 4;
  static constexpr int LAZY_K = Coop<C>::LAZY_K;
  __device__ static __forceinline__ int coef(int slot, int k, int xi) { return slot * SLOT + (2 * k + xi) * S2; }
};

// slot numbers
enum { FE_F = 0, FE_T = 1, FE_U = 2, FE_A = 3, FE_B = 4, FE_C = 5, FE_Y0 = 6, FE_Y1, FE_Y2, FE_Y3, FE_Y4, FE_Y5, FE_Y6, FE_T0, FE_T1, FE_X };

template <int W>
__device__ __forceinline__ void lds_load_w(u32 (&r)[W], int off) {
  extern __shared__ u32 lds[];
  const uint4* p = reinterpret_cast<const uint4*>(lds + off);
#pragma unroll
  for (int k = 0; k < W / 4; ++k) {
    uint4 v = p[k];
    r[4 * k] = v.x; r[4 * k + 1] = v.y; r[4 * k + 2] = v.z; r[4 * k + 3] = v.w;
  }
}
template <int W>
__device__ __forceinline__ void lds_store_w(int off, const u32 (&a)[W]) {
  extern __shared__ u32 lds[];
  uint4* p = reinterpret_cast<uint4*>(lds + off);
#pragma unroll
  for (int k = 0; k < W / 4; ++k) p[k] = make_uint4(a[4 * k], a[4 * k + 1], a[4 * k + 2], a[4 * k + 3]);
}

// write coefficient k of `slot` (plain) and its xi multiple
template <class C>
__device__ __forceinline__ void fe_put(int slot, int k, const Fp2<C>& v) {
  lds_store_f2<C>(FE<C>::coef(slot, k, 0), v);
  lds_store_f2<C>(FE<C>::coef(slot, k, 1), f2_mulxi<C>(v));
}

// dst <- a * b   (all 64 lanes must call; lanes >= 36 idle)
template <class C>
__device__ __noinline__ void fe_mul(int dst, int a, int b) {
  typedef FE<C> E;
  constexpr int W = E::W;
  extern __shared__ u32 lds[];
  const int lane = threadIdx.x & 63;
  const int j = lane / 6, t = lane % 6;
  if (lane < 36) {
    int k = j - t;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    Fp2<C> x = lds_load_f2<C>(E::coef(a, t, 0));
    Fp2<C> y = lds_load_f2<C>(E::coef(b, k, wrap));
    u32 tmp[W];
    const int o = E::SCR + lane * 3 * W;
    mul_wide<C>(tmp, x.c0.v, y.c0.v);
    lds_store_w<W>(o, tmp);
    mul_wide<C>(tmp, x.c1.v, y.c1.v);
    lds_store_w<W>(o + W, tmp);
    Fp<C> sa = fp_add_nr<C>(x.c0, x.c1), sb = fp_add_nr<C>(y.c0, y.c1);
    mul_wide<C>(tmp, sa.v, sb.v);
    lds_store_w<W>(o + 2 * W, tmp);
  }
  wave_sync();
  if (lane < 36 && t < 2) {
    // t == 0: real part  sum v0 - sum v1 + 6 p^2 ;  t == 1: imaginary part  sum s - sum v0 - sum v1
    u32 pos[W], neg[W], tmp[W];
#pragma unroll
    for (int q = 0; q < W; ++q) pos[q] = neg[q] = 0;
    const int base = E::SCR + (6 * j) * 3 * W;
#pragma unroll 1
    for (int u = 0; u < 6; ++u) {
      const int o = base + u * 3 * W;
      lds_load_w<W>(tmp, o + (t == 0 ? 0 : 2 * W));
      w_add<W>(pos, pos, tmp);
      lds_load_w<W>(tmp, o + W);
      w_add<W>(neg, neg, tmp);
      if (t == 1) {
        lds_load_w<W>(tmp, o);
        w_add<W>(neg, neg, tmp);
      }
    }
    if (t == 0) w_add<W>(pos, pos, C::P2W6);
    w_sub<W>(pos, pos, neg);
    Fp<C> r = redc_k<C, E::LAZY_K>(pos);
    u32* p = lds + E::coef(dst, j, 0) + (t == 0 ? 0 : C::L);
#pragma unroll
    for (int q = 0; q < C::L; ++q) p[q] = r.v[q];
  }
  wave_sync();
  if (lane < 36 && t == 2) lds_store_f2<C>(E::coef(dst, j, 1), f2_mulxi<C>(lds_load_f2<C>(E::coef(dst, j, 0))));
  wave_sync();
}

template <class C>
__device__ __forceinline__ void fe_conj(int dst, int a) {          // a^(p^6): w -> -w
  const int lane = threadIdx.x & 63;
  if (lane < 6) {
    Fp2<C> v = lds_load_f2<C>(FE<C>::coef(a, lane, 0));
    fe_put<C>(dst, lane, (lane & 1) ? f2_neg<C>(v) : v);
  }
  wave_sync();
}
template <class C>
__device__ __noinline__ void fe_frob(int dst, int a, int k) {       // a^(p^k), k = 1..3
  const int lane = threadIdx.x & 63;
  if (lane < 6) {
    Fp2<C> v = lds_load_f2<C>(FE<C>::coef(a, lane, 0));
    if (k & 1) v = f2_conj<C>(v);
    fe_put<C>(dst, lane, f2_mul<C>(v, gamma_const<C>(k, lane)));
  }
  wave_sync();
}
// dst <- a^-1 through norms:  a^-1 = conj(a) * (N^(p^2) N^(p^4)) / Norm_{Fp6/Fp2}(N),  N = a conj(a) in Fp6.
// Four cooperative products, two Frobenius maps and ONE Fp inversion (on lane 0).  Uses FE_X, FE_Y5, FE_Y6 as scratch.
template <class C>
__device__ __noinline__ void fe_inv(int dst, int a) {
  typedef FE<C> E;
  const int lane = threadIdx.x & 63;
  const int N = FE_X, A = FE_Y5, B = FE_Y6;
  fe_conj<C>(B, a);
  fe_mul<C>(N, a, B);                    // N = a * conj(a): odd w-coefficients vanish
  fe_frob<C>(A, N, 2);                   // N^(p^2)
  fe_frob<C>(B, A, 2);                   // N^(p^4)
  fe_mul<C>(A, A, B);                    // M = N^(p^2) N^(p^4)
  fe_mul<C>(B, N, A);                    // Norm(N) in Fp2: only coefficient 0
  if (lane < 6) {
    Fp2<C> dinv = f2_inv<C>(lds_load_f2<C>(E::coef(B, 0, 0)));
    fe_put<C>(A, lane, f2_mul<C>(lds_load_f2<C>(E::coef(A, lane, 0)), dinv));     // N^-1
  }
  wave_sync();
  fe_conj<C>(B, a);
  fe_mul<C>(dst, B, A);
}
// dst <- a^e, public exponent with its top bit at nbits-1; dst != a
template <class C>
__device__ __noinline__ void fe_pow(int dst, int a, const u32* e, int nbits) {
  const int lane = threadIdx.x & 63;
  if (lane < 6) fe_put<C>(dst, lane, lds_load_f2<C>(FE<C>::coef(a, lane, 0)));
  wave_sync();
  for (int i = nbits - 2; i >= 0; --i) {
    fe_mul<C>(dst, dst, dst);
    if ((e[i >> 5] >> (i & 31)) & 1u) fe_mul<C>(dst, dst, a);
  }
}

// slot FE_F <- FE_F ^ ((p^12 - 1) / r)
template <class C>
__device__ __noinline__ void fe_final_exp() {
  // easy part: (p^6 - 1)(p^2 + 1)
  fe_conj<C>(FE_T, FE_F);
  fe_inv<C>(FE_U, FE_F);
  fe_mul<C>(FE_T, FE_T, FE_U);
  fe_frob<C>(FE_U, FE_T, 2);
  fe_mul<C>(FE_F, FE_U, FE_T);
  if constexpr (C::CURVE_ID == 0) {
    // hard part, y0..y6 vectorial chain (pairing.hpp final_exp)
    fe_pow<C>(FE_A, FE_F, C::U_ABS, C::U_BITS);     // ft1
    fe_pow<C>(FE_B, FE_A, C::U_ABS, C::U_BITS);     // ft2
    fe_pow<C>(FE_C, FE_B, C::U_ABS, C::U_BITS);     // ft3
    fe_frob<C>(FE_Y0, FE_F, 1);
    fe_frob<C>(FE_T, FE_F, 2);
    fe_mul<C>(FE_Y0, FE_Y0, FE_T);
    fe_frob<C>(FE_T, FE_F, 3);
    fe_mul<C>(FE_Y0, FE_Y0, FE_T);
    fe_conj<C>(FE_Y1, FE_F);
    fe_frob<C>(FE_Y2, FE_B, 2);
    fe_frob<C>(FE_Y3, FE_A, 1);
    fe_conj<C>(FE_Y3, FE_Y3);
    fe_frob<C>(FE_T, FE_B, 1);
    fe_mul<C>(FE_Y4, FE_A, FE_T);
    fe_conj<C>(FE_Y4, FE_Y4);
    fe_conj<C>(FE_Y5, FE_B);
    fe_frob<C>(FE_T, FE_C, 1);
    fe_mul<C>(FE_Y6, FE_C, FE_T);
    fe_conj<C>(FE_Y6, FE_Y6);
    fe_mul<C>(FE_T0, FE_Y6, FE_Y6);
    fe_mul<C>(FE_T0, FE_T0, FE_Y4);
    fe_mul<C>(FE_T0, FE_T0, FE_Y5);
    fe_mul<C>(FE_T1, FE_Y3, FE_Y5);
    fe_mul<C>(FE_T1, FE_T1, FE_T0);
    fe_mul<C>(FE_T0, FE_T0, FE_Y2);
    fe_mul<C>(FE_T1, FE_T1, FE_T1);
    fe_mul<C>(FE_T1, FE_T1, FE_T0);
    fe_mul<C>(FE_T1, FE_T1, FE_T1);
    fe_mul<C>(FE_T0, FE_T1, FE_Y1);
    fe_mul<C>(FE_T1, FE_T1, FE_Y0);
    fe_mul<C>(FE_T0, FE_T0, FE_T0);
    fe_mul<C>(FE_F, FE_T1, FE_T0);
  } else {
    // (p^4 - p^2 + 1)/r = c (x + p)(x^2 + p^2 - 1) + 1,  c = (x-1)^2/3,  x < 0
    fe_pow<C>(FE_A, FE_F, C::COFACTOR, C::COFACTOR_BITS);   // a = f^c
    fe_pow<C>(FE_T, FE_A, C::U_ABS, C::U_BITS);
    fe_conj<C>(FE_T, FE_T);                                 // a^x
    fe_frob<C>(FE_U, FE_A, 1);
    fe_mul<C>(FE_B, FE_T, FE_U);                            // b = a^x a^p
    fe_pow<C>(FE_T, FE_B, C::U_ABS, C::U_BITS);
    fe_conj<C>(FE_T, FE_T);                                 // b^x
    fe_pow<C>(FE_U, FE_T, C::U_ABS, C::U_BITS);
    fe_conj<C>(FE_U, FE_U);                                 // b^(x^2)
    fe_frob<C>(FE_T, FE_B, 2);
    fe_mul<C>(FE_U, FE_U, FE_T);
    fe_conj<C>(FE_T, FE_B);
    fe_mul<C>(FE_U, FE_U, FE_T);                            // d
    fe_mul<C>(FE_F, FE_U, FE_F);
  }
}

}  // namespace bgls
