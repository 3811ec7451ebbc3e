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
  static constexpr int SCR_DW = 36 * 3 * W;
  static constexpr int LDS_DW = SCR + SCR_DW;
  static constexpr int LDS_BYTES = LDS_DW * 4;
  static constexpr int LDS_BYTES2 = (SCR + 2 * SCR_DW) * 4;      // two waves, each with its own scratch (and its own slots)
  __device__ static __forceinline__ int scr() { return SCR + (int)(threadIdx.x >> 6) * SCR_DW; }
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

// xi * c for a value whose real part sits on an even lane and imaginary part on the next lane:
// returns this lane's half of xi*(re + i im) (re-lane: XI_RE*re - im, im-lane: XI_RE*im + re).
template <class C>
__device__ __forceinline__ Fp<C> mulxi_half(const Fp<C>& mine, bool is_im) {
  Fp<C> other;
#pragma unroll
  for (int q = 0; q < C::L; ++q) other.v[q] = __shfl_xor(mine.v[q], 1);
  Fp<C> m = mine;
  if constexpr (C::XI_RE == 9) {
    m = fp_dbl<C>(fp_dbl<C>(fp_dbl<C>(mine)));
    m = fp_add<C>(m, mine);
  }
  return fp_select<C>(is_im, fp_add<C>(m, other), fp_sub<C>(m, other));
}

// dst <- a * b   (all 64 lanes must call; lanes >= 36 idle).  A = a (plain coefficients only), B = b (needs
// its xi variants).  lane 6j+t: term t of coefficient j.  The six partial triples (a0 b0, a1 b1,
// (a0+a1)(b0+b1)) of a coefficient are summed as a two-level tree through LDS: lanes t<3 absorb lane t+3,
// then lane t=0 finishes the real part and lane t=1 the imaginary part -- the SAME instruction stream on
// both (operand selects, no divergent branches: a lone wave pays every dependent carry chain in full).
// want_xi = false skips the xi variants of the result (enough when it is only squared or used as A next).
template <class C>
__device__ __noinline__ void fe_mul(int dst, int a, int b, bool want_xi = true) {
  typedef FE<C> E;
  constexpr int W = E::W;
  extern __shared__ u32 lds[];
  const int lane = threadIdx.x & 63;
  const int j = lane / 6, t = lane % 6;
  const bool act = lane < 36;
  const int o = E::scr() + lane * 3 * W;
  u32 v0[W], v1[W], s[W], tmp[W];
  if (act) {
    int k = j - t;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    Fp2<C> x = lds_load_f2<C>(E::coef(a, t, 0));
    Fp2<C> y = lds_load_f2<C>(E::coef(b, k, wrap));
    mul_wide<C>(v0, x.c0.v, y.c0.v);
    mul_wide<C>(v1, x.c1.v, y.c1.v);
    Fp<C> sa = fp_add_nr<C>(x.c0, x.c1), sb = fp_add_nr<C>(y.c0, y.c1);
    mul_wide<C>(s, sa.v, sb.v);
    if (t >= 3) {
      lds_store_w<W>(o, v0);
      lds_store_w<W>(o + W, v1);
      lds_store_w<W>(o + 2 * W, s);
    }
  }
  wave_sync();
  if (act && t < 3) {
    const int po = o + 3 * 3 * W;          // partner lane t+3
    lds_load_w<W>(tmp, po);
    w_add<W>(v0, v0, tmp);
    lds_load_w<W>(tmp, po + W);
    w_add<W>(v1, v1, tmp);
    lds_load_w<W>(tmp, po + 2 * W);
    w_add<W>(s, s, tmp);
    lds_store_w<W>(o, v0);
    lds_store_w<W>(o + W, v1);
    lds_store_w<W>(o + 2 * W, s);
  }
  wave_sync();
  if (act && t < 2) {
    const int base = E::scr() + (6 * j) * 3 * W;
#pragma unroll
    for (int r = 1; r <= 2; ++r) {          // the two other partial sums (own ones are still in registers)
      int u = t + r;
      u -= u >= 3 ? 3 : 0;
      const int q = base + u * 3 * W;
      lds_load_w<W>(tmp, q);
      w_add<W>(v0, v0, tmp);
      lds_load_w<W>(tmp, q + W);
      w_add<W>(v1, v1, tmp);
      lds_load_w<W>(tmp, q + 2 * W);
      w_add<W>(s, s, tmp);
    }
    // uniform form R = X + K - V:  t == 0: (sum v0) + 6 p^2 - (sum v1)      t == 1: (sum s) + 0 - (sum v0 + sum v1)
    const u32 m0 = t == 0 ? 0xFFFFFFFFu : 0u;
    u32 X[W], V[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
      X[q] = (v0[q] & m0) | (s[q] & ~m0);
      tmp[q] = v0[q] & ~m0;
    }
    w_add<W>(V, v1, tmp);
#pragma unroll
    for (int q = 0; q < W; ++q) tmp[q] = C::P2W6[q] & m0;
    w_add<W>(X, X, tmp);
    w_sub<W>(X, X, V);
    Fp<C> r = redc_k<C, E::LAZY_K>(X);
    u32* p = lds + E::coef(dst, j, 0) + (t == 0 ? 0 : C::L);
#pragma unroll
    for (int q = 0; q < C::L; ++q) p[q] = r.v[q];
    if (want_xi) {
      Fp<C> z = mulxi_half<C>(r, t == 1);
      u32* px = lds + E::coef(dst, j, 1) + (t == 0 ? 0 : C::L);
#pragma unroll
      for (int q = 0; q < C::L; ++q) px[q] = z.v[q];
    }
  }
  wave_sync();
}

// recompute the xi variants of a slot (after a chain of want_xi = false operations)
template <class C>
__device__ __forceinline__ void fe_fix_xi(int slot) {
  const int lane = threadIdx.x & 63;
  if (lane < 6) lds_store_f2<C>(FE<C>::coef(slot, lane, 1), f2_mulxi<C>(lds_load_f2<C>(FE<C>::coef(slot, lane, 0))));
  wave_sync();
}

// dst <- a^2 for a in the cyclotomic subgroup (Granger-Scott, as f12_cyclo_sqr in tower.hpp).
// Nine Fp2 squarings -- the pairs (e_q, e_{q+3}), q = 0..2, give A^2, B^2, (A+B)^2 -- on 18 lanes (one Fp
// product + one reduction each: real part (x0+x1)(x0-x1), imaginary part 2 x0 x1), then 12 lanes
// assemble the six coefficients, real and imaginary halves side by side.  Uniform instruction stream.
template <class C>
__device__ __noinline__ void fe_cyclo_sqr(int dst, int a, bool want_xi = true) {
  typedef FE<C> E;
  constexpr int W = E::W, L = C::L;
  extern __shared__ u32 lds[];
  const int lane = threadIdx.x & 63;
  if (lane < 18) {
    const int sq = lane >> 1, part = lane & 1, q = sq / 3, which = sq % 3;
    Fp2<C> x = lds_load_f2<C>(E::coef(a, q, 0)), y = lds_load_f2<C>(E::coef(a, q + 3, 0));
    Fp2<C> in = which == 0 ? x : which == 1 ? y : f2_add<C>(x, y);
    Fp<C> X = fp_select<C>(part == 1, in.c0, fp_add_nr<C>(in.c0, in.c1));
    Fp<C> Y = fp_select<C>(part == 1, in.c1, fp_sub<C>(in.c0, in.c1));
    u32 tw[W], t2[W];
    mul_wide<C>(tw, X.v, Y.v);
    const u32 md = part ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int k = 0; k < W; ++k) t2[k] = tw[k] & md;
    w_add<W>(tw, tw, t2);                                   // imaginary part: 2 x0 x1
    Fp<C> r = redc<C>(tw);
    u32* p = lds + E::scr() + sq * E::S2 + part * L;
#pragma unroll
    for (int k = 0; k < L; ++k) p[k] = r.v[k];
  }
  wave_sync();
  if (lane < 12) {
    const int k = lane >> 1;                                // output coefficient
    const bool im = lane & 1;
    // e0,e3 <- pair 0; e2,e5 <- pair 1; e1,e4 <- pair 2
    const int q = (k == 0 || k == 3) ? 0 : (k == 2 || k == 5) ? 1 : 2;
    const int off = im ? L : 0;
    Fp<C> t0 = fp_load<C>(lds + E::scr() + (3 * q) * E::S2 + off);
    Fp<C> t1 = fp_load<C>(lds + E::scr() + (3 * q + 1) * E::S2 + off);
    Fp<C> t2 = fp_load<C>(lds + E::scr() + (3 * q + 2) * E::S2 + off);
    Fp<C> z = fp_load<C>(lds + E::coef(a, k, 0) + off);
    Fp<C> c1 = fp_sub<C>(fp_sub<C>(t2, t0), t1);            // c1 = (A+B)^2 - A^2 - B^2
    const bool even = (k & 1) == 0;
    Fp<C> m = mulxi_half<C>(fp_select<C>(even, t1, c1), im); // xi*t1 (even k) or xi*c1 (k == 1)
    Fp<C> tt = even ? fp_add<C>(m, t0) : (k == 1 ? m : c1);  // c0 = xi t1 + t0 | xi c1 | c1
    Fp<C> r = even ? fp_sub<C>(tt, z) : fp_add<C>(tt, z);
    r = fp_add<C>(fp_dbl<C>(r), tt);                         // 3 tt -/+ 2 z
    wave_sync();
    u32* p = lds + E::coef(dst, k, 0) + off;
#pragma unroll
    for (int w = 0; w < L; ++w) p[w] = r.v[w];
    if (want_xi) {
      Fp<C> zx = mulxi_half<C>(r, im);
      u32* px = lds + E::coef(dst, k, 1) + off;
#pragma unroll
      for (int w = 0; w < L; ++w) px[w] = zx.v[w];
    }
  } else {
    wave_sync();
  }
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
    fe_cyclo_sqr<C>(dst, dst, false);
    if ((e[i >> 5] >> (i & 31)) & 1u) fe_mul<C>(dst, dst, a, false);
  }
  fe_fix_xi<C>(dst);
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
    fe_cyclo_sqr<C>(FE_T0, FE_Y6);
    fe_mul<C>(FE_T0, FE_T0, FE_Y4);
    fe_mul<C>(FE_T0, FE_T0, FE_Y5);
    fe_mul<C>(FE_T1, FE_Y3, FE_Y5);
    fe_mul<C>(FE_T1, FE_T1, FE_T0);
    fe_mul<C>(FE_T0, FE_T0, FE_Y2);
    fe_cyclo_sqr<C>(FE_T1, FE_T1);
    fe_mul<C>(FE_T1, FE_T1, FE_T0);
    fe_cyclo_sqr<C>(FE_T1, FE_T1);
    fe_mul<C>(FE_T0, FE_T1, FE_Y1);
    fe_mul<C>(FE_T1, FE_T1, FE_Y0);
    fe_cyclo_sqr<C>(FE_T0, FE_T0);
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
