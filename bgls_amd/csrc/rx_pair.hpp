// The G2 point steps of the Miller loop on a LANE PAIR, in the signed carry-free form of rx.hpp.
//
// One pairing's running point T = (X, Y, Z) over Fp2 is spread over two neighbouring lanes: the even lane holds the real
// parts, the odd lane the imaginary parts.  An Fp2 product a b = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) i then costs each lane
// two limb-product piles and ONE reduction (the partner's halves arrive through DPP quad permutes, 2 NL v_mov_dpp per
// product against 3 NL^2 multiplier instructions), a squaring one pile and one reduction.  Compared with a whole point
// step on one lane this halves the live state (3 NL registers of T per lane instead of 6 NL, one pile instead of three) --
// the point-step wave of the 32-bit kernels spilled 150-650 registers -- and halves the latency of a step.
//
// Formulas and line coefficients are exactly those of pairing.hpp (dbl_step_emit / add_step_emit), so the Miller values
// are bit-identical to the other kernels'.  Replaces, with the rest of the kernel, bn256.Pair / bls12 GT.Pair behind
// curves/altbn128.go:130-145 and curves/bls12_381.go:228-240.
#pragma once
#include "rx.hpp"

#if defined(__HIPCC__)
#define RX_DEV __device__ __forceinline__
#else
#define RX_DEV inline __attribute__((always_inline))
// host emulation (unit tests only): two threads in lock-step, see tests/harness/host_harness.cpp
int rx_host_pair_swap(int v);
#endif

namespace bgls {

// the partner lane's value (lanes 2k <-> 2k+1)
RX_DEV i32 pair_swap1(i32 v) {
#if defined(__HIPCC__)
  return __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
#else
  return rx_host_pair_swap(v);
#endif
}
template <class C, int LA>
RX_DEV Sx<C, LA> pair_swap(const Sx<C, LA>& a) {
  Sx<C, LA> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = pair_swap1(a.v[i]);
  return r;
}

// own half of a * b
template <class C, int LA, int LB>
RX_DEV Sx<C, SX_T> pair_mul(const Sx<C, LA>& a, const Sx<C, LB>& b, bool odd) {
  const Sx<C, LA> pa = pair_swap<C>(a);
  const Sx<C, LB> pb = pair_swap<C>(b);
  const Sx<C, LB> p = sx_select<C>(odd, pb, b);
  const Sx<C, LB> q = sx_select<C>(odd, b, sx_neg<C>(pb));
  return sx_mont<C>(a, p, pa, q);
}
// own half of a * k for a constant k = (k_re, k_im) known to every lane
template <class C, int LA>
RX_DEV Sx<C, SX_T> pair_mul_const(const Sx<C, LA>& a, const u32* k_re, const u32* k_im, bool odd) {
  const Sx<C, LA> pa = pair_swap<C>(a);
  const Sx<C, SX_T> kr = sx_const<C>(k_re), ki = sx_const<C>(k_im);
  // even: a0 kr - a1 ki      odd: a1 kr + a0 ki
  const Sx<C, SX_T> q = sx_select<C>(odd, ki, sx_neg<C>(ki));
  return sx_mont<C>(a, kr, pa, q);
}
// own half of a^2:  (a0 + a1)(a0 - a1)  |  2 a0 a1
template <class C, int LA>
RX_DEV Sx<C, SX_T> pair_sqr(const Sx<C, LA>& a, bool odd) {
  const Sx<C, LA> pa = pair_swap<C>(a);
  const Sx<C, 2 * LA> u = sx_add<C>(a, sx_select<C>(odd, a, pa));
  const Sx<C, 2 * LA> p = sx_select<C>(odd, sx_as<2 * LA, C>(pa), sx_sub<C>(a, pa));
  return sx_mont<C>(u, p);
}
// own half of a * s, s in Fp (the same value on both lanes)
template <class C, int LA, int LS>
RX_DEV Sx<C, SX_T> pair_muls(const Sx<C, LA>& a, const Sx<C, LS>& s) {
  return sx_mont<C>(a, s);
}
// own half of a b - c d, one reduction
template <class C, int LA, int LB, int LC, int LD>
RX_DEV Sx<C, SX_T> pair_mulsub(const Sx<C, LA>& a, const Sx<C, LB>& b, const Sx<C, LC>& c, const Sx<C, LD>& d, bool odd) {
  const Sx<C, LA> pa = pair_swap<C>(a);
  const Sx<C, LB> pb = pair_swap<C>(b);
  const Sx<C, LC> pc = pair_swap<C>(c);
  const Sx<C, LD> pd = pair_swap<C>(d);
  // minus (c d): the same two products with the left factors negated
  return sx_mont<C>(a, sx_select<C>(odd, pb, b), pa, sx_select<C>(odd, b, sx_neg<C>(pb)), sx_neg<C>(c), sx_select<C>(odd, pd, d), sx_neg<C>(pc),
                    sx_select<C>(odd, d, sx_neg<C>(pd)));
}
// own half of g^2 - e f, one reduction
template <class C, int LG, int LE, int LF>
RX_DEV Sx<C, SX_T> pair_sqrsub(const Sx<C, LG>& g, const Sx<C, LE>& e, const Sx<C, LF>& f, bool odd) {
  const Sx<C, LG> pg = pair_swap<C>(g);
  const Sx<C, 2 * LG> u = sx_add<C>(g, sx_select<C>(odd, g, pg));
  const Sx<C, 2 * LG> p = sx_select<C>(odd, sx_as<2 * LG, C>(pg), sx_sub<C>(g, pg));
  const Sx<C, LE> pe = pair_swap<C>(e);
  const Sx<C, LF> pf = pair_swap<C>(f);
  return sx_mont<C>(u, p, sx_neg<C>(e), sx_select<C>(odd, pf, f), sx_neg<C>(pe), sx_select<C>(odd, f, sx_neg<C>(pf)));
}
// own half of xi * a  (alt-bn128: 9 + i, BLS12-381: 1 + i):  XI_RE a0 - a1  |  XI_RE a1 + a0
template <class C, int LA>
RX_DEV Sx<C, (C::XI_RE + 1) * LA> pair_mulxi(const Sx<C, LA>& a, bool odd) {
  const Sx<C, LA> pa = pair_swap<C>(a);
  return sx_add<C>(sx_mulc<C::XI_RE, C>(a), sx_select<C>(odd, pa, sx_neg<C>(pa)));
}
// 3 b' z of the doubling step: a product by the constant on both curves.  (BLS12-381's 3 b' = 12 (1 + i) could be formed with
// additions, as pairing.hpp does, but the VALUE would grow to 25 p and the P-free line coefficient E - B must reach the
// consumer below 32 p after the fat multiple of p that makes it non-negative; a reduction is what brings it back.)
template <class C, int LA>
RX_DEV Sx<C, SX_T> pair_mul_3b(const Sx<C, LA>& z, bool odd) {
  return pair_mul_const<C>(z, C::RX_B2X3_RE, C::RX_B2X3_IM, odd);
}

template <class C>
struct PointX {
  Sx<C, SX_T> X, Y, Z;     // own halves
};

// Doubling step.  emit(slot, value): slot 2 = the P-free coefficient (any bound), slots 0 / 1 = the coefficients already
// scaled by yP / xP (tight).  env.nyP() = -yP and env.xP() are the hash point's coordinates in this form (same on both
// lanes); they are fetched where they are used so that they do not occupy registers through the step.
template <class C, class Env, class Emit>
RX_DEV void dbl_step_x(PointX<C>& R, Env&& env, bool odd, Emit&& emit) {
  const Sx<C, SX_T> B = pair_sqr<C>(R.Y, odd);
  const Sx<C, SX_T> Cc = pair_sqr<C>(R.Z, odd);
  const auto H = sx_sub<C>(pair_sqr<C>(sx_normf<C>(sx_add<C>(R.Y, R.Z)), odd), sx_add<C>(B, Cc));      // 2 Y Z
  const Sx<C, SX_T> E = pair_mul_3b<C>(Cc, odd);
  emit(2, sx_sub<C>(E, B));                                  // I = E - B
  emit(0, pair_muls<C>(H, env.nyP()));                       // (-H) yP
  R.Z = pair_mul<C>(B, H, odd);
  emit(1, pair_muls<C>(sx_mulc<3, C>(pair_sqr<C>(R.X, odd)), env.xP()));   // 3 X^2 xP
  const auto A = sx_half<C>(pair_mul<C>(R.X, R.Y, odd));
  const Sx<C, SX_F> Fv = sx_normf<C>(sx_mulc<3, C>(E));
  R.X = pair_mul<C>(A, sx_sub<C>(B, Fv), odd);
  const Sx<C, SX_F> G = sx_normf<C>(sx_half<C>(sx_add<C>(B, Fv)));
  R.Y = pair_sqrsub<C>(G, E, Fv, odd);                       // G^2 - 3 E^2
}

// Mixed addition step with the affine point (env.xq(), env.yq()) (own halves, tight).
template <class C, class Env, class Emit>
RX_DEV void add_step_x(PointX<C>& R, Env&& env, bool odd, Emit&& emit) {
  const Sx<C, SX_F> th = sx_normf<C>(sx_sub<C>(R.Y, pair_mul<C>(env.yq(), R.Z, odd)));
  const Sx<C, SX_F> la = sx_normf<C>(sx_sub<C>(R.X, pair_mul<C>(env.xq(), R.Z, odd)));
  emit(2, pair_mulsub<C>(th, env.xq(), la, env.yq(), odd));  // th xq - la yq
  emit(0, pair_muls<C>(sx_neg<C>(la), env.nyP()));           // la yP
  emit(1, pair_muls<C>(sx_neg<C>(th), env.xP()));            // (-th) xP
  const Sx<C, SX_T> D = pair_sqr<C>(la, odd);
  const Sx<C, SX_T> G = pair_mul<C>(R.X, D, odd);
  const Sx<C, SX_T> E = pair_mul<C>(la, D, odd);
  const Sx<C, SX_T> Fv = pair_mul<C>(R.Z, pair_sqr<C>(th, odd), odd);
  const Sx<C, SX_F> Hh = sx_normf<C>(sx_sub<C>(sx_add<C>(E, Fv), sx_mulc<2, C>(G)));
  R.X = pair_mul<C>(la, Hh, odd);
  R.Z = pair_mul<C>(R.Z, E, odd);
  R.Y = pair_mulsub<C>(th, sx_sub<C>(G, Hh), E, R.Y, odd);
}

}  // namespace bgls
