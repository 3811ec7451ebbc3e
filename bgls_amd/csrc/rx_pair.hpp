// The G2 point steps of the Miller loop on a LANE PAIR, in the signed carry-free form of rx.hpp.
//
// One pairing's running point T = (X, Y, Z) over Fp2 is spread over two neighbouring lanes: the even lane holds the real
// parts, the odd lane the imaginary parts.  An Fp2 product a b = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) i then costs each lane
// two limb products and ONE reduction, interleaved row by row over NL + 1 columns (the partner's halves arrive through DPP
// quad permutes, 3 NL v_mov_dpp per product against 3 NL^2 multiplier instructions), a squaring one product and one reduction.  Compared with a whole point
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

// values from the lanes of the own pair (lanes 2k, 2k+1): the partner's, the even lane's, the odd lane's
RX_DEV i32 pair_swap1(i32 v) {
#if defined(__HIPCC__)
  return __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
#else
  return rx_host_pair_swap(v);
#endif
}
RX_DEV i32 pair_even1(i32 v, bool odd) {
#if defined(__HIPCC__)
  (void)odd;
  return __builtin_amdgcn_update_dpp(0, v, 0xA0 /* quad_perm [0,0,2,2] */, 0xF, 0xF, true);
#else
  const i32 o = rx_host_pair_swap(v);
  return odd ? o : v;
#endif
}
RX_DEV i32 pair_odd1(i32 v, bool odd) {
#if defined(__HIPCC__)
  (void)odd;
  return __builtin_amdgcn_update_dpp(0, v, 0xF5 /* quad_perm [1,1,3,3] */, 0xF, 0xF, true);
#else
  const i32 o = rx_host_pair_swap(v);
  return odd ? v : o;
#endif
}
// the partner's half, negated on the even lane: the "- a1 b1" of the real part is then a plain product
template <class C, int LA>
RX_DEV Sx<C, LA> pair_swap_neg_even(const Sx<C, LA>& a, bool odd) {
  const i32 sgn = odd ? 0 : -1;
  Sx<C, LA> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = (pair_swap1(a.v[i]) ^ sgn) - sgn;
  return r;
}

// own half of a * b:   even lane  a0 b0 - a1 b1,   odd lane  a1 b0 + a0 b1.
// Column factors: the own half of a and the partner's (sign-adjusted); row factors: limb i of b's even-lane half and of its
// odd-lane half, fetched per row by quad permutes -- no select, and b's neighbour half never occupies NL registers.
template <class C, int LA, int LB>
RX_DEV Sx<C, SX_T> pair_mul(const Sx<C, LA>& a, const Sx<C, LB>& b, bool odd) {
  const Sx<C, LA> pa = pair_swap_neg_even<C>(a, odd);
  const i32* const cols[2] = {a.v, pa.v};
  return sx_montr<C, 2, 2 * LA * LB>(cols, [&](int k, int i) { return k == 0 ? pair_even1(b.v[i], odd) : pair_odd1(b.v[i], odd); });
}
// own half of a * k for a constant k = (k_re, k_im) known to every lane:  a0 kr - a1 ki  |  a1 kr + a0 ki
template <class C, int LA>
RX_DEV Sx<C, SX_T> pair_mul_const(const Sx<C, LA>& a, const u32* k_re, const u32* k_im, bool odd) {
  const Sx<C, LA> pa = pair_swap_neg_even<C>(a, odd);
  const i32* const cols[2] = {a.v, pa.v};
  return sx_montr<C, 2, 2 * LA * SX_T>(cols, [&](int k, int i) { return (i32)(k == 0 ? k_re[i] : k_im[i]); });
}
// own half of a^2:  (a0 + a1)(a0 - a1)  |  2 a1 a0
// Column budget: 2 LA x 2 LA in general.  The 29-bit form has no room for that and does not need it: every value squared in the
// point steps is a reduction's output or one parallel carry step behind a sum (limbs 0..NL-2 non-negative up to 2^4: LA <= SX_F),
// so the DIFFERENCE a0 - a1 stays inside LA + 1 and only the sum doubles.
template <class C, int LA>
RX_DEV Sx<C, SX_T> pair_sqr(const Sx<C, LA>& a, bool odd) {
  static_assert(rx_lazy<C> || LA <= SX_F, "29-bit form: squares of (nearly) non-negative limbs only");
  constexpr int BUDGET = rx_lazy<C> ? 4 * LA * LA : 2 * LA * (LA + 1);
  Sx<C, 2 * LA> u;                                            // a0 + a1 | 2 a1
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) u.v[i] = a.v[i] + pair_odd1(a.v[i], odd);
  const i32 even = odd ? 0 : -1;
  const i32* const cols[1] = {u.v};
  return sx_montr<C, 1, BUDGET>(cols, [&](int, int i) { return pair_even1(a.v[i], odd) - (pair_odd1(a.v[i], odd) & even); });   // a0 - a1 | a0
}
// own half of a * s, s in Fp (the same value on both lanes)
template <class C, int LA, int LS>
RX_DEV Sx<C, SX_T> pair_muls(const Sx<C, LA>& a, const Sx<C, LS>& s) {
  const i32* const cols[1] = {s.v};
  return sx_montr<C, 1, LA * LS>(cols, [&](int, int i) { return a.v[i]; });
}
// own half of a b - c d, one reduction (the subtracted products enter with negated ROW factors: one instruction per row
// instead of NL registers of negated columns)
template <class C, int LA, int LB, int LC, int LD>
RX_DEV Sx<C, SX_T> pair_mulsub(const Sx<C, LA>& a, const Sx<C, LB>& b, const Sx<C, LC>& c, const Sx<C, LD>& d, bool odd) {
  static_assert(rx_fits<C>(2 * LA * LB + 2 * LC * LD), "four products per reduction: 28-bit forms (the 29-bit form reduces the two products apart)");
  const Sx<C, LA> pa = pair_swap_neg_even<C>(a, odd);
  const Sx<C, LC> pc = pair_swap_neg_even<C>(c, odd);
  const i32* const cols[4] = {a.v, pa.v, c.v, pc.v};
  return sx_montr<C, 4, 2 * LA * LB + 2 * LC * LD>(cols, [&](int k, int i) {
    return k == 0 ? pair_even1(b.v[i], odd) : (k == 1 ? pair_odd1(b.v[i], odd) : (k == 2 ? -pair_even1(d.v[i], odd) : -pair_odd1(d.v[i], odd)));
  });
}
// own half of g^2 - e f, one reduction
template <class C, int LG, int LE, int LF>
RX_DEV Sx<C, SX_T> pair_sqrsub(const Sx<C, LG>& g, const Sx<C, LE>& e, const Sx<C, LF>& f, bool odd) {
  static_assert(rx_fits<C>(4 * LG * LG + 2 * LE * LF), "three products per reduction: 28-bit forms");
  Sx<C, 2 * LG> u;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) u.v[i] = g.v[i] + pair_odd1(g.v[i], odd);
  const i32 even = odd ? 0 : -1;
  const Sx<C, LE> pe = pair_swap_neg_even<C>(e, odd);
  const i32* const cols[3] = {u.v, e.v, pe.v};
  return sx_montr<C, 3, 4 * LG * LG + 2 * LE * LF>(cols, [&](int k, int i) {
    return k == 0 ? pair_even1(g.v[i], odd) - (pair_odd1(g.v[i], odd) & even) : (k == 1 ? -pair_even1(f.v[i], odd) : -pair_odd1(f.v[i], odd));
  });
}
// own half of g^2 - 3 e^2, one reduction over TWO products (round 6; rounds 3-5 formed e * 3 e as a general product: two column factors, NL^2
// multiplier instructions more per lane and doubling step).  Both are squares: own half of a^2 = (a0 + a1)(a0 - a1) | (2 a1) a0, so the column factors
// are g + its odd-lane half and e + its odd-lane half, the row factors g0 - g1 | g0 and -3 (e0 - e1) | -3 e0 -- the multiple formed in the ROW factors,
// so that 3 e never occupies NL registers next to e (round 5: the doubling step's register peak sits on this product; BLS12-381 spilled there).
template <class C, int LG, int LE>
RX_DEV Sx<C, SX_T> pair_sqrsub3(const Sx<C, LG>& g, const Sx<C, LE>& e, bool odd) {
  // Budget (units 2^(2W-8)): g is one parallel carry step behind a sum (limbs in (-2^4, 2^W + 2^4)), so the sum g0 + g1 stays below
  // 2 (2^W + 2^4) and the difference g0 - g1 inside +-(2^W + 2^5): 512.0002 units, counted as 513, where the bounds' product
  // (2 * 17) * (2 * 17) would say 1156.  e is a reduction's (or the centred quasi-reduction's) output: limbs 0 .. NL-2 in [0, 2^W), so e0 + e1 is
  // below 2 * 2^W and e0 - e1 inside +-2^W: 3 * (2 LE) * LE units.
  static_assert(LG <= SX_F && LE <= SX_T, "g: a reduction's output or one carry step behind a sum; e: a reduction's output");
  constexpr int BUDGET = 513 + 6 * LE * LE;
  static_assert(rx_fits<C>(BUDGET), "two squares per reduction: 28-bit forms");
  Sx<C, 2 * LG> u;
  Sx<C, 2 * LE> w;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) {
    u.v[i] = g.v[i] + pair_odd1(g.v[i], odd);
    w.v[i] = e.v[i] + pair_odd1(e.v[i], odd);
  }
  const i32 even = odd ? 0 : -1;
  const i32* const cols[2] = {u.v, w.v};
  return sx_montr<C, 2, BUDGET>(cols, [&](int k, int i) {
    return k == 0 ? pair_even1(g.v[i], odd) - (pair_odd1(g.v[i], odd) & even) : -3 * (pair_even1(e.v[i], odd) - (pair_odd1(e.v[i], odd) & even));
  });
}
// 3 b' z of the doubling step.  alt-bn128: a product by the constant (3 b' = 9 / (9 + i) is a full-size element).  BLS12-381: 3 b' = 12 (1 + i)
// (the M-type twist y^2 = x^3 + 4 (1 + i): curves/bls12_381.go), so the product is 12 (z0 - z1) + 12 (z0 + z1) i -- the partner's half
// by one DPP move per limb, an addition, x 3, x 4 with parallel carry steps between them (the limb bounds of Sx) -- and, since that integer is up to
// 25 p large, a centred quasi-reduction (sx_quasi_center: NL multiplier instructions) that hands every later use the small tight representative a
// Montgomery product would have: the same field element, so the same lines and partial products bit for bit.  Round 5: 2 NL^2 + NL^2 multiplier
// instructions and a reduction's bookkeeping (~725 instructions per lane on fourteen limbs) become ~250; same-box A/B with a timing stand-in
// 86.3 -> 84.8 ms per 2^20 pairings.  (Rounds 3-4 kept the constant product because the VALUE had to reach the consumer below 32 p.)
template <class C, int LA>
RX_DEV Sx<C, SX_T> pair_mul_3b(const Sx<C, LA>& z, bool odd) {
  if constexpr (C::CURVE_ID == 1) {
    static_assert(C::XI_RE == 1 && !C::TWIST_D, "3 b' = 12 (1 + i): BLS12-381's M-type twist");
    const auto t = sx_normf<C>(sx_add<C>(z, pair_swap_neg_even<C>(z, odd)));            // z0 - z1 | z1 + z0
    const auto t12 = sx_mulc<4, C>(sx_normf<C>(sx_mulc<3, C>(t)));
    return sx_quasi_center<C>(t12);
  } else {
    return pair_mul_const<C>(z, C::RX_B2X3_RE, C::RX_B2X3_IM, odd);
  }
}

template <class C>
struct PointX {
  Sx<C, SX_T> X, Y, Z;     // own halves
};

// Doubling step.  emit(slot, value): slot 2 = the P-free coefficient (any bound), slots 0 / 1 = the coefficients already
// scaled by yP / xP (tight).  env.nyP() = -yP and env.xP() are the hash point's coordinates in this form (same on both
// lanes); they are fetched where they are used so that they do not occupy registers through the step.
//
// 29-bit form (BN254W): the column budget holds TWO products of near-tight factors per reduction, so every factor that is a sum or
// a multiple goes through one parallel carry step first (sx_normf: three instructions per limb) and the two results that the 28-bit
// forms reduce lazily as differences of products (Y3 here, Y3 and the P-free line coefficient of the addition) are reduced apart
// and subtracted: one reduction more per doubling step (28 units of NL^2 = 81 multiplier instructions against 27 of 100 / 196), two more
// per addition step (41 against 39).  Same values mod p at every step, hence the same lines.
template <class C, class T>
RX_DEV auto rx_nf(const T& a) {              // one carry step where the form has no head-room for the raw sum
  if constexpr (rx_lazy<C>) return a;
  else return sx_normf<C>(a);
}
template <class C, class Env, class Emit>
RX_DEV void dbl_step_x(PointX<C>& R, Env&& env, bool odd, Emit&& emit) {
  const Sx<C, SX_T> B = pair_sqr<C>(R.Y, odd);
  const Sx<C, SX_T> Cc = pair_sqr<C>(R.Z, odd);
  const auto H = rx_nf<C>(sx_sub<C>(pair_sqr<C>(sx_normf<C>(sx_add<C>(R.Y, R.Z)), odd), sx_add<C>(B, Cc)));      // 2 Y Z
  const Sx<C, SX_T> E = pair_mul_3b<C>(Cc, odd);
  const auto J3 = rx_nf<C>(sx_mulc<3, C>(pair_sqr<C>(R.X, odd)));      // 3 X^2
  const auto A = rx_nf<C>(sx_half<C>(pair_mul<C>(R.X, R.Y, odd)));
  const Sx<C, SX_F> Fv = sx_normf<C>(sx_mulc<3, C>(E));
  R.X = pair_mul<C>(A, rx_nf<C>(sx_sub<C>(B, Fv)), odd);
  const Sx<C, SX_F> G = sx_normf<C>(sx_half<C>(sx_add<C>(B, Fv)));
  // G^2 - 3 E^2: the step's register peak (Z3 and I come after it).  The 28-bit forms take 3 E in the row factors, so that Fv is dead here.
  if constexpr (rx_lazy<C>) R.Y = pair_sqrsub3<C>(G, E, odd);
  else R.Y = sx_norm<C>(sx_sub<C>(pair_sqr<C>(G, odd), sx_mulc<3, C>(pair_sqr<C>(E, odd))));      // 3 E^2 as a square (round 6; E * 3 E before)
  const auto I = sx_sub<C>(E, B);
  R.Z = pair_mul<C>(B, H, odd);
  // the line, last: its coefficients go straight from registers to the hand-over
  emit(2, I);                                                // I = E - B
  emit(0, pair_muls<C>(H, env.nyP()));                       // (-H) yP
  emit(1, pair_muls<C>(J3, env.xP()));                       // 3 X^2 xP
}

// Mixed addition step with the affine point (env.xq(), env.yq()) (own halves, tight).
template <class C, class Env, class Emit>
RX_DEV void add_step_x(PointX<C>& R, Env&& env, bool odd, Emit&& emit) {
  const Sx<C, SX_F> th = sx_normf<C>(sx_sub<C>(R.Y, pair_mul<C>(env.yq(), R.Z, odd)));
  const Sx<C, SX_F> la = sx_normf<C>(sx_sub<C>(R.X, pair_mul<C>(env.xq(), R.Z, odd)));
  const Sx<C, SX_T> D = pair_sqr<C>(la, odd);
  const Sx<C, SX_T> G = pair_mul<C>(R.X, D, odd);
  const Sx<C, SX_T> E = pair_mul<C>(la, D, odd);
  const Sx<C, SX_T> Fv = pair_mul<C>(R.Z, pair_sqr<C>(th, odd), odd);
  const Sx<C, SX_F> Hh = sx_normf<C>(sx_sub<C>(sx_add<C>(E, Fv), sx_mulc<2, C>(G)));
  R.X = pair_mul<C>(la, Hh, odd);
  R.Z = pair_mul<C>(R.Z, E, odd);
  if constexpr (rx_lazy<C>) {
    R.Y = pair_mulsub<C>(th, sx_sub<C>(G, Hh), E, R.Y, odd);
    // the line, last (la and th are live to the end anyway)
    emit(2, pair_mulsub<C>(th, env.xq(), la, env.yq(), odd));  // th xq - la yq
  } else {
    R.Y = sx_norm<C>(sx_sub<C>(pair_mul<C>(th, sx_normf<C>(sx_sub<C>(G, Hh)), odd), pair_mul<C>(E, R.Y, odd)));
    emit(2, sx_sub<C>(pair_mul<C>(th, env.xq(), odd), pair_mul<C>(la, env.yq(), odd)));
  }
  emit(0, pair_muls<C>(sx_neg<C>(la), env.nyP()));           // la yP
  emit(1, pair_muls<C>(sx_neg<C>(th), env.xP()));            // (-th) xP
}

}  // namespace bgls
