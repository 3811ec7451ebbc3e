// G2 key sums on LANE PAIRS in the carry-free form: the Jacobian mixed addition of rx_jac.hpp (madd-2007-bl) with every Fp2 value
// split over two neighbouring lanes -- the even lane holds the real parts, the odd lane the imaginary parts (rx_pair.hpp).
//
// Why a pair: one lane owning a whole G2 addition carries 6 NL registers of running point plus the temporaries of Fp2 products
// (k_sum_main: 256 registers, 70 spilled, two waves per SIMD, 0.24 of the multiplier peak; rx_jac.hpp on one lane: 224 registers).
// On a pair each lane holds half of everything, an Fp2 product is two limb products sharing ONE reduction per lane (28 units of
// NL^2 multiplier instructions per lane and addition instead of 56 on one lane), and the kernel runs three waves per SIMD (alt-bn128: 7 spilled registers).
//
// Same group law and exceptional cases as the reference's chain of Add (AggregatePoints, curves/curve.go:73-121): P = Q doubles,
// P = -Q and points at infinity are exact, so the sum is the same point.  Control flow is uniform over a pair (the conditions
// are AND-ed over both halves), which is what the quad permutes need.
#pragma once
#include "curve.hpp"
#include "rx_pair.hpp"

namespace bgls {

template <class C>
struct JacP {          // own halves of a Jacobian point; inf is the same on both lanes
  Sx<C, SX_F> X, Y, Z;
  bool inf;
};
template <class C>
struct AffP {
  Sx<C, SX_T> x, y;
  bool inf;
};

// true on both lanes iff it is true on both
RX_DEV bool pair_both(bool own) { return own && pair_swap1(own ? 1 : 0) != 0; }
RX_DEV bool pair_any(bool own) { return own || pair_swap1(own ? 1 : 0) != 0; }

template <class C>
RX_DEV JacP<C> jacp_inf() {
  JacP<C> r;
  r.X = r.Y = r.Z = sx_as<SX_F, C>(ux_to_sx<C>(ux_zero<C>()));
  r.inf = true;
  return r;
}
template <class C>
RX_DEV Sx<C, SX_T> pair_one(bool odd) {
  return sx_select<C>(odd, ux_to_sx<C>(ux_zero<C>()), sx_const<C>(C::RX_ONE));
}

// y^2 = x^3 + b' on the twist, decided for the pair
template <class C>
RX_DEV bool affp_on_curve(const AffP<C>& q, bool odd) {
  const Sx<C, SX_T> b2 = sx_const<C>(odd ? C::RX_B2_IM : C::RX_B2_RE);
  const auto d = sx_sub<C>(pair_sqr<C>(q.y, odd), sx_add<C>(pair_mul<C>(pair_sqr<C>(q.x, odd), q.x, odd), b2));
  return pair_both(sx_is_zero_mod_p<C>(d)) || q.inf;
}

// own half of a b - c d as a lazy value (SX_F): one interleaved reduction on the 28-bit forms; on the 29-bit form, whose columns hold
// two products of near-tight factors, the two products are reduced apart and subtracted (c, whose bound is doubled by the callers,
// goes through a parallel carry step first)
template <class C, int LA, int LB, int LC, int LD>
RX_DEV Sx<C, SX_F> pair_mulsub_f(const Sx<C, LA>& a, const Sx<C, LB>& b, const Sx<C, LC>& c, const Sx<C, LD>& d, bool odd) {
  if constexpr (rx_fits<C>(2 * LA * LB + 2 * LC * LD)) return sx_as<SX_F, C>(pair_mulsub<C>(a, b, c, d, odd));
  else return sx_normf<C>(sx_sub<C>(pair_mul<C>(a, b, odd), pair_mul<C>(sx_normf<C>(c), d, odd)));
}

// 2 p (dbl-2009-l); the rare branch of the mixed addition (a key met twice in one pair's slice)
template <class C>
RX_DEV JacP<C> jacp_dbl(const JacP<C>& p, bool odd) {
  if (p.inf) return p;
  const Sx<C, SX_T> A = pair_sqr<C>(p.X, odd), B = pair_sqr<C>(p.Y, odd), Cc = pair_sqr<C>(B, odd);
  const Sx<C, SX_F> D = sx_normf<C>(sx_mulc<2, C>(sx_sub<C>(sx_sub<C>(pair_sqr<C>(sx_normf<C>(sx_add<C>(p.X, B)), odd), A), Cc)));
  const Sx<C, SX_F> E = sx_normf<C>(sx_mulc<3, C>(A));
  const Sx<C, SX_T> F = pair_sqr<C>(E, odd);
  JacP<C> r;
  r.X = sx_normf<C>(sx_sub<C>(F, sx_mulc<2, C>(D)));
  r.Y = sx_normf<C>(sx_sub<C>(pair_mul<C>(E, sx_normf<C>(sx_sub<C>(D, r.X)), odd), sx_mulc<2, C>(sx_normf<C>(sx_mulc<4, C>(Cc)))));
  r.Z = sx_normf<C>(sx_mulc<2, C>(pair_mul<C>(p.Y, p.Z, odd)));
  r.inf = false;
  return r;
}

// p + q, q affine (madd-2007-bl): 28 units of NL^2 multiplier instructions per lane
template <class C>
RX_DEV JacP<C> jacp_madd(const JacP<C>& p, const AffP<C>& q, bool odd) {
  if (q.inf) return p;
  if (p.inf) {
    JacP<C> r;
    r.X = sx_as<SX_F, C>(q.x);
    r.Y = sx_as<SX_F, C>(q.y);
    r.Z = sx_as<SX_F, C>(pair_one<C>(odd));
    r.inf = false;
    return r;
  }
  const Sx<C, SX_T> Z1Z1 = pair_sqr<C>(p.Z, odd);
  const Sx<C, SX_T> U2 = pair_mul<C>(q.x, Z1Z1, odd);
  const Sx<C, SX_T> S2 = pair_mul<C>(pair_mul<C>(q.y, p.Z, odd), Z1Z1, odd);
  const auto Hd = sx_sub<C>(U2, p.X);
  const auto Rd = sx_sub<C>(S2, p.Y);
  if (pair_both(sx_is_zero_mod_p<C>(Hd))) {                     // same x: P = Q (double) or P = -Q (infinity)
    if (pair_both(sx_is_zero_mod_p<C>(Rd))) return jacp_dbl<C>(p, odd);
    return jacp_inf<C>();
  }
  const Sx<C, SX_F> H = sx_normf<C>(Hd);
  const Sx<C, SX_F> rr = sx_normf<C>(sx_mulc<2, C>(Rd));
  const Sx<C, SX_T> HH = pair_sqr<C>(H, odd);
  const Sx<C, SX_F> I = sx_normf<C>(sx_mulc<4, C>(HH));
  const Sx<C, SX_T> J = pair_mul<C>(H, I, odd);
  const Sx<C, SX_T> V = pair_mul<C>(p.X, I, odd);
  JacP<C> r;
  r.X = sx_normf<C>(sx_sub<C>(sx_sub<C>(pair_sqr<C>(rr, odd), J), sx_mulc<2, C>(V)));
  r.Y = pair_mulsub_f<C>(rr, sx_normf<C>(sx_sub<C>(V, r.X)), sx_mulc<2, C>(p.Y), J, odd);
  r.Z = sx_normf<C>(sx_sub<C>(sx_sub<C>(pair_sqr<C>(sx_normf<C>(sx_add<C>(p.Z, H)), odd), Z1Z1), HH));
  r.inf = false;
  return r;
}

// p + q, both Jacobian (add-2007-bl, 11 products + 5 squarings): the in-block tree above the per-pair partial sums
template <class C>
RX_DEV JacP<C> jacp_add(const JacP<C>& p, const JacP<C>& q, bool odd) {
  if (q.inf) return p;
  if (p.inf) return q;
  const Sx<C, SX_T> Z1Z1 = pair_sqr<C>(p.Z, odd), Z2Z2 = pair_sqr<C>(q.Z, odd);
  const Sx<C, SX_T> U1 = pair_mul<C>(p.X, Z2Z2, odd), U2 = pair_mul<C>(q.X, Z1Z1, odd);
  const Sx<C, SX_T> S1 = pair_mul<C>(pair_mul<C>(p.Y, q.Z, odd), Z2Z2, odd);
  const Sx<C, SX_T> S2 = pair_mul<C>(pair_mul<C>(q.Y, p.Z, odd), Z1Z1, odd);
  const auto Hd = sx_sub<C>(U2, U1);
  const auto Rd = sx_sub<C>(S2, S1);
  if (pair_both(sx_is_zero_mod_p<C>(Hd))) {                     // same x: P = Q (double) or P = -Q (infinity)
    if (pair_both(sx_is_zero_mod_p<C>(Rd))) return jacp_dbl<C>(p, odd);
    return jacp_inf<C>();
  }
  const Sx<C, SX_F> H = sx_normf<C>(Hd);
  const Sx<C, SX_F> rr = sx_normf<C>(sx_mulc<2, C>(Rd));
  const Sx<C, SX_T> I = pair_sqr<C>(sx_normf<C>(sx_mulc<2, C>(H)), odd);
  const Sx<C, SX_T> J = pair_mul<C>(H, I, odd);
  const Sx<C, SX_T> V = pair_mul<C>(U1, I, odd);
  JacP<C> r;
  r.X = sx_normf<C>(sx_sub<C>(sx_sub<C>(pair_sqr<C>(rr, odd), J), sx_mulc<2, C>(V)));
  r.Y = pair_mulsub_f<C>(rr, sx_normf<C>(sx_sub<C>(V, r.X)), sx_mulc<2, C>(S1), J, odd);
  const Sx<C, SX_F> zz = sx_normf<C>(sx_sub<C>(sx_sub<C>(pair_sqr<C>(sx_normf<C>(sx_add<C>(p.Z, q.Z)), odd), Z1Z1), Z2Z2));
  r.Z = sx_as<SX_F, C>(pair_mul<C>(zz, H, odd));
  r.inf = false;
  return r;
}

// own halves of a key from its wire bytes (x_im || x_re || y_im || y_re, big-endian; all zero = infinity); ok = canonical
template <class C>
RX_DEV bool affp_from_bytes(AffP<C>& out, const uint8_t* b, bool odd) {
  constexpr int NB = C::FP_BYTES;
  const Fp<C> xw = fp_from_be<C>(b + (odd ? 0 : NB)), yw = fp_from_be<C>(b + (odd ? 2 * NB : 3 * NB));
  const bool canon = pair_both(!fp_geq_p<C>(xw) && !fp_geq_p<C>(yw));
  out.inf = pair_both(fp_is_zero<C>(xw) && fp_is_zero<C>(yw));
  out.x = sx_from_plain<C>(xw);
  out.y = sx_from_plain<C>(yw);
  return canon;
}
// own halves of a parsed key (the library's Montgomery form)
template <class C>
RX_DEV AffP<C> affp_from_mont(const Aff<F2<C>>& a, bool odd) {
  AffP<C> r;
  r.inf = a.inf;
  r.x = ux_to_sx<C>(to_ux<C>(odd ? a.x.c1 : a.x.c0));
  r.y = ux_to_sx<C>(to_ux<C>(odd ? a.y.c1 : a.y.c0));
  return r;
}
// R' form (limbs below 2^(W+1), |value| < 8 p) -> the library's form (see rx_jac.hpp sx_to_mont)
template <class C, int LA>
RX_DEV Fp<C> sxp_to_mont(const Sx<C, LA>& a) {
  constexpr int N = C::RX_NL;
  const Sx<C, SX_T> one = sx_const<C>(C::RX_ONE);
  const i32* const cols[1] = {one.v};
  const Sx<C, SX_T> r = sx_montr<C, 1, LA * SX_T>(cols, [&](int, int i) { return a.v[i]; });
  Sx<C, 2 * SX_T> t;
#pragma unroll
  for (int i = 0; i < N; ++i) t.v[i] = r.v[i] + (i32)C::RX_PK[N + i];      // + p
  const Sx<C, SX_T> n = sx_norm<C>(t);
  Ux<C> u;
#pragma unroll
  for (int i = 0; i < N; ++i) u.v[i] = (u32)n.v[i];
  return from_ux<C>(u);
}

}  // namespace bgls
