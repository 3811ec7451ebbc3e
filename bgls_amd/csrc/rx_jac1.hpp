// G1 arithmetic on ONE lane in the carry-free 28-bit-limb form of rx.hpp (signed limbs, compile-time bounds): the scalar
// multiplications at the seam that work on G1 -- Sign = HashToG1(m).Mul(sk) (bgls/bgls.go:46-56), ScalePoints / Point.Mul
// on G1 (curves/curve.go:190-214), and the cofactor clearing inside BLS12-381's HashToG1 (curves/bls12_381.go:361-376).
//
// Why: these ran on curve.hpp's 32-bit Montgomery form -- three VALU instructions per multiplier instruction and dependent
// carry chains (k_scale_aff<BLS381, G1> 139 ms and k_bls_combine<false> 51 ms per 2^20 points in round 3).  Here a field
// product is NL^2 bare multiplier instructions into NL + 1 live columns plus NL^2 for its reduction (sx_montr), sums and
// differences are limb-wise, and a difference of two products shares one reduction.  The formulas are rx_jac.hpp's (the
// same ones over Fp2 serve the G2 key sums): dbl-2009-l, madd-2007-bl, add-2007-bl; exceptional cases (P = Q, P = -Q,
// infinity) are exact, so every chain gives the same POINT as curve.hpp's and, after normalisation, the same bytes.
#pragma once
#include "curve.hpp"
#include "rx.hpp"
#include "rx_jac.hpp"

namespace bgls {

template <class C, int LA, int LB>
BGLS_HD Sx<C, SX_T> s1_mul(const Sx<C, LA>& a, const Sx<C, LB>& b) {
  const i32* const cols[1] = {b.v};
  return sx_montr<C, 1, LA * LB>(cols, [&](int, int i) { return a.v[i]; });
}
template <class C, int LA>
BGLS_HD Sx<C, SX_T> s1_sqr(const Sx<C, LA>& a) {
  const i32* const cols[1] = {a.v};
  return sx_montr<C, 1, LA * LA>(cols, [&](int, int i) { return a.v[i]; });
}
// a b - c d, one reduction
template <class C, int LA, int LB, int LC, int LD>
BGLS_HD Sx<C, SX_T> s1_mulsub(const Sx<C, LA>& a, const Sx<C, LB>& b, const Sx<C, LC>& c, const Sx<C, LD>& d) {
  const i32* const cols[2] = {b.v, d.v};
  return sx_montr<C, 2, LA * LB + LC * LD>(cols, [&](int k, int i) { return k == 0 ? a.v[i] : -c.v[i]; });
}

template <class C>
struct Jac1 {
  Sx<C, SX_F> X, Y, Z;
  bool inf;
};
template <class C>
struct Aff1 {
  Sx<C, SX_T> x, y;
  bool inf;
};
template <class C>
BGLS_HD Jac1<C> jac1_inf() {
  Jac1<C> r;
  const Sx<C, SX_F> z = sx_as<SX_F, C>(ux_to_sx<C>(ux_zero<C>()));
  r.X = z; r.Y = z; r.Z = z;
  r.inf = true;
  return r;
}
template <class C>
BGLS_HD Aff1<C> aff1_from_mont(const Aff<F1<C>>& a) {
  Aff1<C> r;
  r.inf = a.inf;
  r.x = sx_from_mont<C>(a.x);
  r.y = sx_from_mont<C>(a.y);
  return r;
}
template <class C>
BGLS_HD Aff1<C> aff1_neg(const Aff1<C>& a) {
  Aff1<C> r = a;
  r.y = sx_norm<C>(sx_neg<C>(a.y));          // limbs tight again, the sign sits in the top limb
  return r;
}
template <class C>
BGLS_HD Jac1<C> jac1_from_aff(const Aff1<C>& q) {
  Jac1<C> r;
  r.X = sx_as<SX_F, C>(q.x);
  r.Y = sx_as<SX_F, C>(q.y);
  r.Z = sx_as<SX_F, C>(sx_const<C>(C::RX_ONE));
  r.inf = q.inf;
  return r;
}
template <class C>
BGLS_HD Jac<F1<C>> jac1_to_mont(const Jac1<C>& p) {
  if (p.inf) return jac_inf<F1<C>>();
  return {sx_to_mont<C>(p.X), sx_to_mont<C>(p.Y), sx_to_mont<C>(p.Z)};
}

// 2 p (dbl-2009-l, a = 0): 2 products + 5 squarings
template <class C>
BGLS_HD Jac1<C> jac1_dbl(const Jac1<C>& p) {
  if (p.inf) return p;
  const Sx<C, SX_T> A = s1_sqr<C>(p.X), B = s1_sqr<C>(p.Y), Cc = s1_sqr<C>(B);
  const Sx<C, SX_F> D = sx_normf<C>(sx_mulc<2, C>(sx_sub<C>(sx_sub<C>(s1_sqr<C>(sx_normf<C>(sx_add<C>(p.X, B))), A), Cc)));
  const Sx<C, SX_F> E = sx_normf<C>(sx_mulc<3, C>(A));
  const Sx<C, SX_T> F = s1_sqr<C>(E);
  Jac1<C> r;
  r.X = sx_normf<C>(sx_sub<C>(F, sx_mulc<2, C>(D)));
  r.Y = sx_normf<C>(sx_sub<C>(s1_mul<C>(E, sx_normf<C>(sx_sub<C>(D, r.X))), sx_mulc<2, C>(sx_normf<C>(sx_mulc<4, C>(Cc)))));
  r.Z = sx_normf<C>(sx_mulc<2, C>(s1_mul<C>(p.Y, p.Z)));
  r.inf = false;
  return r;
}

// p + q, q affine (madd-2007-bl): 7 products + 4 squarings
template <class C>
BGLS_HD Jac1<C> jac1_madd(const Jac1<C>& p, const Aff1<C>& q) {
  if (q.inf) return p;
  if (p.inf) return jac1_from_aff<C>(q);
  const Sx<C, SX_T> Z1Z1 = s1_sqr<C>(p.Z);
  const Sx<C, SX_T> U2 = s1_mul<C>(q.x, Z1Z1);
  const Sx<C, SX_T> S2 = s1_mul<C>(s1_mul<C>(q.y, p.Z), Z1Z1);
  const auto Hd = sx_sub<C>(U2, p.X);
  const auto Rd = sx_sub<C>(S2, p.Y);
  if (sx_is_zero_mod_p<C>(Hd)) {                               // same x: P = Q (double) or P = -Q (infinity)
    if (sx_is_zero_mod_p<C>(Rd)) return jac1_dbl<C>(p);
    return jac1_inf<C>();
  }
  const Sx<C, SX_F> H = sx_normf<C>(Hd);
  const Sx<C, SX_F> rr = sx_normf<C>(sx_mulc<2, C>(Rd));
  const Sx<C, SX_T> HH = s1_sqr<C>(H);
  const Sx<C, SX_F> I = sx_normf<C>(sx_mulc<4, C>(HH));
  const Sx<C, SX_T> J = s1_mul<C>(H, I);
  const Sx<C, SX_T> V = s1_mul<C>(p.X, I);
  Jac1<C> r;
  r.X = sx_normf<C>(sx_sub<C>(sx_sub<C>(s1_sqr<C>(rr), J), sx_mulc<2, C>(V)));
  r.Y = sx_as<SX_F, C>(s1_mulsub<C>(rr, sx_normf<C>(sx_sub<C>(V, r.X)), sx_mulc<2, C>(p.Y), J));
  r.Z = sx_normf<C>(sx_sub<C>(sx_sub<C>(s1_sqr<C>(sx_normf<C>(sx_add<C>(p.Z, H))), Z1Z1), HH));
  r.inf = false;
  return r;
}

// p + q, both Jacobian (add-2007-bl): 11 products + 5 squarings
template <class C>
BGLS_HD Jac1<C> jac1_add(const Jac1<C>& p, const Jac1<C>& q) {
  if (q.inf) return p;
  if (p.inf) return q;
  const Sx<C, SX_T> Z1Z1 = s1_sqr<C>(p.Z), Z2Z2 = s1_sqr<C>(q.Z);
  const Sx<C, SX_T> U1 = s1_mul<C>(p.X, Z2Z2), U2 = s1_mul<C>(q.X, Z1Z1);
  const Sx<C, SX_T> S1 = s1_mul<C>(s1_mul<C>(p.Y, q.Z), Z2Z2), S2 = s1_mul<C>(s1_mul<C>(q.Y, p.Z), Z1Z1);
  const auto Hd = sx_sub<C>(U2, U1);
  const auto Rd = sx_sub<C>(S2, S1);
  if (sx_is_zero_mod_p<C>(Hd)) {
    if (sx_is_zero_mod_p<C>(Rd)) return jac1_dbl<C>(p);
    return jac1_inf<C>();
  }
  const Sx<C, SX_F> H = sx_normf<C>(Hd);
  const Sx<C, SX_F> rr = sx_normf<C>(sx_mulc<2, C>(Rd));
  const Sx<C, SX_F> I = sx_normf<C>(sx_mulc<4, C>(s1_sqr<C>(H)));
  const Sx<C, SX_T> J = s1_mul<C>(H, I);
  const Sx<C, SX_T> V = s1_mul<C>(U1, I);
  Jac1<C> r;
  r.X = sx_normf<C>(sx_sub<C>(sx_sub<C>(s1_sqr<C>(rr), J), sx_mulc<2, C>(V)));
  r.Y = sx_as<SX_F, C>(s1_mulsub<C>(rr, sx_normf<C>(sx_sub<C>(V, r.X)), sx_mulc<2, C>(S1), J));
  r.Z = sx_as<SX_F, C>(s1_mul<C>(sx_normf<C>(sx_sub<C>(sx_sub<C>(s1_sqr<C>(sx_normf<C>(sx_add<C>(p.Z, q.Z))), Z1Z1), Z2Z2)), H));
  r.inf = false;
  return r;
}

// k * P for a per-lane scalar of up to 256 bits: curve.hpp's jac_mul_w4 (signed radix-16 digits against P .. 8P, four
// doublings and ONE general addition per window whatever the digit) on this arithmetic.  Same point.
template <class C>
BGLS_FN Jac1<C> jac1_mul_w4(const Aff1<C>& p, const u32* k, int nbits) {
  if (p.inf || nbits <= 0) return jac1_inf<C>();
  u32 w[9];
  const int nl = (nbits + 31) >> 5;
#pragma unroll
  for (int j = 0; j < 9; ++j) w[j] = j < nl && j < 8 ? k[j] : 0u;
  if (nbits & 31) w[nl - 1] &= (1u << (nbits & 31)) - 1u;
  Jac1<C> tab[8];                                  // tab[a - 1] = a P
  tab[0] = jac1_from_aff<C>(p);
  tab[1] = jac1_dbl<C>(tab[0]);
  tab[2] = jac1_madd<C>(tab[1], p);
  tab[3] = jac1_dbl<C>(tab[1]);
  tab[4] = jac1_madd<C>(tab[3], p);
  tab[5] = jac1_dbl<C>(tab[2]);
  tab[6] = jac1_madd<C>(tab[5], p);
  tab[7] = jac1_dbl<C>(tab[3]);
  const int nw = (nbits + 4) >> 2;                 // one bit above the scalar: the top window's sign bit is clear
  Jac1<C> r = jac1_inf<C>();
#pragma unroll 1
  for (int i = nw - 1; i >= 0; --i) {
    if (i != nw - 1) {
#pragma unroll 1
      for (int d = 0; d < 4; ++d) r = jac1_dbl<C>(r);
    }
    const int pos = 4 * i - 1;                     // bits pos .. pos + 4
    u32 b5;
    if (pos < 0) {
      b5 = (w[0] << 1) & 31u;
    } else {
      const int q = pos >> 5, sh = pos & 31;
      u32 lo = w[q] >> sh;
      if (sh > 27) lo |= w[q + 1 < 9 ? q + 1 : 8] << (32 - sh);
      b5 = lo & 31u;
    }
    const int mag = (int)(((b5 & 15u) + 1u) >> 1), neg8 = (int)(b5 >> 4) * 8;
    const int val = mag - neg8;
    const int a = val < 0 ? -val : val;
    if (a) {
      Jac1<C> q = tab[a - 1];
      if (val < 0) q.Y = sx_as<SX_F, C>(sx_norm<C>(sx_neg<C>(q.Y)));
      r = jac1_add<C>(r, q);
    }
  }
  return r;
}

}  // namespace bgls
