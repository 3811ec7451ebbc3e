// G2 key sums on the carry-free 28-bit-limb arithmetic of rx.hpp: Fp2 on ONE lane (signed limbs, compile-time bounds) and the
// Jacobian mixed addition a key sum is made of (AggregatePoints, curves/curve.go:73-121: n - 1 additions of unit-weight keys).
//
// Why: the 32-bit form of this loop (points_inl.hpp jac_madd_inl) runs at 0.24 of the multiplier's peak -- three VALU
// instructions per multiplier instruction, 256 registers with 70 spilled.  Here an Fp2 product is four interleaved
// limb products and two reductions (sx_montr: bare v_mad_i64_i32 into NL + 1 live columns), differences are limb-wise, and
// the sum of two products shares one reduction.  Same group law, same exceptional cases (P = Q doubles, P = -Q and the
// point at infinity are exact), so the sum is the same point; the result leaves in the library's 32-bit Montgomery form.
#pragma once
#include "curve.hpp"
#include "rx.hpp"

namespace bgls {

template <class C, int LB>
struct X2 {
  Sx<C, LB> c0, c1;
};

template <class C, int LA, int LB>
BGLS_HD X2<C, LA + LB> x2_add(const X2<C, LA>& a, const X2<C, LB>& b) { return {sx_add<C>(a.c0, b.c0), sx_add<C>(a.c1, b.c1)}; }
template <class C, int LA, int LB>
BGLS_HD X2<C, LA + LB> x2_sub(const X2<C, LA>& a, const X2<C, LB>& b) { return {sx_sub<C>(a.c0, b.c0), sx_sub<C>(a.c1, b.c1)}; }
template <int K, class C, int LA>
BGLS_HD X2<C, K * LA> x2_mulc(const X2<C, LA>& a) { return {sx_mulc<K, C>(a.c0), sx_mulc<K, C>(a.c1)}; }
template <class C, int LA>
BGLS_HD X2<C, SX_F> x2_normf(const X2<C, LA>& a) { return {sx_normf<C>(a.c0), sx_normf<C>(a.c1)}; }
template <class C, int LA>
BGLS_HD X2<C, LA> x2_select(bool c, const X2<C, LA>& a, const X2<C, LA>& b) { return {sx_select<C>(c, a.c0, b.c0), sx_select<C>(c, a.c1, b.c1)}; }
template <int LB, class C, int LA>
BGLS_HD X2<C, LB> x2_as(const X2<C, LA>& a) { return {sx_as<LB, C>(a.c0), sx_as<LB, C>(a.c1)}; }

// a * b: (a0 b0 - a1 b1) + (a0 b1 + a1 b0) i, two products and one reduction per component
template <class C, int LA, int LB>
BGLS_HD X2<C, SX_T> x2_mul(const X2<C, LA>& a, const X2<C, LB>& b) {
  X2<C, SX_T> r;
  {
    const i32* const cols[2] = {b.c0.v, b.c1.v};
    r.c0 = sx_montr<C, 2, 2 * LA * LB>(cols, [&](int k, int i) { return k == 0 ? a.c0.v[i] : -a.c1.v[i]; });
  }
  {
    const i32* const cols[2] = {b.c1.v, b.c0.v};
    r.c1 = sx_montr<C, 2, 2 * LA * LB>(cols, [&](int k, int i) { return k == 0 ? a.c0.v[i] : a.c1.v[i]; });
  }
  return r;
}
// a^2: (a0 + a1)(a0 - a1) + 2 a0 a1 i
template <class C, int LA>
BGLS_HD X2<C, SX_T> x2_sqr(const X2<C, LA>& a) {
  X2<C, SX_T> r;
  {
    const Sx<C, 2 * LA> u = sx_add<C>(a.c0, a.c1);
    const i32* const cols[1] = {u.v};
    r.c0 = sx_montr<C, 1, 4 * LA * LA>(cols, [&](int, int i) { return a.c0.v[i] - a.c1.v[i]; });
  }
  {
    const i32* const cols[1] = {a.c1.v};
    r.c1 = sx_montr<C, 1, 2 * LA * LA>(cols, [&](int, int i) { return 2 * a.c0.v[i]; });
  }
  return r;
}
// a b - c d, one reduction per component
template <class C, int LA, int LB, int LC, int LD>
BGLS_HD X2<C, SX_T> x2_mulsub(const X2<C, LA>& a, const X2<C, LB>& b, const X2<C, LC>& c, const X2<C, LD>& d) {
  X2<C, SX_T> r;
  {
    const i32* const cols[4] = {b.c0.v, b.c1.v, d.c0.v, d.c1.v};
    r.c0 = sx_montr<C, 4, 2 * LA * LB + 2 * LC * LD>(cols, [&](int k, int i) { return k == 0 ? a.c0.v[i] : (k == 1 ? -a.c1.v[i] : (k == 2 ? -c.c0.v[i] : c.c1.v[i])); });
  }
  {
    const i32* const cols[4] = {b.c1.v, b.c0.v, d.c1.v, d.c0.v};
    r.c1 = sx_montr<C, 4, 2 * LA * LB + 2 * LC * LD>(cols, [&](int k, int i) { return k == 0 ? a.c0.v[i] : (k == 1 ? a.c1.v[i] : (k == 2 ? -c.c0.v[i] : -c.c1.v[i])); });
  }
  return r;
}

template <class C, int LA>
BGLS_HD bool x2_is_zero(const X2<C, LA>& a) { return sx_is_zero_mod_p<C>(a.c0) && sx_is_zero_mod_p<C>(a.c1); }

// x R (the library's Montgomery form) -> R' form
template <class C>
BGLS_HD Sx<C, SX_T> sx_from_mont(const Fp<C>& y) { return ux_to_sx<C>(to_ux<C>(y)); }
// R' form (limbs below 2^29, |value| < 8 p) -> the library's form: a product by one brings the value into (-eps p, (1 + eps) p),
// adding p makes it positive, from_ux does the rest
template <class C, int LA>
BGLS_HD Fp<C> sx_to_mont(const Sx<C, LA>& a) {
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

// running sum of a key-sum thread: Jacobian (X, Y, Z) over Fp2, limbs almost tight, plus the infinity flag
template <class C>
struct JacX {
  X2<C, SX_F> X, Y, Z;
  bool inf;
};
template <class C>
struct AffX {
  X2<C, SX_T> x, y;
  bool inf;
};
template <class C>
BGLS_HD JacX<C> jacx_inf() {
  JacX<C> r;
  const Sx<C, SX_F> z = sx_as<SX_F, C>(ux_to_sx<C>(ux_zero<C>()));
  r.X = {z, z}; r.Y = {z, z}; r.Z = {z, z};
  r.inf = true;
  return r;
}

// y^2 = x^3 + b' on the twist
template <class C>
BGLS_HD bool affx_on_curve(const AffX<C>& q) {
  if (q.inf) return true;
  const X2<C, SX_T> b2 = {sx_const<C>(C::RX_B2_RE), sx_const<C>(C::RX_B2_IM)};
  const X2<C, SX_T> x3 = x2_mul<C>(x2_sqr<C>(q.x), q.x);
  return x2_is_zero<C>(x2_sub<C>(x2_sqr<C>(q.y), x2_add<C>(x3, b2)));
}

// 2 p (dbl-2009-l); the rare branch of the mixed addition (a key met twice in one thread's slice)
template <class C>
BGLS_FN JacX<C> jacx_dbl(const JacX<C>& p) {
  if (p.inf) return p;
  const X2<C, SX_T> A = x2_sqr<C>(p.X), B = x2_sqr<C>(p.Y), Cc = x2_sqr<C>(B);
  const X2<C, SX_F> D = x2_normf<C>(x2_mulc<2, C>(x2_sub<C>(x2_sub<C>(x2_sqr<C>(x2_normf<C>(x2_add<C>(p.X, B))), A), Cc)));
  const X2<C, SX_F> E = x2_normf<C>(x2_mulc<3, C>(A));
  const X2<C, SX_T> F = x2_sqr<C>(E);
  JacX<C> r;
  r.X = x2_normf<C>(x2_sub<C>(F, x2_mulc<2, C>(D)));
  r.Y = x2_normf<C>(x2_sub<C>(x2_mul<C>(E, x2_normf<C>(x2_sub<C>(D, r.X))), x2_mulc<2, C>(x2_normf<C>(x2_mulc<4, C>(Cc)))));
  r.Z = x2_normf<C>(x2_mulc<2, C>(x2_mul<C>(p.Y, p.Z)));
  r.inf = false;
  return r;
}

// p + q, q affine (madd-2007-bl), 56 units of NL^2 multiplier instructions
template <class C>
BGLS_HD JacX<C> jacx_madd(const JacX<C>& p, const AffX<C>& q) {
  if (q.inf) return p;
  if (p.inf) {
    JacX<C> r;
    r.X = x2_as<SX_F, C>(q.x);
    r.Y = x2_as<SX_F, C>(q.y);
    const Sx<C, SX_F> z = sx_as<SX_F, C>(ux_to_sx<C>(ux_zero<C>()));
    r.Z = {sx_as<SX_F, C>(sx_const<C>(C::RX_ONE)), z};
    r.inf = false;
    return r;
  }
  const X2<C, SX_T> Z1Z1 = x2_sqr<C>(p.Z);
  const X2<C, SX_T> U2 = x2_mul<C>(q.x, Z1Z1);
  const X2<C, SX_T> S2 = x2_mul<C>(x2_mul<C>(q.y, p.Z), Z1Z1);
  const auto Hd = x2_sub<C>(U2, p.X);
  const auto Rd = x2_sub<C>(S2, p.Y);
  if (x2_is_zero<C>(Hd)) {                                     // same x: P = Q (double) or P = -Q (infinity)
    if (x2_is_zero<C>(Rd)) return jacx_dbl<C>(p);
    return jacx_inf<C>();
  }
  const X2<C, SX_F> H = x2_normf<C>(Hd);
  const X2<C, SX_F> rr = x2_normf<C>(x2_mulc<2, C>(Rd));
  const X2<C, SX_T> HH = x2_sqr<C>(H);
  const X2<C, SX_F> I = x2_normf<C>(x2_mulc<4, C>(HH));
  const X2<C, SX_T> J = x2_mul<C>(H, I);
  const X2<C, SX_T> V = x2_mul<C>(p.X, I);
  JacX<C> r;
  r.X = x2_normf<C>(x2_sub<C>(x2_sub<C>(x2_sqr<C>(rr), J), x2_mulc<2, C>(V)));
  r.Y = x2_as<SX_F, C>(x2_mulsub<C>(rr, x2_normf<C>(x2_sub<C>(V, r.X)), x2_mulc<2, C>(p.Y), J));
  r.Z = x2_normf<C>(x2_sub<C>(x2_sub<C>(x2_sqr<C>(x2_normf<C>(x2_add<C>(p.Z, H))), Z1Z1), HH));
  r.inf = false;
  return r;
}

// wire bytes (x_im || x_re || y_im || y_re, big-endian; all zero = infinity) -> R' form; false when a coordinate is not canonical
template <class C>
BGLS_HD bool affx_from_bytes(AffX<C>& out, const uint8_t* b) {
  constexpr int N = C::FP_BYTES;
  const Fp<C> xi = fp_from_be<C>(b), xr = fp_from_be<C>(b + N), yi = fp_from_be<C>(b + 2 * N), yr = fp_from_be<C>(b + 3 * N);
  const bool ok = !fp_geq_p<C>(xi) && !fp_geq_p<C>(xr) && !fp_geq_p<C>(yi) && !fp_geq_p<C>(yr);
  out.inf = fp_is_zero<C>(xi) && fp_is_zero<C>(xr) && fp_is_zero<C>(yi) && fp_is_zero<C>(yr);
  out.x = {sx_from_plain<C>(xr), sx_from_plain<C>(xi)};
  out.y = {sx_from_plain<C>(yr), sx_from_plain<C>(yi)};
  return ok;
}
template <class C>
BGLS_HD AffX<C> affx_from_mont(const Aff<F2<C>>& a) {
  AffX<C> r;
  r.inf = a.inf;
  r.x = {sx_from_mont<C>(a.x.c0), sx_from_mont<C>(a.x.c1)};
  r.y = {sx_from_mont<C>(a.y.c0), sx_from_mont<C>(a.y.c1)};
  return r;
}
template <class C>
BGLS_HD Jac<F2<C>> jacx_to_mont(const JacX<C>& p) {
  Jac<F2<C>> r;
  if (p.inf) return jac_inf<F2<C>>();
  r.X = {sx_to_mont<C>(p.X.c0), sx_to_mont<C>(p.X.c1)};
  r.Y = {sx_to_mont<C>(p.Y.c0), sx_to_mont<C>(p.Y.c1)};
  r.Z = {sx_to_mont<C>(p.Z.c0), sx_to_mont<C>(p.Z.c1)};
  return r;
}

}  // namespace bgls
