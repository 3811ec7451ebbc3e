// Fp2 / Fp6 / Fp12 tower:  Fp2 = Fp[i]/(i^2+1), Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v).
// xi = 9+i (alt-bn128), 1+i (BLS12-381).  Fp2 convention matches the reference's own
// curves/complexNum.go:12-93 (i^2 = -1); the upper tower lives in the absent upstream
// libraries, so it is derived from the standard construction.
//
// GT "Add" in the reference's PointT interface (curves/altbn128.go:264-271,
// curves/bls12_381.go:160-168) is f12_mul here.
#pragma once
#include "fp.hpp"

namespace bgls {

template <class C>
struct Fp2 {
  Fp<C> c0, c1;
};
template <class C>
struct Fp6 {
  Fp2<C> a0, a1, a2;
};
template <class C>
struct Fp12 {
  Fp6<C> g, h;  // g + h*w
};

// ---------------------------------------------------------------- Fp2
template <class C>
BGLS_HD Fp2<C> f2_zero() {
  return {fp_zero<C>(), fp_zero<C>()};
}
template <class C>
BGLS_HD Fp2<C> f2_one() {
  return {fp_one<C>(), fp_zero<C>()};
}
template <class C>
BGLS_HD Fp2<C> f2_load(const u32* p) {  // re limbs then im limbs
  return {fp_load<C>(p), fp_load<C>(p + C::L)};
}
template <class C>
BGLS_HD bool f2_is_zero(const Fp2<C>& a) {
  return fp_is_zero<C>(a.c0) && fp_is_zero<C>(a.c1);
}
template <class C>
BGLS_HD bool f2_eq(const Fp2<C>& a, const Fp2<C>& b) {
  return fp_eq<C>(a.c0, b.c0) && fp_eq<C>(a.c1, b.c1);
}
template <class C>
BGLS_HD Fp2<C> f2_select(bool c, const Fp2<C>& a, const Fp2<C>& b) {
  return {fp_select<C>(c, a.c0, b.c0), fp_select<C>(c, a.c1, b.c1)};
}
template <class C>
BGLS_HD Fp2<C> f2_add(const Fp2<C>& a, const Fp2<C>& b) {
  return {fp_add<C>(a.c0, b.c0), fp_add<C>(a.c1, b.c1)};
}
template <class C>
BGLS_HD Fp2<C> f2_sub(const Fp2<C>& a, const Fp2<C>& b) {
  return {fp_sub<C>(a.c0, b.c0), fp_sub<C>(a.c1, b.c1)};
}
template <class C>
BGLS_HD Fp2<C> f2_neg(const Fp2<C>& a) {
  return {fp_neg<C>(a.c0), fp_neg<C>(a.c1)};
}
template <class C>
BGLS_HD Fp2<C> f2_conj(const Fp2<C>& a) {
  return {a.c0, fp_neg<C>(a.c1)};
}
template <class C>
BGLS_HD Fp2<C> f2_dbl(const Fp2<C>& a) {
  return {fp_dbl<C>(a.c0), fp_dbl<C>(a.c1)};
}
template <class C>
BGLS_HD Fp2<C> f2_mul3(const Fp2<C>& a) {
  return {fp_mul3<C>(a.c0), fp_mul3<C>(a.c1)};
}
template <class C>
BGLS_HD Fp2<C> f2_muls(const Fp2<C>& a, const Fp<C>& s) {  // by an Fp scalar
  return {fp_mul<C>(a.c0, s), fp_mul<C>(a.c1, s)};
}

// Karatsuba with lazy reduction: 3 wide products, 2 Montgomery reductions.
// Bounds: (a0+a1)(b0+b1) < 4p^2 < 2^(64L); both reduced inputs < 2p^2 < p*2^(32L).
template <class C>
BGLS_HD Fp2<C> f2_mul_inl(const Fp2<C>& a, const Fp2<C>& b) {
  constexpr int W = 2 * C::L;
  u32 v0[W], v1[W], s[W];
  mul_wide<C>(v0, a.c0.v, b.c0.v);
  mul_wide<C>(v1, a.c1.v, b.c1.v);
  Fp<C> sa = fp_add_nr<C>(a.c0, a.c1);
  Fp<C> sb = fp_add_nr<C>(b.c0, b.c1);
  mul_wide<C>(s, sa.v, sb.v);
  w_sub<W>(s, s, v0);
  w_sub<W>(s, s, v1);  // s = a0 b1 + a1 b0
  w_add<W>(v0, v0, C::P2W);
  w_sub<W>(v0, v0, v1);  // v0 = a0 b0 - a1 b1 + p^2  in (0, 2p^2)
  Fp2<C> r;
  r.c0 = redc<C>(v0);
  r.c1 = redc<C>(s);
  return r;
}

template <class C>
BGLS_HD Fp2<C> f2_sqr_inl(const Fp2<C>& a) {
  constexpr int W = 2 * C::L;
  Fp<C> s = fp_add_nr<C>(a.c0, a.c1);
  Fp<C> d = fp_sub<C>(a.c0, a.c1);
  u32 t0[W], t1[W];
  mul_wide<C>(t0, s.v, d.v);        // (a0+a1)(a0-a1) < 2p^2
  mul_wide<C>(t1, a.c0.v, a.c1.v);  // a0 a1 < p^2
  w_add<W>(t1, t1, t1);
  Fp2<C> r;
  r.c0 = redc<C>(t0);
  r.c1 = redc<C>(t1);
  return r;
}

// a b - c d with ONE Montgomery reduction per component: six wide products, two reductions instead of four.  The wide
// differences carry a bias of 3 p^2 and stay below 5 p^2 < 2 p 2^(32L) on both curves.
template <class C>
BGLS_HD Fp2<C> f2_mulsub_inl(const Fp2<C>& a, const Fp2<C>& b, const Fp2<C>& c, const Fp2<C>& d) {
  constexpr int W = 2 * C::L;
  u32 re[W], im[W], t[W], u[W];
  mul_wide<C>(re, a.c0.v, b.c0.v);
  mul_wide<C>(t, a.c1.v, b.c1.v);
  {
    const Fp<C> sa = fp_add_nr<C>(a.c0, a.c1), sb = fp_add_nr<C>(b.c0, b.c1);
    mul_wide<C>(im, sa.v, sb.v);
  }
  w_sub<W>(im, im, re);
  w_sub<W>(im, im, t);                 // a0 b1 + a1 b0                     in [0, 2 p^2)
  w_add<W>(re, re, C::P2W3);
  w_sub<W>(re, re, t);                 // a0 b0 - a1 b1 + 3 p^2             in (2 p^2, 4 p^2)
  mul_wide<C>(t, c.c0.v, d.c0.v);
  w_sub<W>(re, re, t);
  mul_wide<C>(u, c.c1.v, d.c1.v);
  w_add<W>(re, re, u);                 // ... - c0 d0 + c1 d1               in (p^2, 5 p^2)
  w_add<W>(t, t, u);                   // c0 d0 + c1 d1
  {
    const Fp<C> sc = fp_add_nr<C>(c.c0, c.c1), sd = fp_add_nr<C>(d.c0, d.c1);
    mul_wide<C>(u, sc.v, sd.v);
  }
  w_sub<W>(u, u, t);                   // c0 d1 + c1 d0                     in [0, 2 p^2)
  w_add<W>(im, im, C::P2W3);
  w_sub<W>(im, im, u);                 // imaginary part + 3 p^2            in (p^2, 5 p^2)
  Fp2<C> r;
  r.c0 = redc_k<C, 2>(re);
  r.c1 = redc_k<C, 2>(im);
  return r;
}
// g^2 - 3 e^2, same idea (bias 6 p^2, values below 8 p^2 < 2 p 2^(32L))
template <class C>
BGLS_HD Fp2<C> f2_sqrsub3_inl(const Fp2<C>& g, const Fp2<C>& e) {
  constexpr int W = 2 * C::L;
  u32 re[W], im[W], t[W], t3[W];
  {
    const Fp<C> s = fp_add_nr<C>(g.c0, g.c1), d = fp_sub<C>(g.c0, g.c1);
    mul_wide<C>(re, s.v, d.v);         // g0^2 - g1^2 (as (g0+g1)(g0-g1))   in [0, 2 p^2)
  }
  mul_wide<C>(im, g.c0.v, g.c1.v);
  w_add<W>(im, im, im);                // 2 g0 g1                           in [0, 2 p^2)
  {
    const Fp<C> s = fp_add_nr<C>(e.c0, e.c1), d = fp_sub<C>(e.c0, e.c1);
    mul_wide<C>(t, s.v, d.v);
  }
  w_add<W>(t3, t, t);
  w_add<W>(t3, t3, t);                 // 3 (e0^2 - e1^2)                   in [0, 6 p^2)
  w_add<W>(re, re, C::P2W6);
  w_sub<W>(re, re, t3);
  mul_wide<C>(t, e.c0.v, e.c1.v);
  w_add<W>(t3, t, t);
  w_add<W>(t3, t3, t3);
  w_add<W>(t3, t3, t);
  w_add<W>(t3, t3, t);                 // 6 e0 e1                           in [0, 6 p^2)
  w_add<W>(im, im, C::P2W6);
  w_sub<W>(im, im, t3);
  Fp2<C> r;
  r.c0 = redc_k<C, 2>(re);
  r.c1 = redc_k<C, 2>(im);
  return r;
}

template <class C>
BGLS_FN Fp2<C> f2_mul(const Fp2<C>& a, const Fp2<C>& b) {
  return f2_mul_inl<C>(a, b);
}
template <class C>
BGLS_FN Fp2<C> f2_sqr(const Fp2<C>& a) {
  return f2_sqr_inl<C>(a);
}

// multiply by the sextic non-residue xi = XI_RE + i
template <class C>
BGLS_HD Fp2<C> f2_mulxi(const Fp2<C>& a) {
  if constexpr (C::XI_RE == 1) {
    return {fp_sub<C>(a.c0, a.c1), fp_add<C>(a.c0, a.c1)};
  } else {  // 9 + i
    Fp<C> a8 = fp_dbl<C>(fp_dbl<C>(fp_dbl<C>(a.c0)));
    Fp<C> b8 = fp_dbl<C>(fp_dbl<C>(fp_dbl<C>(a.c1)));
    Fp<C> a9 = fp_add<C>(a8, a.c0);
    Fp<C> b9 = fp_add<C>(b8, a.c1);
    return {fp_sub<C>(a9, a.c1), fp_add<C>(b9, a.c0)};
  }
}

template <class C>
BGLS_FN Fp2<C> f2_inv(const Fp2<C>& a) {
  Fp<C> n = fp_add<C>(fp_sqr<C>(a.c0), fp_sqr<C>(a.c1));
  Fp<C> ni = fp_inv<C>(n);
  return {fp_mul<C>(a.c0, ni), fp_neg<C>(fp_mul<C>(a.c1, ni))};
}

// ---------------------------------------------------------------- Fp6
template <class C>
BGLS_HD Fp6<C> f6_zero() {
  return {f2_zero<C>(), f2_zero<C>(), f2_zero<C>()};
}
template <class C>
BGLS_HD Fp6<C> f6_one() {
  return {f2_one<C>(), f2_zero<C>(), f2_zero<C>()};
}
template <class C>
BGLS_HD Fp6<C> f6_add(const Fp6<C>& a, const Fp6<C>& b) {
  return {f2_add<C>(a.a0, b.a0), f2_add<C>(a.a1, b.a1), f2_add<C>(a.a2, b.a2)};
}
template <class C>
BGLS_HD Fp6<C> f6_sub(const Fp6<C>& a, const Fp6<C>& b) {
  return {f2_sub<C>(a.a0, b.a0), f2_sub<C>(a.a1, b.a1), f2_sub<C>(a.a2, b.a2)};
}
template <class C>
BGLS_HD Fp6<C> f6_neg(const Fp6<C>& a) {
  return {f2_neg<C>(a.a0), f2_neg<C>(a.a1), f2_neg<C>(a.a2)};
}
template <class C>
BGLS_HD Fp6<C> f6_mulv(const Fp6<C>& a) {  // * v
  return {f2_mulxi<C>(a.a2), a.a0, a.a1};
}
template <class C>
BGLS_HD bool f6_eq(const Fp6<C>& a, const Fp6<C>& b) {
  return f2_eq<C>(a.a0, b.a0) && f2_eq<C>(a.a1, b.a1) && f2_eq<C>(a.a2, b.a2);
}

template <class C>
BGLS_FN Fp6<C> f6_mul(const Fp6<C>& a, const Fp6<C>& b) {
  Fp2<C> t0 = f2_mul<C>(a.a0, b.a0);
  Fp2<C> t1 = f2_mul<C>(a.a1, b.a1);
  Fp2<C> t2 = f2_mul<C>(a.a2, b.a2);
  Fp2<C> c0 = f2_mul<C>(f2_add<C>(a.a1, a.a2), f2_add<C>(b.a1, b.a2));
  c0 = f2_add<C>(t0, f2_mulxi<C>(f2_sub<C>(f2_sub<C>(c0, t1), t2)));
  Fp2<C> c1 = f2_mul<C>(f2_add<C>(a.a0, a.a1), f2_add<C>(b.a0, b.a1));
  c1 = f2_add<C>(f2_sub<C>(f2_sub<C>(c1, t0), t1), f2_mulxi<C>(t2));
  Fp2<C> c2 = f2_mul<C>(f2_add<C>(a.a0, a.a2), f2_add<C>(b.a0, b.a2));
  c2 = f2_add<C>(f2_sub<C>(f2_sub<C>(c2, t0), t2), t1);
  return {c0, c1, c2};
}

template <class C>
BGLS_FN Fp6<C> f6_sqr(const Fp6<C>& a) {  // CH-SQR2
  Fp2<C> s0 = f2_sqr<C>(a.a0);
  Fp2<C> s1 = f2_dbl<C>(f2_mul<C>(a.a0, a.a1));
  Fp2<C> s2 = f2_sqr<C>(f2_add<C>(f2_sub<C>(a.a0, a.a1), a.a2));
  Fp2<C> s3 = f2_dbl<C>(f2_mul<C>(a.a1, a.a2));
  Fp2<C> s4 = f2_sqr<C>(a.a2);
  Fp2<C> c0 = f2_add<C>(s0, f2_mulxi<C>(s3));
  Fp2<C> c1 = f2_add<C>(s1, f2_mulxi<C>(s4));
  Fp2<C> c2 = f2_sub<C>(f2_sub<C>(f2_add<C>(f2_add<C>(s1, s2), s3), s0), s4);
  return {c0, c1, c2};
}

// a * (x0 + x1 v)
template <class C>
BGLS_FN Fp6<C> f6_mul_by_01(const Fp6<C>& a, const Fp2<C>& x0, const Fp2<C>& x1) {
  Fp2<C> t0 = f2_mul<C>(a.a0, x0);
  Fp2<C> t1 = f2_mul<C>(a.a1, x1);
  Fp2<C> c0 = f2_sub<C>(f2_mul<C>(f2_add<C>(a.a1, a.a2), x1), t1);
  c0 = f2_add<C>(f2_mulxi<C>(c0), t0);
  Fp2<C> c1 = f2_mul<C>(f2_add<C>(a.a0, a.a1), f2_add<C>(x0, x1));
  c1 = f2_sub<C>(f2_sub<C>(c1, t0), t1);
  Fp2<C> c2 = f2_mul<C>(f2_add<C>(a.a0, a.a2), x0);
  c2 = f2_add<C>(f2_sub<C>(c2, t0), t1);
  return {c0, c1, c2};
}
// a * x0
template <class C>
BGLS_HD Fp6<C> f6_mul_by_0(const Fp6<C>& a, const Fp2<C>& x0) {
  return {f2_mul<C>(a.a0, x0), f2_mul<C>(a.a1, x0), f2_mul<C>(a.a2, x0)};
}
// a * (x1 v)
template <class C>
BGLS_HD Fp6<C> f6_mul_by_1(const Fp6<C>& a, const Fp2<C>& x1) {
  return {f2_mulxi<C>(f2_mul<C>(a.a2, x1)), f2_mul<C>(a.a0, x1), f2_mul<C>(a.a1, x1)};
}

template <class C>
BGLS_FN Fp6<C> f6_inv(const Fp6<C>& a) {
  Fp2<C> t0 = f2_sub<C>(f2_sqr<C>(a.a0), f2_mulxi<C>(f2_mul<C>(a.a1, a.a2)));
  Fp2<C> t1 = f2_sub<C>(f2_mulxi<C>(f2_sqr<C>(a.a2)), f2_mul<C>(a.a0, a.a1));
  Fp2<C> t2 = f2_sub<C>(f2_sqr<C>(a.a1), f2_mul<C>(a.a0, a.a2));
  Fp2<C> d = f2_add<C>(f2_mul<C>(a.a0, t0),
                       f2_mulxi<C>(f2_add<C>(f2_mul<C>(a.a2, t1), f2_mul<C>(a.a1, t2))));
  Fp2<C> di = f2_inv<C>(d);
  return {f2_mul<C>(t0, di), f2_mul<C>(t1, di), f2_mul<C>(t2, di)};
}

// ---------------------------------------------------------------- Fp12
template <class C>
BGLS_HD Fp12<C> f12_one() {
  return {f6_one<C>(), f6_zero<C>()};
}
template <class C>
BGLS_HD bool f12_eq(const Fp12<C>& a, const Fp12<C>& b) {
  return f6_eq<C>(a.g, b.g) && f6_eq<C>(a.h, b.h);
}
template <class C>
BGLS_HD bool f12_is_one(const Fp12<C>& a) {
  return f12_eq<C>(a, f12_one<C>());
}
template <class C>
BGLS_HD Fp12<C> f12_conj(const Fp12<C>& a) {  // a^(p^6)
  return {a.g, f6_neg<C>(a.h)};
}

template <class C>
BGLS_FN Fp12<C> f12_mul(const Fp12<C>& a, const Fp12<C>& b) {
  Fp6<C> t0 = f6_mul<C>(a.g, b.g);
  Fp6<C> t1 = f6_mul<C>(a.h, b.h);
  Fp6<C> c1 = f6_mul<C>(f6_add<C>(a.g, a.h), f6_add<C>(b.g, b.h));
  c1 = f6_sub<C>(f6_sub<C>(c1, t0), t1);
  return {f6_add<C>(t0, f6_mulv<C>(t1)), c1};
}

template <class C>
BGLS_FN Fp12<C> f12_sqr(const Fp12<C>& a) {  // complex squaring, 2 Fp6 products
  Fp6<C> gh = f6_mul<C>(a.g, a.h);
  Fp6<C> t = f6_mul<C>(f6_add<C>(a.g, a.h), f6_add<C>(a.g, f6_mulv<C>(a.h)));
  t = f6_sub<C>(f6_sub<C>(t, gh), f6_mulv<C>(gh));
  return {t, f6_add<C>(gh, gh)};
}

template <class C>
BGLS_FN Fp12<C> f12_inv(const Fp12<C>& a) {
  Fp6<C> d = f6_sub<C>(f6_sqr<C>(a.g), f6_mulv<C>(f6_sqr<C>(a.h)));
  Fp6<C> di = f6_inv<C>(d);
  return {f6_mul<C>(a.g, di), f6_neg<C>(f6_mul<C>(a.h, di))};
}

// Sparse line multiplication.  A Miller line is  l = e0 + e1 w + e3 w^3  (D-type twist, alt-bn128)
// or  l = e0 + e2 w^2 + e3 w^3  (M-type, BLS12-381); with w^2 = v these are
//   D:  (e0, 0, 0) + (e1, e3, 0) w          M:  (e0, e2, 0) + (0, e3, 0) w .
// 13 Fp2 products instead of 18.
template <class C>
BGLS_FN Fp12<C> f12_mul_line(const Fp12<C>& f, const Fp2<C>& ea, const Fp2<C>& eb, const Fp2<C>& ec) {
  if constexpr (C::TWIST_D) {  // ea=e0, eb=e1, ec=e3
    Fp6<C> a = f6_mul_by_0<C>(f.g, ea);
    Fp6<C> b = f6_mul_by_01<C>(f.h, eb, ec);
    Fp6<C> e = f6_mul_by_01<C>(f6_add<C>(f.g, f.h), f2_add<C>(ea, eb), ec);
    return {f6_add<C>(a, f6_mulv<C>(b)), f6_sub<C>(f6_sub<C>(e, a), b)};
  } else {  // ea=e0, eb=e2, ec=e3
    Fp6<C> a = f6_mul_by_01<C>(f.g, ea, eb);
    Fp6<C> b = f6_mul_by_1<C>(f.h, ec);
    Fp6<C> e = f6_mul_by_01<C>(f6_add<C>(f.g, f.h), ea, f2_add<C>(eb, ec));
    return {f6_add<C>(a, f6_mulv<C>(b)), f6_sub<C>(f6_sub<C>(e, a), b)};
  }
}

// Frobenius a^(p^j), j in 1..3: w-basis coefficient k is conj^j(e_k) * GAMMA[j-1][k].
template <class C>
BGLS_HD Fp2<C> gamma_const(int j, int k) {
  return f2_load<C>(C::GAMMA + ((j - 1) * 6 + k) * 2 * C::L);
}
template <class C>
BGLS_FN Fp12<C> f12_frob(const Fp12<C>& a, int j) {
  Fp2<C> e[6] = {a.g.a0, a.h.a0, a.g.a1, a.h.a1, a.g.a2, a.h.a2};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    Fp2<C> x = (j & 1) ? f2_conj<C>(e[k]) : e[k];
    e[k] = (k == 0) ? x : f2_mul<C>(x, gamma_const<C>(j, k));
  }
  return {{e[0], e[2], e[4]}, {e[1], e[3], e[5]}};
}

// Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part of the
// final exponentiation); same value as f12_sqr there.  9 Fp2 squarings.
template <class C>
BGLS_HD void fp4_sqr(Fp2<C>& c0, Fp2<C>& c1, const Fp2<C>& a, const Fp2<C>& b) {
  Fp2<C> t0 = f2_sqr<C>(a);
  Fp2<C> t1 = f2_sqr<C>(b);
  c0 = f2_add<C>(f2_mulxi<C>(t1), t0);
  c1 = f2_sub<C>(f2_sub<C>(f2_sqr<C>(f2_add<C>(a, b)), t0), t1);
}
template <class C>
BGLS_FN Fp12<C> f12_cyclo_sqr(const Fp12<C>& f) {
  Fp2<C> z0 = f.g.a0, z4 = f.g.a1, z3 = f.g.a2, z2 = f.h.a0, z1 = f.h.a1, z5 = f.h.a2;
  Fp2<C> t0, t1, t2, t3;
  fp4_sqr<C>(t0, t1, z0, z1);
  z0 = f2_add<C>(f2_dbl<C>(f2_sub<C>(t0, z0)), t0);
  z1 = f2_add<C>(f2_dbl<C>(f2_add<C>(t1, z1)), t1);
  fp4_sqr<C>(t0, t1, z2, z3);
  fp4_sqr<C>(t2, t3, z4, z5);
  z4 = f2_add<C>(f2_dbl<C>(f2_sub<C>(t0, z4)), t0);
  z5 = f2_add<C>(f2_dbl<C>(f2_add<C>(t1, z5)), t1);
  t0 = f2_mulxi<C>(t3);
  z2 = f2_add<C>(f2_dbl<C>(f2_add<C>(t0, z2)), t0);
  z3 = f2_add<C>(f2_dbl<C>(f2_sub<C>(t2, z3)), t2);
  return {{z0, z4, z3}, {z2, z1, z5}};
}

}  // namespace bgls
