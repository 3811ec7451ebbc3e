// G1 / G2 group arithmetic (short Weierstrass, a = 0), generic over the coordinate field.
//
// Reference operations served: Point.Add (curves/altbn128.go:59-66,181-188;
// curves/bls12_381.go:33-41,94-102), Point.Mul incl. Mul(-1) negation (altbn128.go:107-128,
// 235-249; bls12_381.go:65-83,126-137), AggregatePoints (curves/curve.go:73-121),
// ScalePoints (curves/curve.go:190-214), the G1 cofactor multiplication inside BLS12-381
// hash-to-G1 (curves/hash.go:91).  The upstream libraries' internals are absent, so the
// formulas are the standard Jacobian ones (dbl-2009-l, madd-2007-bl, add-2007-bl) with the
// exceptional cases handled explicitly: affine results are unique, hence bit-identical.
#pragma once
#include "tower.hpp"

namespace bgls {

template <class C>
struct F1 {  // coordinate field of G1
  typedef Fp<C> T;
  static BGLS_HD T zero() { return fp_zero<C>(); }
  static BGLS_HD T one() { return fp_one<C>(); }
  static BGLS_HD T add(const T& a, const T& b) { return fp_add<C>(a, b); }
  static BGLS_HD T sub(const T& a, const T& b) { return fp_sub<C>(a, b); }
  static BGLS_HD T neg(const T& a) { return fp_neg<C>(a); }
  static BGLS_HD T dbl(const T& a) { return fp_dbl<C>(a); }
  static BGLS_HD T mul(const T& a, const T& b) { return fp_mul<C>(a, b); }
  static BGLS_HD T sqr(const T& a) { return fp_sqr<C>(a); }
  static BGLS_HD T mul_inl(const T& a, const T& b) { return fp_mul_inl<C>(a, b); }      // expanded in place (hot loops)
  static BGLS_HD T sqr_inl(const T& a) { return fp_sqr_inl<C>(a); }
  static BGLS_HD T inv(const T& a) { return fp_inv<C>(a); }
  static BGLS_HD bool is_zero(const T& a) { return fp_is_zero<C>(a); }
  static BGLS_HD bool eq(const T& a, const T& b) { return fp_eq<C>(a, b); }
  static BGLS_HD T select(bool c, const T& a, const T& b) { return fp_select<C>(c, a, b); }
  static BGLS_HD T curve_b() { return fp_load<C>(C::B); }
  static constexpr int NFP = 1;
};

template <class C>
struct F2 {  // coordinate field of G2 (the twist)
  typedef Fp2<C> T;
  static BGLS_HD T zero() { return f2_zero<C>(); }
  static BGLS_HD T one() { return f2_one<C>(); }
  static BGLS_HD T add(const T& a, const T& b) { return f2_add<C>(a, b); }
  static BGLS_HD T sub(const T& a, const T& b) { return f2_sub<C>(a, b); }
  static BGLS_HD T neg(const T& a) { return f2_neg<C>(a); }
  static BGLS_HD T dbl(const T& a) { return f2_dbl<C>(a); }
  static BGLS_HD T mul(const T& a, const T& b) { return f2_mul<C>(a, b); }
  static BGLS_HD T sqr(const T& a) { return f2_sqr<C>(a); }
  static BGLS_HD T mul_inl(const T& a, const T& b) { return f2_mul_inl<C>(a, b); }
  static BGLS_HD T sqr_inl(const T& a) { return f2_sqr_inl<C>(a); }
  static BGLS_HD T inv(const T& a) { return f2_inv<C>(a); }
  static BGLS_HD bool is_zero(const T& a) { return f2_is_zero<C>(a); }
  static BGLS_HD bool eq(const T& a, const T& b) { return f2_eq<C>(a, b); }
  static BGLS_HD T select(bool c, const T& a, const T& b) { return f2_select<C>(c, a, b); }
  static BGLS_HD T curve_b() { return {fp_load<C>(C::B2_RE), fp_load<C>(C::B2_IM)}; }
  static constexpr int NFP = 2;
};

template <class F>
struct Aff {
  typename F::T x, y;
  bool inf;
};
template <class F>
struct Jac {
  typename F::T X, Y, Z;  // Z == 0  <=>  infinity
};

template <class F>
BGLS_HD Jac<F> jac_inf() {
  return {F::one(), F::one(), F::zero()};
}
template <class F>
BGLS_HD Jac<F> jac_from_aff(const Aff<F>& a) {
  if (a.inf) return jac_inf<F>();
  return {a.x, a.y, F::one()};
}
template <class F>
BGLS_HD bool jac_is_inf(const Jac<F>& p) {
  return F::is_zero(p.Z);
}

template <class F>
BGLS_HD bool aff_on_curve(const Aff<F>& a) {
  if (a.inf) return true;
  typename F::T l = F::sqr(a.y);
  typename F::T r = F::add(F::mul(F::sqr(a.x), a.x), F::curve_b());
  return F::eq(l, r);
}

template <class F>
BGLS_FN Jac<F> jac_dbl(const Jac<F>& p) {
  typedef typename F::T T;
  T A = F::sqr(p.X);
  T B = F::sqr(p.Y);
  T Cc = F::sqr(B);
  T D = F::sub(F::sub(F::sqr(F::add(p.X, B)), A), Cc);
  D = F::dbl(D);
  T E = F::add(F::dbl(A), A);
  T Fv = F::sqr(E);
  T X3 = F::sub(Fv, F::dbl(D));
  T C8 = F::dbl(F::dbl(F::dbl(Cc)));
  T Y3 = F::sub(F::mul(E, F::sub(D, X3)), C8);
  T Z3 = F::dbl(F::mul(p.Y, p.Z));
  return {X3, Y3, Z3};
}

// Jacobian + affine
template <class F>
BGLS_FN Jac<F> jac_add_aff(const Jac<F>& p, const Aff<F>& q) {
  typedef typename F::T T;
  if (q.inf) return p;
  if (jac_is_inf<F>(p)) return {q.x, q.y, F::one()};
  T Z1Z1 = F::sqr(p.Z);
  T U2 = F::mul(q.x, Z1Z1);
  T S2 = F::mul(F::mul(q.y, p.Z), Z1Z1);
  T H = F::sub(U2, p.X);
  T rr = F::sub(S2, p.Y);
  if (F::is_zero(H)) {
    if (F::is_zero(rr)) return jac_dbl<F>(p);
    return jac_inf<F>();
  }
  rr = F::dbl(rr);
  T HH = F::sqr(H);
  T I = F::dbl(F::dbl(HH));
  T J = F::mul(H, I);
  T V = F::mul(p.X, I);
  T X3 = F::sub(F::sub(F::sqr(rr), J), F::dbl(V));
  T Y3 = F::sub(F::mul(rr, F::sub(V, X3)), F::dbl(F::mul(p.Y, J)));
  T Z3 = F::sub(F::sub(F::sqr(F::add(p.Z, H)), Z1Z1), HH);
  return {X3, Y3, Z3};
}

// Jacobian + Jacobian
template <class F>
BGLS_FN Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
  typedef typename F::T T;
  if (jac_is_inf<F>(p)) return q;
  if (jac_is_inf<F>(q)) return p;
  T Z1Z1 = F::sqr(p.Z);
  T Z2Z2 = F::sqr(q.Z);
  T U1 = F::mul(p.X, Z2Z2);
  T U2 = F::mul(q.X, Z1Z1);
  T S1 = F::mul(F::mul(p.Y, q.Z), Z2Z2);
  T S2 = F::mul(F::mul(q.Y, p.Z), Z1Z1);
  T H = F::sub(U2, U1);
  T rr = F::sub(S2, S1);
  if (F::is_zero(H)) {
    if (F::is_zero(rr)) return jac_dbl<F>(p);
    return jac_inf<F>();
  }
  rr = F::dbl(rr);
  T I = F::sqr(F::dbl(H));
  T J = F::mul(H, I);
  T V = F::mul(U1, I);
  T X3 = F::sub(F::sub(F::sqr(rr), J), F::dbl(V));
  T Y3 = F::sub(F::mul(rr, F::sub(V, X3)), F::dbl(F::mul(S1, J)));
  T Z3 = F::mul(F::sub(F::sub(F::sqr(F::add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
  return {X3, Y3, Z3};
}

template <class F>
BGLS_FN Aff<F> jac_to_aff(const Jac<F>& p) {
  typedef typename F::T T;
  if (jac_is_inf<F>(p)) return {F::zero(), F::zero(), true};
  T zi = F::inv(p.Z);
  T zi2 = F::sqr(zi);
  return {F::mul(p.X, zi2), F::mul(F::mul(p.Y, zi2), zi), false};
}

template <class F>
BGLS_HD Aff<F> aff_neg(const Aff<F>& a) {
  return {a.x, F::neg(a.y), a.inf};
}

// k * P, k given as NL little-endian u32 limbs (non-negative); MSB-first double-and-add.
template <class F>
BGLS_FN Jac<F> jac_mul(const Aff<F>& p, const u32* k, int nbits) {
  Jac<F> r = jac_inf<F>();
  for (int i = nbits - 1; i >= 0; --i) {
    r = jac_dbl<F>(r);
    if ((k[i >> 5] >> (i & 31)) & 1u) r = jac_add_aff<F>(r, p);
  }
  return r;
}

// k * P for a per-lane scalar of up to 256 bits (Point.Mul at the seam, curves/curve.go:190-214: the reference hands the
// scalar to the curve library): signed radix-16 digits (Booth recoding: d_i = -8 k_{4i+3} + 4 k_{4i+2} + 2 k_{4i+1} + k_{4i} +
// k_{4i-1}, in [-8, 8]) against a table P .. 8P.  Every window is four doublings and ONE general addition whatever the digit,
// which is what a wave wants: the lanes of a wave hold different scalars and execute the union of their paths, so a sparse
// recoding (NAF: fewer additions per scalar) buys nothing -- some lane adds at every position, and three kinds of addition
// side by side cost 3.6x (measured: a width-4 NAF ran 2x slower than double-and-add) -- while the fixed schedule costs
// 64 x (4 x 7 + 16) field products against 256 x (7 + 11) for double-and-add.  Same point as jac_mul (the group law does not
// care about the chain), so every golden vector stands.
template <class F>
BGLS_FN Jac<F> jac_mul_w4(const Aff<F>& p, const u32* k, int nbits) {
  if (nbits <= 8 || p.inf) return jac_mul<F>(p, k, nbits);
  u32 w[9];
  const int nl = (nbits + 31) >> 5;
#pragma unroll
  for (int j = 0; j < 9; ++j) w[j] = j < nl && j < 8 ? k[j] : 0u;
  if (nbits & 31) w[nl - 1] &= (1u << (nbits & 31)) - 1u;
  Jac<F> tab[8];                                   // tab[a - 1] = a P
  tab[0] = jac_from_aff<F>(p);
  tab[1] = jac_dbl<F>(tab[0]);
  tab[2] = jac_add_aff<F>(tab[1], p);
  tab[3] = jac_dbl<F>(tab[1]);
  tab[4] = jac_add_aff<F>(tab[3], p);
  tab[5] = jac_dbl<F>(tab[2]);
  tab[6] = jac_add_aff<F>(tab[5], p);
  tab[7] = jac_dbl<F>(tab[3]);
  const int nw = (nbits + 4) >> 2;                 // one bit above the scalar: the top window's sign bit is clear
  Jac<F> r = jac_inf<F>();
  for (int i = nw - 1; i >= 0; --i) {
    if (i != nw - 1) {
      r = jac_dbl<F>(r);
      r = jac_dbl<F>(r);
      r = jac_dbl<F>(r);
      r = jac_dbl<F>(r);
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
      Jac<F> q = tab[a - 1];
      if (val < 0) q.Y = F::neg(q.Y);
      r = jac_add<F>(r, q);
    }
  }
  return r;
}

// ---- G2 subgroup membership ------------------------------------------------------------------------------------------
// The reference validates G2 inputs when a Point is constructed: alt-bn128 MakeG2Point / UnmarshalG2
// (curves/altbn128.go:157-179,329-376) reach upstream bn256's G2.Unmarshal, which rejects twist points outside the
// order-r subgroup, BLS12-381 has Check() (curves/bls12_381.go:242-264).  A twist point outside G2 must never reach the
// Miller loop: the pairing is not bilinear there.  Membership is decided with the endomorphism psi = twist o Frobenius o
// untwist (psi acts on G2 as multiplication by p), one 63/64-bit scalar multiplication instead of a 254/255-bit one:
//     alt-bn128   [u+1]Q + psi([u]Q) + psi^2([u]Q) = psi^3([2u]Q)
//     BLS12-381   psi(Q) = [x]Q
// Both accept EXACTLY the points with [r]Q = infinity: oracle/pyref/subgroup.py proves it for these two curves by
// checking every prime-order component of the cofactor part of E'(Fp2) (tests/test_oracle.py).
template <class C>
BGLS_HD Jac<F2<C>> g2_psi(const Jac<F2<C>>& q) {        // conjugation is a field automorphism: acts coordinate-wise, Z included
  const Fp2<C> px = f2_load<C>(C::PSI_X), py = f2_load<C>(C::PSI_Y);
  return {f2_mul<C>(f2_conj<C>(q.X), px), f2_mul<C>(f2_conj<C>(q.Y), py), f2_conj<C>(q.Z)};
}
template <class F>
BGLS_FN bool jac_eq(const Jac<F>& a, const Jac<F>& b) {
  const bool ai = jac_is_inf<F>(a), bi = jac_is_inf<F>(b);
  if (ai || bi) return ai && bi;
  typedef typename F::T T;
  const T za = F::sqr(a.Z), zb = F::sqr(b.Z);
  if (!F::eq(F::mul(a.X, zb), F::mul(b.X, za))) return false;
  return F::eq(F::mul(a.Y, F::mul(zb, b.Z)), F::mul(b.Y, F::mul(za, a.Z)));
}
// q: on the twist (caller checked).  Exact also for points of small order: the group law below handles P = +-Q and
// infinity explicitly.
template <class C>
BGLS_FN bool g2_in_subgroup(const Aff<F2<C>>& q) {
  typedef F2<C> F;
  if (q.inf) return true;
  Jac<F> xq = jac_mul<F>(q, C::U_ABS, C::U_BITS);
  if constexpr (C::CURVE_ID == 0) {
    const Jac<F> p1 = g2_psi<C>(xq), p2 = g2_psi<C>(p1);
    const Jac<F> lhs = jac_add<F>(jac_add<F>(jac_add_aff<F>(xq, q), p1), p2);
    const Jac<F> rhs = g2_psi<C>(g2_psi<C>(g2_psi<C>(jac_dbl<F>(xq))));
    return jac_eq<F>(lhs, rhs);
  } else {
    xq.Y = f2_neg<C>(xq.Y);                              // x < 0
    return jac_eq<F>(g2_psi<C>(jac_from_aff<F>(q)), xq);
  }
}

// G1 membership.  alt-bn128's G1 is the whole curve (cofactor 1).  BLS12-381's E(Fp) has cofactor (x-1)^2/3 and the
// reference validates G1 points exactly like G2 points when they are constructed (MakeG1Point with check, UnmarshalG1:
// curves/bls12_381.go:196-264 -> pt.Check()): a point sigma + T with T of cofactor order must not become a signature
// (it would verify like sigma: malleability, and scalars reduced mod r would act wrongly on it).  Decided by the
// definition, [r]P = infinity -- construction-time work, one 255-bit scalar multiplication in Fp.
template <class C>
BGLS_FN bool g1_in_subgroup(const Aff<F1<C>>& p) {
  if constexpr (C::CURVE_ID == 0) {
    return true;
  } else {
    if (p.inf) return true;
    return jac_is_inf<F1<C>>(jac_mul_w4<F1<C>>(p, C::ORDER, 255));
  }
}

// ---- (de)serialisation of affine points at the seam (uncompressed wire formats) ----
//  G1: x || y big-endian (curves/altbn128.go:42-57; bls12G1Hash.dat)
//  G2: x_im || x_re || y_im || y_re (curves/altbn128.go:157-179, altbn128_test.go:26-38;
//      curves/bls12_381.go:147-158,209-226)
//  infinity = all-zero bytes (curves/altbn128.go:431-439)
// Returns false when a coordinate is not a canonical field element (>= p).
template <class C>
BGLS_HD bool g1_from_bytes(Aff<F1<C>>& out, const uint8_t* b) {
  Fp<C> x = fp_from_be<C>(b), y = fp_from_be<C>(b + C::FP_BYTES);
  bool ok = !fp_geq_p<C>(x) && !fp_geq_p<C>(y);
  out.inf = fp_is_zero<C>(x) && fp_is_zero<C>(y);
  out.x = fp_to_mont<C>(x);
  out.y = fp_to_mont<C>(y);
  return ok;
}
template <class C>
BGLS_HD void g1_to_bytes(uint8_t* b, const Aff<F1<C>>& a) {
  Fp<C> x = a.inf ? fp_zero<C>() : fp_from_mont<C>(a.x);
  Fp<C> y = a.inf ? fp_zero<C>() : fp_from_mont<C>(a.y);
  fp_to_be<C>(b, x);
  fp_to_be<C>(b + C::FP_BYTES, y);
}
template <class C>
BGLS_HD bool g2_from_bytes(Aff<F2<C>>& out, const uint8_t* b) {
  constexpr int N = C::FP_BYTES;
  Fp<C> xi = fp_from_be<C>(b), xr = fp_from_be<C>(b + N), yi = fp_from_be<C>(b + 2 * N), yr = fp_from_be<C>(b + 3 * N);
  bool ok = !fp_geq_p<C>(xi) && !fp_geq_p<C>(xr) && !fp_geq_p<C>(yi) && !fp_geq_p<C>(yr);
  out.inf = fp_is_zero<C>(xi) && fp_is_zero<C>(xr) && fp_is_zero<C>(yi) && fp_is_zero<C>(yr);
  out.x = {fp_to_mont<C>(xr), fp_to_mont<C>(xi)};
  out.y = {fp_to_mont<C>(yr), fp_to_mont<C>(yi)};
  return ok;
}
template <class C>
BGLS_HD void g2_to_bytes(uint8_t* b, const Aff<F2<C>>& a) {
  constexpr int N = C::FP_BYTES;
  Fp<C> z = fp_zero<C>();
  fp_to_be<C>(b, a.inf ? z : fp_from_mont<C>(a.x.c1));
  fp_to_be<C>(b + N, a.inf ? z : fp_from_mont<C>(a.x.c0));
  fp_to_be<C>(b + 2 * N, a.inf ? z : fp_from_mont<C>(a.y.c1));
  fp_to_be<C>(b + 3 * N, a.inf ? z : fp_from_mont<C>(a.y.c0));
}

}  // namespace bgls
