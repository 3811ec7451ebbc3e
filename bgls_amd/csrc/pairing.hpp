// Optimal-ate pairing pieces: Miller-loop point steps + line evaluation, and the final
// exponentiation with exponent exactly (p^12 - 1)/r.
//
// Replaces what the reference reaches through CurveSystem.Pair / PairingProduct
// (curves/altbn128.go:130-145 -> bn256.Pair; curves/bls12_381.go:228-240 -> bls12 GT.Pair;
// curves/curve.go:125-170).  The reference runs one FULL pairing (Miller loop + final
// exponentiation) per (H(m_i), pk_i); the final exponentiation is a group homomorphism, so the
// product of the raw Miller values followed by ONE final exponentiation is the same GT element.
//
// Miller loop: homogeneous projective (X,Y,Z) on the twist E': y^2 = x^3 + b'
//  doubling:  A=XY/2 B=Y^2 C=Z^2 E=3b'C F=3E G=(B+F)/2 H=(Y+Z)^2-(B+C) I=E-B J=X^2
//             X3=A(B-F) Y3=G^2-3E^2 Z3=BH ;  line coefficients (-H, 3J, I)
//  mixed add: th=Y-yQ Z, la=X-xQ Z, C=th^2 D=la^2 E=la D F=Z C G=X D Hh=E+F-2G
//             X3=la Hh, Y3=th(G-Hh)-E Y, Z3=Z E ; line coefficients (la, -th, th xQ - la yQ)
//  D-type twist (alt-bn128):  l = c0 yP + c1 xP w + c2 w^3
//  M-type twist (BLS12-381):  l = c2 + c1 xP w^2 + c0 yP w^3
// Lines are defined up to factors in proper subfields of Fp12, which the final exponentiation
// kills; the reduced pairing value does not depend on these choices.
#pragma once
#include "curve.hpp"

namespace bgls {

template <class C>
struct G2Proj {
  Fp2<C> X, Y, Z;
};
template <class C>
struct LineCoeffs {
  Fp2<C> c0, c1, c2;
};

// INL = true expands the Fp2 products in place (straight-line code, operands stay in VGPRs);
// INL = false calls the shared out-of-line f2_mul / f2_sqr (small code, operands via the stack).
template <class C, bool INL>
BGLS_HD Fp2<C> f2m(const Fp2<C>& a, const Fp2<C>& b) {
  if constexpr (INL) return f2_mul_inl<C>(a, b); else return f2_mul<C>(a, b);
}
template <class C, bool INL>
BGLS_HD Fp2<C> f2s(const Fp2<C>& a) {
  if constexpr (INL) return f2_sqr_inl<C>(a); else return f2_sqr<C>(a);
}
// 3 b' Z^2 of the doubling step.  BLS12-381: b' = 4 xi with xi = 1 + i, so 3 b' = 12 (1 + i) and the product is two
// additions for xi and a short doubling chain for 12 = 8 + 4 -- no multiplication.  alt-bn128: b' = 3 / (9 + i) has no
// such structure and stays a full Fp2 multiplication by the constant.
template <class C, bool INL>
BGLS_HD Fp2<C> f2_mul_3b(const Fp2<C>& z2) {
  if constexpr (!C::TWIST_D && C::XI_RE == 1) {
    const Fp2<C> t4 = f2_dbl<C>(f2_dbl<C>(f2_mulxi<C>(z2)));
    return f2_add<C>(f2_dbl<C>(t4), t4);
  } else {
    const Fp2<C> b3 = {fp_load<C>(C::B2X3_RE), fp_load<C>(C::B2X3_IM)};
    if constexpr (INL) return f2_mul_inl<C>(b3, z2); else return f2_mul<C>(b3, z2);
  }
}

template <class C>
BGLS_HD Fp2<C> f2_half(const Fp2<C>& a) { return {fp_half<C>(a.c0), fp_half<C>(a.c1)}; }   // a / 2: shifts, no multiplication
template <class C, bool INL>
BGLS_HD Fp2<C> f2ms(const Fp2<C>& a, const Fp<C>& s) {      // by an Fp scalar
  if constexpr (INL) {
    u32 t0[2 * C::L], t1[2 * C::L];
    mul_wide<C>(t0, a.c0.v, s.v);
    mul_wide<C>(t1, a.c1.v, s.v);
    return {redc<C>(t0), redc<C>(t1)};
  } else {
    return f2_muls<C>(a, s);
  }
}

template <class C, bool INL>
BGLS_HD LineCoeffs<C> dbl_step_t(G2Proj<C>& R) {
  Fp2<C> A = f2_half<C>(f2m<C, INL>(R.X, R.Y));
  Fp2<C> B = f2s<C, INL>(R.Y);
  Fp2<C> Cc = f2s<C, INL>(R.Z);
  Fp2<C> E = f2_mul_3b<C, INL>(Cc);
  Fp2<C> Fv = f2_mul3<C>(E);
  Fp2<C> G = f2_half<C>(f2_add<C>(B, Fv));
  Fp2<C> H = f2_sub<C>(f2s<C, INL>(f2_add<C>(R.Y, R.Z)), f2_add<C>(B, Cc));
  Fp2<C> I = f2_sub<C>(E, B);
  Fp2<C> J = f2s<C, INL>(R.X);
  Fp2<C> Esq = f2s<C, INL>(E);
  R.X = f2m<C, INL>(A, f2_sub<C>(B, Fv));
  R.Y = f2_sub<C>(f2s<C, INL>(G), f2_mul3<C>(Esq));
  R.Z = f2m<C, INL>(B, H);
  return {f2_neg<C>(H), f2_mul3<C>(J), I};
}

template <class C, bool INL>
BGLS_HD LineCoeffs<C> add_step_t(G2Proj<C>& R, const Fp2<C>& xq, const Fp2<C>& yq) {
  Fp2<C> th = f2_sub<C>(R.Y, f2m<C, INL>(yq, R.Z));
  Fp2<C> la = f2_sub<C>(R.X, f2m<C, INL>(xq, R.Z));
  Fp2<C> Cc = f2s<C, INL>(th);
  Fp2<C> D = f2s<C, INL>(la);
  Fp2<C> E = f2m<C, INL>(la, D);
  Fp2<C> Fv = f2m<C, INL>(R.Z, Cc);
  Fp2<C> G = f2m<C, INL>(R.X, D);
  Fp2<C> Hh = f2_sub<C>(f2_add<C>(E, Fv), f2_dbl<C>(G));
  Fp2<C> j = f2_sub<C>(f2m<C, INL>(th, xq), f2m<C, INL>(la, yq));
  R.X = f2m<C, INL>(la, Hh);
  R.Y = f2_sub<C>(f2m<C, INL>(th, f2_sub<C>(G, Hh)), f2m<C, INL>(E, R.Y));
  R.Z = f2m<C, INL>(R.Z, E);
  return {la, f2_neg<C>(th), j};
}

// Register-lean forms for the producer wave: values are consumed as early as possible and each line
// coefficient is handed to `emit(slot, value)` the moment it exists (slot 0/1/2 = c0, c1, c2 of LineCoeffs),
// so at most five Fp2 temporaries are live next to the running point.
template <class C, class Emit>
BGLS_HD void dbl_step_emit(G2Proj<C>& R, Emit&& emit) {
  Fp2<C> B = f2_sqr_inl<C>(R.Y);
  Fp2<C> Cc = f2_sqr_inl<C>(R.Z);
  Fp2<C> H = f2_sub<C>(f2_sqr_inl<C>(f2_add<C>(R.Y, R.Z)), f2_add<C>(B, Cc));
  Fp2<C> E = f2_mul_3b<C, true>(Cc);
  emit(2, f2_sub<C>(E, B));
  emit(0, f2_neg<C>(H));
  emit(1, f2_mul3<C>(f2_sqr_inl<C>(R.X)));
  Fp2<C> A = f2_half<C>(f2_mul_inl<C>(R.X, R.Y));
  R.Z = f2_mul_inl<C>(B, H);
  Fp2<C> Fv = f2_mul3<C>(E);
  R.X = f2_mul_inl<C>(A, f2_sub<C>(B, Fv));
  Fp2<C> G = f2_half<C>(f2_add<C>(B, Fv));
  R.Y = f2_sqrsub3_inl<C>(G, E);
}
template <class C, class Emit>
BGLS_HD void add_step_emit(G2Proj<C>& R, const Fp2<C>& xq, const Fp2<C>& yq, Emit&& emit) {
  Fp2<C> th = f2_sub<C>(R.Y, f2_mul_inl<C>(yq, R.Z));
  Fp2<C> la = f2_sub<C>(R.X, f2_mul_inl<C>(xq, R.Z));
  emit(2, f2_mulsub_inl<C>(th, xq, la, yq));
  emit(0, la);
  emit(1, f2_neg<C>(th));
  Fp2<C> D = f2_sqr_inl<C>(la);
  Fp2<C> G = f2_mul_inl<C>(R.X, D);
  Fp2<C> E = f2_mul_inl<C>(la, D);
  Fp2<C> Hh = f2_sub<C>(f2_add<C>(E, f2_mul_inl<C>(R.Z, f2_sqr_inl<C>(th))), f2_dbl<C>(G));
  R.X = f2_mul_inl<C>(la, Hh);
  R.Z = f2_mul_inl<C>(R.Z, E);
  R.Y = f2_mulsub_inl<C>(th, f2_sub<C>(G, Hh), E, R.Y);
}

template <class C>
BGLS_FN LineCoeffs<C> dbl_step(G2Proj<C>& R) {
  return dbl_step_t<C, false>(R);
}
template <class C>
BGLS_FN LineCoeffs<C> add_step(G2Proj<C>& R, const Fp2<C>& xq, const Fp2<C>& yq) {
  return add_step_t<C, false>(R, xq, yq);
}
// the same steps with every Fp2 product expanded in place
template <class C>
BGLS_FN LineCoeffs<C> dbl_step_inl(G2Proj<C>& R) {
  return dbl_step_t<C, true>(R);
}
template <class C>
BGLS_FN LineCoeffs<C> add_step_inl(G2Proj<C>& R, const Fp2<C>& xq, const Fp2<C>& yq) {
  return add_step_t<C, true>(R, xq, yq);
}

// f * line(P)
template <class C>
BGLS_FN Fp12<C> mul_by_line(const Fp12<C>& f, const LineCoeffs<C>& l, const Fp<C>& xP, const Fp<C>& yP) {
  Fp2<C> a = f2_muls<C>(l.c0, yP);
  Fp2<C> b = f2_muls<C>(l.c1, xP);
  if constexpr (C::TWIST_D)
    return f12_mul_line<C>(f, a, b, l.c2);
  else
    return f12_mul_line<C>(f, l.c2, b, a);
}

// Miller value of one (P, Q) pair, P in G1 affine, Q in G2 affine; 1 when either is infinity
// (the reference defines the GT identity as Pair(g1, inf) / Pair(inf, g2): curves/altbn128.go:478,
// curves/bls12_381.go:341).
template <class C>
BGLS_FN Fp12<C> miller_loop(const Aff<F1<C>>& P, const Aff<F2<C>>& Q) {
  Fp12<C> f = f12_one<C>();
  if (P.inf || Q.inf) return f;
  G2Proj<C> R = {Q.x, Q.y, f2_one<C>()};
  Fp2<C> nyq = f2_neg<C>(Q.y);
  for (int i = 1; i < C::LOOP_LEN; ++i) {
    LineCoeffs<C> l = dbl_step<C>(R);
    f = f12_sqr<C>(f);
    f = mul_by_line<C>(f, l, P.x, P.y);
    int d = C::LOOP_NAF[i];
    if (d != 0) {
      l = add_step<C>(R, Q.x, d > 0 ? Q.y : nyq);
      f = mul_by_line<C>(f, l, P.x, P.y);
    }
  }
  if constexpr (C::CURVE_ID == 0) {
    // Q1 = pi(Q), -Q2 = -pi^2(Q) on the twist
    Fp2<C> x1 = f2_mul<C>(f2_conj<C>(Q.x), gamma_const<C>(1, 2));
    Fp2<C> y1 = f2_mul<C>(f2_conj<C>(Q.y), gamma_const<C>(1, 3));
    Fp2<C> x2 = f2_mul<C>(Q.x, gamma_const<C>(2, 2));
    Fp2<C> y2 = f2_neg<C>(f2_mul<C>(Q.y, gamma_const<C>(2, 3)));
    LineCoeffs<C> l = add_step<C>(R, x1, y1);
    f = mul_by_line<C>(f, l, P.x, P.y);
    l = add_step<C>(R, x2, y2);
    f = mul_by_line<C>(f, l, P.x, P.y);
  } else {
    f = f12_conj<C>(f);  // x < 0
  }
  return f;
}

// a^e for unitary a (cyclotomic subgroup), e = NL limbs, public wave-uniform exponent
template <class C>
BGLS_FN Fp12<C> f12_pow_cyclo(const Fp12<C>& a, const u32* e, int nbits) {
  Fp12<C> r = a;  // top bit is set by construction
  for (int i = nbits - 2; i >= 0; --i) {
    r = f12_cyclo_sqr<C>(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = f12_mul<C>(r, a);
  }
  return r;
}

template <class C>
BGLS_FN Fp12<C> final_exp(const Fp12<C>& fin) {
  // easy part: (p^6 - 1)(p^2 + 1)
  Fp12<C> f = f12_mul<C>(f12_conj<C>(fin), f12_inv<C>(fin));
  f = f12_mul<C>(f12_frob<C>(f, 2), f);
  if constexpr (C::CURVE_ID == 0) {
    // hard part (p^4 - p^2 + 1)/r = l0 + l1 p + l2 p^2 + p^3, y0..y6 vectorial addition chain
    Fp12<C> ft1 = f12_pow_cyclo<C>(f, C::U_ABS, C::U_BITS);
    Fp12<C> ft2 = f12_pow_cyclo<C>(ft1, C::U_ABS, C::U_BITS);
    Fp12<C> ft3 = f12_pow_cyclo<C>(ft2, C::U_ABS, C::U_BITS);
    Fp12<C> y0 = f12_mul<C>(f12_mul<C>(f12_frob<C>(f, 1), f12_frob<C>(f, 2)), f12_frob<C>(f, 3));
    Fp12<C> y1 = f12_conj<C>(f);
    Fp12<C> y2 = f12_frob<C>(ft2, 2);
    Fp12<C> y3 = f12_conj<C>(f12_frob<C>(ft1, 1));
    Fp12<C> y4 = f12_conj<C>(f12_mul<C>(ft1, f12_frob<C>(ft2, 1)));
    Fp12<C> y5 = f12_conj<C>(ft2);
    Fp12<C> y6 = f12_conj<C>(f12_mul<C>(ft3, f12_frob<C>(ft3, 1)));
    Fp12<C> t0 = f12_mul<C>(f12_mul<C>(f12_cyclo_sqr<C>(y6), y4), y5);
    Fp12<C> t1 = f12_mul<C>(f12_mul<C>(y3, y5), t0);
    t0 = f12_mul<C>(t0, y2);
    t1 = f12_cyclo_sqr<C>(f12_mul<C>(f12_cyclo_sqr<C>(t1), t0));
    t0 = f12_mul<C>(t1, y1);
    t1 = f12_mul<C>(t1, y0);
    t0 = f12_cyclo_sqr<C>(t0);
    return f12_mul<C>(t1, t0);
  } else {
    // (p^4 - p^2 + 1)/r = c (x + p)(x^2 + p^2 - 1) + 1,  c = (x-1)^2/3 = G1 cofactor, x < 0
    Fp12<C> a = f12_pow_cyclo<C>(f, C::COFACTOR, C::COFACTOR_BITS);
    Fp12<C> ax = f12_conj<C>(f12_pow_cyclo<C>(a, C::U_ABS, C::U_BITS));
    Fp12<C> b = f12_mul<C>(ax, f12_frob<C>(a, 1));
    Fp12<C> bx = f12_conj<C>(f12_pow_cyclo<C>(b, C::U_ABS, C::U_BITS));
    Fp12<C> bxx = f12_conj<C>(f12_pow_cyclo<C>(bx, C::U_ABS, C::U_BITS));
    Fp12<C> d = f12_mul<C>(f12_mul<C>(bxx, f12_frob<C>(b, 2)), f12_conj<C>(b));
    return f12_mul<C>(d, f);
  }
}

// GT wire format (UNPINNED against the upstream libraries; layout modelled on bn256's
// GT.Marshal: tower coefficients from the top position down, imaginary part first):
//   h.a2.im h.a2.re h.a1.im h.a1.re h.a0.im h.a0.re g.a2.im ... g.a0.re, each FP_BYTES big-endian.
template <class C>
BGLS_HD void gt_to_bytes(uint8_t* b, const Fp12<C>& a) {
  const Fp2<C>* e[6] = {&a.h.a2, &a.h.a1, &a.h.a0, &a.g.a2, &a.g.a1, &a.g.a0};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    fp_to_be<C>(b + (2 * k) * C::FP_BYTES, fp_from_mont<C>(e[k]->c1));
    fp_to_be<C>(b + (2 * k + 1) * C::FP_BYTES, fp_from_mont<C>(e[k]->c0));
  }
}
template <class C>
BGLS_HD bool gt_from_bytes(Fp12<C>& a, const uint8_t* b) {
  Fp2<C>* e[6] = {&a.h.a2, &a.h.a1, &a.h.a0, &a.g.a2, &a.g.a1, &a.g.a0};
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    Fp<C> im = fp_from_be<C>(b + (2 * k) * C::FP_BYTES);
    Fp<C> re = fp_from_be<C>(b + (2 * k + 1) * C::FP_BYTES);
    ok = ok && !fp_geq_p<C>(im) && !fp_geq_p<C>(re);
    e[k]->c1 = fp_to_mont<C>(im);
    e[k]->c0 = fp_to_mont<C>(re);
  }
  return ok;
}

}  // namespace bgls
