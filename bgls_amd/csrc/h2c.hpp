// Hash-to-G1, both curves, result-identical to the reference.
//
// alt-bn128:  HashToG1 (curves/altbn128.go:509-513) -> AltbnKeccak3 (:494-497) ->
//   tryAndIncrementEvm (curves/hash.go:53-77): x = Keccak256(counter || msg) mod q,
//   y^2 = x^3 + 3 (altbn128.go:409-414), r = (y^2)^((q+1)/4) (hash.go:178-190), accept when
//   r^2 == y^2; sign bit = Keccak256(0xFF || msg)[31] & 1 selects y = q - r.
//   (The reference compares against the UNREDUCED x^3+3; that differs from the reduced value
//   only when x^3 mod q >= q-3, which needs a Keccak preimage -- not reproduced.)
// BLS12-381:  HashToG1 (curves/bls12_381.go:349-351) -> hashToG1BlindingAbstracted(msg,false)
//   (:361-376): t_k = BLAKE2b-512(msg || "G1_k") mod q, P_k = bls12FouqueTibouchi(t_k)
//   (:378-393) = cofactor * sw(t_k) (curves/hash.go:86-167) with the degenerate t handled as
//   t=0 -> inf, t=FTRoot1 -> +g1, t=FTRoot2 -> -g1 (no cofactor); result P_0 + P_1.
//   Restructured without changing any output: (i) the Euler test and the square root share one
//   exponentiation (s = g^((q+1)/4); s^2 == g  <=>  g is a square, 0 included: hash.go:254-265);
//   (ii) 1/(1+b+t^2) and 1/w^2 come from a single inversion; (iii) the cofactor is applied once
//   to sw(t_0)+sw(t_1) (scalar multiplication is linear).
#pragma once
#include "curve.hpp"
#include "hashes.hpp"

namespace bgls {

// ------------------------------------------------------------------ alt-bn128
// One try-and-increment candidate.  Returns true and (x, r) in Montgomery form when x^3+3 is a square.
inline BGLS_FN bool bn_h2c_try(const uint8_t* msg, size_t len, u32 counter, Fp<BN254>& x, Fp<BN254>& r) {
  typedef BN254 C;
  ByteSrc src;
  src.msg = msg;
  src.len = len;
  src.pre[0] = (uint8_t)counter;
  src.npre = 1;
  src.nsuf = 0;
  u32 d[8];
  keccak256_legacy(src, d);
  Fp<C> h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h.v[j] = d[7 - j];
  x = fp_to_mont<C>(h);  // h mod q, any 256-bit h
  Fp<C> y2 = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), fp_load<C>(C::B));
  r = fp_sqrt_candidate<C>(y2);
  return fp_eq<C>(fp_sqr<C>(r), y2);
}

// Candidate test by Legendre symbol: returns true when x^3+3 is a square (or 0), with x and y2 = x^3+3.
// Accepts exactly the candidates bn_h2c_try accepts; the square root is taken later, once.
inline BGLS_FN bool bn_h2c_test(const uint8_t* msg, size_t len, u32 counter, Fp<BN254>& x, Fp<BN254>& y2) {
  typedef BN254 C;
  ByteSrc src;
  src.msg = msg;
  src.len = len;
  src.pre[0] = (uint8_t)counter;
  src.npre = 1;
  src.nsuf = 0;
  u32 d[8];
  keccak256_legacy(src, d);
  Fp<C> h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h.v[j] = d[7 - j];
  x = fp_to_mont<C>(h);
  y2 = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), fp_load<C>(C::B));
  return fp_jacobi<C>(y2) >= 0;
}

inline BGLS_FN u32 bn_h2c_sign(const uint8_t* msg, size_t len) {
  ByteSrc src;
  src.msg = msg;
  src.len = len;
  src.pre[0] = 0xFF;
  src.npre = 1;
  src.nsuf = 0;
  u32 d[8];
  keccak256_legacy(src, d);
  return d[7] & 1u;  // last digest byte, low bit
}

// Full per-message loop (used for single points and by the straggler path).
inline BGLS_FN bool bn_hash_to_g1(const uint8_t* msg, size_t len, Aff<F1<BN254>>& out) {
  typedef BN254 C;
  for (u32 c = 0; c < 256; ++c) {
    Fp<C> x, r;
    if (bn_h2c_try(msg, len, c, x, r)) {
      if (bn_h2c_sign(msg, len)) r = fp_neg<C>(r);
      out.x = x;
      out.y = r;
      out.inf = false;
      return true;
    }
  }
  return false;  // the reference would spin forever here (probability 2^-256)
}

// ------------------------------------------------------------------ BLS12-381
// plain integer a > q - a  (parity(), curves/hash.go:169-172)
template <class C>
BGLS_HD bool fp_plain_parity(const Fp<C>& a) {
  Fp<C> d;
  u32 bw = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) d.v[j] = subb(C::P[j], a.v[j], bw);
  // a > d ?
  u32 b2 = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) (void)subb(d.v[j], a.v[j], b2);
  return b2 != 0;
}

// BLAKE2b-512(msg || "G1_" || k), k in {0,1}: the sixteen digest words, d[0] the most significant (curves/hash.go:97-107)
inline BGLS_FN void bls_h2c_digest(const uint8_t* msg, size_t len, int k, u32 (&d)[16]) {
  ByteSrc src;
  src.msg = msg;
  src.len = len;
  src.npre = 0;
  src.pre[0] = 0;
  src.suf[0] = 'G';
  src.suf[1] = '1';
  src.suf[2] = '_';
  src.suf[3] = (uint8_t)('0' + k);
  src.nsuf = 4;
  blake2b512(src, d);
}
// t (plain, reduced) for tag k in {0,1}
inline BGLS_FN Fp<BLS381> bls_h2c_t(const uint8_t* msg, size_t len, int k, Fp<BLS381>& t_mont) {
  typedef BLS381 C;
  u32 d[16];
  bls_h2c_digest(msg, len, k, d);
  Fp<C> lo, hi = fp_zero<C>();
#pragma unroll
  for (int j = 0; j < 12; ++j) lo.v[j] = d[15 - j];
#pragma unroll
  for (int j = 0; j < 4; ++j) hi.v[j] = d[3 - j];
  // (lo + hi 2^384) R mod q
  t_mont = fp_add<C>(fp_mul<C>(lo, fp_load<C>(C::R2)), fp_mul<C>(hi, fp_load<C>(C::R3)));
  return fp_from_mont<C>(t_mont);
}

// Shallue-van de Woestijne encoding of a non-degenerate t; affine result, Montgomery form.
inline BGLS_FN Aff<F1<BLS381>> bls_sw_encode(const Fp<BLS381>& t_mont, bool t_parity) {
  typedef BLS381 C;
  const Fp<C> one = fp_one<C>();
  const Fp<C> b = fp_load<C>(C::B);
  Fp<C> t2 = fp_sqr<C>(t_mont);
  Fp<C> u = fp_add<C>(fp_add<C>(t2, one), b);       // 1 + b + t^2
  Fp<C> v = fp_mul3<C>(t2);                          // 3 t^2
  Fp<C> I = fp_inv<C>(fp_mul<C>(u, v));              // 1/(u v)
  Fp<C> inv_u = fp_mul<C>(I, v);
  Fp<C> w = fp_mul<C>(fp_mul<C>(fp_load<C>(C::SQRT_M3), t_mont), inv_u);
  Fp<C> x = fp_sub<C>(fp_load<C>(C::Z_SW), fp_mul<C>(t_mont, w));  // x0
  Fp<C> y;
  bool found = false;
  for (int i = 0; i < 3 && !found; ++i) {
    if (i == 1) x = fp_sub<C>(fp_neg<C>(x), one);                               // x1 = -1 - x0
    if (i == 2) x = fp_sub<C>(one, fp_mul<C>(fp_mul<C>(fp_sqr<C>(u), u), I));   // x2 = 1 + 1/w^2 = 1 - u^3/(u v)
    Fp<C> g = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), b);
    y = fp_sqrt_candidate<C>(g);
    found = (i == 2) || fp_eq<C>(fp_sqr<C>(y), g);
  }
  if (fp_plain_parity<C>(fp_from_mont<C>(y)) != t_parity) y = fp_neg<C>(y);
  return {x, y, false};
}

// ---- staged form (one work item per (message, tag); candidates x0, x1, x2 tested in separate,
// compacted rounds so that no lane waits for another lane's extra exponentiations) ----
// kind of a per-tag contribution
enum : u32 { H2C_INF = 0, H2C_PLUS_G1 = 1, H2C_MINUS_G1 = 2, H2C_SW = 3, H2C_PENDING = 4 };

struct BlsSwPrep {
  Fp<BLS381> x0, x2;   // candidates (x1 = -1 - x0)
};
// t (Montgomery) -> candidates; t must be non-degenerate
inline BGLS_FN BlsSwPrep bls_sw_prep(const Fp<BLS381>& t_mont) {
  typedef BLS381 C;
  const Fp<C> one = fp_one<C>();
  Fp<C> t2 = fp_sqr<C>(t_mont);
  Fp<C> u = fp_add<C>(fp_add<C>(t2, one), fp_load<C>(C::B));
  Fp<C> v = fp_mul3<C>(t2);
  Fp<C> I = fp_inv<C>(fp_mul<C>(u, v));
  Fp<C> w = fp_mul<C>(fp_mul<C>(fp_load<C>(C::SQRT_M3), t_mont), fp_mul<C>(I, v));
  BlsSwPrep r;
  r.x0 = fp_sub<C>(fp_load<C>(C::Z_SW), fp_mul<C>(t_mont, w));
  r.x2 = fp_sub<C>(one, fp_mul<C>(fp_mul<C>(fp_sqr<C>(u), u), I));
  return r;
}
// candidate test: y = (x^3+4)^((q+1)/4); ok iff y^2 == x^3+4 (0 counts as a square, hash.go:254-265)
inline BGLS_FN bool bls_sw_try(const Fp<BLS381>& x, Fp<BLS381>& y) {
  typedef BLS381 C;
  Fp<C> g = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), fp_load<C>(C::B));
  y = fp_sqrt_candidate<C>(g);
  return fp_eq<C>(fp_sqr<C>(y), g);
}

template <class F>
BGLS_FN Jac<F> jac_mul_jac(const Jac<F>& p, const u32* k, int nbits) {
  Jac<F> r = jac_inf<F>();
  for (int i = nbits - 1; i >= 0; --i) {
    r = jac_dbl<F>(r);
    if ((k[i >> 5] >> (i & 31)) & 1u) r = jac_add<F>(r, p);
  }
  return r;
}

inline BGLS_FN Jac<F1<BLS381>> bls_hash_to_g1_jac(const uint8_t* msg, size_t len) {
  typedef BLS381 C;
  typedef F1<C> F;
  Jac<F> sw_sum = jac_inf<F>();   // sum of the SW-encoded points (cofactor pending)
  Jac<F> special = jac_inf<F>();  // +-g1 contributions (no cofactor)
  const Aff<F> g1 = {fp_load<C>(C::G1X), fp_load<C>(C::G1Y), false};
  for (int k = 0; k < 2; ++k) {
    Fp<C> tm;
    Fp<C> t = bls_h2c_t(msg, len, k, tm);
    if (fp_is_zero<C>(t)) continue;
    if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT1))) {
      special = jac_add_aff<F>(special, g1);
    } else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT2))) {
      special = jac_add_aff<F>(special, aff_neg<F>(g1));
    } else {
      sw_sum = jac_add_aff<F>(sw_sum, bls_sw_encode(tm, fp_plain_parity<C>(t)));
    }
  }
  Jac<F> r = jac_mul_jac<F>(sw_sum, C::COFACTOR, C::COFACTOR_BITS);
  return jac_add<F>(r, special);
}

inline BGLS_FN Aff<F1<BLS381>> bls_hash_to_g1(const uint8_t* msg, size_t len) {
  return jac_to_aff<F1<BLS381>>(bls_hash_to_g1_jac(msg, len));
}

}  // namespace bgls
