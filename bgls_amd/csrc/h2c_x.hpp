// BLS12-381 hash-to-G1, one Shallue-van de Woestijne work item (message, tag) on the carry-free limbs (rx.hpp: fourteen limbs of 28 bits,
// radix R' = 2^392): what k_bls_sw_jacobi (k_hash.hip) runs per lane, in a header so that the CPU tier runs the same code with every column
// accumulation checked (tests/harness/host_harness.cpp: ht_bls_sw_x, against h2c.hpp's 32-bit form of curves/hash.go:97-167).
//
// The reference computes the three candidates through 1 / (u v), u = t^2 + 1 + b, v = 3 t^2; they are the fractions
//     x0 = (Z u - sqrt(-3) t^2) / u,    x1 = -x0 - 1 = (-N0 - u) / u,    x2 = 1 - u^2 / v = (v - u^2) / v,
// and for x = N / D:  g(x) = x^3 + b = G / D^3 with G = N^3 + b D^3, so that
//     g(x) is a square  <=>  chi(G D) >= 0                         (D^4 is a square; g = 0 <=> G = 0, which counts as one),
//     sqrt(G / D^3) = G D^3 (G D^9)^((p-3)/4)                      (p = 3 mod 4; either root: the parity rule picks the sign),
// and the point leaves as Jacobian (X, Y, Z) = (N D, y D^3, D): same affine x and y as the reference's, no inversion.
//
// Round 6: the 512-bit digest goes into the carry-free form by ONE two-product reduction (lo R'^2 + hi R'^2 2^384), every product of the candidates
// is NL^2 multiplier instructions + NL^2 for its reduction instead of the 32-bit Montgomery product's three instructions per limb product, b = 4 is a
// limb-wise multiple, the Legendre symbols are taken on the carry-free residue itself (R' is a perfect square like R), the square-root chain starts
// and ends on these limbs without a conversion, and the three coordinates leave as the library's Montgomery words by folding R mod p (RX_TOM) into
// their last product.  Per 2^20 messages (2^21 work items) on one box: 17.79 ms in the 32-bit form -> 16.79 ms; by stage: digest + t 0.58 ms, the three
// candidates and two Legendre symbols 2.90 ms, the exponentiation 13.1 ms (379 squarings of 301 multiplier instructions in 411, 108 products:
// issue-bound), rest 0.2 ms.
#pragma once
#include "h2c.hpp"
#include "rx_pow.hpp"
#include "rx_jac1.hpp"

namespace bgls {

// d: BLAKE2b-512(msg || "G1_" || k), d[0] most significant (bls_h2c_digest).  ld / st: the lane's window table of sx_pow_sqrt (W-bit windows).
// Returns the kind of the contribution; pt is written for H2C_SW only.
template <int W, bool E0REG, class Ld, class St>
BGLS_HD u32 bls_sw_jac_x(const u32 (&d)[16], Jac<F1<BLS381>>& pt, Ld&& ld, St&& st) {
  typedef BLS381 C;
  constexpr int N = C::RX_NL;
  typedef Sx<C, SX_T> ST;
  typedef Sx<C, SX_F> SF;
  ST tx;
  {
    // value = lo + hi 2^384, lo = d[4..15] (384 bits), hi = d[0..3] (128 bits), d[0] most significant
    i32 sl[N], sh[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int lo = C::RX_W * i, q = lo >> 5, r = lo & 31;
      u64 two = q < 12 ? (u64)d[15 - q] : 0;
      if (q + 1 < 12) two |= (u64)d[14 - q] << 32;
      sl[i] = (i32)((u32)(two >> r) & C::RX_MASK);
      u64 th = q < 4 ? (u64)d[3 - q] : 0;
      if (q + 1 < 4) th |= (u64)d[2 - q] << 32;
      sh[i] = (i32)((u32)(th >> r) & C::RX_MASK);
    }
    const ST r2 = sx_const<C>(C::RX_R2), h384 = sx_const<C>(C::RX_SW_H384);
    const i32* const cols[2] = {r2.v, h384.v};
    tx = sx_montr<C, 2, 2 * SX_T * SX_T>(cols, [&](int k, int i) { return k == 0 ? sl[i] : sh[i]; });      // t R', value in (0, 3 p)
  }
  const Fp<C> t = sx_over_r_words<C>(tx);                                                                  // t itself, canonical
  uint32_t kind = H2C_SW;
  if (fp_is_zero<C>(t)) kind = H2C_INF;
  else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT1))) kind = H2C_PLUS_G1;
  else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT2))) kind = H2C_MINUS_G1;
  if (kind != H2C_SW) return kind;
  const bool t_par = fp_plain_parity<C>(t);
  const ST t2 = sx_sqr<C>(tx);
  const SF u = sx_normf<C>(sx_add<C>(t2, sx_const<C>(C::RX_SW_U0)));                 // t^2 + 1 + b
  const SF v = sx_normf<C>(sx_mulc<3, C>(t2));                                       // 3 t^2
  const ST N0 = s1_mulsub<C>(sx_const<C>(C::RX_SW_Z), u, sx_const<C>(C::RX_SW_S3), t2);
  // chi(G D) on the carry-free residue: non-negative, canonical, 32-bit words
  auto chi = [&](const SF& G, const SF& D) { return fp_jacobi<C>(ux_to_words<C>(sx_to_ux_p<C>(s1_mul<C>(G, D)))); };
  auto g_of = [&](const SF& Nn, const ST& D3) { return sx_normf<C>(sx_add<C>(s1_mul<C>(s1_sqr<C>(Nn), Nn), sx_mulc<4, C>(D3))); };   // N^3 + 4 D^3
  SF Nn = sx_as<SX_F, C>(N0), D = u;
  ST D3 = s1_mul<C>(s1_sqr<C>(D), D);
  SF G = g_of(Nn, D3);
  if (chi(G, D) < 0) {
    Nn = sx_normf<C>(sx_sub<C>(sx_neg<C>(N0), u));
    G = g_of(Nn, D3);
    if (chi(G, D) < 0) {
      D = v;
      Nn = sx_normf<C>(sx_sub<C>(v, s1_sqr<C>(u)));
      D3 = s1_mul<C>(s1_sqr<C>(D), D);
      G = g_of(Nn, D3);
    }
  }
  const ST D9 = sx_mul<C>(sx_sqr<C>(D3), D3);
  const ST A = ux_to_sx<C>(sx_to_ux_p<C>(s1_mul<C>(G, D9)));                          // non-negative, below 2.1 p
  const ST rt = sx_pow_sqrt<C, W, true, E0REG>(A, ld, st);              // (G D^9)^((p - 3) / 4)
  ST y = s1_mul<C>(s1_mul<C>(rt, G), D3);
  if (fp_plain_parity<C>(sx_over_r_words<C>(y)) != t_par) y = sx_neg<C>(y);
  // (X, Y, Z) = (N D, y D^3, D) as Montgomery words: Z R = D * RX_TOM / R', X R = N * (Z R) / R', Y R = (y D^3) * RX_TOM / R'
  const ST tom = sx_const<C>(C::RX_TOM);
  const ST Zw = s1_mul<C>(D, tom);
  const ST Xw = s1_mul<C>(Nn, Zw);
  const ST Yw = s1_mul<C>(s1_mul<C>(y, D3), tom);
  pt = {ux_to_words<C>(sx_to_ux_p<C>(Xw)), ux_to_words<C>(sx_to_ux_p<C>(Yw)), ux_to_words<C>(sx_to_ux_p<C>(Zw))};
  return kind;
}

}  // namespace bgls
