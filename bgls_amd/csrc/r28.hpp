// Carry-free 28-bit-limb arithmetic for the consumer half of the Miller kernel (alt-bn128).
//
// The consumer's unit of work is a short dot product of Fp2 elements with ONE reduction per output half
// (coop.hpp).  With 32-bit limbs every limb product drags a carry add, a re-zeroed addend and a hazard nop along and
// every term needs wide additions; with ten 28-bit limbs all limb products of all terms go straight into 64-bit column
// accumulators (v_mad_u64_u32 d = a*b + d, no carry-out), 2^8 of head-room per column: measured 1.46x on the five-term
// dot product (tools/mb_radix.hip, D).  Montgomery radix R' = 2^280 (26 spare bits above p, so operands may stay
// several p large and nothing needs a conditional subtraction).
//
// The producer half keeps the library's 32-bit limbs (Montgomery radix R = 2^256) and converts each line coefficient as
// it stores it: x R' = (x R) 2^24 mod p, i.e. a shift into 28-bit limbs and ONE Barrett step with a 25-bit quotient
// (to_r28) -- ten small multiplications instead of a field multiplication.
#pragma once
#include "tower.hpp"

namespace bgls {

struct F28 {
  u32 v[10];
};
struct F28x2 {
  F28 c0, c1;
};
constexpr u32 R28_MASK = (1u << 28) - 1;

template <class C>
BGLS_HD F28 r28_load(const u32 (&k)[10]) {
  F28 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.v[i] = k[i];
  return r;
}

// limbs back below 2^28 (the top limb takes what is left); value unchanged
BGLS_HD F28 r28_norm(const F28& a) {
  F28 r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const u32 t = a.v[i] + c;
    r.v[i] = t & R28_MASK;
    c = t >> 28;
  }
  r.v[9] = a.v[9] + c;
  return r;
}

// columns += a * b (all limb products; nothing else)
BGLS_HD void r28_acc(u64 (&c)[20], const F28& a, const F28& b) {
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 10; ++j) c[i + j] = (u64)a.v[i] * b.v[j] + c[i + j];
}

// Montgomery reduction of the columns by R' = 2^280: returns T / R' mod p with limbs below 2^28, value < T / 2^280 + p
template <class C>
BGLS_HD F28 r28_redc(u64 (&c)[20]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const u32 m = ((u32)c[i] * C::R28_NP) & R28_MASK;
#pragma unroll
    for (int j = 0; j < 10; ++j) c[i + j] = (u64)m * C::R28_P[j] + c[i + j];
    c[i + 1] += c[i] >> 28;
  }
  F28 r;
#pragma unroll
  for (int k = 10; k < 19; ++k) {
    r.v[k - 10] = (u32)c[k] & R28_MASK;
    c[k + 1] += c[k] >> 28;
  }
  r.v[9] = (u32)c[19];
  return r;
}

// FAT - b limb by limb: a representative of -b with non-negative limbs (< 2^29); needs b tight and b < 64 p
template <class C>
BGLS_HD F28 r28_fatneg(const F28& b) {
  F28 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.v[i] = C::R28_FAT[i] - b.v[i];
  return r;
}

// x R (32-bit limbs, reduced) -> x R' (28-bit limbs, value < 3p):  z = y 2^24 - q p,  q = (top32(y) MU) >> 32 <= y 2^24 / p
template <class C>
BGLS_HD F28 to_r28(const Fp<C>& y) {
  static_assert(C::L == 8, "alt-bn128 only");
  // y 2^24 as ten 28-bit limbs: bit b of y lands at bit b + 24
  F28 s;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int lo = 28 * i - 24;                      // first bit of y in this limb (may be negative)
    u32 w = 0;
    if (lo < 0) {
      w = (y.v[0] << (-lo)) & R28_MASK;              // i == 0: low 4 bits of y shifted up by 24
    } else {
      const int q = lo >> 5, r = lo & 31;
      u64 two = q < 8 ? (u64)y.v[q] : 0;
      if (q + 1 < 8) two |= (u64)y.v[q + 1] << 32;
      w = (u32)(two >> r) & R28_MASK;
    }
    s.v[i] = w;
  }
  const u32 top = (y.v[7] << 2) | (y.v[6] >> 30);    // bits 222..253 of y
  const u32 q = (u32)(((u64)top * C::R28_MU) >> 32);
  F28 z;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int64_t t = (int64_t)s.v[i] - (int64_t)((u64)q * C::R28_P[i]) + carry;
    z.v[i] = (u32)((u64)t & R28_MASK);
    carry = t >> 28;                                 // arithmetic shift: borrows travel as negative carries
  }
  return z;                                          // q never overestimates, so the total is >= 0 and carry ends at 0
}
template <class C>
BGLS_HD F28x2 to_r28(const Fp2<C>& a) {
  return {to_r28<C>(a.c0), to_r28<C>(a.c1)};
}

// x R' (tight limbs, value < 16 p) -> x R in the library's 32-bit form
template <class C>
BGLS_HD Fp<C> from_r28(const F28& a) {
  // value as 9 x 32-bit limbs (280 bits), then subtract 8p, 4p, 2p, p where possible
  u32 w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int lo = 32 * k;                           // first bit of this word
    const int i = lo / 28, r = lo % 28;
    u64 acc = (u64)a.v[i] >> r;
    int have = 28 - r;
    if (i + 1 < 10) { acc |= (u64)a.v[i + 1] << have; have += 28; }
    if (have < 32 && i + 2 < 10) acc |= (u64)a.v[i + 2] << have;
    w[k] = (u32)acc;
  }
#pragma unroll
  for (int sh = 3; sh >= 0; --sh) {
    // d = w - (p << sh) over 9 words; keep it when it does not borrow
    u32 d[9];
    u32 bw = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const u32 pk = (k < 8 ? (C::P[k] << sh) : 0u) | ((sh && k >= 1) ? (C::P[k - 1] >> (32 - sh)) : 0u);
      d[k] = subb(w[k], pk, bw);
    }
    const u32 keep = bw ? 0u : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = (d[k] & keep) | (w[k] & ~keep);
  }
  Fp<C> y;
#pragma unroll
  for (int k = 0; k < 8; ++k) y.v[k] = w[k];         // now the integer x R' mod p
  return fp_mul<C>(y, fp_load<C>(C::R28_BACK));      // (x R') (R^2 / R') / R = x R
}
template <class C>
BGLS_HD Fp2<C> from_r28(const F28x2& a) {
  return {from_r28<C>(a.c0), from_r28<C>(a.c1)};
}

// xi * a for alt-bn128 (xi = 9 + i): (9 a0 - a1) + (9 a1 + a0) i, limb-wise, then normalised.  a tight, a < 6 p.
template <class C>
BGLS_HD F28x2 r28_mulxi(const F28x2& a) {
  static_assert(C::XI_RE == 9, "alt-bn128 only");
  F28x2 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    r.c0.v[i] = 9u * a.c0.v[i] + (C::R28_FAT[i] - a.c1.v[i]);
    r.c1.v[i] = 9u * a.c1.v[i] + a.c0.v[i];
  }
  r.c0 = r28_norm(r.c0);
  r.c1 = r28_norm(r.c1);
  return r;
}

// a * b in Fp2 (R' form): schoolbook over i, the subtraction as a fat negation; one reduction per half
template <class C>
BGLS_HD F28x2 r28_f2_mul(const F28x2& a, const F28x2& b) {
  F28x2 r;
  {
    u64 c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
    r28_acc(c, a.c0, b.c0);
    r28_acc(c, a.c1, r28_fatneg<C>(b.c1));
    r.c0 = r28_redc<C>(c);
  }
  {
    u64 c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
    r28_acc(c, a.c0, b.c1);
    r28_acc(c, a.c1, b.c0);
    r.c1 = r28_redc<C>(c);
  }
  return r;
}

// ---- Karatsuba form of a three-term dot product: three piles per term, one combination per output ----
BGLS_HD void r28_kara_term(u64 (&v0)[20], u64 (&v1)[20], u64 (&ss)[20], const F28x2& a, const F28x2& b) {
  F28 sa, sb;
#pragma unroll
  for (int q = 0; q < 10; ++q) { sa.v[q] = a.c0.v[q] + a.c1.v[q]; sb.v[q] = b.c0.v[q] + b.c1.v[q]; }
  r28_acc(v0, a.c0, b.c0);
  r28_acc(v1, a.c1, b.c1);
  r28_acc(ss, sa, sb);
}
// at most three terms of tight operands (limbs < 2^28, top limb < 2^12): see R28_BIAS3 in tools/gen_constants.py
template <class C>
BGLS_HD F28x2 r28_kara_finish(u64 (&v0)[20], u64 (&v1)[20], u64 (&ss)[20]) {
#pragma unroll
  for (int k = 0; k < 20; ++k) {
    const u64 bias = C::R28_BIAS3[k];
    const u64 both = v0[k] + v1[k];
    ss[k] = ss[k] + bias - both;            // imaginary part: sum (a0 b1 + a1 b0)
    v0[k] = v0[k] + bias - v1[k];           // real part: sum (a0 b0 - a1 b1)
  }
  F28x2 r;
  r.c0 = r28_redc<C>(v0);
  r.c1 = r28_redc<C>(ss);
  return r;
}

}  // namespace bgls
