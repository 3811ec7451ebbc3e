// Fixed-exponent powers in Fp on the carry-free limbs (rx.hpp; any limb width C::RX_W): the square roots of the hash-to-G1 maps.
//
// The reference takes these roots with big.Int arithmetic per message (curves/hash.go:53-77 try-and-increment on alt-bn128,
// curves/hash.go:109-139 Fouque-Tibouchi on BLS12-381: y = sqrt(x^3 + b) by the exponent (p + 1) / 4); on the device one lane owns
// one message and the exponentiation is a chain of ~NE*32 dependent squarings -- the longest serial piece of the hashing
// stage.  On 28-bit limbs a squaring is NL (NL + 1) / 2 multiplier instructions into 64-bit columns plus NL^2 for the reduction
// and nothing else (no carry adds), against three VALU instructions per limb product in fp.hpp's 32-bit form.
//
// Sliding windows of W bits over the public exponent: 2^(W-1) odd powers live in LDS (column layout: entry e, limb i of lane l
// at ((e * NL + i) * 64 + l) words -- conflict-free, 4-byte reads as row factors of the interleaved product), the running
// value stays in registers.  The scan of the exponent is wave-uniform scalar work.
#pragma once
#include "rx.hpp"

namespace bgls {

// t: 2 NL - 1 signed columns of limb products (t[2 NL - 1] is never read), |column| < 2^62  ->  T / R' mod p, tight, value in [T / R', T / R' + p)
template <class C>
BGLS_HD Sx<C, SX_T> sx_redc_cols(i64 (&t)[2 * C::RX_NL]) {
  constexpr int N = C::RX_NL;
  constexpr i32 W = C::RX_W;
  constexpr u32 RX_MASK = C::RX_MASK;
  i32 m = (i32)(((u32)t[0] * C::RX_NP) & RX_MASK);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    // the next row's factor needs this row's first two products only: they go first, the scalar chain hides under the rest
    rx_rows_blk<2, true>(t + i, m, (const i32*)C::RX_P);
    t[i + 1] += t[i] >> W;
    const i32 m_next = (i32)(((u32)t[i + 1] * C::RX_NP) & RX_MASK);
    rx_rows<N - 2, true>(t + i + 2, m, (const i32*)C::RX_P + 2);
    m = m_next;
  }
  Sx<C, SX_T> r;
#pragma unroll
  for (int k = N; k < 2 * N - 1; ++k) {
    r.v[k - N] = (i32)((u32)t[k] & RX_MASK);
    if (k + 1 < 2 * N - 1) t[k + 1] += t[k] >> W;
  }
  r.v[N - 1] = (i32)(t[2 * N - 2] >> W);     // column 2 NL - 1 would receive this carry and nothing else: it does not exist
  return r;
}

// rows of the symmetric squaring: row I = a_I^2 into column 2 I and a_I * (2 a_j), j > I, into columns 2 I + 1 ..
template <class C, int I>
BGLS_HD void sx_sqr_rows(i64* t, const i32* a, const i32* d) {
  constexpr int N = C::RX_NL;
  if constexpr (I < N) {
    // columns are written for the first time here (row 0: all of its columns; row I: its last one), never zeroed
    constexpr int K = N - 1 - I;
    if constexpr (I == 0 || K == 0) rx_muls(t[2 * I], a[I], a[I]);
    else rx_macs(t[2 * I], a[I], a[I]);
    if constexpr (K >= 2) rx_rows_new<K, (I == 0 ? 1 : 2)>(t + 2 * I + 1, a[I], d + I + 1);
    else if constexpr (K == 1) rx_muls(t[2 * I + 1], a[I], d[I + 1]);
    sx_sqr_rows<C, I + 1>(t, a, d);
  }
}

// a^2 / R': the NL squares and the NL (NL - 1) / 2 cross products against the doubled limbs.  a tight (|limb| < 2^W):
// a column collects at most NL products below 2^(2W+1) and the reduction's NL below 2^(2W).
template <class C>
BGLS_HD Sx<C, SX_T> sx_sqr(const Sx<C, SX_T>& a) {
  constexpr int N = C::RX_NL;
  // exact count: a column holds at most (NL - 1) / 2 doubled cross products and one square = NL products' worth of 2^(2W), the reduction adds NL more
  static_assert(2 * N < (1 << (63 - 2 * C::RX_W)), "column budget: 2 NL 2^(2W) < 2^63");
  i64 t[2 * N];
  i32 d[N];
#pragma unroll
  for (int k = 0; k < N; ++k) d[k] = 2 * a.v[k];
  sx_sqr_rows<C, 0>(t, a.v, d);
  return sx_redc_cols<C>(t);
}

template <class C>
BGLS_HD Sx<C, SX_T> sx_mul(const Sx<C, SX_T>& a, const Sx<C, SX_T>& b) {
  const i32* const cols[1] = {a.v};
  return sx_montr<C, 1, SX_T * SX_T>(cols, [&](int, int i) { return b.v[i]; });
}

// a (tight, value in (-p, 4 p)) / R' as the canonical plain integer in 32-bit words: the reduction rows alone (a product by one without the product)
template <class C>
BGLS_HD Fp<C> sx_over_r_words(const Sx<C, SX_T>& a) {
  constexpr int N = C::RX_NL;
  i64 t[2 * N];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = a.v[i];
#pragma unroll
  for (int i = N; i < 2 * N; ++i) t[i] = 0;
  return ux_to_words<C>(sx_to_ux_p<C>(sx_redc_cols<C>(t)));
}

// a^e, e = sum word(k) 2^(32 k) over NBITS bits (public, the same for every lane), a tight with value in [0, 2p).
// ld(e, i) / st(e, i, v): limb i of table entry e (entry e holds a^(2 e + 1)), 2^(W-1) entries.  Result tight, value in [0, 1.01 p).
template <class C, int W, int NBITS, class Word, class Ld, class St>
BGLS_HD Sx<C, SX_T> sx_pow_sw(const Sx<C, SX_T>& a, Word&& word, Ld&& ld, St&& st) {
  constexpr int N = C::RX_NL;
  constexpr int TE = 1 << (W - 1);
  {
    const Sx<C, SX_T> a2 = sx_sqr<C>(a);
    Sx<C, SX_T> o = a;
#pragma unroll
    for (int i = 0; i < N; ++i) st(0, i, o.v[i]);
#pragma unroll 1
    for (int e = 1; e < TE; ++e) {
      o = sx_mul<C>(o, a2);
#pragma unroll
      for (int i = 0; i < N; ++i) st(e, i, o.v[i]);
    }
  }
  auto bit = [&](int i) -> u32 { return (word(i >> 5) >> (i & 31)) & 1u; };
  int i = NBITS - 1;
  while (i >= 0 && !bit(i)) --i;
  Sx<C, SX_T> r;
  bool have = false;
#pragma unroll 1
  while (i >= 0) {
    int nsq = 1, idx = -1;
    if (bit(i)) {
      int l = i - W + 1;
      if (l < 0) l = 0;
      while (!bit(l)) ++l;
      u32 v = 0;
      for (int k = i; k >= l; --k) v = (v << 1) | bit(k);
      nsq = i - l + 1;
      idx = (int)(v >> 1);
      i = l - 1;
    } else {
      --i;
    }
    if (have) {
#pragma unroll 1
      for (int s = 0; s < nsq; ++s) r = sx_sqr<C>(r);
      if (idx >= 0) {
        const i32* const cols[1] = {r.v};
        r = sx_montr<C, 1, SX_T * SX_T>(cols, [&](int, int k) { return ld(idx, k); });
      }
    } else {
      // the leading window: r = table entry
#pragma unroll
      for (int k = 0; k < N; ++k) r.v[k] = ld(idx, k);
      have = true;
    }
  }
  return r;
}

// The square-root chains again ((p + 1) / 4, or (p - 3) / 4 with M1), from a window schedule computed at COMPILE time.  sx_pow_sw scans
// the exponent at run time: every bit it looks at is a scalar load from the constant array and a wait (~570 of them per root); on
// a lone wave -- the small-batch hash, one chain per lane and nothing else on the SIMD -- that is 15 % of the root.  Here an
// operation is one 16-bit word: "square nsq times (the zero bits in front of a window and the window's own length), then
// multiply by table entry idx"; the same squarings and products in the same order as sx_pow_sw<C, W, 32 L>, hence the same limbs.
template <class C, int W, bool M1>
struct SqrtSched {
  static constexpr int NB = 32 * C::L;
  struct Tab {
    unsigned short op[NB + 1];                               // nsq | (idx + 1) << 10; op[0]: the leading window (nsq unused)
    int n;
  };
  static constexpr u32 bitc(int i) {
    const u32 w = C::EXP_SQRT[i >> 5] - ((M1 && (i >> 5) == 0) ? 1u : 0u);
    return (w >> (i & 31)) & 1u;
  }
  static constexpr Tab make() {
    Tab t{};
    int n = 0, i = NB - 1, pend = 0;
    while (i >= 0 && !bitc(i)) --i;
    while (i >= 0) {
      if (bitc(i)) {
        int l = i - W + 1;
        if (l < 0) l = 0;
        while (!bitc(l)) ++l;
        u32 v = 0;
        for (int k = i; k >= l; --k) v = (v << 1) | bitc(k);
        t.op[n++] = (unsigned short)((u32)(pend + (i - l + 1)) | (((v >> 1) + 1u) << 10));
        pend = 0;
        i = l - 1;
      } else {
        ++pend;
        --i;
      }
    }
    if (pend) t.op[n++] = (unsigned short)pend;
    t.n = n;
    return t;
  }
};
template <class C, int W, bool M1>
struct SqrtSchedTab {
  static constexpr typename SqrtSched<C, W, M1>::Tab tab = SqrtSched<C, W, M1>::make();
};

//
// E0REG: table entry 0 (a itself) stays in the caller's registers and the table holds a^3, a^5, .. in slots 0 .. 2^(W-1) - 2 (ld / st see slot numbers):
// BLS12-381's four entries of fourteen limbs are 14 336 B per wave, eleven waves per CU; three entries are 10 752 B, and the register limit's
// twelve waves fit (k_bls_sw_jacobi: fourteen more live registers, still under the 168 of three waves per SIMD).
template <class C, int W, bool M1, bool E0REG = false, class Ld, class St>
BGLS_HD Sx<C, SX_T> sx_pow_sqrt(const Sx<C, SX_T>& a, Ld&& ld, St&& st) {
  constexpr int N = C::RX_NL;
  constexpr int TE = 1 << (W - 1);
  constexpr int E0 = E0REG ? 1 : 0;
  {
    const Sx<C, SX_T> a2 = sx_sqr<C>(a);
    Sx<C, SX_T> o = a;
    if constexpr (!E0REG) {
#pragma unroll
      for (int i = 0; i < N; ++i) st(0, i, o.v[i]);
    }
#pragma unroll 1
    for (int e = 1; e < TE; ++e) {
      o = sx_mul<C>(o, a2);
#pragma unroll
      for (int i = 0; i < N; ++i) st(e - E0, i, o.v[i]);
    }
  }
  const int nops = SqrtSchedTab<C, W, M1>::tab.n;
  Sx<C, SX_T> r;
  {
    const int idx = (int)(SqrtSchedTab<C, W, M1>::tab.op[0] >> 10) - 1;
    if (E0REG && idx == 0) r = a;
    else {
#pragma unroll
      for (int k = 0; k < N; ++k) r.v[k] = ld(idx - E0, k);
    }
  }
#pragma unroll 1
  for (int o = 1; o < nops; ++o) {
    const int op = SqrtSchedTab<C, W, M1>::tab.op[o];
    const int nsq = op & 1023, idx = (op >> 10) - 1;
#pragma unroll 1
    for (int s = 0; s < nsq; ++s) r = sx_sqr<C>(r);
    if (E0REG && idx == 0) r = sx_mul<C>(r, a);
    else if (idx >= 0) {
      const i32* const cols[1] = {r.v};
      r = sx_montr<C, 1, SX_T * SX_T>(cols, [&](int, int k) { return ld(idx - E0, k); });
    }
  }
  return r;
}

}  // namespace bgls
