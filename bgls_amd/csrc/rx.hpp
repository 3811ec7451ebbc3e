// Carry-free 28-bit-limb field arithmetic for both curves: NL limbs of 28 bits in 32-bit registers, Montgomery radix
// R' = 2^(28 NL)  (alt-bn128: NL = 10, 26 spare bits above p;  BLS12-381: NL = 14, 11 spare bits).
//
// Why: with 32-bit limbs (fp.hpp) every limb product drags a carry add and a re-zeroed addend along -- three VALU
// instructions per multiplier instruction, and the multiplier (v_mad_u64_u32, a quarter-rate instruction) is the resource
// that bounds this path.  With 28-bit limbs all limb products of a whole dot product go straight into 64-bit column
// accumulators (d = a * b + d, nothing else), 2^8 of head-room per column; carries are resolved once per reduction.  The
// Miller kernel built on this header (miller_x.hpp) runs BOTH roles of the loop on it:
//
//   Ux / Ux2   unsigned, "tight" (limbs 0..NL-2 below 2^28, small non-negative top limb): the form values have in LDS and
//              the form the consumer's dot products work on.  Differences are formed with a column bias (a multiple of p).
//   Sx<LB>     signed limbs with a compile-time magnitude bound LB (units of 2^24; tight = 16): the producer's point
//              steps.  Additions and subtractions are limb-wise and free of carries; every product checks at compile time
//              that its column sums stay inside the signed 64-bit budget (Pile<B>).
//
// The reference has no field arithmetic (it imports bn256/cloudflare and dis2/bls12: curves/altbn128.go:11,
// curves/bls12_381.go:11); results here are the same field elements as fp.hpp's, in another radix.
//
// The header compiles for the host as well (tests/harness): with BGLS_RX_CHECK every column accumulation is checked for
// overflow there, which is how the worst-case operands of the unit tests prove the bounds.
#pragma once
#include "tower.hpp"

namespace bgls {

typedef int32_t i32;
typedef int64_t i64;
// limb width: C::RX_W bits (28 everywhere but the alt-bn128 Miller kernel's form BN254W: nine limbs of 29 bits, round 5), mask C::RX_MASK
constexpr u32 RX_MASK = (1u << 28) - 1;      // the 28-bit forms' mask, for the headers that only exist on them (rx_pow.hpp, finalx.hpp)
// "lazy" forms leave 2^8 of head-room in a 64-bit column (W = 28): six term-equivalents per pile, products of sums, four
// products per interleaved reduction.  The 29-bit form has 2^6: three-term piles of TIGHT operands only (63 of the 64 units of
// 2^58 a column holds: tools/gen_constants.py), and its Montgomery radix is only 169 p, so values are kept small explicitly
// (ux_quasi) where the 28-bit forms simply never get near R'.
template <class C>
constexpr bool rx_lazy = 2 * C::RX_W + 8 <= 64;
// does an interleaved product with this bound sum (units 2^(2 W - 8)) fit the signed 64-bit columns?  (MontAcc below)
template <class C>
constexpr bool rx_fits(long long budget) { return (long long)C::RX_NL * (budget + 256) + 64 < (1ll << (71 - 2 * C::RX_W)); }

#if !defined(__HIP_DEVICE_COMPILE__) && defined(BGLS_RX_CHECK)
extern int g_rx_overflow;
#define RX_HOST_CHECK 1
#else
#define RX_HOST_CHECK 0
#endif

// d <- a * b + d.  On the device this is ONE v_mad_u64_u32 / v_mad_i64_i32 accumulating in place, written as inline
// assembly: left to itself the compiler renames the loop-carried column accumulators (out-of-place multiply-adds plus a
// v_mov_b64 per column and iteration) and schedules every operand load of a loop body up front, which costs a copy per
// multiplier instruction and pushed the kernels hundreds of registers over their budget.  The statements are volatile, so
// the multiplier instructions issue in program order: consecutive ones write different columns and a column is revisited
// NL - 1 instructions later, far beyond the instruction's latency.  A row of limb products goes out as ONE asm statement
// (rx_rows_gen.hpp): the compiler separates two adjacent asm statements by an s_nop.
BGLS_HD void rx_macu(u64& c, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
#elif RX_HOST_CHECK
  u64 r;
  if (__builtin_add_overflow(c, (u64)a * b, &r)) g_rx_overflow = 1;
  c = r;
#else
  c = (u64)a * b + c;
#endif
}
BGLS_HD void rx_macs(i64& c, i32 a, i32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
#elif RX_HOST_CHECK
  i64 r;
  if (__builtin_add_overflow(c, (i64)a * b, &r)) g_rx_overflow = 1;
  c = r;
#else
  c = (i64)a * b + c;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
#define RX_M1(OP, k, A, B) OP " %" #k ", vcc, %" #A ", %" #B ", %" #k "\n\t"
#define RX_M0(OP, k, A, B) OP " %" #k ", vcc, %" #A ", %" #B ", 0\n\t"
// K accumulators c[0..K), one left factor a, K right factors b[0..K) (BC = "v": registers, "s": scalar constants)
#include "rx_rows_gen.hpp"
#endif

// c[0..K) += a * b[0..K)   (2 <= K <= 14; CONST: the b are compile-time constants, held in scalar registers)
template <int K, bool CONST>
BGLS_HD void rx_rowu_blk(u64* c, u32 a, const u32* b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CONST) RX_ROW_DISPATCH("v_mad_u64_u32", "s", K, c, a, b);
  else RX_ROW_DISPATCH("v_mad_u64_u32", "v", K, c, a, b);
#else
  for (int j = 0; j < K; ++j) rx_macu(c[j], a, b[j]);
#endif
}
template <int K, bool CONST>
BGLS_HD void rx_rows_blk(i64* c, i32 a, const i32* b) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CONST) RX_ROW_DISPATCH("v_mad_i64_i32", "s", K, c, a, b);
  else RX_ROW_DISPATCH("v_mad_i64_i32", "v", K, c, a, b);
#else
  for (int j = 0; j < K; ++j) rx_macs(c[j], a, b[j]);
#endif
}
// MODE 1: c[0..K) = a * b[0..K) (every column written for the first time); MODE 2: c[0..K-1) += .., c[K-1] = a * b[K-1].
// A product's columns start life in these rows, so that no register is zeroed in front of a product (a 64-bit move per
// column: 7 % of the Miller kernel's vector instructions).
template <int K, int MODE>
BGLS_HD void rx_rows_new(i64* c, i32 a, const i32* b) {
  static_assert(MODE == 1 || MODE == 2, "mode");
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (MODE == 1) RX_ROWF_DISPATCH("v_mad_i64_i32", "v", K, c, a, b);
  else RX_ROWL_DISPATCH("v_mad_i64_i32", "v", K, c, a, b);
#else
  for (int j = 0; j < K; ++j) {
    if (MODE == 1 || j == K - 1) c[j] = 0;
    rx_macs(c[j], a, b[j]);
  }
#endif
}
template <int K, int MODE>
BGLS_HD void rx_rowu_new(u64* c, u32 a, const u32* b) {
  static_assert(MODE == 1 || MODE == 2, "mode");
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (MODE == 1) RX_ROWF_DISPATCH("v_mad_u64_u32", "v", K, c, a, b);
  else RX_ROWL_DISPATCH("v_mad_u64_u32", "v", K, c, a, b);
#else
  for (int j = 0; j < K; ++j) {
    if (MODE == 1 || j == K - 1) c[j] = 0;
    rx_macu(c[j], a, b[j]);
  }
#endif
}
// c[0..K) += a * b[0..K) with SIGNED factors on columns that are read as unsigned totals: two's complement, every intermediate taken mod 2^64
// (the cross pile of ux_dot_k2p: its final totals are non-negative and inside the budget, its partial sums need not be)
template <int K>
BGLS_HD void rx_rows_wrap(u64* c, i32 a, const i32* b) {
#if defined(__HIP_DEVICE_COMPILE__)
  RX_ROW_DISPATCH("v_mad_i64_i32", "v", K, c, a, b);
#else
  for (int j = 0; j < K; ++j) c[j] += (u64)((i64)a * (i64)b[j]);
#endif
}
// c = a * b (first write of the column)
BGLS_HD void rx_muls(i64& c, i32 a, i32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b) : "vcc");
#else
  c = (i64)a * b;
#endif
}
BGLS_HD void rx_mulu(u64& c, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b) : "vcc");
#else
  c = (u64)a * b;
#endif
}
// c[0..K) += a * b[0..K): one asm statement per row (K <= 14)
template <int K, bool CONST>
BGLS_HD void rx_rowu(u64* c, u32 a, const u32* b) { rx_rowu_blk<K, CONST>(c, a, b); }
template <int K, bool CONST>
BGLS_HD void rx_rows(i64* c, i32 a, const i32* b) { rx_rows_blk<K, CONST>(c, a, b); }

// ===================================================================================================== unsigned, tight
template <class C>
struct Ux {
  u32 v[C::RX_NL];
};
template <class C>
struct Ux2 {
  Ux<C> c0, c1;
};

template <class C>
BGLS_HD Ux<C> ux_load(const u32* k) {
  Ux<C> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = k[i];
  return r;
}
template <class C>
BGLS_HD Ux<C> ux_zero() {
  Ux<C> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = 0;
  return r;
}

// columns += a * b : NL^2 multiplier instructions and nothing else
template <class C>
BGLS_HD void ux_acc(u64 (&c)[2 * C::RX_NL], const Ux<C>& a, const Ux<C>& b) {
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) rx_rowu<C::RX_NL, false>(c + i, a.v[i], b.v);
}

// columns = a * b, every column written for the first time by its first product (no zeroing; the top column, which no
// product reaches, is set)
template <class C>
BGLS_HD void ux_acc_new(u64 (&c)[2 * C::RX_NL], const Ux<C>& a, const Ux<C>& b) {
  constexpr int N = C::RX_NL;
  rx_rowu_new<N, 1>(c, a.v[0], b.v);
#pragma unroll
  for (int i = 1; i < N; ++i) rx_rowu_new<N, 2>(c + i, a.v[i], b.v);
  c[2 * N - 1] = 0;
}

// Montgomery reduction of the columns by R': returns T / R' mod p, tight, value < T / R' + p.
// Row i adds m_i p to columns i .. i+NL-1, m_i = -c[i] / p mod 2^28.  The next row's factor depends on this row's first two
// products only (c[i] gives the carry, c[i+1] the factor), so those two go first and the dependent scalar chain -- shift,
// add, multiply, mask -- resolves under the row's other NL - 2 multiplier instructions; with the row issued in plain
// column order a lone wave sat out that chain on every row (half of a reduction's time).
template <class C>
BGLS_HD Ux<C> ux_redc(u64 (&c)[2 * C::RX_NL]) {
  constexpr int N = C::RX_NL;
  constexpr int W = C::RX_W;
  constexpr u32 RX_MASK = C::RX_MASK;
  u32 m = ((u32)c[0] * C::RX_NP) & RX_MASK;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    rx_rowu_blk<2, true>(c + i, m, C::RX_P);
    c[i + 1] += c[i] >> W;
    const u32 m_next = ((u32)c[i + 1] * C::RX_NP) & RX_MASK;
    rx_rowu<N - 2, true>(c + i + 2, m, C::RX_P + 2);
    m = m_next;
  }
  Ux<C> r;
#pragma unroll
  for (int k = N; k < 2 * N - 1; ++k) {
    r.v[k - N] = (u32)c[k] & RX_MASK;
    c[k + 1] += c[k] >> W;
  }
  r.v[N - 1] = (u32)c[2 * N - 1];
  return r;
}

// limbs back below 2^28 (the top limb takes what is left); value unchanged.  Limbs must be non-negative.
template <class C>
BGLS_HD Ux<C> ux_norm(const Ux<C>& a) {
  Ux<C> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < C::RX_NL - 1; ++i) {
    const u32 t = a.v[i] + c;
    r.v[i] = t & C::RX_MASK;
    c = t >> C::RX_W;
  }
  r.v[C::RX_NL - 1] = a.v[C::RX_NL - 1] + c;
  return r;
}

// xi * a, normalised; a tight with value < 2 p (a reduction's output); the result is tight, non-negative and below 32 p.  alt-bn128: xi = 9 + i; BLS12-381: 1 + i.
//
// 29-bit form: the radix is only 169 p, so the ten-fold growth cannot be left to the next reduction (the real part's column bias is
// worth 3 A B / 169 p for operand bounds A p, B p: with xi copies of 40 p the folds' fixed point runs away).  An estimate q of the
// quotient is read off the top limbs before the pass -- q <= value / p, at most 3 short -- and subtracted IN the pass:
//      xi a - q p  =  xi a + (32 - q) p + G,    G0 = 4 p - 32 p (low limbs dominating the a1 that the real part subtracts, a1 below 4 p), G1 = -32 p
// one more multiplier instruction per limb; 9 a_i leaves 32 bits anyway, so the carry runs through 64-bit sums limb by limb.
// a tight, value below 3 p: the result is tight, non-negative and below 3.001 p.
template <class C>
BGLS_HD Ux2<C> ux_mulxi(const Ux2<C>& a) {
  Ux2<C> r;
  if constexpr (rx_lazy<C>) {
#pragma unroll
    for (int i = 0; i < C::RX_NL; ++i) {
      r.c0.v[i] = (u32)C::XI_RE * a.c0.v[i] + (C::RX_FAT[i] - a.c1.v[i]);
      r.c1.v[i] = (u32)C::XI_RE * a.c1.v[i] + a.c0.v[i];
    }
    r.c0 = ux_norm<C>(r.c0);
    r.c1 = ux_norm<C>(r.c1);
  } else {
    constexpr int T = C::RX_NL - 1;
    constexpr float QI = 1.0f / (float)(C::RX_P[T] + 1u);
#if RX_HOST_CHECK
    // preconditions (advice r5: they were enforced by the callers' discipline only): tight limbs, both halves below 3 p -- the quotient is read
    // off the TOP limbs alone, in single precision, and is allowed to be 3 short, no more
    for (int i = 0; i < T; ++i)
      if (a.c0.v[i] > C::RX_MASK || a.c1.v[i] > C::RX_MASK) g_rx_overflow = 1;
    if (a.c0.v[T] > 3u * C::RX_P[T] + 3u || a.c1.v[T] > 3u * C::RX_P[T] + 3u) g_rx_overflow = 1;
#endif
    // un-normalised top limbs of xi a (+ 4 p, not counting the low limbs' 2^W each: an under-estimate): the lower limbs add less than 12 units to them
    const i32 s0 = (i32)((u32)C::XI_RE * a.c0.v[T]) - (i32)a.c1.v[T] + 4 * (i32)C::RX_P[T];
    const i32 s1 = (i32)((u32)C::XI_RE * a.c1.v[T] + a.c0.v[T]);
    auto quot = [&](i32 s) {              // floor(s / (top limb of p + 1)) - 2 (one for the float's rounding), clamped to 0 .. 31
      i32 q = (i32)((float)s * QI) - 2;
      q = q < 0 ? 0 : q;
      return (u32)(32 - (q > 31 ? 31 : q));
    };
    const u32 k0 = quot(s0), k1 = quot(s1);
    u64 c0 = 0, c1 = 0;
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const u64 t0 = (u64)(u32)C::XI_RE * a.c0.v[i] + (u64)k0 * C::RX_P[i] + (u64)((u32)C::RX_XIG0[i] - a.c1.v[i]) + c0;
      const u64 t1 = (u64)(u32)C::XI_RE * a.c1.v[i] + (u64)k1 * C::RX_P[i] + (u64)((u32)C::RX_XIG1[i] + a.c0.v[i]) + c1;
      r.c0.v[i] = (u32)t0 & C::RX_MASK;
      r.c1.v[i] = (u32)t1 & C::RX_MASK;
      c0 = t0 >> C::RX_W;
      c1 = t1 >> C::RX_W;
    }
    r.c0.v[T] = (u32)((i32)((u32)C::XI_RE * a.c0.v[T] + k0 * C::RX_P[T]) + (i32)C::RX_XIG0[T] - (i32)a.c1.v[T] + (i32)c0);
    r.c1.v[T] = (u32)((i32)((u32)C::XI_RE * a.c1.v[T] + k1 * C::RX_P[T] + a.c0.v[T]) + (i32)C::RX_XIG1[T] + (i32)c1);
#if RX_HOST_CHECK
    // postcondition: non-negative and below 3.001 p, i.e. a top limb in [0, 3 p_top + 3]
    if ((i32)r.c0.v[T] < 0 || (i32)r.c1.v[T] < 0 || r.c0.v[T] > 3u * C::RX_P[T] + 3u || r.c1.v[T] > 3u * C::RX_P[T] + 3u) g_rx_overflow = 1;
#endif
  }
  return r;
}

// Quasi-reduction (29-bit form): subtract 2^k p, k = KHI .. KLO, wherever the top limb shows that the value is above it, then one
// carry pass.  a: non-negative limbs below 2^31 - 2^(W+2) (a few tight values added up), any value below 2^(KHI+1) p; the result
// is tight and below 2^KLO p + 8 * 2^(W (NL-1))  (= 2^KLO p (1 + 2^-19)).  The margin of 5 on the top limb covers the lower limbs
// going negative by the constants of the levels already subtracted (at most four levels).
template <class C, int KHI, int KLO>
BGLS_HD Ux<C> ux_quasi(const Ux<C>& a) {
  constexpr int N = C::RX_NL;
  static_assert(KLO >= 1 && KHI <= 6 && KHI - KLO <= 3, "ladder levels");
  i32 t[N];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = (i32)a.v[i];
#pragma unroll
  for (int k = KHI; k >= KLO; --k) {
    const u32* L = C::RX_LAD + (k - 1) * N;
    const i32 hit = t[N - 1] >= (i32)L[N - 1] + 5 ? -1 : 0;
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] -= (i32)L[i] & hit;
  }
  Ux<C> r;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    const i32 s = t[i] + c;
    r.v[i] = (u32)(s & (i32)C::RX_MASK);
    c = s >> C::RX_W;
  }
  r.v[N - 1] = (u32)(t[N - 1] + c);
  return r;
}
template <class C, int KHI, int KLO>
BGLS_HD Ux2<C> ux_quasi(const Ux2<C>& a) {
  return {ux_quasi<C, KHI, KLO>(a.c0), ux_quasi<C, KHI, KLO>(a.c1)};
}

// ---- the consumer's dot products.  Operands are fetched through functors (LDS on the device, arrays in the unit tests):
//      lda(t, h) / ldb(t, h) return half h (0 = real, 1 = imaginary part) of the t-th left / right operand.
//
// c = sum_{t<NT} A_t * B_t in Fp2, NT <= 3, Karatsuba over i in TWO passes so that only two column piles are ever live:
//   pass 1:  D = sum a0 b0,  E = sum a1 b1        ->  real part = D + BIAS - E   (BIAS: a multiple of p above every column of E)
//   pass 2:  X = (D + E) + sum (a1 - a0)(b0 - b1)  ->  imaginary part            (wrap-around arithmetic: the column totals
//            sum (a0 b1 + a1 b0) are non-negative because every limb is, so the 64-bit result is exact; round 5: the
//            DIFFERENCES' products instead of the sums' -- D + E enters as it is instead of negated, two instructions per column
//            less, and a product of differences is a quarter of a product of sums)
// 3 NT NL^2 + 2 NL^2 multiplier instructions.  Operands tight, values < 32 p (column budget: tools/gen_constants.py).
// Loops over the terms: the plain pass-1 loop and the pass-2 loop are unrolled (round 5, same-box A/B in tools/mb_x60.bin: BLS12-381 88.2 -> 86.5 ms,
// alt-bn128 54.8 -> 54.4 ms per 2^20 pairings); the software-pipelined pass-1 loop stays rolled (unrolled it speeds the consumer alone up by 9 % and
// the whole kernel not at all: the kernel's text is many times the instruction cache).
template <class C, int NT, bool PF = false, class LA, class LB>
BGLS_HD Ux2<C> ux_dot_k2p(LA&& lda, LB&& ldb) {
  constexpr int N = C::RX_NL;
  static_assert(NT >= 1 && NT <= 3, "column budget");
  u64 d[2 * N], e[2 * N];
  // the first term's products create the columns (ux_acc_new), the others accumulate
  if constexpr (PF) {
    // software-pipelined fetches: the operands of the next pile's products are requested before this pile's multiplier
    // instructions are issued (2 NL more live registers: alt-bn128 has them, BLS12-381 at 168 registers does not)
    Ux<C> a0 = lda(0, 0), b0 = ldb(0, 0);
    {
      const Ux<C> a1 = lda(0, 1), b1 = ldb(0, 1);
      ux_acc_new<C>(d, a0, b0);
      if (NT > 1) { a0 = lda(1, 0); b0 = ldb(1, 0); }
      ux_acc_new<C>(e, a1, b1);
    }
#pragma unroll 1
    for (int t = 1; t < NT; ++t) {
      const Ux<C> a1 = lda(t, 1), b1 = ldb(t, 1);
      ux_acc<C>(d, a0, b0);
      if (t + 1 < NT) { a0 = lda(t + 1, 0); b0 = ldb(t + 1, 0); }
      ux_acc<C>(e, a1, b1);
    }
  } else {
    {
      const Ux<C> a0 = lda(0, 0), b0 = ldb(0, 0);
      ux_acc_new<C>(d, a0, b0);
    }
    {
      const Ux<C> a1 = lda(0, 1), b1 = ldb(0, 1);
      ux_acc_new<C>(e, a1, b1);
    }
#pragma unroll
    for (int t = 1; t < NT; ++t) {
      {
        const Ux<C> a0 = lda(t, 0), b0 = ldb(t, 0);
        ux_acc<C>(d, a0, b0);
      }
      {
        const Ux<C> a1 = lda(t, 1), b1 = ldb(t, 1);
        ux_acc<C>(e, a1, b1);
      }
    }
  }
  Ux2<C> r;
#pragma unroll
  for (int k = 0; k < 2 * N; ++k) {
    const u64 dk = d[k], ek = e[k];
    e[k] = dk + C::RX_BIAS_D3[k] - ek;
    d[k] = dk + ek;
  }
  r.c0 = ux_redc<C>(e);
#if RX_HOST_CHECK
  u64 chk[2 * N];
  for (int k = 0; k < 2 * N; ++k) chk[k] = 0;
#endif
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    i32 sa[N], sb[N];
    {
      const Ux<C> a0 = lda(t, 0), a1 = lda(t, 1);
#pragma unroll
      for (int q = 0; q < N; ++q) sa[q] = (i32)a1.v[q] - (i32)a0.v[q];
    }
    {
      const Ux<C> b0 = ldb(t, 0), b1 = ldb(t, 1);
#pragma unroll
      for (int q = 0; q < N; ++q) sb[q] = (i32)b0.v[q] - (i32)b1.v[q];
    }
#if RX_HOST_CHECK
    // the TRUE cross pile sum (a0 b1 + a1 b0) is accumulated with every addition checked; the pile of the differences' products is
    // taken mod 2^64 on purpose (its partial sums dip below zero) and must land on the true one column by column
    ux_acc<C>(chk, lda(t, 0), ldb(t, 1));
    ux_acc<C>(chk, lda(t, 1), ldb(t, 0));
#endif
#pragma unroll
    for (int i = 0; i < N; ++i) rx_rows_wrap<N>(d + i, sa[i], sb);
  }
#if RX_HOST_CHECK
  for (int k = 0; k < 2 * N; ++k)
    if (chk[k] != d[k]) g_rx_overflow = 1;
#endif
  r.c1 = ux_redc<C>(d);
  return r;
}

// The same dot product with every operand fetch issued ONE STAGE AHEAD of its use (round 6).  The multiplier rows are volatile asm statements,
// so the compiler leaves each fetch where the source puts it: ux_dot_k2p's second pass (fetch two halves, subtract, fetch two halves, subtract,
// multiply) waits out an LDS round trip in front of every term, six times per line fold -- tools/mb_stamps.bin: a consumer wave alone runs at
// 60 % of its issue rate.  Here the halves of term t + 1 are requested before the 81 / 196 multiplier instructions of term t go out, pass 2's
// first term before the first reduction, and (PRE) the caller may hand in halves it requested even earlier.  Same products, same piles, same
// order of accumulation: bit-identical results.  Live at the peak (pass 1): two piles, the two halves being multiplied, the two in flight.
template <class C, int NT, class LA, class LB>
BGLS_HD Ux2<C> ux_dot_k2q(LA&& lda, LB&& ldb) {
  constexpr int N = C::RX_NL;
  static_assert(NT >= 1 && NT <= 3, "column budget");
  u64 d[2 * N], e[2 * N];
  Ux<C> a0 = lda(0, 0), b0 = ldb(0, 0);
  Ux<C> a1 = lda(0, 1), b1 = ldb(0, 1);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t == 0) ux_acc_new<C>(d, a0, b0); else ux_acc<C>(d, a0, b0);
    if (t + 1 < NT) { a0 = lda(t + 1, 0); b0 = ldb(t + 1, 0); }
    else { a0 = lda(0, 0); b0 = ldb(0, 0); }                       // pass 2, term 0
    if (t == 0) ux_acc_new<C>(e, a1, b1); else ux_acc<C>(e, a1, b1);
    if (t + 1 < NT) { a1 = lda(t + 1, 1); b1 = ldb(t + 1, 1); }
    else { a1 = lda(0, 1); b1 = ldb(0, 1); }
  }
  Ux2<C> r;
#pragma unroll
  for (int k = 0; k < 2 * N; ++k) {
    const u64 dk = d[k], ek = e[k];
    e[k] = dk + C::RX_BIAS_D3[k] - ek;
    d[k] = dk + ek;
  }
  r.c0 = ux_redc<C>(e);
#if RX_HOST_CHECK
  u64 chk[2 * N];
  for (int k = 0; k < 2 * N; ++k) chk[k] = 0;
#endif
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    i32 sa[N], sb[N];
#pragma unroll
    for (int q = 0; q < N; ++q) { sa[q] = (i32)a1.v[q] - (i32)a0.v[q]; sb[q] = (i32)b0.v[q] - (i32)b1.v[q]; }
#if RX_HOST_CHECK
    ux_acc<C>(chk, lda(t, 0), ldb(t, 1));
    ux_acc<C>(chk, lda(t, 1), ldb(t, 0));
#endif
    if (t + 1 < NT) { a0 = lda(t + 1, 0); a1 = lda(t + 1, 1); b0 = ldb(t + 1, 0); b1 = ldb(t + 1, 1); }
#pragma unroll
    for (int i = 0; i < N; ++i) rx_rows_wrap<N>(d + i, sa[i], sb);
  }
#if RX_HOST_CHECK
  for (int k = 0; k < 2 * N; ++k)
    if (chk[k] != d[k]) g_rx_overflow = 1;
#endif
  r.c1 = ux_redc<C>(d);
  return r;
}

// c = sum over up to four slots of A_t * B_t with doubled slots (the symmetric squaring of coop.hpp: at most six
// term-equivalents per lane), Karatsuba over i in the two passes of ux_dot_k2p (round 4; schoolbook until then: 16 NL^2
// multiplier instructions per lane and squaring, now 12):
//   pass 1:  D = sum k a0 b0,  E = sum k a1 b1          ->  real part = D + BIAS_S6 - E
//   pass 2:  X = -(D + E) + sum k (a0 + a1)(b0 + b1)    ->  imaginary part, in wrap-around arithmetic
// The pile sum k (a0 + a1)(b0 + b1) ALONE would leave 64 bits on NL = 14 (six term-equivalents of four limb products: 2^64.4),
// which is why round 3 kept the schoolbook form -- but it is never formed alone: X starts at -(D + E) mod 2^64 and every
// intermediate value is taken mod 2^64; the FINAL column totals are those of sum k (a0 b1 + a1 b0), non-negative and inside
// the budget the schoolbook cross pile already had (tools/gen_constants.py "sqr cross pile": 12 col + red + carry < 2^64 on
// both curves), so the 64-bit result is exact.  The host build checks exactly that: the true cross pile is accumulated with
// overflow checks next to the wrap-around one and the two must agree column by column.
//   kind(t) = 0 (unused slot), 1 (plain) or 2 (doubled);  lda / ldb as above (halves are fetched one pair at a time so that
//   two piles and two halves are all that is live)
template <class C, class KD, class LA, class LB>
BGLS_HD Ux2<C> ux_sqr_dot(KD&& kind, LA&& lda, LB&& ldb) {
  constexpr int N = C::RX_NL;
  static_assert(rx_lazy<C>, "six term-equivalents per pile: 28-bit forms only (the 29-bit form squares with ux_sqr_dot3)");
  u64 d[2 * N], e[2 * N];
  auto scaled = [&](int t, int h) __attribute__((always_inline)) {       // left operand: doubled or masked out
    const int kd = kind(t);
    const u32 keep = kd ? 0xFFFFFFFFu : 0u;
    const u32 sh = kd == 2 ? 1u : 0u;
    Ux<C> a = lda(t, h);
#pragma unroll
    for (int q = 0; q < N; ++q) a.v[q] = (a.v[q] << sh) & keep;
    return a;
  };
  // slot 0 creates the columns (ux_acc_new: no zeroing), slots 1..3 accumulate
  {
    const Ux<C> a0 = scaled(0, 0), b0 = ldb(0, 0);
    ux_acc_new<C>(d, a0, b0);
  }
  {
    const Ux<C> a1 = scaled(0, 1), b1 = ldb(0, 1);
    ux_acc_new<C>(e, a1, b1);
  }
#pragma unroll 1
  for (int t = 1; t < 4; ++t) {
    {
      const Ux<C> a0 = scaled(t, 0), b0 = ldb(t, 0);
      ux_acc<C>(d, a0, b0);
    }
    {
      const Ux<C> a1 = scaled(t, 1), b1 = ldb(t, 1);
      ux_acc<C>(e, a1, b1);
    }
  }
#pragma unroll
  for (int k = 0; k < 2 * N; ++k) {
    const u64 dk = d[k], ek = e[k];
    e[k] = dk + C::RX_BIAS_S6[k] - ek;
    d[k] = 0 - (dk + ek);
  }
  Ux2<C> r;
  r.c0 = ux_redc<C>(e);
#if RX_HOST_CHECK
  u64 chk[2 * N];
  for (int k = 0; k < 2 * N; ++k) chk[k] = 0;
#endif
#pragma unroll 1
  for (int t = 0; t < 4; ++t) {
    Ux<C> sa, sb;
    {
      const Ux<C> a0 = scaled(t, 0), a1 = scaled(t, 1);
#pragma unroll
      for (int q = 0; q < N; ++q) sa.v[q] = a0.v[q] + a1.v[q];
#if RX_HOST_CHECK
      ux_acc<C>(chk, a0, ldb(t, 1));    // the true cross pile sum k (a0 b1 + a1 b0), every accumulation checked
      ux_acc<C>(chk, a1, ldb(t, 0));
#endif
    }
    {
      const Ux<C> b0 = ldb(t, 0), b1 = ldb(t, 1);
#pragma unroll
      for (int q = 0; q < N; ++q) sb.v[q] = b0.v[q] + b1.v[q];
    }
#if RX_HOST_CHECK
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) d[i + j] += (u64)sa.v[i] * sb.v[j];          // mod 2^64 on purpose
#else
    ux_acc<C>(d, sa, sb);
#endif
  }
#if RX_HOST_CHECK
  for (int k = 0; k < 2 * N; ++k)
    if (chk[k] != d[k]) g_rx_overflow = 1;                                      // the wrap-around pile is the true cross pile
#endif
  r.c1 = ux_redc<C>(d);
  return r;
}

// The symmetric squaring on the 29-bit form.  A pile holds three term-equivalents and the row has six -- (2, 2, 2) on odd coefficients,
// (1, 2, 2, 1) on even ones -- so it is TWO piles of two slots each, four slot products and four reductions (the 28-bit form: four and two):
//     even row:  A = 2 d0 + p0,   B = 2 d1 + p1           (the doubling inside the pile: the doubled slot's left operand is shifted)
//     odd row:   A = d0 + d1 (undoubled), B = 2 d2          c = 2 A + B
// one instruction stream for both kinds of rows: the lane's `twice` says whether pile A is doubled after its reduction (odd rows) or
// its first slot inside (even rows).  Operands below 2 p / 3 p: each pile reduces below 1.5 p, c below 4.5 p, then a quasi-reduction
// back below 2 p.  (The first version ran the doubled and the plain products as a three-slot and a two-slot pile: five slot products.)
//   lda(t, side, h) / ldb(t, side, h): slot t < 2 of pile A / B, side 0 / 1 = left / right operand, half h; zeros for an unused slot.
//   sha(t) / shb(t): 1 if the slot's left operand is doubled.
template <class C, class LA, class LB, class SA, class SB>
BGLS_HD Ux2<C> ux_sqr_dot3(LA&& lda, LB&& ldb, SA&& sha, SB&& shb, bool twice) {
  constexpr int N = C::RX_NL;
  auto left = [&](auto& ld, auto& sh, int t, int h) __attribute__((always_inline)) {
    Ux<C> a = ld(t, 0, h);
    const u32 s = sh(t) ? 1u : 0u;
#pragma unroll
    for (int q = 0; q < N; ++q) a.v[q] <<= s;
    return a;
  };
#ifdef MX_EXP_Q
  const Ux2<C> d = ux_dot_k2q<C, 2>([&](int t, int h) { return left(lda, sha, t, h); }, [&](int t, int h) { return lda(t, 1, h); });
  const Ux2<C> p = ux_dot_k2q<C, 2>([&](int t, int h) { return left(ldb, shb, t, h); }, [&](int t, int h) { return ldb(t, 1, h); });
#else
  const Ux2<C> d = ux_dot_k2p<C, 2>([&](int t, int h) { return left(lda, sha, t, h); }, [&](int t, int h) { return lda(t, 1, h); });
  const Ux2<C> p = ux_dot_k2p<C, 2>([&](int t, int h) { return left(ldb, shb, t, h); }, [&](int t, int h) { return ldb(t, 1, h); });
#endif
  const u32 ps = twice ? 1u : 0u;
  Ux2<C> s;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    s.c0.v[i] = (d.c0.v[i] << ps) + p.c0.v[i];
    s.c1.v[i] = (d.c1.v[i] << ps) + p.c1.v[i];
  }
  return ux_quasi<C, 2, 1>(s);
}

// ---- conversions between the library's form (32-bit limbs, Montgomery radix R = 2^(32 L)) and this one
// x R (reduced, 32-bit limbs) -> x R' (tight, value < 2 p): split into 28-bit limbs, one product by R'^2 / R
template <class C>
BGLS_HD Ux<C> to_ux(const Fp<C>& y) {
  constexpr int N = C::RX_NL;
  Ux<C> s;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int lo = C::RX_W * i, q = lo >> 5, r = lo & 31;
    u64 two = q < C::L ? (u64)y.v[q] : 0;
    if (q + 1 < C::L) two |= (u64)y.v[q + 1] << 32;
    s.v[i] = (u32)(two >> r) & C::RX_MASK;
  }
  u64 c[2 * N];
  ux_acc_new<C>(c, s, ux_load<C>(C::RX_TO));
  return ux_redc<C>(c);
}
template <class C>
BGLS_HD Ux2<C> to_ux(const Fp2<C>& a) {
  return {to_ux<C>(a.c0), to_ux<C>(a.c1)};
}
// x R' (tight, value < 4 p: a reduction's output) -> x R in the library's form
template <class C>
BGLS_HD Fp<C> from_ux(const Ux<C>& a) {
  constexpr int N = C::RX_NL;
  constexpr int L = C::L;
  u32 w[L + 1];
#pragma unroll
  for (int k = 0; k <= L; ++k) {
    const int lo = 32 * k;
    const int i = lo / C::RX_W, r = lo % C::RX_W;
    u64 acc = 0;
    if (i < N) acc = (u64)a.v[i] >> r;
    int have = C::RX_W - r;
    if (i + 1 < N) { acc |= (u64)a.v[i + 1] << have; have += C::RX_W; }
    if (have < 32 && i + 2 < N) acc |= (u64)a.v[i + 2] << have;
    w[k] = (u32)acc;
  }
  // subtract 2p, then p, where possible (value < 4 p < 2^(32 L + 1))
#pragma unroll
  for (int sh = 1; sh >= 0; --sh) {
    u32 d[L + 1];
    u32 bw = 0;
#pragma unroll
    for (int k = 0; k <= L; ++k) {
      const u32 pk = (k < L ? (C::P[k] << sh) : 0u) | ((sh && k >= 1) ? (C::P[k - 1] >> (32 - sh)) : 0u);
      d[k] = subb(w[k], pk, bw);
    }
    const u32 keep = bw ? 0u : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k <= L; ++k) w[k] = (d[k] & keep) | (w[k] & ~keep);
  }
  Fp<C> y;
#pragma unroll
  for (int k = 0; k < L; ++k) y.v[k] = w[k];
  return fp_mul<C>(y, fp_load<C>(C::RX_BACK));
}
template <class C>
BGLS_HD Fp2<C> from_ux(const Ux2<C>& a) {
  return {from_ux<C>(a.c0), from_ux<C>(a.c1)};
}

// ===================================================================================================== signed, bounded
// |limb i| < LB * 2^24 for i < NL-1 (tight: limbs in [0, 2^28), LB = 16); the top limb is small (|value| stays below ~64 p
// everywhere in the point steps: R' / p > 2^11 leaves room for products of such values).
template <class C, int LB>
struct Sx {
  i32 v[C::RX_NL];
};
constexpr int SX_T = 16;       // tight
constexpr int SX_F = 17;       // after one parallel carry step (sx_normf)

template <class C, int LA, int LB>
BGLS_HD Sx<C, LA + LB> sx_add(const Sx<C, LA>& a, const Sx<C, LB>& b) {
  static_assert(LA + LB < 128, "limb leaves 31 bits");
  Sx<C, LA + LB> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
template <class C, int LA, int LB>
BGLS_HD Sx<C, LA + LB> sx_sub(const Sx<C, LA>& a, const Sx<C, LB>& b) {
  static_assert(LA + LB < 128, "limb leaves 31 bits");
  Sx<C, LA + LB> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = a.v[i] - b.v[i];
  return r;
}
template <class C, int LA>
BGLS_HD Sx<C, LA> sx_neg(const Sx<C, LA>& a) {
  Sx<C, LA> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = -a.v[i];
  return r;
}
template <int K, class C, int LA>
BGLS_HD Sx<C, K * LA> sx_mulc(const Sx<C, LA>& a) {
  static_assert(K * LA < 128, "limb leaves 31 bits");
  Sx<C, K * LA> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = K * a.v[i];
  return r;
}
template <class C, int LA>
BGLS_HD Sx<C, LA> sx_select(bool c, const Sx<C, LA>& a, const Sx<C, LA>& b) {
  Sx<C, LA> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
// widen the static bound (no code)
template <int LB, class C, int LA>
BGLS_HD Sx<C, LB> sx_as(const Sx<C, LA>& a) {
  static_assert(LB >= LA, "bound can only grow");
  Sx<C, LB> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = a.v[i];
  return r;
}
// full carry propagation: limbs 0..NL-2 into [0, 2^28), the (signed) top limb takes the rest
template <class C, int LA>
BGLS_HD Sx<C, SX_T> sx_norm(const Sx<C, LA>& a) {
  Sx<C, SX_T> r;
  i32 c = 0;
#pragma unroll
  for (int i = 0; i < C::RX_NL - 1; ++i) {
    const i32 t = a.v[i] + c;
    r.v[i] = t & (i32)C::RX_MASK;
    c = t >> C::RX_W;
  }
  r.v[C::RX_NL - 1] = a.v[C::RX_NL - 1] + c;
  return r;
}
// one parallel carry step (no dependency chain): limbs into (-2^4, 2^28 + 2^4)
template <class C, int LA>
BGLS_HD Sx<C, SX_F> sx_normf(const Sx<C, LA>& a) {
  static_assert(LA < 128, "limb leaves 31 bits");
  Sx<C, SX_F> r;
  constexpr int W = C::RX_W;
  constexpr i32 RX_MASK = (i32)C::RX_MASK;
  r.v[0] = a.v[0] & RX_MASK;
#pragma unroll
  for (int i = 1; i < C::RX_NL - 1; ++i) r.v[i] = (a.v[i] & RX_MASK) + (a.v[i - 1] >> W);
  r.v[C::RX_NL - 1] = a.v[C::RX_NL - 1] + (a.v[C::RX_NL - 2] >> W);
  return r;
}
// a / 2 mod p: add p when odd (the parity of the value is the parity of limb 0), then shift every limb, the dropped bit of
// limb i+1 entering limb i at 2^27
template <class C, int LA>
BGLS_HD Sx<C, (LA + 16 + 1) / 2 + 8> sx_half(const Sx<C, LA>& a) {
  constexpr int N = C::RX_NL;
  const i32 odd = -(a.v[0] & 1);
  i32 t[N];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = a.v[i] + ((i32)C::RX_P[i] & odd);
  Sx<C, (LA + 16 + 1) / 2 + 8> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = (t[i] >> 1) + (i + 1 < N ? (t[i + 1] & 1) << (C::RX_W - 1) : 0);
  return r;
}

// Interleaved Montgomery product (coarsely integrated operand scanning on carry-free columns): the running accumulator
// is NL + 1 columns of 64 bits -- 30 registers on BLS12-381 where a full double-width pile is 56 -- and one call folds
// up to four limb-product rows per step into it before the step's reduction row:
//      t <- (t + sum_k a_k[i] * b_k + m p) / 2^28,   m = -t[0] / p mod 2^28,    i = 0 .. NL-1
// Result: sum_k a_k b_k / R' mod p, tight limbs, signed top limb, value in (T / R', T / R' + p).
// Static budget: a column collects at most NL rows of sum_k LA_k LB_k 2^48 plus NL 2^56 of the reduction's own products
// and the carries; the total must stay below 2^63.
template <class C, int B>
struct MontAcc {
  // units 2^(2 W - 8) (bounds count 2^(W - 4)): NL rows of the products' bounds plus the reduction's own 2^(2 W) and the carries, below 2^63
  static_assert((long long)C::RX_NL * (B + 256) + 64 < (1ll << (71 - 2 * C::RX_W)), "column budget (signed 64-bit)");
};
// NP products (1..4).  cols[k] = the limbs of product k's COLUMN factor (register-resident); row(k, i) = limb i of its ROW
// factor, produced where it is used -- the point steps derive it from a neighbour lane's register, so a row factor never
// occupies NL registers.  BUDGET = sum over the products of (column bound) * (row bound), units 2^48.
template <class C, int NP, int BUDGET, class Row>
BGLS_HD Sx<C, SX_T> sx_montr(const i32* const (&cols)[NP], Row&& row) {
  constexpr int N = C::RX_NL;
  static_assert(NP >= 1 && NP <= 4, "products per reduction");
  (void)sizeof(MontAcc<C, BUDGET>);
  // t[i .. i+N) are the live columns of row i (column k is dead once row k is done): written as one array of 2 N columns so
  // that no value ever moves between registers -- a shifting window of N + 1 columns cost a v_mov_b64 per column and row.
  // Columns are written for the first time by the products themselves (row 0: all of them, row i: its last one), never zeroed.
  i64 t[2 * N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    i32 r[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) r[k] = row(k, i);
    // column 0 first: the row's Montgomery factor m depends on it and is ready by the time the row's other products are issued
    if (i == 0) rx_muls(t[0], r[0], cols[0][0]);
    else rx_macs(t[i], r[0], cols[0][0]);
#pragma unroll
    for (int k = 1; k < NP; ++k) rx_macs(t[i], r[k], cols[k][0]);
    const i32 m = (i32)(((u32)t[i] * C::RX_NP) & C::RX_MASK);
    if (i == 0) rx_rows_new<N - 1, 1>(t + 1, r[0], cols[0] + 1);
    else rx_rows_new<N - 1, 2>(t + i + 1, r[0], cols[0] + 1);
#pragma unroll
    for (int k = 1; k < NP; ++k) rx_rows<N - 1, false>(t + i + 1, r[k], cols[k] + 1);
    rx_rows<N, true>(t + i, m, (const i32*)C::RX_P);
    t[i + 1] += t[i] >> C::RX_W;
  }
  Sx<C, SX_T> r;
#pragma unroll
  for (int k = N; k < 2 * N - 1; ++k) {
    r.v[k - N] = (i32)((u32)t[k] & C::RX_MASK);
    if (k + 1 < 2 * N - 1) t[k + 1] += t[k] >> C::RX_W;
  }
  r.v[N - 1] = (i32)(t[2 * N - 2] >> C::RX_W);     // column 2 NL - 1 receives this carry and nothing else
  return r;
}

// Centred quasi-reduction of a signed value: subtract round(value / p) p, read off the top limb, and carry-normalise -- tight limbs, signed
// top limb, |result| below 0.51 p.  a: limbs of any bound below 2^31 (a few tight values added up or multiplied by small constants), |value|
// below 2^6 p.  The point steps use it where a product by a SMALL curve constant is formed with additions instead of a Montgomery product
// (BLS12-381's 3 b' = 12 (1 + i), rx_pair.hpp): the sum is 25 p large, and every later use -- the P-free line coefficient E - B above all,
// which must reach the consumer below 32 p -- wants the small representative a reduction would have returned.  NL multiplier instructions.
template <class C, int LA>
BGLS_HD Sx<C, SX_T> sx_quasi_center(const Sx<C, LA>& a) {
  constexpr int N = C::RX_NL;
  constexpr int W = C::RX_W;
  constexpr float QI = 1.0f / ((float)C::RX_P[N - 1] + 0.5f);
  // the limbs below the top one are worth a few units of the top limb together (LA / 16 of them); p's top limb is 2^17 (BLS12-381) / 2^21 (alt-bn128) units
#if RX_HOST_CHECK
  // preconditions (advice r5): |value| below 2^6 p, and a top limb the single-precision conversion holds exactly
  {
    const i64 top = a.v[N - 1] < 0 ? -(i64)a.v[N - 1] : (i64)a.v[N - 1];
    if (top >= (1 << 24) || top > 64 * (i64)C::RX_P[N - 1] + 64) g_rx_overflow = 1;
  }
#endif
  const float qf = (float)a.v[N - 1] * QI;
  const i32 q = (i32)(qf + (qf < 0.0f ? -0.5f : 0.5f));
  Sx<C, SX_T> r;
  i64 c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    const i64 t = (i64)a.v[i] - (i64)q * (i64)(i32)C::RX_P[i] + c;
    r.v[i] = (i32)((u32)t & C::RX_MASK);
    c = t >> W;
  }
  r.v[N - 1] = (i32)((i64)a.v[N - 1] - (i64)q * (i64)(i32)C::RX_P[N - 1] + c);
#if RX_HOST_CHECK
  {          // postcondition: the centred representative, |value| below 0.51 p
    const i64 top = r.v[N - 1] < 0 ? -(i64)r.v[N - 1] : (i64)r.v[N - 1];
    if (2 * top > (i64)C::RX_P[N - 1] + (i64)C::RX_P[N - 1] / 32 + 64) g_rx_overflow = 1;
  }
#endif
  return r;
}

// value == 0 (mod p)?  a: limbs below 2^30, value in (-3 p, 5 p) (differences of a few reductions' outputs).  v + 3p lies in
// (0, 8p): after a full carry it equals one of 0, p, .. 8p limb for limb iff v is a multiple of p.
template <class C, int LA>
BGLS_HD bool sx_is_zero_mod_p(const Sx<C, LA>& a) {
  constexpr int N = C::RX_NL;
  Sx<C, LA + 32> t;
#pragma unroll
  for (int i = 0; i < N; ++i) t.v[i] = a.v[i] + (i32)C::RX_PK[3 * N + i];   // + 3p (tight limbs)
  const Sx<C, SX_T> n = sx_norm<C>(t);
  bool hit = false;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) d |= (u32)n.v[i] ^ C::RX_PK[k * N + i];
    hit = hit || d == 0;
  }
  return hit;
}
// plain integer (canonical, < p, 32-bit limbs) -> R' form, tight: split into 28-bit limbs, one product by R'^2
template <class C>
BGLS_HD Sx<C, SX_T> sx_from_plain(const Fp<C>& y);

// signed, bounded  ->  tight and NON-NEGATIVE (the form LDS holds): add the fat multiple of p that dominates every limb,
// then carry.  |value| must be below K RX_FAT_VB p with K = ceil(LA / 16) (the point steps hand over reductions' outputs
// and one difference of two: |value| < 2.2 p, K <= 2); the result is below (2 K RX_FAT_VB + 2) p + |value| <= 21 p.
template <class C, int LA>
BGLS_HD Ux<C> sx_to_ux(const Sx<C, LA>& a) {
  constexpr int K = (LA + 15) / 16;
  static_assert(K >= 1 && K <= C::RX_FAT_KMAX, "fat constants cover limbs below RX_FAT_KMAX * 2^W");
  Ux<C> t;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) t.v[i] = (u32)((i32)C::RX_FAT[(K - 1) * C::RX_NL + i] + a.v[i]);
  return ux_norm<C>(t);
}
// the same with the fat multiple chosen by the caller: K 2^W must dominate the limbs' magnitudes (a difference of two tight
// non-negative values has limbs inside (-2^W, 2^W) whatever its static bound says) and K RX_FAT_VB p the value's negative side
template <int K, class C, int LA>
BGLS_HD Ux<C> sx_to_ux_k(const Sx<C, LA>& a) {
  static_assert(K >= 1 && K <= C::RX_FAT_KMAX, "fat constants");
  Ux<C> t;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) t.v[i] = (u32)((i32)C::RX_FAT[(K - 1) * C::RX_NL + i] + a.v[i]);
  return ux_norm<C>(t);
}
// a reduction's output (limbs 0..NL-2 non-negative, value above -p): adding p itself makes it non-negative; result below value + p
template <class C>
BGLS_HD Ux<C> sx_to_ux_p(const Sx<C, SX_T>& a) {
  Ux<C> t;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) t.v[i] = (u32)((i32)C::RX_P[i] + a.v[i]);
  return ux_norm<C>(t);
}
template <class C>
BGLS_HD Sx<C, SX_T> ux_to_sx(const Ux<C>& a) {
  Sx<C, SX_T> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = (i32)a.v[i];
  return r;
}
template <class C>
BGLS_HD Sx<C, SX_T> sx_const(const u32* k) {
  Sx<C, SX_T> r;
#pragma unroll
  for (int i = 0; i < C::RX_NL; ++i) r.v[i] = (i32)k[i];
  return r;
}

template <class C>
BGLS_HD Sx<C, SX_T> sx_from_plain(const Fp<C>& y) {
  constexpr int N = C::RX_NL;
  i32 s[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int lo = C::RX_W * i, q = lo >> 5, r = lo & 31;
    u64 two = q < C::L ? (u64)y.v[q] : 0;
    if (q + 1 < C::L) two |= (u64)y.v[q + 1] << 32;
    s[i] = (i32)((u32)(two >> r) & C::RX_MASK);
  }
  const Sx<C, SX_T> k = sx_const<C>(C::RX_R2);
  const i32* const cols[1] = {k.v};
  return sx_montr<C, 1, SX_T * SX_T>(cols, [&](int, int i) { return s[i]; });
}

// tight, non-negative, value < 4 p  ->  the canonical residue in [0, p) as a plain integer in the library's 32-bit words (no multiplication:
// repack, subtract 2 p and p where possible).  With RX_TOM folded into the last carry-free product this IS the library's Montgomery residue.
template <class C>
BGLS_HD Fp<C> ux_to_words(const Ux<C>& a) {
  constexpr int N = C::RX_NL;
  constexpr int L = C::L;
  u32 w[L + 1];
#pragma unroll
  for (int k = 0; k <= L; ++k) {
    const int lo = 32 * k;
    const int i = lo / C::RX_W, r = lo % C::RX_W;
    u64 acc = 0;
    if (i < N) acc = (u64)a.v[i] >> r;
    int have = C::RX_W - r;
    if (i + 1 < N) { acc |= (u64)a.v[i + 1] << have; have += C::RX_W; }
    if (have < 32 && i + 2 < N) acc |= (u64)a.v[i + 2] << have;
    w[k] = (u32)acc;
  }
#pragma unroll
  for (int sh = 1; sh >= 0; --sh) {
    u32 d[L + 1];
    u32 bw = 0;
#pragma unroll
    for (int k = 0; k <= L; ++k) {
      const u32 pk = (k < L ? (C::P[k] << sh) : 0u) | ((sh && k >= 1) ? (C::P[k - 1] >> (32 - sh)) : 0u);
      d[k] = subb(w[k], pk, bw);
    }
    const u32 keep = bw ? 0u : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k <= L; ++k) w[k] = (d[k] & keep) | (w[k] & ~keep);
  }
  Fp<C> y;
#pragma unroll
  for (int k = 0; k < L; ++k) y.v[k] = w[k];
  return y;
}

// x R' (tight, value < 4 p) -> x R in the library's form, everything expanded in place (no call, no stack: the Miller kernel's
// epilogue must not drag a scratch frame along)
template <class C>
BGLS_HD Fp<C> from_ux_inl(const Ux<C>& a) {
  constexpr int N = C::RX_NL;
  constexpr int L = C::L;
  u32 w[L + 1];
#pragma unroll
  for (int k = 0; k <= L; ++k) {
    const int lo = 32 * k;
    const int i = lo / C::RX_W, r = lo % C::RX_W;
    u64 acc = 0;
    if (i < N) acc = (u64)a.v[i] >> r;
    int have = C::RX_W - r;
    if (i + 1 < N) { acc |= (u64)a.v[i + 1] << have; have += C::RX_W; }
    if (have < 32 && i + 2 < N) acc |= (u64)a.v[i + 2] << have;
    w[k] = (u32)acc;
  }
#pragma unroll
  for (int sh = 1; sh >= 0; --sh) {
    u32 d[L + 1];
    u32 bw = 0;
#pragma unroll
    for (int k = 0; k <= L; ++k) {
      const u32 pk = (k < L ? (C::P[k] << sh) : 0u) | ((sh && k >= 1) ? (C::P[k - 1] >> (32 - sh)) : 0u);
      d[k] = subb(w[k], pk, bw);
    }
    const u32 keep = bw ? 0u : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k <= L; ++k) w[k] = (d[k] & keep) | (w[k] & ~keep);
  }
  Fp<C> y;
#pragma unroll
  for (int k = 0; k < L; ++k) y.v[k] = w[k];
  return fp_mul_inl<C>(y, fp_load<C>(C::RX_BACK));
}

}  // namespace bgls
