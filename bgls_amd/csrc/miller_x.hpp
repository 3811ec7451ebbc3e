// k_miller_x60: the fused Miller-loop kernel on carry-free 28-bit limbs, both curves (rx.hpp, rx_pair.hpp).
//
// Block = 3 waves, 60 pairings, 168 registers per lane (three waves per SIMD, four blocks per CU):
//   two PRODUCER waves  30 pairings each, one pairing per LANE PAIR (rx_pair.hpp): parse the key, walk the G2 point steps,
//                       scale the line by the hash point and hand its three Fp2 coefficients over through LDS
//   one CONSUMER wave   10 groups x 6 lanes; a group shares ONE Fp12 accumulator among its six pairings
//                       (f <- f^2 l_1 ... l_6, coop.hpp); lane j owns coefficient j of  f = sum e_j w^j
// Per line step:   producers: point step (lines stay in registers) | barrier A | store lines | barrier B
//                  consumer:  fold six lines | f <- f^2 (doubling steps) | barrier A | barrier B
// so the producers' step s+1 overlaps the consumer's folds of step s AND its squaring (round 6: rounds 3-5 squared between A
// and B, where both producer waves parked for it -- 15 % of their time by the barrier stamps of tools/mb_stamps.hip; ahead of A
// nobody does: 53.9 -> 53.3 ms alt-bn128, 84.3 -> 83.1 ms BLS12-381 per 2^20 pairings, same bytes); one line buffer instead of
// two (38.6 KB of LDS per block: four blocks per CU).
//
// Why the hand-over stays two block barriers on ONE buffer (round 6, profiles/r6/miller_handover.md).  The stamps show who waits:
// the consumer arrives last at A in 70 % of the steps and waits there 9.5 % of its time -- all of it in the first twenty steps of
// a block's life, when its two producers are the YOUNGEST waves of their SIMDs (the issue arbiter serves the oldest ready wave
// first: a block's period falls from 112 k clocks to 62 k as it ages).  -DMX_EXP_FLAGS=D replaces the barriers by LDS sequence
// words and lets the producers store D / 2 steps ahead (wrong results beyond D = 2: the single buffer is overwritten -- timing
// only): D = 2 / 3 / 4 / 6 measured 53.3 / 53.2 / 52.9 / 52.6 ms (alt-bn128), 84.2 / 84.6 / 84.6 / 84.2 ms (BLS12-381) -- a ring of
// three half-step slots, the most that 160 KB of LDS hold at four blocks per CU, would buy 0.3 %, two whole buffers 0.9 %.
// Likewise the consumer's exposed LDS round trips: fetching every operand one stage ahead (-DMX_EXP_Q, rx.hpp ux_dot_k2q) takes a
// consumer wave ALONE from 38.9 to 34.9 ms and the whole kernel from 53.9 to 55.2 ms: the old blocks' consumers issue denser and
// the young blocks' producers starve longer.  What bounds the kernel is instruction issue: 2.73e10 vector instructions per launch
// at the 4.4-4.5 clocks a mixed stream of three waves per SIMD pays per instruction (probe in tools/mb_stamps.hip) are 50 of the 54 ms.
//
// What the 32-bit kernels (k_miller_ab64 / k_miller_s60) lose and this one does not: three VALU instructions per multiplier
// instruction (carry add, re-zeroed addend) -- here every dot product is bare v_mad_u64_u32 into 64-bit columns; a producer
// lane holding a whole G2 point plus three piles (256 registers, 150-650 spilled) -- here half a point and one pile.
//
// Work per pairing in units of NL^2 multiplier instructions (NL = 10 / 14): producer 2 x (27 per doubling since round 6 -- 3 E^2 as a square: 53.8 -> 53.5 ms alt-bn128, 82.8 -> 82.0 ms BLS12-381 per 2^20 pairings, same box -- 39 per
// addition step), consumer 66 per line + 84 / 6 per squaring (round 4: the squaring's cross terms in the Karatsuba form, rx.hpp).  Same line coefficients and the same product order as the
// other kernels: the partial products are bit-identical to k_miller_ab64's.
//
// Replaces the n calls of CurveSystem.Pair behind PairingProduct: curves/curve.go:125-170, curves/altbn128.go:130-145,
// curves/bls12_381.go:228-240.
#pragma once
#include <type_traits>
#include "dev_common.hpp"
#include "coop.hpp"
#include "rx_pair.hpp"

namespace bgls {

// LDS layout of a block (round 4).  The consumer's operand fetches are ds_read_b128, which the LDS serves in four fixed groups
// of sixteen lanes, conflict-free when the sixteen 16-byte slots differ mod 256 bytes (MI355X_MICROARCH.md, LDS).  Round 3's
// layout (lane = 6 g + j, entries 24 / 32 dwords apart) spent half (alt-bn128) to two thirds (BLS12-381) of its LDS cycles
// on conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.50 / 0.70).  tools/lds_conflicts.py models the accesses; with
//   * consumer lane = 10 j + g (coefficient-major: the lanes of a 16-lane group are runs of consecutive groups g of two or
//     three coefficients j),
//   * slot strides (in 16-byte slots mod 16): group GS, accumulator entry KS, xi copy WS with WS = -6 KS -- the entry a lane
//     reads for shift s is then slot KS (j - s) whether it wraps or not, the same pattern for every shift,
//   * (GS, KS, WS) one of the eight solutions of the exhaustive search, e.g. (15, 6, 12) and (3, 14, 12),
// every fetch of a line fold is conflict-free and the squaring's table-driven fetches cost about a third extra.
//   alt-bn128  (halves padded to 12 dwords): plain e_k at 24 k, xi e_k at 176 + 24 k, lines at 320 + 24 e, GROUP_DW 764 / 828
//   BLS12-381  (halves packed, 14 dwords; a half that starts 8 bytes off a 16-byte boundary is fetched from 8 bytes before):
//              plain e_k at 56 k, xi e_k at 368 + 56 k, the lines in the 28-dword gaps between them, GROUP_DW 844 / 940
// which also frees the 6.7 KB that hold the BLS12-381 hash points' coordinates (round 3 re-read them from HBM every step).
//
// NP = pairings per block: 60 (six lines per group and step) or 64 (all 32 lane pairs of both producer waves: groups 0..3 take
// a seventh line, the other groups' seventh line is the constant 1) -- 1024 resident blocks of the 64-form are exactly 2^16
// pairings, the batch a lone VerifyAggregateSignature of BASELINE configs 2 / 3 submits.
// The number form a curve's kernel runs on: alt-bn128 on nine limbs of 29 bits (BN254W, round 5: 81 instead of 100 multiplier
// instructions per limb product), BLS12-381 on fourteen of 28 (thirteen of 30 leave no head-room in a 64-bit column at all).
// -DMX_BN_W28 keeps alt-bn128 on ten limbs of 28 bits (A/B measurements).
template <class C>
struct MxForm { typedef C type; };
#ifndef MX_BN_W28
template <>
struct MxForm<BN254> { typedef BN254W type; };
#endif

template <class C, int NP = 60>
struct MX {
  static_assert(NP == 60 || NP == 64, "pairings per block");
  static constexpr int NL = C::RX_NL;
  static constexpr bool PACKED = (C::RX_NL % 4) != 0 && C::CURVE_ID == 1;   // BLS12-381: 14-dword halves back to back
  static constexpr int HS = PACKED ? NL : ((NL + 3) & ~3);                  // dwords per half: 12 / 14
  static constexpr int ES = 2 * HS;                                         // per Fp2 entry: 24 / 28
  static constexpr int NLINES = NP == 64 ? 7 : 6;
  static constexpr int FOLD_UNROLL = (!rx_lazy<C> && NLINES % 3 == 0) ? 3 : 1;                   // the consumer's loop over a step's lines (k_miller_x60)
  static constexpr int KS = PACKED ? 2 * ES : ES;                           // accumulator entry stride: 24 / 56
  static constexpr int WS = PACKED ? 368 : 176;                             // xi copies
  static constexpr int GROUP_DW = PACKED ? (NP == 64 ? 940 : 844) : (NP == 64 ? 828 : 764);
  static constexpr int THREADS = 192;
  // accumulator entry (k, wrap) and line entry e = 3 m + t of a group, in dwords from the group's base
  static __device__ __forceinline__ int acc_off(int k, int wrap) { return k * KS + wrap * WS; }
  static __device__ __forceinline__ int line_off(int e) {
    if constexpr (PACKED) return e < 12 ? 28 + 56 * e + (e >= 6 ? 32 : 0) : (e == 12 ? 336 : 704 + 28 * (e - 13));
    else return 320 + ES * e;
  }
  static_assert(!PACKED || (ES == 28 && 704 + 28 * (3 * NLINES - 13) <= GROUP_DW), "BLS12-381 line slots");
  static_assert(PACKED || 320 + ES * 3 * NLINES <= GROUP_DW, "alt-bn128 line slots");
  // the hash points' coordinates (-yP, xP per pairing, read by both lanes of its pair at every step) live in LDS where a
  // quarter of a CU's 160 KB has room for them next to the groups, else in the lanes' global workspace (BLS12-381, NP = 64)
  static constexpr int PQ = 10 * GROUP_DW;            // [NP pairings][2] halves
#ifndef MX_P_LDS_LIMIT
#define MX_P_LDS_LIMIT 40960
#endif
  static constexpr bool P_IN_LDS = (10 * GROUP_DW + NP * 2 * HS) * 4 <= MX_P_LDS_LIMIT;
  static constexpr int SEQ_DW = 10 * GROUP_DW + (P_IN_LDS ? NP * 2 * HS : 0);          // MX_EXP_FLAGS: four sequence words behind everything else
#ifdef MX_EXP_FLAGS
  static constexpr int BLOCK_BYTES = (SEQ_DW + 16) * 4;
#else
  static constexpr int BLOCK_BYTES = (10 * GROUP_DW + (P_IN_LDS ? NP * 2 * HS : 0)) * 4;
#endif
  static constexpr int NPARK_Q = C::CURVE_ID == 0 ? 6 : 2;                  // xq yq [x1 y1 x2 y2]
  static constexpr int NPARK = NPARK_Q + (P_IN_LDS ? 0 : 2);                // ... nyP xP   (PS dwords each)
  static constexpr int PS = (NL + 3) & ~3;                                  // parked values stay 16-byte aligned
  static constexpr size_t park_bytes(size_t nblocks) { return nblocks * 128 * NPARK * PS * 4; }
};

// one half (NL limbs) from / to LDS.  `second`: the half is the c1 of a packed entry (starts 8 bytes off a 16-byte boundary).
// Fetches are ds_read_b128 ONLY: the compiler is told that the address is 16-byte aligned (without that it emits ds_read2_b64
// -- two 8-byte accesses per instruction at half the LDS rate, banked mod 32 -- for every 16-byte load from the unsized
// extern array: that is what round 3's kernel ran on).  The load that holds a half's last two limbs is narrowed by the
// compiler to a ds_read_b64 (served in two 32-lane groups; the 8-byte slots of 16-byte aligned entries all have the same
// parity, so it is two-way conflicted: 4 LDS cycles, what the full ds_read_b128 would cost -- keeping it wide needs an asm
// use of the unused dwords, which puts a wait behind every load, or a volatile access, which loses the LDS address space).
__device__ __forceinline__ uint4 mx_ld16(int dw) {
  extern __shared__ u32 lds[];
  typedef u32 v4u __attribute__((ext_vector_type(4)));      // a native vector: HIP's uint4 is a struct, whose load is split into scalars
  const v4u v = *reinterpret_cast<const v4u*>(__builtin_assume_aligned(lds + dw, 16));
  return make_uint4(v.x, v.y, v.z, v.w);
}
template <class C, bool PACKED>
__device__ __forceinline__ Ux<C> mx_ld_half(int off, bool second) {
  constexpr int N = C::RX_NL;
  Ux<C> r;
  if constexpr (!PACKED) {
    (void)second;
#pragma unroll
    for (int k = 0; k < (N + 3) / 4; ++k) {
      const uint4 v = mx_ld16(off + 4 * k);
      r.v[4 * k] = v.x;
      if (4 * k + 1 < N) r.v[4 * k + 1] = v.y;
      if (4 * k + 2 < N) r.v[4 * k + 2] = v.z;
      if (4 * k + 3 < N) r.v[4 * k + 3] = v.w;
    }
  } else {
    static_assert(!PACKED || N % 4 == 2, "packed halves: NL = 2 mod 4");
    if (!second) {
#pragma unroll
      for (int k = 0; k < (N + 2) / 4; ++k) {
        const uint4 v = mx_ld16(off + 4 * k);
        r.v[4 * k] = v.x; r.v[4 * k + 1] = v.y;
        if (4 * k + 2 < N) { r.v[4 * k + 2] = v.z; r.v[4 * k + 3] = v.w; }
      }
    } else {
      const uint4 v0 = mx_ld16(off - 2);
      r.v[0] = v0.z; r.v[1] = v0.w;
#pragma unroll
      for (int k = 1; k < (N + 2) / 4; ++k) {
        const uint4 v = mx_ld16(off - 2 + 4 * k);
        r.v[4 * k - 2] = v.x; r.v[4 * k - 1] = v.y; r.v[4 * k] = v.z; r.v[4 * k + 1] = v.w;
      }
    }
  }
  return r;
}
template <class C, bool PACKED>
__device__ __forceinline__ void mx_st_half(int off, bool second, const Ux<C>& a) {
  extern __shared__ u32 lds[];
  constexpr int N = C::RX_NL;
  if (!PACKED || !second) {
    uint4* p = reinterpret_cast<uint4*>(lds + off);
#pragma unroll
    for (int k = 0; k < N / 4; ++k) p[k] = make_uint4(a.v[4 * k], a.v[4 * k + 1], a.v[4 * k + 2], a.v[4 * k + 3]);
    if constexpr (N % 4 == 2) *reinterpret_cast<uint2*>(lds + off + (N & ~3)) = make_uint2(a.v[N - 2], a.v[N - 1]);
    if constexpr (N % 4 == 1) lds[off + N - 1] = a.v[N - 1];
    static_assert(N % 4 != 3, "tail store");
  } else {
    *reinterpret_cast<uint2*>(lds + off) = make_uint2(a.v[0], a.v[1]);
    uint4* p = reinterpret_cast<uint4*>(lds + off + 2);
#pragma unroll
    for (int k = 0; k < N / 4; ++k) p[k] = make_uint4(a.v[4 * k + 2], a.v[4 * k + 3], a.v[4 * k + 4], a.v[4 * k + 5]);
  }
}

// ---- consumer pieces (one output coefficient per lane); gb = the group's base.  K = the LDS layout of a group: MX<C, NP> for the Miller kernel,
// PrepX<C> (prepared.hpp) for the prepared-key fold -- PACKED, HS, acc_off(k, wrap), line_off(e).  XI3: only the xi copies of coefficients 3..5 are
// stored (nothing ever reads the others: every wrapped factor of a fold or a squaring is one of e_3, e_4, e_5); the Miller kernel stores all six
// because its layout has the room and a predicated store costs what it saves.
template <class C, class K, bool XI3 = false>
__device__ __forceinline__ void mxk_publish(int gb, int j, const Ux2<C>& v, bool live) {
  if (live) {
    mx_st_half<C, K::PACKED>(gb + K::acc_off(j, 0), false, v.c0);
    mx_st_half<C, K::PACKED>(gb + K::acc_off(j, 0) + K::HS, true, v.c1);
    const Ux2<C> x = ux_mulxi<C>(v);
    if (!XI3 || j >= 3) {
      mx_st_half<C, K::PACKED>(gb + K::acc_off(j, 1), false, x.c0);
      mx_st_half<C, K::PACKED>(gb + K::acc_off(j, 1) + K::HS, true, x.c1);
    }
  }
  wave_sync();
}
template <class C, int NP>
__device__ __forceinline__ void mx_publish(int gb, int j, const Ux2<C>& v, bool live) { mxk_publish<C, MX<C, NP>>(gb, j, v, live); }
// f <- f * line_m:  c_j = sum_t L[3m + t] * B[(j - sh[t]) mod 6] * xi^[sh[t] > j]
template <class C, int NP>
__device__ __forceinline__ Ux2<C> mx_fold(int gb, int m, int j) {
  typedef MX<C, NP> K;
  // powers of w the line's three entries sit at: {0, 1, 3} (D-type twist) / {0, 2, 3} (M-type), as arithmetic on t: a table in
  // constant memory costs a scalar load and a wait for it in front of every operand fetch
  auto sh = [](int t) { return C::TWIST_D ? t + (t == 2 ? 1 : 0) : t + (t >= 1 ? 1 : 0); };
#ifdef MX_EXP_Q
  if constexpr (C::RX_NL <= 10)
    return ux_dot_k2q<C, 3>(
      [&](int t, int h) { return mx_ld_half<C, K::PACKED>(gb + K::line_off(3 * m + t) + h * K::HS, h != 0); },
      [&](int t, int h) {
        int k = j - sh(t);
        const int wrap = k < 0 ? 1 : 0;
        k += 6 * wrap;
        return mx_ld_half<C, K::PACKED>(gb + K::acc_off(k, wrap) + h * K::HS, h != 0);
      });
  else
#endif
  return ux_dot_k2p<C, 3, (C::RX_NL <= 10)>(
      [&](int t, int h) { return mx_ld_half<C, K::PACKED>(gb + K::line_off(3 * m + t) + h * K::HS, h != 0); },
      [&](int t, int h) {
        int k = j - sh(t);
        const int wrap = k < 0 ? 1 : 0;
        k += 6 * wrap;
        return mx_ld_half<C, K::PACKED>(gb + K::acc_off(k, wrap) + h * K::HS, h != 0);
      });
}
// f <- f^2 with the symmetric terms merged (COOP_SQ_TAB)
template <class C, class K>
__device__ __forceinline__ Ux2<C> mxk_sqr(int gb, int j) {
  const unsigned row = COOP_SQ_TAB[j];
  // table entry per slot: bits 0-2 = i (7 = unused), bits 3-5 = k, bit 6 = wrap (xi copy), bit 7 = doubled
  return ux_sqr_dot<C>(
      [&](int t) {
        const unsigned e = (row >> (8 * t)) & 0xFFu;
        return (e & 7u) == 7u ? 0 : (((e >> 7) & 1u) ? 2 : 1);
      },
      [&](int t, int h) {
        const unsigned e = (row >> (8 * t)) & 0xFFu;
        const int i = (e & 7u) == 7u ? 0 : (int)(e & 7u);
        return mx_ld_half<C, K::PACKED>(gb + K::acc_off(i, 0) + h * K::HS, h != 0);
      },
      [&](int t, int h) {
        const unsigned e = (row >> (8 * t)) & 0xFFu;
        const bool unused = (e & 7u) == 7u;
        return mx_ld_half<C, K::PACKED>(gb + K::acc_off(unused ? 0 : (int)((e >> 3) & 7u), unused ? 0 : (int)((e >> 6) & 1u)) + h * K::HS, h != 0);
      });
}
template <class C, int NP>
__device__ __forceinline__ Ux2<C> mx_sqr(int gb, int j) { return mxk_sqr<C, MX<C, NP>>(gb, j); }

// The 29-bit form's squaring (rx.hpp ux_sqr_dot3): the row's slots as two piles of two.  The split of the row is a per-lane constant, made once
// before the step loop: pa / pb = the table bytes of pile A's / pile B's two slots (0xff = unused); bit 16 of pa = "pile A is doubled after its
// reduction" (odd rows: A = d0 + d1 undoubled, B = 2 d2), else the first slot of each pile is doubled inside (even rows: A = 2 d0 + p0, B = 2 d1 + p1).
__device__ __forceinline__ void mx_sq_split(unsigned row, unsigned& pa, unsigned& pb) {
  unsigned d[3] = {0xffu, 0xffu, 0xffu}, p[2] = {0xffu, 0xffu};
  int nd = 0, np = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned e = (row >> (8 * t)) & 0xFFu;
    if ((e & 7u) == 7u) continue;
    if (e & 0x80u) {
      if (nd == 0) d[0] = e; else if (nd == 1) d[1] = e; else d[2] = e;
      ++nd;
    } else {
      if (np == 0) p[0] = e; else p[1] = e;
      ++np;
    }
  }
  if (nd == 3) {               // odd coefficient: (2, 2, 2)
    pa = d[0] | (d[1] << 8) | (1u << 16);
    pb = d[2] | (0xffu << 8);
  } else {                     // even coefficient: (1, 2, 2, 1)
    pa = d[0] | (p[0] << 8);
    pb = d[1] | (p[1] << 8);
  }
}
template <class C, class K>
__device__ __forceinline__ Ux2<C> mxk_sqr3(int gb, unsigned pa, unsigned pb) {
  const bool twice = (pa >> 16) & 1u;
  auto fetch = [&](unsigned sl, int t, int side, int h) __attribute__((always_inline)) {
    const unsigned e = (sl >> (8 * t)) & 0xFFu;
    const bool unused = (e & 7u) == 7u;
    const int i = unused ? 0 : (int)(e & 7u), k = unused ? 0 : (int)((e >> 3) & 7u), wrap = unused ? 0 : (int)((e >> 6) & 1u);
    Ux<C> a = mx_ld_half<C, K::PACKED>(gb + (side == 0 ? K::acc_off(i, 0) : K::acc_off(k, wrap)) + h * K::HS, h != 0);
    if (side == 0) {                                      // an unused slot's product vanishes with its left operand
      const u32 keep = unused ? 0u : 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < C::RX_NL; ++q) a.v[q] &= keep;
    }
    return a;
  };
  return ux_sqr_dot3<C>([&](int t, int side, int h) { return fetch(pa, t, side, h); }, [&](int t, int side, int h) { return fetch(pb, t, side, h); },
                        [&](int t) { return t == 0 && !twice; }, [&](int t) { return t == 0; }, twice);
}

template <class C, int NP>
__device__ __forceinline__ Ux2<C> mx_sqr3(int gb, unsigned pa, unsigned pb) { return mxk_sqr3<C, MX<C, NP>>(gb, pa, pb); }

// A producer lane's parked values: its own region of the workspace, NPARK slots of HS dwords (16-byte aligned), moved with
// 16-byte accesses off ONE address register.  (A lane-interleaved layout coalesces better but needs a separate 64-bit
// address per limb -- the element stride exceeds the instructions' immediate offsets -- and those addresses cost more
// registers than the values they fetch.)
template <class C>
struct MxPark {
  static constexpr int NL = C::RX_NL;
  static constexpr int HS = MX<C>::PS;                 // slot stride: 16-byte aligned
  static __device__ __forceinline__ void st_raw(u32* lane_base, int slot, const u32 (&v)[NL]) {
    uint4* p = reinterpret_cast<uint4*>(lane_base + slot * HS);
#pragma unroll
    for (int k = 0; k < NL / 4; ++k) p[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    if constexpr (NL % 4 == 2) *reinterpret_cast<uint2*>(lane_base + slot * HS + (NL & ~3)) = make_uint2(v[NL - 2], v[NL - 1]);
    if constexpr (NL % 4 == 1) lane_base[slot * HS + NL - 1] = v[NL - 1];
  }
  static __device__ __forceinline__ void ld_raw(const u32* lane_base, int slot, u32 (&v)[NL]) {
    const uint4* p = reinterpret_cast<const uint4*>(lane_base + slot * HS);
#pragma unroll
    for (int k = 0; k < NL / 4; ++k) {
      const uint4 q = p[k];
      v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
    }
    if constexpr (NL % 4 == 2) {
      const uint2 q = *reinterpret_cast<const uint2*>(lane_base + slot * HS + (NL & ~3));
      v[NL - 2] = q.x; v[NL - 1] = q.y;
    }
    if constexpr (NL % 4 == 1) v[NL - 1] = lane_base[slot * HS + NL - 1];
  }
  static __device__ __forceinline__ void st(u32* base, int slot, const Sx<C, SX_T>& a) {
    u32 w[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) w[k] = (u32)a.v[k];
    st_raw(base, slot, w);
  }
  static __device__ __forceinline__ Sx<C, SX_T> ld(const u32* base, int slot) {
    u32 w[NL];
    ld_raw(base, slot, w);
    Sx<C, SX_T> r;
#pragma unroll
    for (int k = 0; k < NL; ++k) r.v[k] = (i32)w[k];
    return r;
  }
  static __device__ __forceinline__ void st_u(u32* base, int slot, const Ux<C>& a) { st_raw(base, slot, a.v); }
  static __device__ __forceinline__ Ux<C> ld_u(const u32* base, int slot) {
    Ux<C> r;
    ld_raw(base, slot, r.v);
    return r;
  }
  static __device__ __forceinline__ const u32* launder(const u32* p) {       // opaque to the optimiser: no hoisting out of the step loop
    asm volatile("" : "+v"(p));
    return p;
  }
};

// rot_mode: how the three roles are dealt to the three waves of a block (the hardware places wave w of a block on some
// SIMD; rotating the roles from block to block keeps each SIMD's mix of producers and consumers even)
// DBG (development tools only, never instantiated in the library): 1 = producer work only, 2 = consumer work only -- wrong
// results, used to time the two roles separately.
// (Round 3's variant with two consumer waves per block -- 20 partial products, the squaring done twice -- measured slower in
// steady state and is gone.)
// DBG == 4 (development tools only, tools/mb_stamps.hip): time stamps (s_memtime, low word) at the arrival at and the release from the two
// barriers of every line step, per wave of the sampled blocks: who waits for whom.  Record of a wave: MX_STAMP_DW dwords,
// [0] HW_ID [1] XCC_ID [2..3] s_memrealtime at the start [4..5] at the end [6] role [7] s_memtime at the start [8] at the end [9] line steps,
// [16 + 4 s + k]: step s, k = 0 arrival at A, 1 release from A, 2 arrival at B, 3 release from B.
// MX_EXP_FLAGS = D (experiment): the hand-over through LDS sequence words instead of the two block barriers.  cons (word 0) counts the HALF steps the
// consumer has finished with (three lines each), prod[w] (words 1, 2) the steps producer wave w has stored.  A producer may store step s once
// cons >= 2 s + 2 - D: D = 2 is the single line buffer (what the barriers enforce), D = 3 a ring of three half-step slots, D = 4 two whole buffers.
// With D > 2 on the single-buffer layout the lines are overwritten early: WRONG RESULTS, timing only -- it prices the decoupling before the ring is built.
__device__ __forceinline__ u32 mx_seq_ld(int dw) {
  u32 v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(dw * 4) : "memory");
  return (u32)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void mx_seq_st(int dw, u32 val) {
  asm volatile("ds_write_b32 %0, %1" : : "v"(dw * 4), "v"(val) : "memory");
}
__device__ __forceinline__ void mx_seq_wait(int dw, int target) {
  while ((int)mx_seq_ld(dw) - target < 0) __builtin_amdgcn_s_sleep(1);
}

constexpr int MX_STAMP_STEPS = 96;
constexpr int MX_STAMP_DW = 16 + 4 * MX_STAMP_STEPS;
constexpr int MX_STAMP_EVERY = 16;          // every sixteenth block is sampled
template <int DBG>
__device__ __forceinline__ void mx_stamp(u32* rec, int slot) {
  if constexpr (DBG == 4) {
    if (rec != nullptr) {
      const unsigned long long t = __builtin_readcyclecounter();
      if ((threadIdx.x & 63) == 0) rec[slot] = (u32)t;
    }
  } else {
    (void)rec; (void)slot;
  }
}

#ifndef MX_WAVES
#define MX_WAVES 3          // waves per SIMD the register allocation is made for (tools: -DMX_WAVES=4 tries five blocks per CU)
#endif
template <class C, int DBG = 0, int NP = 60>
__global__ void __launch_bounds__(192, MX_WAVES) k_miller_x60(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, Fp2<C>* out, uint32_t* flags, u32* park,
                                                       int rot_mode, u32* rec_dbg) {
  typedef MX<C, NP> K;
  constexpr int NL = C::RX_NL;
  constexpr int PPW = NP / 2;                                    // pairings (lane pairs at work) per producer wave: 30 / 32
  const int w = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  // Which wave consumes?  The hardware puts the three waves of a block on three of the CU's four SIMDs, and a full CU holds
  // four blocks (168 registers: three waves per SIMD), so every SIMD is the one "missing" from exactly one block.  Taking
  // the consumer on SIMD (missing + 1) mod 4 therefore gives every SIMD exactly one consumer and two producers -- the
  // consumer is the long pole of a block's step, and two of them on one SIMD make that block (and, at one round of blocks,
  // the launch) run at half speed: measured on 1024 resident blocks, wave index alone leaves 10 % of the SIMDs with two
  // consumers and 10 % with none.  rot_mode & 3: 0 = by SIMD (default), 1 = wave 2 consumes, 2 = rotate by block index.
  extern __shared__ u32 lds_roles[];
  const int rmode = rot_mode & 3;
  int role;
  if constexpr (DBG == 3) {          // development tools only (tools/mb_lone.hip): where and when every wave runs; rec_dbg is the record buffer (nullptr in the library)
    if (lane == 0) {
      u32* rec = rec_dbg + ((size_t)blockIdx.x * 3 + w) * 8;
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      rec[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      rec[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      rec[2] = (u32)t; rec[3] = (u32)(t >> 32);
    }
  }
  if (rmode == 0) {
    const u32 hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);                             // HW_REG_HW_ID
    const int simd = (int)((hw >> 4) & 3u);                                               // bits 5:4
    if (lane == 0) lds_roles[w] = (u32)simd;
    __syncthreads();
    const int s0 = (int)lds_roles[0], s1 = (int)lds_roles[1], s2 = (int)lds_roles[2];
    int cw = 2;                                                // fallback: wave 2
    if (s0 != s1 && s0 != s2 && s1 != s2) {
      const int want = (6 - (s0 + s1 + s2) + 1) & 3;
      cw = s0 == want ? 0 : (s1 == want ? 1 : 2);
    }
    __syncthreads();                                           // the words are reused by the accumulator region
    role = w == cw ? 2 : (w > cw ? w - 1 : w);
  } else {
    const int rot = rmode == 2 ? (int)(blockIdx.x % 3u) : 0;
    role = w + rot;
    if (role >= 3) role -= 3;
  }
#ifdef MX_EXP_FLAGS
  if (threadIdx.x < 4) lds_roles[K::SEQ_DW + threadIdx.x] = 0u;
  __syncthreads();
#endif
  u32* srec = nullptr;                 // DBG == 4: this wave's stamp record (sampled blocks only)
  int sstep = 0;
  if constexpr (DBG == 4) {
    if (rec_dbg != nullptr && blockIdx.x % MX_STAMP_EVERY == 0) {
      srec = rec_dbg + ((size_t)(blockIdx.x / MX_STAMP_EVERY) * 3 + w) * MX_STAMP_DW;
      if (lane == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        srec[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        srec[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        srec[2] = (u32)t; srec[3] = (u32)(t >> 32);
        srec[6] = (u32)role;
        srec[7] = (u32)__builtin_readcyclecounter();
      }
    }
  }
  if (role < 2) {
    // ---------------------------------------------------------------- producer: PPW pairings, one per lane pair
    if (rot_mode & 8) __builtin_amdgcn_s_setprio(3);
    const int q = lane >> 1;
    const bool odd = lane & 1;
    const bool owner = q < PPW;
    const int pi = role * PPW + (owner ? q : 0);
    const size_t idx = (size_t)blockIdx.x * NP + pi;
    // group and line slot of this pairing: six per group; the 64-form's last four pairings are the seventh line of groups 0..3
    const int tg = pi < 60 ? pi / 6 : pi - 60, m = pi < 60 ? pi % 6 : 6;
    u32* const mypark = park + ((size_t)blockIdx.x * 128 + (role * 64 + lane)) * (K::NPARK * K::PS);
    constexpr int P_NYP = K::NPARK_Q, P_XP = K::NPARK_Q + 1;
    bool valid = owner && idx < n;
    PointX<C> T;
    {
      // Setup, all in the carry-free form on the lane pair (no out-of-line 32-bit helper: the kernel has no stack frame besides
      // its spills): the even lane parses the real parts of the key (x_re, y_re), the odd lane the imaginary parts; canonical
      // encoding, the curve equation y^2 = x^3 + b' and the point at infinity are decided jointly.
      constexpr int NB = C::FP_BYTES;
      const uint8_t* kb = g2s + (valid ? idx : 0) * 4 * NB;
      const Fp<C> xw = fp_from_be<C>(kb + (odd ? 0 : NB)), yw = fp_from_be<C>(kb + (odd ? 2 * NB : 3 * NB));     // wire order: x_im x_re y_im y_re
      const bool canon_own = !fp_geq_p<C>(xw) && !fp_geq_p<C>(yw);
      const bool zero_own = fp_is_zero<C>(xw) && fp_is_zero<C>(yw);
      const bool canon = canon_own && pair_swap1(canon_own ? 1 : 0) != 0;
      const bool qinf = zero_own && pair_swap1(zero_own ? 1 : 0) != 0;
      Sx<C, SX_T> xq = sx_from_plain<C>(xw), yq = sx_from_plain<C>(yw);
      {
        const Sx<C, SX_T> b2 = sx_const<C>(odd ? C::RX_B2_IM : C::RX_B2_RE);
        const auto d = sx_sub<C>(pair_sqr<C>(yq, odd), sx_add<C>(pair_mul<C>(pair_sqr<C>(xq, odd), xq, odd), b2));
        const bool on_own = sx_is_zero_mod_p<C>(d);
        const bool on_curve = on_own && pair_swap1(on_own ? 1 : 0) != 0;
        if (valid && !(canon && (qinf || on_curve))) atomicOr(flags, FLAG_ENC);
      }
      Aff<F1<C>> P = g1s[valid ? idx : 0];
      valid = valid && !P.inf && !qinf;
      if (!valid) {                         // harmless stand-ins: the generators (the lane's line is replaced by the constant 1)
        xq = ux_to_sx<C>(to_ux<C>(fp_load<C>(C::G2 + (odd ? C::L : 0))));
        yq = ux_to_sx<C>(to_ux<C>(fp_load<C>(C::G2 + 2 * C::L + (odd ? C::L : 0))));
        P.x = fp_load<C>(C::G1X);
        P.y = fp_load<C>(C::G1Y);
      }
      MxPark<C>::st(mypark, 0, xq);
      MxPark<C>::st(mypark, 1, yq);
      if constexpr (C::CURVE_ID == 0) {
        // Q1 = pi(Q) = (conj(x) g12, conj(y) g13), -Q2 = -pi^2(Q) = (x g22, -y g23) on the twist (pairing.hpp miller_loop)
        const Sx<C, SX_T> cx = sx_select<C>(odd, sx_neg<C>(xq), xq), cy = sx_select<C>(odd, sx_neg<C>(yq), yq);
        constexpr int N2 = 2 * C::RX_NL;
        MxPark<C>::st(mypark, 2, pair_mul_const<C>(cx, C::RX_GAMMA + 0 * N2, C::RX_GAMMA + 0 * N2 + C::RX_NL, odd));
        MxPark<C>::st(mypark, 3, pair_mul_const<C>(cy, C::RX_GAMMA + 1 * N2, C::RX_GAMMA + 1 * N2 + C::RX_NL, odd));
        MxPark<C>::st(mypark, 4, pair_mul_const<C>(xq, C::RX_GAMMA + 2 * N2, C::RX_GAMMA + 2 * N2 + C::RX_NL, odd));
        // -y g23: negate, then one carry pass so that the parked limbs are tight again
        MxPark<C>::st(mypark, 5, sx_norm<C>(sx_neg<C>(pair_mul_const<C>(yq, C::RX_GAMMA + 3 * N2, C::RX_GAMMA + 3 * N2 + C::RX_NL, odd))));
      }
      if constexpr (K::P_IN_LDS) {          // the even lane stores -yP, the odd lane xP; both read both (same wave: no barrier)
        if (owner) mx_st_half<C, K::PACKED>(K::PQ + (2 * pi + (odd ? 1 : 0)) * K::HS, odd, odd ? to_ux<C>(P.x) : to_ux<C>(fp_neg<C>(P.y)));
        wave_sync();
      } else {
        MxPark<C>::st(mypark, P_NYP, ux_to_sx<C>(to_ux<C>(fp_neg<C>(P.y))));
        MxPark<C>::st(mypark, P_XP, ux_to_sx<C>(to_ux<C>(P.x)));
      }
      T.X = xq;
      T.Y = yq;
      T.Z = sx_select<C>(odd, ux_to_sx<C>(ux_zero<C>()), sx_const<C>(C::RX_ONE));
    }
    const int rl_base = tg * K::GROUP_DW + (odd ? K::HS : 0);
    // the three line coefficients of a step: computed last in the step, held in registers over barrier A
    Ux<C> e[3];
    auto emit = [&](int which, const auto& v) __attribute__((always_inline)) {
      const int entry = which == 1 ? 1 : ((which == 0) == C::TWIST_D ? 0 : 2);     // D-type: c0 yP, c1 xP, c2;  M-type: c2, c1 xP, c0 yP
      if constexpr (rx_lazy<C>) e[entry] = sx_to_ux<C>(v);
      else if constexpr (std::is_same<std::decay_t<decltype(v)>, Sx<C, SX_T>>::value) e[entry] = sx_to_ux_p<C>(v);    // a reduction's output: + p, below 2.1 p
      else e[entry] = sx_to_ux_k<1, C>(v);              // the P-free coefficient, a difference of two reductions' outputs: + 2 p, below 3.1 p
    };
    int pstep = 0;                        // MX_EXP_FLAGS: line steps stored so far
    auto hand_over = [&]() __attribute__((always_inline)) {
      // the constant line 1 for a pairing that is not there (a batch's ragged end, a key or hash point at infinity).  Decided per WAVE first: the lanes
      // that own no pairing never store their line, so only an OWNED invalid pairing needs the selects, and a wave without one -- all but the last
      // block of a batch -- branches over the 3 NL of them
      if (DBG == 2 || __builtin_amdgcn_ballot_w64(owner && !valid) != 0) {
        if (DBG == 2 || !valid) {
          e[0] = odd ? ux_zero<C>() : ux_load<C>(C::RX_ONE);
          e[1] = ux_zero<C>();
          e[2] = ux_zero<C>();
        }
      }
      mx_stamp<DBG>(srec, 16 + 4 * sstep);
#ifdef MX_EXP_FLAGS
      mx_seq_wait(K::SEQ_DW, 2 * pstep + 2 - (MX_EXP_FLAGS));
#else
      __syncthreads();                    // A: the consumer has finished with the previous lines
#endif
      mx_stamp<DBG>(srec, 16 + 4 * sstep + 1);
      if (owner) {
        mx_st_half<C, K::PACKED>(rl_base + K::line_off(3 * m), odd, e[0]);
        mx_st_half<C, K::PACKED>(rl_base + K::line_off(3 * m + 1), odd, e[1]);
        mx_st_half<C, K::PACKED>(rl_base + K::line_off(3 * m + 2), odd, e[2]);
      }
      mx_stamp<DBG>(srec, 16 + 4 * sstep + 2);
#ifdef MX_EXP_FLAGS
      ++pstep;
      mx_seq_st(K::SEQ_DW + 1 + role, (u32)pstep);          // LDS serves a wave's accesses in order: the lines are in place when this word is
#else
      __syncthreads();                    // B: lines visible
#endif
      mx_stamp<DBG>(srec, 16 + 4 * sstep + 3);
      if constexpr (DBG == 4) ++sstep;
    };
    struct Env {
      const u32* pk;
      int xs, ys;
      bool neg_y;
      int pq;                             // LDS offset of this pairing's (-yP, xP)
      __device__ __forceinline__ Sx<C, SX_T> nyP() const {
        if constexpr (K::P_IN_LDS) return ux_to_sx<C>(mx_ld_half<C, K::PACKED>(pq, false));
        else return MxPark<C>::ld(MxPark<C>::launder(pk), K::NPARK_Q);
      }
      __device__ __forceinline__ Sx<C, SX_T> xP() const {
        if constexpr (K::P_IN_LDS) return ux_to_sx<C>(mx_ld_half<C, K::PACKED>(pq + K::HS, true));
        else return MxPark<C>::ld(MxPark<C>::launder(pk), K::NPARK_Q + 1);
      }
      __device__ __forceinline__ Sx<C, SX_T> xq() const { return MxPark<C>::ld(MxPark<C>::launder(pk), xs); }
      __device__ __forceinline__ Sx<C, SX_T> yq() const {
        const Sx<C, SX_T> y = MxPark<C>::ld(MxPark<C>::launder(pk), ys);
        return sx_select<C>(neg_y, sx_neg<C>(y), y);
      }
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      if constexpr (DBG != 2) dbl_step_x<C>(T, Env{mypark, 0, 1, false, K::PQ + 2 * pi * K::HS}, odd, emit);
      hand_over();
      const int d = C::LOOP_NAF[i];
      if (d != 0) {
        if constexpr (DBG != 2) add_step_x<C>(T, Env{mypark, 0, 1, d < 0, K::PQ + 2 * pi * K::HS}, odd, emit);
        hand_over();
      }
    }
    if constexpr (C::CURVE_ID == 0) {
#pragma unroll 1
      for (int s = 0; s < 2; ++s) {
        if constexpr (DBG != 2) add_step_x<C>(T, Env{mypark, 2 + 2 * s, 3 + 2 * s, false, K::PQ + 2 * pi * K::HS}, odd, emit);
        hand_over();
      }
    }
    // Degenerate point steps (T = +-Q in an addition, a 2-torsion point or infinity in a doubling) leave Z = 0, and Z = 0 stays:
    // Z3 = 2 Y^3 Z (doubling), Z3 = Z la^3 (addition).  A key of order r never gets there; a small-order twist point handed in
    // without the subgroup check does, and the reference refuses such points at construction (curves/bls12_381.go:196-264,
    // curves/altbn128.go:157-179) -- so the verification reports an encoding error instead of an unspecified verdict.
    if constexpr (DBG == 0) {
      const bool z_own = sx_is_zero_mod_p<C>(T.Z);
      const bool z_zero = z_own && pair_swap1(z_own ? 1 : 0) != 0;
      if (valid && z_zero && !odd) atomicOr(flags, FLAG_DEGENERATE);
    }
  } else {
    // ---------------------------------------------------------------- consumer: 10 groups x 6 lanes, lane = 10 j + g
    if (rot_mode & 4) __builtin_amdgcn_s_setprio(3);       // the consumer is the long pole of a block's step: let it issue first
#ifdef MX_PRIO_EXP                                         // development tools only (tools/mb_x60.hip): intermediate priorities in mode bits 5-6
    if (((rot_mode >> 5) & 3) == 1) __builtin_amdgcn_s_setprio(1);
    if (((rot_mode >> 5) & 3) == 2) __builtin_amdgcn_s_setprio(2);
#endif
    const bool live = lane < 60;
    const int cl = live ? lane : lane - 12;                // lanes 60..63 shadow lanes 48..51 (same 16-lane group of a ds_read_b128: broadcast)
    const int g = cl % 10;
    const int j = cl / 10;
    const int gb = g * K::GROUP_DW;
    Ux2<C> fj;
    {
      const Ux<C> one = ux_load<C>(C::RX_ONE);
#pragma unroll
      for (int k = 0; k < NL; ++k) { fj.c0.v[k] = j == 0 ? one.v[k] : 0u; fj.c1.v[k] = 0u; }
    }
    if constexpr (NP == 64) {
      // groups 4..9 have six pairings: their seventh line is the constant 1 for the whole loop (no producer writes it)
      if (live && g >= 4 && j < 3) {
        const Ux<C> one = ux_load<C>(C::RX_ONE), zero = ux_zero<C>();
        mx_st_half<C, K::PACKED>(gb + K::line_off(18 + j), false, j == 0 ? one : zero);
        mx_st_half<C, K::PACKED>(gb + K::line_off(18 + j) + K::HS, true, zero);
      }
    }
    mx_publish<C, NP>(gb, j, fj, live);
    unsigned sq_d = 0, sq_p = 0;
    int cstep = 0;                        // MX_EXP_FLAGS: line steps folded so far
    (void)cstep;
    if constexpr (!rx_lazy<C>) mx_sq_split(COOP_SQ_TAB[j], sq_d, sq_p);
    auto fold_all = [&]() __attribute__((always_inline)) {
      if constexpr (DBG == 1) {
#ifdef MX_EXP_FLAGS
        mx_seq_st(K::SEQ_DW, (u32)(2 * cstep + 2));
        ++cstep;
#endif
        return;
      }
      // unrolled by three on the 29-bit form (same-box A/B at 2^20 pairings: 54.4 -> 54.0 ms; by two 54.3, by six 54.1), rolled on BLS12-381
      // (unrolled by two or three: 86.8 -> 87.2 ms)
#pragma unroll K::FOLD_UNROLL
      for (int m = 0; m < K::NLINES; ++m) {
        fj = mx_fold<C, NP>(gb, m, j);
#ifdef MX_EXP_FLAGS
        if (m == 2) mx_seq_st(K::SEQ_DW, (u32)(2 * cstep + 1));                 // the fold's fetches are in LDS's queue ahead of this word
        if (m == K::NLINES - 1) mx_seq_st(K::SEQ_DW, (u32)(2 * cstep + 2));
#endif
        mx_publish<C, NP>(gb, j, fj, live);
      }
#ifdef MX_EXP_FLAGS
      ++cstep;
#endif
    };
#ifdef MX_EXP_FLAGS
    auto lines_ready = [&]() __attribute__((always_inline)) {
      mx_seq_wait(K::SEQ_DW + 1, cstep + 1);
      mx_seq_wait(K::SEQ_DW + 2, cstep + 1);
    };
#endif
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
#ifndef MX_SQR_LATE
      if (DBG != 1 && i > 1) {            // f = 1 before the first step.  The squaring needs no line: it goes AHEAD of A (round 6), so that the producers never park for it at B
        if constexpr (rx_lazy<C>) fj = mx_sqr<C, NP>(gb, j);
        else fj = mx_sqr3<C, NP>(gb, sq_d, sq_p);
        mx_publish<C, NP>(gb, j, fj, live);
      }
#endif
      mx_stamp<DBG>(srec, 16 + 4 * sstep);
#ifdef MX_EXP_FLAGS
      lines_ready();
#else
      __syncthreads();                    // A
#endif
      mx_stamp<DBG>(srec, 16 + 4 * sstep + 1);
#ifdef MX_SQR_LATE                     // -DMX_SQR_LATE: rounds 3-5, the squaring between A and B (A/B measurements)
      if (DBG != 1 && i > 1) {
        if constexpr (rx_lazy<C>) fj = mx_sqr<C, NP>(gb, j);
        else fj = mx_sqr3<C, NP>(gb, sq_d, sq_p);
        mx_publish<C, NP>(gb, j, fj, live);
      }
#endif
      mx_stamp<DBG>(srec, 16 + 4 * sstep + 2);
#ifndef MX_EXP_FLAGS
      __syncthreads();                    // B
#endif
      mx_stamp<DBG>(srec, 16 + 4 * sstep + 3);
      if constexpr (DBG == 4) ++sstep;
      fold_all();
      if (C::LOOP_NAF[i] != 0) {
        mx_stamp<DBG>(srec, 16 + 4 * sstep);
#ifdef MX_EXP_FLAGS
        lines_ready();
#else
        __syncthreads();
#endif
        mx_stamp<DBG>(srec, 16 + 4 * sstep + 1);
        mx_stamp<DBG>(srec, 16 + 4 * sstep + 2);
#ifndef MX_EXP_FLAGS
        __syncthreads();
#endif
        mx_stamp<DBG>(srec, 16 + 4 * sstep + 3);
        if constexpr (DBG == 4) ++sstep;
        fold_all();
      }
    }
    if constexpr (C::CURVE_ID == 0) {
#pragma unroll 1
      for (int s = 0; s < 2; ++s) {
        mx_stamp<DBG>(srec, 16 + 4 * sstep);
#ifdef MX_EXP_FLAGS
        lines_ready();
#else
        __syncthreads();
#endif
        mx_stamp<DBG>(srec, 16 + 4 * sstep + 1);
        mx_stamp<DBG>(srec, 16 + 4 * sstep + 2);
#ifndef MX_EXP_FLAGS
        __syncthreads();
#endif
        mx_stamp<DBG>(srec, 16 + 4 * sstep + 3);
        if constexpr (DBG == 4) ++sstep;
        fold_all();
      }
    }
    if (live) {
      if constexpr (!rx_lazy<C>) fj = ux_quasi<C, 2, 1>(fj);        // the folds' fixed point is 3.7 p; from_ux_inl takes values below 4 p
      Fp2<C> r = {from_ux_inl<C>(fj.c0), from_ux_inl<C>(fj.c1)};
      if constexpr (C::CURVE_ID != 0) {
        if (j & 1) r = f2_neg<C>(r);                      // x < 0: f^(p^6), w -> -w
      }
      out[((size_t)blockIdx.x * 10 + g) * 6 + j] = r;
    }
  }
  if constexpr (DBG == 3) {
    if (lane == 0) {
      u32* rec = rec_dbg + ((size_t)blockIdx.x * 3 + w) * 8;
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      rec[4] = (u32)t; rec[5] = (u32)(t >> 32);
      rec[6] = (u32)role;
    }
  }
  if constexpr (DBG == 4) {
    if (srec != nullptr && lane == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memrealtime();
      srec[4] = (u32)t; srec[5] = (u32)(t >> 32);
      srec[8] = (u32)__builtin_readcyclecounter();
      srec[9] = (u32)sstep;
    }
  }
}

}  // namespace bgls
