// The consumer half of k_miller_ab64 on 28-bit limbs (r28.hpp): the shared accumulators and the line elements live in
// LDS as ten-limb values in Montgomery radix 2^280; the producer converts each coefficient as it stores it.  Same data
// flow as coop.hpp (entry-major regions, one output coefficient per lane, index tables COOP_SH_*, COOP_SQ_TAB), entries
// are 20 dwords instead of 16.
#pragma once
#include "coop.hpp"
#include "r28.hpp"

namespace bgls {

constexpr int R28_S2 = 20;     // dwords per Fp2 entry

__device__ __forceinline__ F28x2 lds_ld28(int off) {
  extern __shared__ u32 lds[];
  const uint4* p = reinterpret_cast<const uint4*>(lds + off);
  u32 w[20];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const uint4 v = p[k];
    w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
  }
  F28x2 r;
#pragma unroll
  for (int k = 0; k < 10; ++k) { r.c0.v[k] = w[k]; r.c1.v[k] = w[10 + k]; }
  return r;
}
__device__ __forceinline__ void lds_st28(int off, const F28x2& a) {
  extern __shared__ u32 lds[];
  uint4* p = reinterpret_cast<uint4*>(lds + off);
  u32 w[20];
#pragma unroll
  for (int k = 0; k < 10; ++k) { w[k] = a.c0.v[k]; w[10 + k] = a.c1.v[k]; }
#pragma unroll
  for (int k = 0; k < 5; ++k) p[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// store entry e of a region: the library's 32-bit form, or converted to the 28-bit form (producer side of the hand-over)
template <class C, bool R28>
__device__ __forceinline__ void st_entry(LReg r, int e, const Fp2<C>& x) {
  if constexpr (R28) lds_st28(r.base + e * R28_S2, to_r28<C>(x));
  else lds_st<C>(r, e, x);
}

// publish this lane's coefficient, plain and xi-multiplied
template <class C>
__device__ __forceinline__ void coop_publish28(int rb_off, int j, const F28x2& v, bool live) {
  if (live) {
    lds_st28(rb_off + (2 * j) * R28_S2, v);
    lds_st28(rb_off + (2 * j + 1) * R28_S2, r28_mulxi<C>(v));
  }
  wave_sync();
}

// c_j = sum_{t<NT} A[a_e0 + t] * B[(j - sh[t]) mod 6] * xi^[sh[t] > j]   (as coop_dot_inl; B = {e_k, xi e_k} pairs)
template <class C, int NT>
__device__ __forceinline__ F28x2 coop_dot28(int ra_off, int a_e0, int rb_off, int j, const int* sh) {
  u64 cr[20], ci[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) cr[k] = ci[k] = 0;
#pragma unroll 1
  for (int t = 0; t < NT; ++t) {
    const int sht = sh[t];
    int k = j - sht;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    const F28x2 a = lds_ld28(ra_off + (a_e0 + t) * R28_S2);
    const F28x2 b = lds_ld28(rb_off + (2 * k + wrap) * R28_S2);
    r28_acc(cr, a.c0, b.c0);
    r28_acc(cr, a.c1, r28_fatneg<C>(b.c1));
    r28_acc(ci, a.c0, b.c1);
    r28_acc(ci, a.c1, b.c0);
  }
  F28x2 r;
  r.c0 = r28_redc<C>(cr);
  r.c1 = r28_redc<C>(ci);
  return r;
}

// The same three-term dot product with Karatsuba over i: three limb-product piles per term instead of four
// (sum a0 b0, sum a1 b1, sum (a0+a1)(b0+b1)), combined column by column at the end.  A column of a difference can be
// negative although the difference is not, hence the bias (a multiple of p that dominates every column, R28_BIAS3).
// Only for NT = 3 and undoubled operands: with more terms the sums would leave the 64-bit column budget.  Worst case for
// operands with limbs < 2^28 (top limb < 2^12): a column reaches 2^63.43 before the reduction and 2^63.51 with the
// reduction's own products and carries, a factor 1.4 below 2^64 (asserted when the constants are generated).
template <class C>
__device__ __forceinline__ F28x2 coop_dot28_k3(int ra_off, int a_e0, int rb_off, int j, const int* sh) {
  u64 v0[20], v1[20], ss[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) v0[k] = v1[k] = ss[k] = 0;
#pragma unroll 1
  for (int t = 0; t < 3; ++t) {
    const int sht = sh[t];
    int k = j - sht;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    const F28x2 a = lds_ld28(ra_off + (a_e0 + t) * R28_S2);
    const F28x2 b = lds_ld28(rb_off + (2 * k + wrap) * R28_S2);
    r28_kara_term(v0, v1, ss, a, b);
  }
  return r28_kara_finish<C>(v0, v1, ss);
}

// f^2 with the symmetric terms merged (COOP_SQ_TAB, see coop_sqr_sym_inl): a doubled term doubles the limbs of its
// left operand (below 2^29, still inside the column budget: at most three doubled terms and one plain one per lane)
template <class C>
__device__ __forceinline__ F28x2 coop_sqr_sym28(int rb_off, int j) {
  u64 cr[20], ci[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) cr[k] = ci[k] = 0;
  const unsigned row = COOP_SQ_TAB[j];
#pragma unroll 1
  for (int t = 0; t < 4; ++t) {
    const unsigned e = (row >> (8 * t)) & 0xFFu;
    const int i = e & 7u, k = (e >> 3) & 7u;
    const bool used = i != 7;
    F28x2 a = lds_ld28(rb_off + (2 * (used ? i : 0)) * R28_S2);
    const F28x2 b = lds_ld28(rb_off + (used ? 2 * k + (int)((e >> 6) & 1u) : 0) * R28_S2);
    const u32 keep = used ? 0xFFFFFFFFu : 0u;          // unused slot: a = 0
    const u32 sh = (e >> 7) & 1u;                      // doubled term: a <<= 1
#pragma unroll
    for (int q = 0; q < 10; ++q) { a.c0.v[q] = (a.c0.v[q] << sh) & keep; a.c1.v[q] = (a.c1.v[q] << sh) & keep; }
    r28_acc(cr, a.c0, b.c0);
    r28_acc(cr, a.c1, r28_fatneg<C>(b.c1));
    r28_acc(ci, a.c0, b.c1);
    r28_acc(ci, a.c1, b.c0);
  }
  F28x2 r;
  r.c0 = r28_redc<C>(cr);
  r.c1 = r28_redc<C>(ci);
  return r;
}

}  // namespace bgls
