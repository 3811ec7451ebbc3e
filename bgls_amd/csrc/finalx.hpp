// Latency-oriented final exponentiation on the carry-free 28-bit limbs: ONE Fp12 value, 36 lanes of each of TWO waves.
//
// Same job and same 36-lane split as finalexp.hpp (the final exponentiation happens once per verification, bgls/bgls.go:115-118
// compares the product of ALL pairings with 1: a serial chain of ~300 Fp12 products that no batch dimension hides; lane 6j + t
// computes term t of output coefficient j), different arithmetic: on a lone wave a 32-bit-limb product is a chain of dependent
// multiply-adds and carries (~40 clocks per limb product, nothing to interleave with), while the rows of a carry-free product
// are NL independent multiplier instructions into 64-bit columns that issue back to back.  Lane 6j + t of wave h computes half h
// (real / imaginary) of the Fp2 product a_t b_(j-t) INCLUDING its reduction (two limb products, one reduction: sx_montr), so what
// the lanes exchange through LDS are reduced values (NL limbs), not double-width piles, and the six terms of a coefficient are
// added limb-wise by twelve lanes (coefficient, half) that also carry-normalise and form the xi multiple the next product's wrapped
// terms need.  The block is 128 threads; the stages of a product are separated by block barriers.
//
// Squarings are plain products here: the Granger-Scott formulas add 2 z to three times a product without a reduction in
// between, and in a redundant (non-reduced) representation that recurrence grows without bound; a product's 36 lanes already
// run in parallel, so there is no latency to win from the cheaper formula.  Values: same Fp12 elements, same chains and exponent
// (p^12 - 1) / r as pairing.hpp final_exp; the result leaves canonically, byte-identical to finalexp.hpp's.
#pragma once
#include "finalexp.hpp"
#include "rx_jac.hpp"
#include "constants_latx_gen.hpp"

namespace bgls {

#ifdef FX_DBG
// development only: shader-clock stamps of the final exponentiation's phases (tools/exp/fx_steps.py)
__device__ unsigned long long g_fx_t[16];
#define FX_T(k) do { if (threadIdx.x == 0) g_fx_t[k] = clock64(); } while (0)
#else
#define FX_T(k) do { } while (0)
#endif
template <class C>
struct FX {
  static constexpr int N = C::RX_NL;
  static constexpr int HS = (N + 3) & ~3;             // one half (Fp), 16-byte granules
  static constexpr int ES = 2 * HS;                   // one Fp2
  static constexpr int SLOT = 12 * ES;                // 6 coefficients x {plain, xi multiple}
  static constexpr int NSLOT = 16;
  static constexpr int SCR = NSLOT * SLOT;
  static constexpr int SCR_DW = 36 * ES;              // 36 lanes x one reduced Fp2 product
  static constexpr int LDS_DW = SCR + SCR_DW;
  static constexpr int LDS_BYTES = LDS_DW * 4;
  static constexpr int LDS_BYTES_PAIR = (LDS_DW + SCR_DW) * 4;   // fx_mul_pair: a second region of half-products
  __device__ static __forceinline__ int coef(int slot, int k, int xi) { return slot * SLOT + (2 * k + xi) * ES; }
};

template <class C>
__device__ __forceinline__ Sx<C, SX_T> fx_ld(int off) {
  extern __shared__ u32 lds[];
  constexpr int N = C::RX_NL;
  Sx<C, SX_T> r;
  const uint4* p = reinterpret_cast<const uint4*>(lds + off);
#pragma unroll
  for (int k = 0; k < N / 4; ++k) {
    const uint4 v = p[k];
    r.v[4 * k] = (i32)v.x; r.v[4 * k + 1] = (i32)v.y; r.v[4 * k + 2] = (i32)v.z; r.v[4 * k + 3] = (i32)v.w;
  }
  if constexpr (N % 4 == 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(lds + off + (N & ~3));
    r.v[N - 2] = (i32)v.x; r.v[N - 1] = (i32)v.y;
  }
  return r;
}
template <class C>
__device__ __forceinline__ void fx_st(int off, const Sx<C, SX_T>& a) {
  extern __shared__ u32 lds[];
  constexpr int N = C::RX_NL;
  uint4* p = reinterpret_cast<uint4*>(lds + off);
#pragma unroll
  for (int k = 0; k < N / 4; ++k) p[k] = make_uint4((u32)a.v[4 * k], (u32)a.v[4 * k + 1], (u32)a.v[4 * k + 2], (u32)a.v[4 * k + 3]);
  if constexpr (N % 4 == 2) *reinterpret_cast<uint2*>(lds + off + (N & ~3)) = make_uint2((u32)a.v[N - 2], (u32)a.v[N - 1]);
}
template <class C>
__device__ __forceinline__ X2<C, SX_T> fx_ld2(int off) {
  return {fx_ld<C>(off), fx_ld<C>(off + FX<C>::HS)};
}

// this half of xi * (re + i im), carry-normalised: the real half is XI_RE re - im, the imaginary half XI_RE im + re
// (alt-bn128: xi = 9 + i, BLS12-381: 1 + i).  `mine` is the own half, `other` the other one; any limbs in (sums of a few tight
// values as well: the carry is 64 bits wide), tight out -- the carry-normalised limbs of an integer are unique, so the result does
// not depend on how the operands were represented.
template <class C, int LA>
__device__ __forceinline__ Sx<C, SX_T> fx_mulxi_half(const Sx<C, LA>& mine, const Sx<C, LA>& other, bool is_im) {
  constexpr int N = C::RX_NL;
  const i32 sgn = is_im ? 1 : -1;
  Sx<C, SX_T> r;
  i64 c = 0;
#pragma unroll
  for (int i = 0; i < N - 1; ++i) {
    c += (i64)C::XI_RE * mine.v[i] + (i64)(sgn * other.v[i]);
    r.v[i] = (i32)((u32)c & RX_MASK);
    c >>= 28;
  }
  r.v[N - 1] = (i32)(c + (i64)C::XI_RE * mine.v[N - 1] + (i64)(sgn * other.v[N - 1]));
  return r;
}

// write coefficient k of `slot` (plain) and its xi multiple; one lane holds both halves
template <class C>
__device__ __forceinline__ void fx_put(int slot, int k, const X2<C, SX_T>& v) {
  typedef FX<C> E;
  fx_st<C>(E::coef(slot, k, 0), v.c0);
  fx_st<C>(E::coef(slot, k, 0) + E::HS, v.c1);
  fx_st<C>(E::coef(slot, k, 1), fx_mulxi_half<C>(v.c0, v.c1, false));
  fx_st<C>(E::coef(slot, k, 1) + E::HS, fx_mulxi_half<C>(v.c1, v.c0, true));
}

// Second stage of a product, shared by the two waves of the pair that computed its half-products (at `scr`): lanes 0..11 of BOTH
// waves add the six terms of coefficient half (j, hh) up; the wave with h == 0 carry-normalises the sum and stores the coefficient,
// the wave with h == 1 forms its xi multiple from the raw sums (the other half's from the neighbour lane) and stores that.  The
// two instruction streams are each about half of what one wave did alone (round 4: the second wave used to wait here).
template <class C>
__device__ __forceinline__ void fx_stage2(int dst, int scr, int lane, int h) {
  typedef FX<C> E;
  if (lane < 12) {
    const int j = lane >> 1, hh = lane & 1;
    const int o = scr + (6 * j) * E::ES + hh * E::HS;
    const Sx<C, SX_T> t0 = fx_ld<C>(o), t1 = fx_ld<C>(o + E::ES), t2 = fx_ld<C>(o + 2 * E::ES);
    const Sx<C, SX_T> t3 = fx_ld<C>(o + 3 * E::ES), t4 = fx_ld<C>(o + 4 * E::ES), t5 = fx_ld<C>(o + 5 * E::ES);
    const auto raw = sx_add<C>(sx_add<C>(sx_add<C>(t0, t1), sx_add<C>(t2, t3)), sx_add<C>(t4, t5));
    if (h == 0) {
      fx_st<C>(E::coef(dst, j, 0) + hh * E::HS, sx_norm<C>(raw));
    } else {
      auto other = raw;
#pragma unroll
      for (int i = 0; i < C::RX_NL; ++i) other.v[i] = __builtin_amdgcn_update_dpp(0, raw.v[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
      fx_st<C>(E::coef(dst, j, 1) + hh * E::HS, fx_mulxi_half<C>(raw, other, hh == 1));
    }
  }
}

// dst <- a * b (all 128 threads must call).  Lane 6j + t of wave h: half h of the reduced product a_t * b_(j - t) (the xi multiple
// of b's coefficient where the index wraps, w^6 = xi): real half a0 b0 - a1 b1, imaginary half a0 b1 + a1 b0, the same instruction
// stream on both waves (column factors selected, the row factor's sign applied arithmetically).  Threads 2j + h of wave 0: half h
// of coefficient j = the limb-wise sum of its six terms, carry-normalised, and its xi multiple (the other half comes from the
// neighbour lane by a quad permute).
// Magnitudes: a term is below (1 + eps) p, a coefficient below 6.1 p, its xi multiple below 61 p (alt-bn128) / 12.2 p
// (BLS12-381); the next product of two such values is below 2^-3 p after the division by R' (R' / p = 2^26 / 2^11).
template <class C>
__device__ __noinline__ void fx_mul(int dst, int a, int b) {
  typedef FX<C> E;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  if (lane < 36 && tid < 128) {                              // waves 0 and 1 (a larger block's other waves only meet the barriers)
    const int j = lane / 6, t = lane % 6;
    int k = j - t;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(a, t, 0));
    const int yo = E::coef(b, k, wrap);
    const Sx<C, SX_T> ya = fx_ld<C>(yo + (h ? E::HS : 0)), yb = fx_ld<C>(yo + (h ? 0 : E::HS));      // y0 y1 | y1 y0
    const i32 sg = h ? 0 : -1;                                                                        // - a1 b1 on the real half
    const i32* const cols[2] = {ya.v, yb.v};
    const Sx<C, SX_T> p = sx_montr<C, 2, 2 * SX_T * SX_T>(cols, [&](int q, int i) { return q == 0 ? x.c0.v[i] : (x.c1.v[i] ^ sg) - sg; });
    fx_st<C>(E::SCR + lane * E::ES + h * E::HS, p);
  }
  __syncthreads();
  if (tid < 128) fx_stage2<C>(dst, E::SCR, lane, h);
  __syncthreads();
}

// TWO independent products side by side in a block of four waves (k_finalx): waves 0, 1 compute d0 <- a0 b0 and waves 2, 3
// d1 <- a1 b1 (d1 < 0: nothing) between the same two barriers, i.e. in the time of one product.  Every operand is read before the
// first barrier and every result written after it, so a destination may be any of the four operands.  The second product's
// half-products use a second scratch region behind the first: dynamic LDS = FX::LDS_BYTES_PAIR.
template <class C>
__device__ __noinline__ void fx_mul_pair(int d0, int a0, int b0, int d1, int a1, int b1) {
  typedef FX<C> E;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = w & 1, g = w >> 1;
  const int dst = g ? d1 : d0, a = g ? a1 : a0, b = g ? b1 : b0;
  const int scr = E::SCR + g * E::SCR_DW;
  if (lane < 36 && dst >= 0) {
    const int j = lane / 6, t = lane % 6;
    int k = j - t;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(a, t, 0));
    const int yo = E::coef(b, k, wrap);
    const Sx<C, SX_T> ya = fx_ld<C>(yo + (h ? E::HS : 0)), yb = fx_ld<C>(yo + (h ? 0 : E::HS));
    const i32 sg = h ? 0 : -1;
    const i32* const cols[2] = {ya.v, yb.v};
    const Sx<C, SX_T> p = sx_montr<C, 2, 2 * SX_T * SX_T>(cols, [&](int q, int i) { return q == 0 ? x.c0.v[i] : (x.c1.v[i] ^ sg) - sg; });
    fx_st<C>(scr + lane * E::ES + h * E::HS, p);
  }
  __syncthreads();
  if (dst >= 0) fx_stage2<C>(dst, scr, lane, h);
  __syncthreads();
}

// The same product on ONE wave (k_miller_latx: the accumulator wave of a latency-form Miller loop cannot share block barriers
// with its producer wave).  Lane 6j + t computes BOTH halves of its term (x2_mul: four limb products, two reductions); the stages
// are separated by wave barriers.
template <class C>
__device__ __noinline__ void fx_mul1(int dst, int a, int b) {
  typedef FX<C> E;
  const int lane = threadIdx.x & 63;
  if (lane < 36) {
    const int j = lane / 6, t = lane % 6;
    int k = j - t;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(a, t, 0));
    const X2<C, SX_T> y = fx_ld2<C>(E::coef(b, k, wrap));
    const X2<C, SX_T> p = x2_mul<C>(x, y);
    fx_st<C>(E::SCR + lane * E::ES, p.c0);
    fx_st<C>(E::SCR + lane * E::ES + E::HS, p.c1);
  }
  wave_sync();
  if (lane < 12) {
    const int j = lane >> 1, hh = lane & 1;
    const int o = E::SCR + (6 * j) * E::ES + hh * E::HS;
    const Sx<C, SX_T> t0 = fx_ld<C>(o), t1 = fx_ld<C>(o + E::ES), t2 = fx_ld<C>(o + 2 * E::ES);
    const Sx<C, SX_T> t3 = fx_ld<C>(o + 3 * E::ES), t4 = fx_ld<C>(o + 4 * E::ES), t5 = fx_ld<C>(o + 5 * E::ES);
    const Sx<C, SX_T> mine = sx_norm<C>(sx_add<C>(sx_add<C>(sx_add<C>(t0, t1), sx_add<C>(t2, t3)), sx_add<C>(t4, t5)));
    Sx<C, SX_T> other;
#pragma unroll
    for (int i = 0; i < C::RX_NL; ++i) other.v[i] = __builtin_amdgcn_update_dpp(0, mine.v[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    fx_st<C>(E::coef(dst, j, 0) + hh * E::HS, mine);
    fx_st<C>(E::coef(dst, j, 1) + hh * E::HS, fx_mulxi_half<C>(mine, other, hh == 1));
  }
  wave_sync();
}

// hand-over words of fx_mul2w: one lane calls this before the block's first barrier; dynamic LDS = FX::LDS_BYTES + FX2W_EXTRA_BYTES
constexpr int FX2W_EXTRA_BYTES = 32;     // four hand-over words + four words for the kernel that uses them (k_miller_latx: its `valid` word)
template <class C>
__device__ __forceinline__ void fx_mul2w_init() {
  extern __shared__ u32 lds[];
  lds[FX<C>::LDS_DW] = 0;
  lds[FX<C>::LDS_DW + 1] = 0;
  lds[FX<C>::LDS_DW + 2] = 0;
  lds[FX<C>::LDS_DW + 3] = 0;
}
// The two-wave product WITHOUT block barriers (round 4): for a block whose third wave does something else and must not be
// dragged into the product's barriers (k_miller_latx: the accumulator on waves 0 and 1, the point steps on wave 2).  The two
// stages of fx_mul are separated by hand-overs through four LDS words instead: each wave publishes "my half-products of product
// number `epoch` are stored" and, after its part of the second stage (fx_stage2), "my part of product `epoch` is stored"; each
// side spins on the other's word.  `epoch` = the number of this product, 1, 2, .. (both waves count their calls; by
// value: a reference parameter of a function that is not inlined lives in scratch memory, and the spin loops then re-read it from there).
template <class C>
__device__ __noinline__ void fx_mul2w(int dst, int a, int b, u32 epoch) {
  typedef FX<C> E;
  extern __shared__ u32 lds[];
  u32* const s_flag = lds + E::LDS_DW;                      // two words behind the slots: the caller zeroes them (fx_mul2w_init) and sizes the block's LDS for them
  const int tid = threadIdx.x, lane = tid & 63, h = (tid >> 6) & 1;
  if (lane < 36) {
    const int j = lane / 6, t = lane % 6;
    int k = j - t;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(a, t, 0));
    const int yo = E::coef(b, k, wrap);
    const Sx<C, SX_T> ya = fx_ld<C>(yo + (h ? E::HS : 0)), yb = fx_ld<C>(yo + (h ? 0 : E::HS));      // y0 y1 | y1 y0
    const i32 sg = h ? 0 : -1;                                                                        // - a1 b1 on the real half
    const i32* const cols[2] = {ya.v, yb.v};
    const Sx<C, SX_T> p = sx_montr<C, 2, 2 * SX_T * SX_T>(cols, [&](int q, int i) { return q == 0 ? x.c0.v[i] : (x.c1.v[i] ^ sg) - sg; });
    fx_st<C>(E::SCR + lane * E::ES + h * E::HS, p);
  }
  // both waves: "my half-products of product `epoch` are stored" -> wait for the other's -> my part of the second stage ->
  // "my part of product `epoch` is stored" -> wait for the other's
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(&s_flag[h], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(&s_flag[1 - h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != epoch) { }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  fx_stage2<C>(dst, E::SCR, lane, h);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(&s_flag[2 + h], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(&s_flag[3 - h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != epoch) { }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <class C>
__device__ __forceinline__ void fx_conj(int dst, int a) {           // a^(p^6): w -> -w
  typedef FX<C> E;
  const int lane = threadIdx.x;
  if (lane < 24) {
    const int k = lane >> 2, xi = (lane >> 1) & 1, h = lane & 1;
    Sx<C, SX_T> v = fx_ld<C>(E::coef(a, k, xi) + h * E::HS);
    if (k & 1) v = sx_neg<C>(v);
    fx_st<C>(E::coef(dst, k, xi) + h * E::HS, v);
  }
  __syncthreads();
}
template <class C>
__device__ __noinline__ void fx_frob(int dst, int a, int kk) {      // a^(p^kk), kk = 1..3
  typedef FX<C> E;
  const int lane = threadIdx.x;
  if (lane < 6) {
    X2<C, SX_T> v = fx_ld2<C>(E::coef(a, lane, 0));
    if (kk & 1) v.c1 = sx_neg<C>(v.c1);
    if (lane != 0) {
      const u32* g = LatxK<C>::FROB_GAMMA + ((kk - 1) * 6 + lane) * 2 * C::RX_NL;      // gamma_kk[lane] in this form
      const X2<C, SX_T> gx = {sx_const<C>(g), sx_const<C>(g + C::RX_NL)};
      v = x2_mul<C>(v, gx);
    }
    fx_put<C>(dst, lane, v);
  }
  __syncthreads();
}
// dst <- a^-1 through norms (as finalexp.hpp fe_inv): four cooperative products, two Frobenius maps and ONE Fp inversion, taken
// in the library's 32-bit form (fp_inv) on the lanes that need it.  Uses FE_X, FE_Y5, FE_Y6 as scratch.
template <class C>
__device__ __noinline__ void fx_inv(int dst, int a, int sN, int sA, int sB) {
  typedef FX<C> E;
  const int lane = threadIdx.x;
  fx_conj<C>(sB, a);
  fx_mul<C>(sN, a, sB);                   // N = a * conj(a): odd w-coefficients vanish
  fx_frob<C>(sA, sN, 2);                  // N^(p^2)
  fx_frob<C>(sB, sA, 2);                  // N^(p^4)
  fx_mul<C>(sA, sA, sB);                  // M = N^(p^2) N^(p^4)
  fx_mul<C>(sB, sN, sA);                  // Norm(N) in Fp2: only coefficient 0
  FX_T(9);
  if (lane < 6) {
    // 1 / d = conj(d) / (d0^2 + d1^2): the norm and the two products on these limbs, ONE value through the 32-bit form and back
    const X2<C, SX_T> d = fx_ld2<C>(E::coef(sB, 0, 0));
    const X2<C, SX_T> dc = {d.c0, sx_norm<C>(sx_neg<C>(d.c1))};
    const X2<C, SX_T> nn = x2_mul<C>(d, dc);
    const Sx<C, SX_T> ni = sx_from_mont<C>(fp_inv<C>(sx_to_mont<C>(nn.c0)));
    const X2<C, SX_T> di = x2_mul<C>(dc, X2<C, SX_T>{ni, ux_to_sx<C>(ux_zero<C>())});
    const X2<C, SX_T> m = fx_ld2<C>(E::coef(sA, lane, 0));
    fx_put<C>(sA, lane, x2_mul<C>(m, di));                         // N^-1 (every lane touches its own coefficient only)
  }
  __syncthreads();
  FX_T(10);
  fx_conj<C>(sB, a);
  fx_mul<C>(dst, sB, sA);
}
// dst <- a^e, public exponent with its top bit at nbits-1; dst != a; sq: a free slot.  Right to left on the two halves of a
// four-wave block (fx_mul_pair): waves 0, 1 walk the squarings a^(2^i) in `sq` while waves 2, 3 multiply the ones the exponent
// selects into dst, so the chain is nbits products long whatever the exponent's weight (left to right on one pair of waves:
// nbits - 1 squarings AND weight - 1 products, one after the other; alt-bn128's u: 89 against 63).
template <class C>
__device__ __noinline__ void fx_pow(int dst, int a, const u32* e, int nbits, int sq) {
  typedef FX<C> E;
  const int lane = threadIdx.x;
  auto copy = [&](int d, int s) {
    if (lane < 24) {
      const int k = lane >> 2, xi = (lane >> 1) & 1, h = lane & 1;
      fx_st<C>(E::coef(d, k, xi) + h * E::HS, fx_ld<C>(E::coef(s, k, xi) + h * E::HS));
    }
    __syncthreads();
  };
  copy(sq, a);
  bool have = false;
  // the exponent's words in registers: a bit test inside the loop is then scalar arithmetic, not a load from the constant array
  // and a wait in front of every product (exponents here have at most 128 bits)
  const u32 w0 = e[0], w1 = nbits > 32 ? e[1] : 0u, w2 = nbits > 64 ? e[2] : 0u, w3 = nbits > 96 ? e[3] : 0u;
  for (int i = 0; i < nbits; ++i) {
    const u32 w = i < 32 ? w0 : i < 64 ? w1 : i < 96 ? w2 : w3;
    const bool bit = (w >> (i & 31)) & 1u;
    const bool more = i + 1 < nbits;
    if (bit && !have) {
      copy(dst, sq);
      have = true;
      if (more) fx_mul_pair<C>(sq, sq, sq, -1, 0, 0);
    } else if (more) {
      fx_mul_pair<C>(sq, sq, sq, bit ? dst : -1, dst, sq);
    } else if (bit) {
      fx_mul_pair<C>(dst, dst, sq, -1, 0, 0);
    }
  }
}

// slot FE_F <- FE_F ^ ((p^12 - 1) / r); slot numbers as in finalexp.hpp
template <class C>
__device__ __noinline__ void fx_final_exp() {
  // easy part: (p^6 - 1)(p^2 + 1)
  FX_T(1);
  fx_conj<C>(FE_T, FE_F);
  fx_inv<C>(FE_U, FE_F, FE_X, FE_Y5, FE_Y6);
  FX_T(2);
  fx_mul<C>(FE_T, FE_T, FE_U);
  fx_frob<C>(FE_U, FE_T, 2);
  fx_mul<C>(FE_F, FE_U, FE_T);
  FX_T(3);
  if constexpr (C::CURVE_ID == 0) {
    // hard part, y0..y6 vectorial chain (pairing.hpp final_exp)
    fx_pow<C>(FE_A, FE_F, C::U_ABS, C::U_BITS, FE_T0);     // ft1
    FX_T(4);
    fx_pow<C>(FE_B, FE_A, C::U_ABS, C::U_BITS, FE_T0);     // ft2
    FX_T(5);
    fx_pow<C>(FE_C, FE_B, C::U_ABS, C::U_BITS, FE_T0);     // ft3
    FX_T(6);
    // (independent products of the chain run two at a time on the two halves of the block)
    fx_frob<C>(FE_Y0, FE_F, 1);
    fx_frob<C>(FE_T, FE_F, 2);
    fx_frob<C>(FE_U, FE_B, 1);
    fx_mul_pair<C>(FE_Y0, FE_Y0, FE_T, FE_Y4, FE_A, FE_U);
    fx_frob<C>(FE_T, FE_F, 3);
    fx_frob<C>(FE_U, FE_C, 1);
    fx_mul_pair<C>(FE_Y0, FE_Y0, FE_T, FE_Y6, FE_C, FE_U);
    fx_conj<C>(FE_Y4, FE_Y4);
    fx_conj<C>(FE_Y6, FE_Y6);
    fx_conj<C>(FE_Y1, FE_F);
    fx_frob<C>(FE_Y2, FE_B, 2);
    fx_frob<C>(FE_Y3, FE_A, 1);
    fx_conj<C>(FE_Y3, FE_Y3);
    fx_conj<C>(FE_Y5, FE_B);
    fx_mul_pair<C>(FE_T0, FE_Y6, FE_Y6, FE_T1, FE_Y3, FE_Y5);
    fx_mul<C>(FE_T0, FE_T0, FE_Y4);
    fx_mul<C>(FE_T0, FE_T0, FE_Y5);
    fx_mul_pair<C>(FE_T1, FE_T1, FE_T0, FE_T0, FE_T0, FE_Y2);
    fx_mul<C>(FE_T1, FE_T1, FE_T1);
    fx_mul<C>(FE_T1, FE_T1, FE_T0);
    fx_mul<C>(FE_T1, FE_T1, FE_T1);
    fx_mul_pair<C>(FE_T0, FE_T1, FE_Y1, FE_T1, FE_T1, FE_Y0);
    fx_mul<C>(FE_T0, FE_T0, FE_T0);
    fx_mul<C>(FE_F, FE_T1, FE_T0);
    FX_T(7);
  } else {
    // (p^4 - p^2 + 1)/r = c (x + p)(x^2 + p^2 - 1) + 1,  c = (x-1)^2/3,  x < 0
    fx_pow<C>(FE_A, FE_F, C::COFACTOR, C::COFACTOR_BITS, FE_T0);   // a = f^c
    fx_pow<C>(FE_T, FE_A, C::U_ABS, C::U_BITS, FE_T0);
    fx_conj<C>(FE_T, FE_T);                                 // a^x
    fx_frob<C>(FE_U, FE_A, 1);
    fx_mul<C>(FE_B, FE_T, FE_U);                            // b = a^x a^p
    fx_pow<C>(FE_T, FE_B, C::U_ABS, C::U_BITS, FE_T0);
    fx_conj<C>(FE_T, FE_T);                                 // b^x
    fx_pow<C>(FE_U, FE_T, C::U_ABS, C::U_BITS, FE_T0);
    fx_conj<C>(FE_U, FE_U);                                 // b^(x^2)
    fx_frob<C>(FE_T, FE_B, 2);
    fx_mul<C>(FE_U, FE_U, FE_T);
    fx_conj<C>(FE_T, FE_B);
    fx_mul<C>(FE_U, FE_U, FE_T);                            // d
    fx_mul<C>(FE_F, FE_U, FE_F);
  }
}

}  // namespace bgls
