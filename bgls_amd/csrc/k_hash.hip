// Hashing kernels: duplicate-message scan and hash-to-G1 for both curves (see h2c.hpp for the algorithms and
// their reference citations), plus the BLAKE2Xb expansion of the hashed-aggregation exponents.
#include <stdlib.h>
#include "dev_common.hpp"
#include "h2c.hpp"
#include "launch.hpp"
#include "rx_pow.hpp"
#include "rx_jac1.hpp"
#include "h2c_x.hpp"

using namespace bgls;

// ---- duplicate-message scan: open-addressing table of (index+1), exact byte comparison ----
// `seed`: drawn by the host per process and call (engine_core.inc dup_scan).  The slot of a record must not be predictable from its bytes: an
// unseeded FNV can be inverted, and 2^20 chosen messages that share one slot turn the scan into 2^40 probes (the reference's Go map is
// seeded per process for the same reason, bgls/bgls.go:139-150).
__device__ __forceinline__ uint64_t msg_hash64(const uint8_t* p, size_t n, uint64_t seed) {
  uint64_t h = 0xcbf29ce484222325ull ^ seed;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  h ^= h >> 29;
  h *= 0xbf58476d1ce4e5b9ull;
  h ^= h >> 32;
  return h;
}

// bucket / n_buckets (n_buckets > 1): only the records whose first byte is `bucket` mod n_buckets enter the table -- equal records
// share a bucket, so n_buckets scans (one per rank of a multi-GPU verification, over the all-gathered digests) find exactly what
// one scan of everything finds, each on 1 / n_buckets of the inserts.  A bucket far above its share (records that are not digests, or an
// adversary's: a signer can grind messages until every digest starts with the same byte) is reported as a hit, which the caller settles
// with the exact scan: never a miss.  Round 6 (advice r5): the bucketed scan gives up after `max_probe` probes instead of walking the
// whole table -- with the table at load factor 1 every insert walked all of it, 10^10 probes at 2^20 records per rank and more with
// every rank added -- and every thread leaves as soon as anybody has raised the flag (the verdict of the scan is decided by then in
// both modes).  The exact scan (n_buckets = 1) has a table of at least 2 n slots and probes without a bound.
__global__ void k_dup_check(MsgView mv, size_t n, uint32_t* table, uint32_t mask, uint32_t* flags, uint32_t bucket, uint32_t n_buckets, uint32_t max_probe,
                            uint64_t seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m = mv.ptr(i);
  const size_t len = mv.size(i);
  if (n_buckets > 1 && (len == 0 ? 0u : (uint32_t)m[0]) % n_buckets != bucket) return;
  uint32_t slot = (uint32_t)msg_hash64(m, len, seed) & mask;
  for (uint32_t probe = 0; probe <= max_probe; ++probe) {
    if ((probe & 31u) == 31u && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & FLAG_DUP)) return;
    uint32_t prev = atomicCAS(&table[slot], 0u, (uint32_t)i + 1u);
    if (prev == 0u) return;
    size_t j = prev - 1u;
    if (mv.size(j) == len) {
      const uint8_t* o = mv.ptr(j);
      bool same = true;
      for (size_t k = 0; k < len; ++k)
        if (o[k] != m[k]) {
          same = false;
          break;
        }
      if (same) {
        atomicOr(flags, FLAG_DUP);
        return;
      }
    }
    slot = (slot + 1u) & mask;
  }
  atomicOr(flags, FLAG_DUP);          // bucketed scan: too long a run of taken slots -- undecided, reported as a hit (see above)
}

// alt-bn128 try-and-increment as compacting rounds (curves/hash.go:53-77 has data-dependent trip
// counts: geometric(1/2) per message, so a per-lane loop idles most of the wave).  Round r tests
// LPM consecutive counters of every still-unfinished message on LPM adjacent lanes; the LOWEST
// successful counter wins (= what the sequential loop would have found), failures are appended to
// the next round's work list.  Counters 0..255 are covered by the fixed schedule in h2c_bn().
template <int LPM>
__global__ void __launch_bounds__(64, 3) k_h2c_bn_round(MsgView mv, size_t n, const uint32_t* list_in, const uint32_t* count_in,
                                                     uint32_t c0, uint32_t* list_out, uint32_t* count_out, int last,
                                                     Aff<F1<BN254>>* out, uint32_t* flags) {
  typedef BN254 C;
  const size_t count = count_in ? (size_t)*count_in : n;
  const size_t slots = (count * LPM + 63) / 64 * 64;
  const int lane = threadIdx.x;
  for (size_t slot = (size_t)blockIdx.x * 64 + lane; slot < slots; slot += (size_t)gridDim.x * 64) {
    const size_t item = slot / LPM;
    const uint32_t sub = (uint32_t)(slot % LPM);
    const uint32_t c = c0 + sub;
    const bool active = item < count && c < 256;
    size_t idx = 0;
    Fp<C> x, r;
    bool ok = false;
    if (item < count) idx = list_in ? list_in[item] : item;
    if (active) ok = bn_h2c_test(mv.ptr(idx), mv.size(idx), c, x, r);
    const unsigned long long ball = __ballot(ok);
    const int seg = (lane / LPM) * LPM;
    const unsigned long long segmask = LPM == 64 ? ball : ((ball >> seg) & ((1ull << (LPM & 63)) - 1ull));
    if (item < count) {
      if (segmask == 0) {
        if (sub == 0) {
          if (last) {
            atomicOr(flags, FLAG_HASH);
            out[idx] = {fp_zero<C>(), fp_zero<C>(), true};
          } else {
            list_out[atomicAdd(count_out, 1u)] = (uint32_t)idx;
          }
        }
      } else if (sub == (uint32_t)__builtin_ctzll(segmask)) {
        out[idx] = {x, r, false};             // r holds x^3+3, k_h2c_bn_finish takes the root
      }
    }
  }
}

// a^((p + 1) / 4) (M1 = false) or a^((p - 3) / 4) (M1 = true) on the carry-free limbs (rx_pow.hpp); `tab` is the wave's
// LDS table, RXP_W-bit sliding windows.  Same field element as fp_pow_w4 on the same exponent.
constexpr int RXP_W = 3;
// the number form of the powers: alt-bn128 on nine limbs of 29 bits (BN254W, round 5: a squaring is 45 + 81 multiplier instructions
// instead of 55 + 100), BLS12-381 on fourteen of 28
template <class C>
struct PowForm { typedef C type; };
template <>
struct PowForm<BN254> { typedef BN254W type; };
// BLS12-381 keeps table entry 0 in registers (sx_pow_sqrt's E0REG): three entries in LDS, twelve waves per CU instead of eleven
template <class C>
constexpr bool rxp_e0reg() { return C::CURVE_ID == 1; }
template <class C>
constexpr int rxp_lds_words() { return ((1 << (RXP_W - 1)) - (rxp_e0reg<C>() ? 1 : 0)) * PowForm<C>::type::RX_NL * 64; }
template <class C, bool M1>
__device__ __forceinline__ Fp<C> rx_sqrt_pow(const Fp<C>& a, i32* tab) {
  typedef typename PowForm<C>::type X;
  constexpr int N = X::RX_NL;
  const int lane = threadIdx.x & 63;
  auto ld = [&](int e, int i) { return tab[(e * N + i) * 64 + lane]; };
  auto st = [&](int e, int i, i32 v) { tab[(e * N + i) * 64 + lane] = v; };
  Fp<X> ax;
#pragma unroll
  for (int i = 0; i < C::L; ++i) ax.v[i] = a.v[i];
  const Sx<X, SX_T> r = sx_pow_sqrt<X, RXP_W, M1, rxp_e0reg<C>()>(ux_to_sx<X>(to_ux<X>(ax)), ld, st);   // (the low word of (p + 1) / 4 is odd: M1 borrows nothing)
  Ux<X> u;
#pragma unroll
  for (int i = 0; i < N; ++i) u.v[i] = (u32)r.v[i];
  const Fp<X> rx = from_ux<X>(u);
  Fp<C> out;
#pragma unroll
  for (int i = 0; i < C::L; ++i) out.v[i] = rx.v[i];
  return out;
}

// second half of the Legendre-symbol rounds: y = sqrt(x^3+3) with the reference's sign rule, once per message
__global__ void __launch_bounds__(64, 3) k_h2c_bn_finish(MsgView mv, size_t n, Aff<F1<BN254>>* out) {
  typedef BN254 C;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F1<C>> p = out[i];
  if (p.inf) return;
  __shared__ i32 tab[rxp_lds_words<C>()];
  Fp<C> r = rx_sqrt_pow<C, false>(p.y, tab);
  if (bn_h2c_sign(mv.ptr(i), mv.size(i))) r = fp_neg<C>(r);
  out[i].y = r;
}

// Small batches (n < 256: the reference's own n = 64 benchmark shape, the single message of a multi-signature): SIXTEEN lanes
// per message.  With one lane per message a wave walks the counters for as long as its unluckiest message needs (7 tries
// expected for 64 messages, ~50 us each on a lone wave: Keccak-f plus a Legendre symbol) and then hashes the sign byte; here
// the first round tests counters 0..14 side by side while lane 15 hashes the 0xFF-prefixed sign input (the same instruction
// stream: only the prefix byte differs), later rounds -- probability 2^-15 per message -- sixteen counters each.  The accepted
// counter is the LOWEST one with x^3 + 3 a square, as in the sequential loop (curves/hash.go:53-77): same (x, y).
#ifdef H2C_DBG
// development only: shader-clock stamps of block 0 of k_h2c_bn_wide (tools/exp/h2c_steps.py)
__device__ unsigned long long g_h2c_t[16];
#define H2C_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_h2c_t[k] = clock64(); } while (0)
extern "C" int bgls_dbg_h2c_dump(unsigned long long* o) { return (int)hipMemcpyFromSymbol(o, HIP_SYMBOL(g_h2c_t), sizeof(g_h2c_t)); }
#else
#define H2C_T(k) do { } while (0)
#endif
__global__ void __launch_bounds__(64, 3) k_h2c_bn_wide(MsgView mv, size_t n, Aff<F1<BN254>>* out, uint32_t* flags) {
  typedef BN254 C;
  const int lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
  const size_t i = (size_t)blockIdx.x * 4 + grp;
  const bool live = i < n;
  const uint8_t* msg = mv.ptr(live ? i : 0);
  const size_t len = mv.size(live ? i : 0);
  __shared__ i32 tab[rxp_lds_words<C>()];
  Fp<C> x, y2;
  bool done = !live, mine = false;
  u32 sign = 0;
  u32 base = 0;
  bool first = true;
  for (;;) {
    const u32 c = first ? (sub == 15 ? 255u : (u32)sub) : base + (u32)sub;
    ByteSrc src;
    src.msg = msg; src.len = len; src.pre[0] = (uint8_t)c; src.npre = 1; src.nsuf = 0;
    u32 d[8];
    H2C_T(0);
    keccak256_legacy(src, d);
    H2C_T(1);
    Fp<C> h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h.v[j] = d[7 - j];
    const Fp<C> xc = fp_to_mont<C>(h);
    const Fp<C> yc = fp_add<C>(fp_mul<C>(fp_sqr<C>(xc), xc), fp_load<C>(C::B));
    H2C_T(2);
    // every lane takes the square root of its own candidate and squares it back instead of asking for the Legendre symbol
    // first: the sixteen exponentiations are one instruction stream, and the accepted lane has its y already
    const Fp<C> rc = rx_sqrt_pow<C, false>(yc, tab);
    H2C_T(3);
    bool ok = fp_eq<C>(fp_sqr<C>(rc), yc) && c < 256u && !done;
    H2C_T(4);
    if (first) {
      sign = __shfl(d[7] & 1u, grp * 16 + 15);          // last digest byte of the 0xFF-prefixed hash, low bit
      if (sub == 15) ok = false;                         // counter 255 is tried in its turn, not in round 0
    }
    const unsigned long long ball = __ballot(ok);
    const u32 field = (u32)(ball >> (16 * grp)) & 0xFFFFu;
    if (!done && field) {
      mine = sub == (int)__builtin_ctz(field);
      done = true;
      if (mine) { x = xc; y2 = rc; }
    }
    base = first ? 15u : base + 16u;
    first = false;
    if (__ballot(!done) == 0ull || base >= 256u) break;
  }
  if (!done) {                                           // the reference would spin forever here (probability 2^-256)
    if (sub == 0) {
      atomicOr(flags, FLAG_HASH);
      out[i] = {fp_zero<C>(), fp_zero<C>(), true};
    }
    return;
  }
  H2C_T(5);
  if (mine) out[i] = {x, sign ? fp_neg<C>(y2) : y2, false};
  H2C_T(6);
}

// BLS12-381: one work item per (message, tag); the Shallue-van de Woestijne candidates as fractions, chosen by Legendre symbols, one square-root
// exponentiation, no inversion, all on the carry-free limbs: h2c_x.hpp (bls_sw_jac_x), which the CPU tier runs as well.
__global__ void __launch_bounds__(64, 3) k_bls_sw_jacobi(MsgView mv, size_t n_items, Jac<F1<BLS381>>* pts, uint32_t* kinds) {
  typedef BLS381 C;
  constexpr int N = C::RX_NL;
  size_t item = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (item >= n_items) return;
  const size_t msg = item >> 1;
  u32 d[16];
  bls_h2c_digest(mv.ptr(msg), mv.size(msg), (int)(item & 1), d);
  __shared__ i32 tab[rxp_lds_words<C>()];
  const int lane = threadIdx.x & 63;
  auto ld = [&](int e, int i) { return tab[(e * N + i) * 64 + lane]; };
  auto st = [&](int e, int i, i32 w) { tab[(e * N + i) * 64 + lane] = w; };
  Jac<F1<C>> pt;
  const uint32_t kind = bls_sw_jac_x<RXP_W, rxp_e0reg<C>()>(d, pt, ld, st);
  kinds[item] = kind;
  if (kind == H2C_SW) pts[item] = pt;
}

// per message: h * (sw_0 + sw_1) + special contributions, to affine.  The sum is normalised once (one
// Euclidean inversion) so that the 126-bit cofactor multiplication runs on a signed-digit (NAF) chain with
// mixed additions: 125 doublings + 42 additions of 11 field products instead of 63 of 16.
//
// RAW = true is the verification path's form: the cofactor is NOT cleared here.  The reduced ate pairing is
// bilinear in its first argument on all of E(Fp), e(h S, Q) = e(S, Q)^h, so the whole batch shares ONE
// exponentiation by h in GT (k_cofactor_epilogue) instead of n 126-bit scalar multiplications; the rare
// "+-generator" outcomes enter as +-G1K, G1K = (h^-1 mod r) g1.
template <bool RAW>
__global__ void __launch_bounds__(64) k_bls_combine(size_t n, const Jac<F1<BLS381>>* pts, const uint32_t* kinds, Aff<F1<BLS381>>* out) {
  typedef BLS381 C;
  typedef F1<C> F;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac<F> sw = jac_inf<F>(), special = jac_inf<F>();
  const Aff<F> g1 = {fp_load<C>(RAW ? C::G1KX : C::G1X), fp_load<C>(RAW ? C::G1KY : C::G1Y), false};
  for (int k = 0; k < 2; ++k) {
    const uint32_t kind = kinds[2 * i + k];
    if (kind == H2C_SW) sw = jac_add<F>(sw, pts[2 * i + k]);
    else if (kind == H2C_PLUS_G1) special = jac_add_aff<F>(special, g1);
    else if (kind == H2C_MINUS_G1) special = jac_add_aff<F>(special, aff_neg<F>(g1));
  }
  if constexpr (RAW) {
    out[i] = jac_to_aff<F>(jac_add<F>(sw, special));
    return;
  }
  const Aff<F> S = jac_to_aff<F>(sw);
  const Aff<F> nS = aff_neg<F>(S);
  Jac<F> r = jac_inf<F>();
  for (int d = 0; d < C::COFACTOR_NAF_LEN; ++d) {
    r = jac_dbl<F>(r);
    const int dig = C::COFACTOR_NAF[d];
    if (dig != 0) r = jac_add_aff<F>(r, dig > 0 ? S : nS);
  }
  out[i] = jac_to_aff<F>(jac_add<F>(r, special));
}

// k_bls_combine<false> with the 126-bit cofactor multiplication on the carry-free limbs (rx_jac1.hpp; round 4): the public
// HashToG1 (curves/bls12_381.go:349-376) clears the cofactor per message -- 125 doublings + 42 mixed additions of the NAF
// chain, 1 337 field products -- and that chain is what this kernel moves off the 32-bit Montgomery form (51 ms per 2^20
// messages there).  Sum of the two encodings, normalisation and the rare special outcomes stay as they were; same chain,
// same group law, same bytes (the reference's 11 KATs).
__global__ void __launch_bounds__(64) k_bls_combine_x(size_t n, const Jac<F1<BLS381>>* pts, const uint32_t* kinds, Aff<F1<BLS381>>* out) {
  typedef BLS381 C;
  typedef F1<C> F;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac<F> sw = jac_inf<F>(), special = jac_inf<F>();
  const Aff<F> g1 = {fp_load<C>(C::G1X), fp_load<C>(C::G1Y), false};
  for (int k = 0; k < 2; ++k) {
    const uint32_t kind = kinds[2 * i + k];
    if (kind == H2C_SW) sw = jac_add<F>(sw, pts[2 * i + k]);
    else if (kind == H2C_PLUS_G1) special = jac_add_aff<F>(special, g1);
    else if (kind == H2C_MINUS_G1) special = jac_add_aff<F>(special, aff_neg<F>(g1));
  }
  const Aff1<C> S = aff1_from_mont<C>(jac_to_aff<F>(sw));
  const Aff1<C> nS = aff1_neg<C>(S);
  Jac1<C> r = jac1_inf<C>();
#pragma unroll 1
  for (int d = 0; d < C::COFACTOR_NAF_LEN; ++d) {
    r = jac1_dbl<C>(r);
    const int dig = C::COFACTOR_NAF[d];
    if (dig != 0) r = jac1_madd<C>(r, dig > 0 ? S : nS);
  }
  out[i] = jac_to_aff<F>(jac_add<F>(jac1_to_mont<C>(r), special));
}

// k_bls_combine<true> on the carry-free limbs (round 4), for the batches that k_bls_combine_raw_batched does not take (below 2^18
// messages: a lone 2^16 verification, n = 64, the one message of a multi-signature).  There the kernel is a few lone waves and
// every field product of the 32-bit form is a dependent carry chain (~5 us each on BLS12-381): the general addition A + B, the
// binary-Euclid inversion and the normalisation took 0.71 ms for 2^16 messages.  Here the addition is rx_jac1.hpp's and the
// inverse is fp_inv's division-step form (uniform control flow: 62 us for a wave of 64 different values where the binary
// Euclid took 670 and Z^(p - 2) on these limbs 510, tools/mb_inv.hip).  Same point, same affine bytes.
__global__ void __launch_bounds__(64) k_bls_combine_raw_x(size_t n, const Jac<F1<BLS381>>* pts, const uint32_t* kinds, Aff<F1<BLS381>>* out) {
  typedef BLS381 C;
  typedef F1<C> F;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Aff1<C> g1 = aff1_from_mont<C>(Aff<F>{fp_load<C>(C::G1KX), fp_load<C>(C::G1KY), false});
  Jac1<C> sw = jac1_inf<C>();
  for (int k = 0; k < 2; ++k) {
    const uint32_t kind = kinds[2 * i + k];
    if (kind == H2C_SW) {
      const Jac<F> q = pts[2 * i + k];
      Jac1<C> qx;
      qx.X = sx_as<SX_F, C>(sx_from_mont<C>(q.X));
      qx.Y = sx_as<SX_F, C>(sx_from_mont<C>(q.Y));
      qx.Z = sx_as<SX_F, C>(sx_from_mont<C>(q.Z));
      qx.inf = jac_is_inf<F>(q);
      sw = jac1_add<C>(sw, qx);
    } else if (kind == H2C_PLUS_G1) {
      sw = jac1_madd<C>(sw, g1);
    } else if (kind == H2C_MINUS_G1) {
      sw = jac1_madd<C>(sw, aff1_neg<C>(g1));
    }
  }
  if (sw.inf) {
    out[i] = {fp_zero<C>(), fp_zero<C>(), true};
    return;
  }
  const Sx<C, SX_T> zi = sx_from_mont<C>(fp_inv<C>(sx_to_mont<C>(sw.Z)));
  const Sx<C, SX_T> zi2 = s1_sqr<C>(zi);
  out[i] = {sx_to_mont<C>(s1_mul<C>(sw.X, zi2)), sx_to_mont<C>(s1_mul<C>(s1_mul<C>(sw.Y, zi2), zi)), false};
}

// The verification path's combine (RAW) with the normalisation shared: a thread sums the two encodings of KB consecutive
// messages, parks the Jacobian sums in the work-item array, and inverts the product of their Z coordinates once
// (Montgomery's trick: the binary-Euclid inversion is ~240 field products' worth of divergent work, the trick costs 3 per
// message).  A sum at infinity contributes 1 to the product and leaves as the point at infinity.
template <int KB>
__global__ void __launch_bounds__(64) k_bls_combine_raw_batched(size_t n, Jac<F1<BLS381>>* pts, const uint32_t* kinds, Aff<F1<BLS381>>* out) {
  typedef BLS381 C;
  typedef F1<C> F;
  const size_t i0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * KB;
  if (i0 >= n) return;
  const Aff<F> g1 = {fp_load<C>(C::G1KX), fp_load<C>(C::G1KY), false};
  Fp<C> pre[KB];                                            // pre[k] = Z_0 ... Z_k (infinite sums skipped)
  Fp<C> run = fp_one<C>();
#pragma unroll 1
  for (int k = 0; k < KB; ++k) {
    const size_t i = i0 + k;
    if (i < n) {
      Jac<F> sw = jac_inf<F>();
      for (int h = 0; h < 2; ++h) {
        const uint32_t kind = kinds[2 * i + h];
        if (kind == H2C_SW) sw = jac_add<F>(sw, pts[2 * i + h]);
        else if (kind == H2C_PLUS_G1) sw = jac_add_aff<F>(sw, g1);
        else if (kind == H2C_MINUS_G1) sw = jac_add_aff<F>(sw, aff_neg<F>(g1));
      }
      pts[2 * i] = sw;
      if (!jac_is_inf<F>(sw)) run = fp_mul<C>(run, sw.Z);
    }
    pre[k] = run;
  }
  Fp<C> inv = fp_inv<C>(run);
#pragma unroll 1
  for (int k = KB - 1; k >= 0; --k) {
    const size_t i = i0 + k;
    if (i >= n) continue;
    const Jac<F> sw = pts[2 * i];
    if (jac_is_inf<F>(sw)) {
      out[i] = {fp_zero<C>(), fp_zero<C>(), true};
      continue;
    }
    const Fp<C> zi = k ? fp_mul<C>(inv, pre[k - 1]) : inv;   // 1 / Z_k
    inv = fp_mul<C>(inv, sw.Z);
    const Fp<C> zi2 = fp_sqr<C>(zi);
    out[i] = {fp_mul<C>(sw.X, zi2), fp_mul<C>(fp_mul<C>(sw.Y, zi2), zi), false};
  }
}

// ---- BLAKE2Xb expansion (bgls/blsHAE.go:80-93): node i of the XOF is one compression of the 64-byte root with its
// own parameter block (hashes.hpp blake2xb_node); one lane per node.  The root is produced on the host side.
__global__ void __launch_bounds__(64) k_blake2x_expand(const u64* root, u32 xof_len, uint8_t* out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nnodes = (xof_len + 63u) / 64u;
  if (i >= nnodes) return;
  u64 r[8], o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = root[k];
  const u32 rest = xof_len - 64u * i;
  const u32 take = rest < 64u ? rest : 64u;
  blake2xb_node(r, i, xof_len, take, o);
  for (u32 b = 0; b < take; ++b) out[(size_t)64 * i + b] = (uint8_t)(o[b >> 3] >> (8 * (b & 7)));
}

// 16-byte digests of fixed-stride messages: the first 16 bytes of BLAKE2b-512 (the multi-GPU duplicate scan exchanges these
// instead of the messages; equal messages have equal digests, so "no two digests equal" proves "no two messages equal" and a
// hit falls back to the exact byte scan)
__global__ void k_msg_digest(MsgView mv, size_t n, uint8_t* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  ByteSrc src = {mv.ptr(i), mv.size(i), {0}, 0, {0, 0, 0, 0}, 0};
  u32 d[16];
  blake2b512(src, d);
  uint4 w = make_uint4(__builtin_bswap32(d[0]), __builtin_bswap32(d[1]), __builtin_bswap32(d[2]), __builtin_bswap32(d[3]));
  *reinterpret_cast<uint4*>(out + 16 * i) = w;
}

// ======================================================================= launchers
namespace bgls {
namespace kl {

void msg_digest(hipStream_t st, MsgView mv, size_t n, uint8_t* out) { k_msg_digest<<<nblk(n, 256), 256, 0, st>>>(mv, n, out); }

// ---- the digest exchange of a multi-GPU verification as an all-to-all by bucket (round 6).  Rank r owns the digests whose first byte is r mod N.
// k_digest_pack sorts a rank's n digests into N send slots of `cap` records each; a slot's unused records are padding whose first byte
// belongs to ANOTHER bucket ((b + 1) mod N: the receiver's bucket filter skips them), so every rank sends and receives N equal chunks --
// no count exchange, no host round trip -- and holds, after the all-to-all, exactly the digests of its bucket: cap x N records instead of
// the n x N an all-gather hands to every rank.  A slot that overflows (a bucket far above its share: an adversary's messages) raises the
// duplicate bit of the probe word: "undecided", settled by the exact scan over the messages like any digest hit.
__global__ void k_digest_pack_fill(uint8_t* out, size_t cap, uint32_t n_buckets) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= cap * n_buckets) return;
  const uint32_t b = (uint32_t)(i / cap);
  uint4 pad = make_uint4((b + 1u) % n_buckets, 0u, 0u, 0u);             // little-endian: byte 0 = the neighbour's bucket
  reinterpret_cast<uint4*>(out)[i] = pad;
}
__global__ void k_digest_pack(const uint8_t* dig, size_t n, uint8_t* out, size_t cap, uint32_t n_buckets, uint32_t* counts, uint32_t* flags) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 d = reinterpret_cast<const uint4*>(dig)[i];
  const uint32_t b = (d.x & 0xFFu) % n_buckets;
  const uint32_t pos = atomicAdd(&counts[b], 1u);
  if (pos >= cap) {
    atomicOr(flags, FLAG_DUP);
    return;
  }
  reinterpret_cast<uint4*>(out)[(size_t)b * cap + pos] = d;
}
void digest_pack(hipStream_t st, const uint8_t* dig, size_t n, uint8_t* out, size_t cap, uint32_t n_buckets, uint32_t* counts, uint32_t* flags) {
  k_digest_pack_fill<<<nblk(cap * n_buckets, 256), 256, 0, st>>>(out, cap, n_buckets);
  if (n) k_digest_pack<<<nblk(n, 256), 256, 0, st>>>(dig, n, out, cap, n_buckets, counts, flags);
}

void dup_check(hipStream_t st, MsgView mv, size_t n, uint32_t* table, uint32_t mask, uint32_t* flags, uint32_t bucket, uint32_t n_buckets, uint64_t seed,
               bool unbounded) {
  // bucketed scan: at most DUP_MAX_PROBE probes per record (a fair bucket's table is at most half full: runs of a thousand taken slots do not occur)
  const uint32_t max_probe = n_buckets > 1 && !unbounded && mask > DUP_MAX_PROBE ? DUP_MAX_PROBE : mask;
  k_dup_check<<<nblk(n, 256), 256, 0, st>>>(mv, n, table, mask, flags, bucket, n_buckets, max_probe, seed);
}

// alt-bn128: (lanes per message, first counter): 1@0, 4@1, 32@5, then 64 lanes per message up to counter 255.  Each
// round costs one try of latency, so the schedule is short: after three rounds a message is still unfinished with
// probability 2^-37.  Acceptance test = Legendre symbol, square root once at the end (k_h2c_bn_finish).
void h2c_bn(hipStream_t st, MsgView mv, size_t n, uint32_t* lists, uint32_t* cn, Aff<F1<BN254>>* out, uint32_t* flags, bool lean) {
  if (n < 256) {
    k_h2c_bn_wide<<<nblk(n, 4), 64, 0, st>>>(mv, n, out, flags);
    return;
  }
  uint32_t* L0 = lists;
  uint32_t* L1 = L0 + n;
  auto grid = [&](double expect_items, int lpm) {
    double lanes = expect_items * lpm * 2.0 + 512.0;
    size_t b = (size_t)(lanes / 64.0) + 1;
    // Never more blocks than the chip holds at once (162 registers: three one-wave blocks per SIMD, 3072 in all; the kernels walk
    // their work lists with grid-stride loops).  Round 4: a launch that OVERSUBSCRIBES the chip leaves the workgroup distributor in
    // a state in which a later launch that fills the chip EXACTLY -- the 1024 blocks of a lone 2^16 Miller launch, four per CU --
    // is dealt three blocks on some CUs and stalls on the rest until the first blocks finish (the same wave-cycles in 1.5x the
    // time; it survives host synchronisation and intervening small kernels).  The 4105 blocks of the second round did that to
    // 45 % of the lone alt-bn128 2^16 verifications: Miller stage 6.6 instead of 4.4 ms (tools/exp/pp_lone.sh: 11 of 24 launches
    // slow with a cap of 4096 or 8192 blocks, 0 of 24 with 3072 or 2048).
    const size_t cap = 3072;
    return (unsigned)(b > cap ? cap : b);
  };
  const double N = (double)n;
  if (lean || n >= ((size_t)1 << 17)) {
    // Large batches -- and any batch that shares the machine with other verifications (throughput mode) -- are bound by
    // the number of tests, not by the latency of a round: test one counter at a time while half of the messages are still
    // open (2.3 n tests in all instead of the 4 n of the schedule below, two more rounds of latency).
    k_h2c_bn_round<1><<<nblk(n, 64), 64, 0, st>>>(mv, n, nullptr, nullptr, 0, L0, cn + 1, 0, out, flags);
    k_h2c_bn_round<1><<<grid(N / 2, 1), 64, 0, st>>>(mv, n, L0, cn + 1, 1, L1, cn + 2, 0, out, flags);
    k_h2c_bn_round<2><<<grid(N / 4, 2), 64, 0, st>>>(mv, n, L1, cn + 2, 2, L0, cn + 3, 0, out, flags);
    k_h2c_bn_round<4><<<grid(N / 16, 4), 64, 0, st>>>(mv, n, L0, cn + 3, 4, L1, cn + 4, 0, out, flags);
    k_h2c_bn_round<8><<<grid(N / 256, 8), 64, 0, st>>>(mv, n, L1, cn + 4, 8, L0, cn + 5, 0, out, flags);
    k_h2c_bn_round<32><<<grid(N / 65536, 32), 64, 0, st>>>(mv, n, L0, cn + 5, 16, L1, cn + 6, 0, out, flags);
    k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L1, cn + 6, 48, L0, cn + 7, 0, out, flags);
    k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L0, cn + 7, 112, L1, cn + 8, 0, out, flags);
    k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L1, cn + 8, 176, L0, cn + 9, 0, out, flags);
    k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L0, cn + 9, 240, L1, cn + 10, 1, out, flags);
    k_h2c_bn_finish<<<nblk(n, 64), 64, 0, st>>>(mv, n, out);
    return;
  }
  k_h2c_bn_round<1><<<nblk(n, 64), 64, 0, st>>>(mv, n, nullptr, nullptr, 0, L0, cn + 1, 0, out, flags);
  k_h2c_bn_round<4><<<grid(N / 2, 4), 64, 0, st>>>(mv, n, L0, cn + 1, 1, L1, cn + 2, 0, out, flags);
  k_h2c_bn_round<32><<<grid(N / 32, 32), 64, 0, st>>>(mv, n, L1, cn + 2, 5, L0, cn + 3, 0, out, flags);
  k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L0, cn + 3, 37, L1, cn + 4, 0, out, flags);
  k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L1, cn + 4, 101, L0, cn + 5, 0, out, flags);
  k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L0, cn + 5, 165, L1, cn + 6, 0, out, flags);
  k_h2c_bn_round<64><<<8, 64, 0, st>>>(mv, n, L1, cn + 6, 229, L0, cn + 7, 1, out, flags);
  k_h2c_bn_finish<<<nblk(n, 64), 64, 0, st>>>(mv, n, out);
}

// BLS12-381: pts / kinds hold 2n work items (message, tag); raw = uncleared sum (verification path, cofactor in GT)
void h2c_bls(hipStream_t st, MsgView mv, size_t n, Jac<F1<BLS381>>* pts, uint32_t* kinds, Aff<F1<BLS381>>* out, bool raw) {
  const size_t items = 2 * n;
  // BGLS_LEGACY bit 32 (engine.hip): the combine step's G1 arithmetic on the 32-bit chain instead of the carry-free limbs (A/B runs)
  static const bool g1x = [] { const char* e = getenv("BGLS_LEGACY"); return !(e && (atoi(e) & 32)); }();
  k_bls_sw_jacobi<<<nblk(items, 64), 64, 0, st>>>(mv, items, pts, kinds);
  if (raw && n >= ((size_t)1 << 18)) k_bls_combine_raw_batched<4><<<nblk((n + 3) / 4, 64), 64, 0, st>>>(n, pts, kinds, out);
  else if (raw) {
    if (g1x) k_bls_combine_raw_x<<<nblk(n, 64), 64, 0, st>>>(n, pts, kinds, out);
    else k_bls_combine<true><<<nblk(n, 64), 64, 0, st>>>(n, pts, kinds, out);
  } else {
    if (g1x) k_bls_combine_x<<<nblk(n, 64), 64, 0, st>>>(n, pts, kinds, out);
    else k_bls_combine<false><<<nblk(n, 64), 64, 0, st>>>(n, pts, kinds, out);
  }
}

void blake2x_expand(hipStream_t st, const uint64_t* root, uint32_t xof_len, uint8_t* out) {
  k_blake2x_expand<<<nblk((xof_len + 63) / 64, 64), 64, 0, st>>>((const u64*)root, xof_len, out);
}

}  // namespace kl
}  // namespace bgls
