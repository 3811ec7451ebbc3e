// HIP kernels + C ABI of the aggregate-verify engine (see include/bgls_hip.h for the contract).
//
// Pipeline of bgls.VerifyAggregateSignature (bgls/bgls.go:94-119) on the device:
//   k_dup_check   exact duplicate-message scan        (containsDuplicateMessage, bgls.go:139-150)
//   k_h2c         H(m_i) for every message            (concurrentHash, bgls.go:107-111,134-137)
//   k_sig_prep    -sigma                              (aggsig.Mul(-1), bgls.go:112)
//   k_miller      Miller value of every (H(m_i), pk_i) and of (-sigma, g2)
//                                                     (concurrentPair, curves/curve.go:132-134,217-223)
//   k_f12_reduce  product of the Miller values        (GT Add tree, curves/curve.go:141-169)
//   k_final       ONE final exponentiation, compare with 1 (Equals(GetGTIdentity), bgls.go:115-118)
// Round-1 mapping: one work-item per pairing / message / point (thread-local tower arithmetic).
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include <atomic>
#include <string>

#include "pairing.hpp"
#include "h2c.hpp"
#include "wire.hpp"
#include "coop.hpp"
#include "coop_r28.hpp"
#include "finalexp.hpp"
#include "../../include/bgls_hip.h"

using namespace bgls;

// ======================================================================= device side
struct MsgView {
  const uint8_t* base;
  const uint64_t* off;  // n+1 offsets, or nullptr for fixed stride
  size_t len, stride;
  __device__ __forceinline__ const uint8_t* ptr(size_t i) const { return off ? base + off[i] : base + i * stride; }
  __device__ __forceinline__ size_t size(size_t i) const { return off ? (size_t)(off[i + 1] - off[i]) : len; }
};

enum : uint32_t { FLAG_DUP = 1u, FLAG_ENC = 2u, FLAG_HASH = 4u };

// ---- duplicate-message scan: open-addressing table of (index+1), exact byte comparison ----
__device__ __forceinline__ uint64_t msg_hash64(const uint8_t* p, size_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  h ^= h >> 29;
  h *= 0xbf58476d1ce4e5b9ull;
  h ^= h >> 32;
  return h;
}

__global__ void k_dup_check(MsgView mv, size_t n, uint32_t* table, uint32_t mask, uint32_t* flags) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* m = mv.ptr(i);
  const size_t len = mv.size(i);
  uint32_t slot = (uint32_t)msg_hash64(m, len) & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
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
}

// ---- hash to G1 ----
template <class C>
__global__ void __launch_bounds__(64) k_h2c(MsgView mv, size_t n, Aff<F1<C>>* out, uint32_t* flags) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if constexpr (C::CURVE_ID == 0) {
    Aff<F1<C>> p;
    if (!bn_hash_to_g1(mv.ptr(i), mv.size(i), p)) {
      atomicOr(flags, FLAG_HASH);
      p = {fp_zero<C>(), fp_zero<C>(), true};
    }
    out[i] = p;
  } else {
    out[i] = bls_hash_to_g1(mv.ptr(i), mv.size(i));
  }
}

// alt-bn128 try-and-increment as compacting rounds (curves/hash.go:53-77 has data-dependent trip
// counts: geometric(1/2) per message, so a per-lane loop idles most of the wave).  Round r tests
// LPM consecutive counters of every still-unfinished message on LPM adjacent lanes; the LOWEST
// successful counter wins (= what the sequential loop would have found), failures are appended to
// the next round's work list.  Counters 0..255 are covered by the fixed schedule in h2c_bn().
template <int LPM, bool JAC>
__global__ void __launch_bounds__(64) k_h2c_bn_round(MsgView mv, size_t n, const uint32_t* list_in, const uint32_t* count_in,
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
    if (active) ok = JAC ? bn_h2c_test(mv.ptr(idx), mv.size(idx), c, x, r) : bn_h2c_try(mv.ptr(idx), mv.size(idx), c, x, r);
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
        if (!JAC && bn_h2c_sign(mv.ptr(idx), mv.size(idx))) r = fp_neg<C>(r);
        out[idx] = {x, r, false};             // JAC: r holds x^3+3, k_h2c_bn_finish takes the root
      }
    }
  }
}

// ---- BLS12-381 hash-to-G1 as a staged pipeline (curves/bls12_381.go:361-393, curves/hash.go:86-167) ----
// item = 2*message + tag.  STAGE 0 hashes, classifies t and tests candidate x0; STAGE 1 tests x1 = -1 - x0
// on the items that failed; STAGE 2 takes x2 (always a square).  Failures are compacted into work lists, so
// every lane of every round does exactly one square-root exponentiation.
template <int STAGE>
__global__ void __launch_bounds__(64) k_bls_sw(MsgView mv, size_t n_items, const uint32_t* list_in, const uint32_t* count_in,
                                               uint32_t* list_out, uint32_t* count_out, Aff<F1<BLS381>>* pts, uint32_t* kinds) {
  typedef BLS381 C;
  const size_t count = count_in ? (size_t)*count_in : n_items;
  for (size_t slot = (size_t)blockIdx.x * 64 + threadIdx.x; slot < count; slot += (size_t)gridDim.x * 64) {
    const size_t item = list_in ? list_in[slot] : slot;
    const size_t msg = item >> 1;
    Fp<C> tm;
    Fp<C> t = bls_h2c_t(mv.ptr(msg), mv.size(msg), (int)(item & 1), tm);
    if (STAGE == 0) {
      uint32_t kind = H2C_PENDING;
      if (fp_is_zero<C>(t)) kind = H2C_INF;
      else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT1))) kind = H2C_PLUS_G1;
      else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT2))) kind = H2C_MINUS_G1;
      if (kind != H2C_PENDING) {
        kinds[item] = kind;
        continue;
      }
    }
    BlsSwPrep pr = bls_sw_prep(tm);
    Fp<C> x = STAGE == 0 ? pr.x0 : STAGE == 1 ? fp_sub<C>(fp_neg<C>(pr.x0), fp_one<C>()) : pr.x2;
    Fp<C> y;
    const bool ok = bls_sw_try(x, y) || STAGE == 2;
    if (ok) {
      if (fp_plain_parity<C>(fp_from_mont<C>(y)) != fp_plain_parity<C>(t)) y = fp_neg<C>(y);
      pts[item] = {x, y, false};
      kinds[item] = H2C_SW;
    } else {
      list_out[atomicAdd(count_out, 1u)] = (uint32_t)item;
    }
  }
}

// second half of the Legendre-symbol rounds: y = sqrt(x^3+3) with the reference's sign rule, once per message
__global__ void __launch_bounds__(64) k_h2c_bn_finish(MsgView mv, size_t n, Aff<F1<BN254>>* out) {
  typedef BN254 C;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F1<C>> p = out[i];
  if (p.inf) return;
  Fp<C> r = fp_sqrt_candidate<C>(p.y);
  if (bn_h2c_sign(mv.ptr(i), mv.size(i))) r = fp_neg<C>(r);
  out[i].y = r;
}

// alt-bn128 try-and-increment with the acceptance test done by the Legendre symbol (fp_jacobi): the lane
// walks the counters with cheap tests only and pays ONE square-root exponentiation, for the accepted x.
// Same accepted counter, same (x, y) as curves/hash.go:53-77.
__global__ void __launch_bounds__(64) k_h2c_bn_jacobi(MsgView mv, size_t n, Aff<F1<BN254>>* out, uint32_t* flags) {
  typedef BN254 C;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* msg = mv.ptr(i);
  const size_t len = mv.size(i);
  Fp<C> x, y2;
  bool found = false;
  for (u32 c = 0; c < 256 && !found; ++c) {
    ByteSrc src;
    src.msg = msg; src.len = len; src.pre[0] = (uint8_t)c; src.npre = 1; src.nsuf = 0;
    u32 d[8];
    keccak256_legacy(src, d);
    Fp<C> h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h.v[j] = d[7 - j];
    x = fp_to_mont<C>(h);
    y2 = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), fp_load<C>(C::B));
    found = fp_jacobi<C>(y2) >= 0;
  }
  if (!found) {
    atomicOr(flags, FLAG_HASH);
    out[i] = {fp_zero<C>(), fp_zero<C>(), true};
    return;
  }
  Fp<C> r = fp_sqrt_candidate<C>(y2);
  if (bn_h2c_sign(msg, len)) r = fp_neg<C>(r);
  out[i] = {x, r, false};
}

// BLS12-381: one work item per (message, tag); candidates chosen by Legendre symbols (isQuadRes,
// curves/hash.go:254-265), then exactly one square-root exponentiation.
__global__ void __launch_bounds__(64) k_bls_sw_jacobi(MsgView mv, size_t n_items, Aff<F1<BLS381>>* pts, uint32_t* kinds) {
  typedef BLS381 C;
  size_t item = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (item >= n_items) return;
  const size_t msg = item >> 1;
  Fp<C> tm;
  Fp<C> t = bls_h2c_t(mv.ptr(msg), mv.size(msg), (int)(item & 1), tm);
  uint32_t kind = H2C_SW;
  if (fp_is_zero<C>(t)) kind = H2C_INF;
  else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT1))) kind = H2C_PLUS_G1;
  else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT2))) kind = H2C_MINUS_G1;
  kinds[item] = kind;
  if (kind != H2C_SW) return;
  BlsSwPrep pr = bls_sw_prep(tm);
  const Fp<C> b = fp_load<C>(C::B);
  Fp<C> x = pr.x0;
  Fp<C> g = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), b);
  if (fp_jacobi<C>(g) < 0) {
    x = fp_sub<C>(fp_neg<C>(pr.x0), fp_one<C>());
    g = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), b);
    if (fp_jacobi<C>(g) < 0) {
      x = pr.x2;
      g = fp_add<C>(fp_mul<C>(fp_sqr<C>(x), x), b);
    }
  }
  Fp<C> y = fp_sqrt_candidate<C>(g);
  if (fp_plain_parity<C>(fp_from_mont<C>(y)) != fp_plain_parity<C>(t)) y = fp_neg<C>(y);
  pts[item] = {x, y, false};
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
__global__ void __launch_bounds__(64) k_bls_combine(size_t n, const Aff<F1<BLS381>>* pts, const uint32_t* kinds, Aff<F1<BLS381>>* out) {
  typedef BLS381 C;
  typedef F1<C> F;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac<F> sw = jac_inf<F>(), special = jac_inf<F>();
  const Aff<F> g1 = {fp_load<C>(RAW ? C::G1KX : C::G1X), fp_load<C>(RAW ? C::G1KY : C::G1Y), false};
  for (int k = 0; k < 2; ++k) {
    const uint32_t kind = kinds[2 * i + k];
    if (kind == H2C_SW) sw = jac_add_aff<F>(sw, pts[2 * i + k]);
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

template <class C>
__global__ void k_g1_to_bytes(const Aff<F1<C>>* in, size_t n, uint8_t* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  g1_to_bytes<C>(out + i * 2 * C::FP_BYTES, in[i]);
}

// parse n G1 points (optionally negating them); bad encodings / off-curve points set FLAG_ENC
template <class C>
__global__ void k_g1_parse(const uint8_t* in, size_t n, int negate, Aff<F1<C>>* out, uint32_t* flags) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F1<C>> p;
  bool ok = g1_from_bytes<C>(p, in + i * 2 * C::FP_BYTES);
  ok = ok && aff_on_curve<F1<C>>(p);
  if (!ok) atomicOr(flags, FLAG_ENC);
  if (negate) p = aff_neg<F1<C>>(p);
  out[i] = p;
}

// ---- Miller loops: one work-item per pair ----
template <class C>
__global__ void __launch_bounds__(64) k_miller(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long gen_at,
                                               Fp12<C>* out, uint32_t* flags) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F2<C>> Q;
  if ((long long)i == gen_at) {
    Q.x = f2_load<C>(C::G2);
    Q.y = f2_load<C>(C::G2 + 2 * C::L);
    Q.inf = false;
  } else {
    size_t k = (gen_at >= 0 && (long long)i > gen_at) ? i - 1 : i;
    bool ok = g2_from_bytes<C>(Q, g2s + k * 4 * C::FP_BYTES);
    ok = ok && aff_on_curve<F2<C>>(Q);
    if (!ok) atomicOr(flags, FLAG_ENC);
  }
  Aff<F1<C>> P = g1s[i];
  out[i] = miller_loop<C>(P, Q);
}

// out[t] = prod in[t*R .. min(n, (t+1)*R))
template <class C>
__global__ void __launch_bounds__(64) k_f12_reduce(const Fp12<C>* in, size_t n, int R, Fp12<C>* out) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t lo = t * (size_t)R;
  if (lo >= n) return;
  size_t hi = lo + R < n ? lo + R : n;
  Fp12<C> acc = in[lo];
  for (size_t k = lo + 1; k < hi; ++k) acc = f12_mul<C>(acc, in[k]);
  out[t] = acc;
}

template <class C>
__global__ void k_f12_to_bytes(const Fp12<C>* in, uint8_t* out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) gt_to_bytes<C>(out, in[0]);
}

// product of `count` serialised partials -> final exponentiation -> GT bytes + (== 1) verdict
template <class C>
__global__ void __launch_bounds__(64) k_final(const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out,
                                              uint32_t* verdict, uint32_t* flags) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Fp12<C> acc = f12_one<C>();
  for (size_t k = 0; k < count; ++k) {
    Fp12<C> x;
    if (!gt_from_bytes<C>(x, partials + k * 12 * C::FP_BYTES)) atomicOr(flags, FLAG_ENC);
    acc = f12_mul<C>(acc, x);
  }
  if (do_final_exp) acc = final_exp<C>(acc);
  if (gt_out) gt_to_bytes<C>(gt_out, acc);
  verdict[0] = f12_is_one<C>(acc) ? 1u : 0u;
}

// ---- point sums ----
template <class F>
__device__ __forceinline__ bool aff_from_bytes(Aff<F>& p, const uint8_t* b);
template <>
__device__ __forceinline__ bool aff_from_bytes<F1<BN254>>(Aff<F1<BN254>>& p, const uint8_t* b) { return g1_from_bytes<BN254>(p, b); }
template <>
__device__ __forceinline__ bool aff_from_bytes<F1<BLS381>>(Aff<F1<BLS381>>& p, const uint8_t* b) { return g1_from_bytes<BLS381>(p, b); }
template <>
__device__ __forceinline__ bool aff_from_bytes<F2<BN254>>(Aff<F2<BN254>>& p, const uint8_t* b) { return g2_from_bytes<BN254>(p, b); }
template <>
__device__ __forceinline__ bool aff_from_bytes<F2<BLS381>>(Aff<F2<BLS381>>& p, const uint8_t* b) { return g2_from_bytes<BLS381>(p, b); }

template <class F>
__device__ __forceinline__ void aff_to_bytes(uint8_t* b, const Aff<F>& p);
template <>
__device__ __forceinline__ void aff_to_bytes<F1<BN254>>(uint8_t* b, const Aff<F1<BN254>>& p) { g1_to_bytes<BN254>(b, p); }
template <>
__device__ __forceinline__ void aff_to_bytes<F1<BLS381>>(uint8_t* b, const Aff<F1<BLS381>>& p) { g1_to_bytes<BLS381>(b, p); }
template <>
__device__ __forceinline__ void aff_to_bytes<F2<BN254>>(uint8_t* b, const Aff<F2<BN254>>& p) { g2_to_bytes<BN254>(b, p); }
template <>
__device__ __forceinline__ void aff_to_bytes<F2<BLS381>>(uint8_t* b, const Aff<F2<BLS381>>& p) { g2_to_bytes<BLS381>(b, p); }

template <class F, int PT_BYTES>
__global__ void __launch_bounds__(64) k_sum_first(const uint8_t* pts, size_t n, int R, Jac<F>* out, uint32_t* flags) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t lo = t * (size_t)R;
  if (lo >= n) return;
  size_t hi = lo + R < n ? lo + R : n;
  Jac<F> acc = jac_inf<F>();
  for (size_t k = lo; k < hi; ++k) {
    Aff<F> p;
    bool ok = aff_from_bytes<F>(p, pts + k * PT_BYTES);
    ok = ok && aff_on_curve<F>(p);
    if (!ok) atomicOr(flags, FLAG_ENC);
    acc = jac_add_aff<F>(acc, p);
  }
  out[t] = acc;
}

template <class F>
__global__ void __launch_bounds__(64) k_sum_next(const Jac<F>* in, size_t n, int R, Jac<F>* out) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t lo = t * (size_t)R;
  if (lo >= n) return;
  size_t hi = lo + R < n ? lo + R : n;
  Jac<F> acc = in[lo];
  for (size_t k = lo + 1; k < hi; ++k) acc = jac_add<F>(acc, in[k]);
  out[t] = acc;
}

template <class F>
__global__ void k_jac_to_bytes(const Jac<F>* in, size_t n, uint8_t* out, int pt_bytes) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  aff_to_bytes<F>(out + i * pt_bytes, jac_to_aff<F>(in[i]));
}

// ---- ScalePoints ----
template <class F, int PT_BYTES>
__global__ void __launch_bounds__(64) k_scale(const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n,
                                              uint8_t* out, uint32_t* flags, int sbytes = 32) {   // sbytes: 32 or 16, big-endian
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p;
  bool ok = aff_from_bytes<F>(p, pts + i * PT_BYTES);
  ok = ok && aff_on_curve<F>(p);
  if (!ok) atomicOr(flags, FLAG_ENC);
  uint8_t sg = signs ? signs[i] : 0;
  if (sg == 2) {  // nil factor: Copy()
    aff_to_bytes<F>(out + i * PT_BYTES, p);
    return;
  }
  u32 k[8];
  const uint8_t* s = scalars + i * (size_t)sbytes;
  const int nw = sbytes / 4;
  int top = -1;
  for (int j = 0; j < 8; ++j) {
    k[j] = 0;
    if (j < nw) {
      const uint8_t* q = s + 4 * (nw - 1 - j);
      k[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
  }
  for (int j = 7; j >= 0 && top < 0; --j)
    if (k[j]) top = j * 32 + (31 - __clz(k[j]));
  if (sg == 1) p = aff_neg<F>(p);
  Jac<F> r = jac_mul<F>(p, k, top + 1);
  aff_to_bytes<F>(out + i * PT_BYTES, jac_to_aff<F>(r));
}

// ---- compressed wire formats of alt-bn128 (wire.hpp; curves/altbn128.go:81-89,203-221,296-376) ----
// GROUP 1: 64-byte points <-> 32-byte forms; GROUP 2: 128 <-> 64.  One point per thread: the decoders are one
// (G1) or two (G2) square-root exponentiations plus a Legendre symbol and an inversion, i.e. about the cost of hashing
// one message; ok[i] = 1 / 0 mirrors the reference's (Point, bool).
template <int GROUP>
__global__ void __launch_bounds__(64) k_decompress_bn(const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
  typedef BN254 C;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int CB = GROUP == BGLS_G1 ? 32 : 64, UB = 2 * CB;
  bool good;
  if constexpr (GROUP == BGLS_G1) {
    Aff<F1<C>> p;
    good = g1_decompress<C>(p, in + i * CB);
    if (good) g1_to_bytes<C>(out + i * UB, p);
  } else {
    Aff<F2<C>> p;
    good = g2_decompress<C>(p, in + i * CB);
    if (good) g2_to_bytes<C>(out + i * UB, p);
  }
  if (!good)
    for (int k = 0; k < UB; ++k) out[i * UB + k] = 0;
  ok[i] = good ? 1 : 0;
}

template <int GROUP>
__global__ void __launch_bounds__(64) k_compress_bn(const uint8_t* in, size_t n, uint8_t* out, uint32_t* flags) {
  typedef BN254 C;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int CB = GROUP == BGLS_G1 ? 32 : 64, UB = 2 * CB;
  if constexpr (GROUP == BGLS_G1) {
    Aff<F1<C>> p;
    bool good = g1_from_bytes<C>(p, in + i * UB) && aff_on_curve<F1<C>>(p);
    if (!good) atomicOr(flags, FLAG_ENC);
    g1_compress<C>(out + i * CB, p);
  } else {
    Aff<F2<C>> p;
    bool good = g2_from_bytes<C>(p, in + i * UB) && aff_on_curve<F2<C>>(p);
    if (!good) atomicOr(flags, FLAG_ENC);
    g2_compress<C>(out + i * CB, p);
  }
}

// ---- hashed aggregation exponents (bgls/blsHAE.go) and weighted key sums ----
// BLAKE2Xb expansion: node i of the XOF is one compression of the 64-byte root with its own parameter block
// (hashes.hpp blake2xb_node); one lane per node.  The root itself is a sequential chain over all key bytes and is
// produced on the host side of the boundary (host_blake2xb_root below).
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

// First pass of sum_i k_i P_i: getAggregatePubKey (blsHAE.go:74-77) = AggregatePoints(ScalePoints(keys, t)) without
// materialising the scaled points: thread t accumulates its R products in Jacobian form.  Weights are 16-byte
// big-endian magnitudes with optional sign bytes (1 = negate the point first, curves/curve.go:190-214).
template <class F, int PT_BYTES>
__global__ void __launch_bounds__(64) k_wsum_first(const uint8_t* pts, const uint8_t* w16, const uint8_t* signs, size_t n, int R,
                                                   Jac<F>* out, uint32_t* flags) {
  size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t lo = t * (size_t)R;
  if (lo >= n) return;
  size_t hi = lo + R < n ? lo + R : n;
  Jac<F> acc = jac_inf<F>();
  for (size_t i = lo; i < hi; ++i) {
    Aff<F> p;
    bool ok = aff_from_bytes<F>(p, pts + i * PT_BYTES);
    ok = ok && aff_on_curve<F>(p);
    if (!ok) atomicOr(flags, FLAG_ENC);
    u32 k[4];
    int top = -1;
    for (int j = 0; j < 4; ++j) {
      const uint8_t* q = w16 + i * 16 + 4 * (3 - j);
      k[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    }
    for (int j = 3; j >= 0 && top < 0; --j)
      if (k[j]) top = j * 32 + (31 - __clz(k[j]));
    if (signs && signs[i] == 1) p = aff_neg<F>(p);
    acc = jac_add<F>(acc, jac_mul<F>(p, k, top + 1));
  }
  out[t] = acc;
}

// ---- batch key generation / signing ----
// out[i] = k_i * P_i with P_i taken from a device array of affine points (the hash-to-G1 output: Sign, bgls/bgls.go:46-56)
// or, when pts == nullptr, the group generator (LoadPublicKey, bgls/bgls.go:40-43: GetG2().Mul(sk)).  Scalars are 32-byte
// big-endian, as everywhere at the seam.
template <class C, class F, int PT_BYTES>
__global__ void __launch_bounds__(64) k_scale_aff(const Aff<F>* pts, const uint8_t* scalars, size_t n, uint8_t* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p;
  if (pts) {
    p = pts[i];
  } else {
    if constexpr (PT_BYTES == 2 * C::FP_BYTES) p = Aff<F>{fp_load<C>(C::G1X), fp_load<C>(C::G1Y), false};
    else p = Aff<F>{f2_load<C>(C::G2), f2_load<C>(C::G2 + 2 * C::L), false};
  }
  u32 k[8];
  int top = -1;
  for (int j = 0; j < 8; ++j) {
    const uint8_t* q = scalars + i * 32 + 4 * (7 - j);
    k[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
  for (int j = 7; j >= 0 && top < 0; --j)
    if (k[j]) top = j * 32 + (31 - __clz(k[j]));
  aff_to_bytes<F>(out + i * PT_BYTES, jac_to_aff<F>(jac_mul<F>(p, k, top + 1)));
}

template <class F, int PT_BYTES>
__global__ void k_check(const uint8_t* pts, size_t n, uint32_t* flags) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p;
  bool ok = aff_from_bytes<F>(p, pts + i * PT_BYTES);
  ok = ok && aff_on_curve<F>(p);
  if (!ok) atomicOr(flags, FLAG_ENC);
}

template <class C>
__global__ void k_generator(int group, uint8_t* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (group == BGLS_G1) {
    Aff<F1<C>> g = {fp_load<C>(C::G1X), fp_load<C>(C::G1Y), false};
    g1_to_bytes<C>(out, g);
  } else {
    Aff<F2<C>> g = {f2_load<C>(C::G2), f2_load<C>(C::G2 + 2 * C::L), false};
    g2_to_bytes<C>(out, g);
  }
}

// ---- peak probe: dependent-free v_mad_u64_u32 chains (roofline denominator) ----
__global__ void __launch_bounds__(256) k_mad_probe(uint32_t seed, int iters, uint64_t* sink) {
  uint32_t a = seed ^ (threadIdx.x * 2654435761u), b = seed + blockIdx.x * 40503u + 1u;
  uint64_t acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = (uint64_t)j * 0x9e3779b97f4a7c15ull + a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (uint64_t)(uint32_t)(a + j) * (uint32_t)(b + it) + acc[j];
  }
  uint64_t x = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) x ^= acc[j];
  if (x == 0x1234567ull) sink[0] = x;
}

// ======================================================================= cooperative (v2) kernels
// One wave per block, 10 groups of 6 lanes; see coop.hpp.  Partial products are kept in the
// "w-basis" layout: 6 consecutive Fp2 per Fp12, coefficient j of w^j.
template <class C>
struct CoopLane {
  int lane, g, j, gb;
  bool live;
  __device__ __forceinline__ CoopLane() {
    lane = threadIdx.x;
    live = lane < 60;
    g = live ? lane / 6 : 9;
    j = live ? lane % 6 : lane - 60;
    gb = g * Coop<C>::GROUP_DW;
  }
};

template <class C>
__global__ void __launch_bounds__(64) k_miller_coop(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long gen_at, int rounds,
                                                    size_t groups_total, Fp2<C>* out, uint32_t* flags) {
  typedef Coop<C> K;
  CoopLane<C> ln;
  const int j = ln.j, gb = ln.gb;
  const bool live = ln.live;
  const size_t G = (size_t)blockIdx.x * K::GROUPS + ln.g;
  Fp2<C> ft = j == 0 ? f2_one<C>() : f2_zero<C>();      // running product over rounds
  for (int r = 0; r < rounds; ++r) {
    const size_t idx = ((size_t)r * groups_total + G) * 6 + j;
    Aff<F2<C>> Q;
    Aff<F1<C>> P;
    bool valid = idx < n && G < groups_total;
    if (valid) {
      if ((long long)idx == gen_at) {
        Q.x = f2_load<C>(C::G2);
        Q.y = f2_load<C>(C::G2 + 2 * C::L);
        Q.inf = false;
      } else {
        size_t k = (gen_at >= 0 && (long long)idx > gen_at) ? idx - 1 : idx;
        bool ok = g2_from_bytes<C>(Q, g2s + k * 4 * C::FP_BYTES);
        ok = ok && aff_on_curve<F2<C>>(Q);
        if (!ok) atomicOr(flags, FLAG_ENC);
      }
      P = g1s[idx];
      valid = !P.inf && !Q.inf;
    }
    if (!valid) {  // keep the arithmetic well defined; the lines are replaced by 1
      Q.x = f2_load<C>(C::G2);
      Q.y = f2_load<C>(C::G2 + 2 * C::L);
      P.x = fp_load<C>(C::G1X);
      P.y = fp_load<C>(C::G1Y);
    }
    G2Proj<C> T = {Q.x, Q.y, f2_one<C>()};
    const Fp2<C> nyq = f2_neg<C>(Q.y);
    Fp2<C> fj = j == 0 ? f2_one<C>() : f2_zero<C>();
    coop_publish<C>(gb + K::RB, j, fj, live);
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      LineCoeffs<C> l = dbl_step<C>(T);
      coop_write_line<C>(gb, j, l, P.x, P.y, valid, live);
      fj = coop_sqr<C>(gb, j);
      coop_publish<C>(gb + K::RB, j, fj, live);
      fj = coop_apply_lines<C>(gb, j, live);
      const int d = C::LOOP_NAF[i];
      if (d != 0) {
        l = add_step<C>(T, Q.x, d > 0 ? Q.y : nyq);
        coop_write_line<C>(gb, j, l, P.x, P.y, valid, live);
        fj = coop_apply_lines<C>(gb, j, live);
      }
    }
    if constexpr (C::CURVE_ID == 0) {
      Fp2<C> x1 = f2_mul<C>(f2_conj<C>(Q.x), gamma_const<C>(1, 2));
      Fp2<C> y1 = f2_mul<C>(f2_conj<C>(Q.y), gamma_const<C>(1, 3));
      Fp2<C> x2 = f2_mul<C>(Q.x, gamma_const<C>(2, 2));
      Fp2<C> y2 = f2_neg<C>(f2_mul<C>(Q.y, gamma_const<C>(2, 3)));
      LineCoeffs<C> l = add_step<C>(T, x1, y1);
      coop_write_line<C>(gb, j, l, P.x, P.y, valid, live);
      fj = coop_apply_lines<C>(gb, j, live);
      l = add_step<C>(T, x2, y2);
      coop_write_line<C>(gb, j, l, P.x, P.y, valid, live);
      fj = coop_apply_lines<C>(gb, j, live);
    } else {
      if (j & 1) fj = f2_neg<C>(fj);                      // x < 0: f^(p^6), w -> -w
      coop_publish<C>(gb + K::RB, j, fj, live);
    }
    ft = coop_mul<C>(gb, j, ft, live);                    // ft <- ft * f
  }
  if (live && G < groups_total) out[G * 6 + j] = ft;
}

// Producer/consumer form of the cooperative Miller loop: a block is TWO waves working on the same
// 60 pairings.  Wave 0 runs the per-lane G2 point steps and publishes the lines of step s into
// line buffer s&1; wave 1 folds them into the shared accumulators while wave 0 already computes
// step s+1.  One workgroup barrier per step.  Doubles the number of waves for a given batch, which
// is what a 2^16-signer batch needs to keep more than one wave per SIMD busy.
template <class C, bool STEP_INL>
__global__ void __launch_bounds__(128, 3) k_miller_ab(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long gen_at, int rounds,
                                                   size_t groups_total, Fp2<C>* out, uint32_t* flags) {
  typedef Coop<C> K;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const bool live = lane < 60;
  const int g = live ? lane / 6 : 9;
  const int j = live ? lane % 6 : lane - 60;
  const int gb = g * K::GROUP_DW_AB;
  const size_t G = (size_t)blockIdx.x * K::GROUPS + g;
  if (wave == 0) {
    // ---------------- producer: point steps + line evaluation
    for (int r = 0; r < rounds; ++r) {
      const size_t idx = ((size_t)r * groups_total + G) * 6 + j;
      Aff<F2<C>> Q;
      Aff<F1<C>> P;
      bool valid = idx < n && G < groups_total;
      if (valid) {
        if ((long long)idx == gen_at) {
          Q.x = f2_load<C>(C::G2);
          Q.y = f2_load<C>(C::G2 + 2 * C::L);
          Q.inf = false;
        } else {
          size_t k = (gen_at >= 0 && (long long)idx > gen_at) ? idx - 1 : idx;
          bool ok = g2_from_bytes<C>(Q, g2s + k * 4 * C::FP_BYTES);
          ok = ok && aff_on_curve<F2<C>>(Q);
          if (!ok) atomicOr(flags, FLAG_ENC);
        }
        P = g1s[idx];
        valid = !P.inf && !Q.inf;
      }
      if (!valid) {
        Q.x = f2_load<C>(C::G2);
        Q.y = f2_load<C>(C::G2 + 2 * C::L);
        P.x = fp_load<C>(C::G1X);
        P.y = fp_load<C>(C::G1Y);
      }
      G2Proj<C> T = {Q.x, Q.y, f2_one<C>()};
      const Fp2<C> nyq = f2_neg<C>(Q.y);
      int buf = 0;
#pragma unroll 1
      for (int i = 1; i < C::LOOP_LEN; ++i) {
        if constexpr (STEP_INL) {
          dbl_step_emit<C>(T, LineEmitter<C>{reg_rl<C>(gb, buf ? K::RL2 : K::RL), j, P.x, P.y, valid, live});
          wave_sync();
        } else {
          LineCoeffs<C> l = dbl_step_t<C, false>(T);
          coop_write_line<C>(gb, j, l, P.x, P.y, valid, live, buf ? K::RL2 : K::RL);
        }
        __syncthreads();
        buf ^= 1;
        const int d = C::LOOP_NAF[i];
        if (d != 0) {
          if constexpr (STEP_INL) {
            add_step_emit<C>(T, Q.x, d > 0 ? Q.y : f2_neg<C>(Q.y), LineEmitter<C>{reg_rl<C>(gb, buf ? K::RL2 : K::RL), j, P.x, P.y, valid, live});
            wave_sync();
          } else {
            LineCoeffs<C> l = add_step_t<C, false>(T, Q.x, d > 0 ? Q.y : nyq);
            coop_write_line<C>(gb, j, l, P.x, P.y, valid, live, buf ? K::RL2 : K::RL);
          }
          __syncthreads();
          buf ^= 1;
        }
      }
      if constexpr (C::CURVE_ID == 0) {
        Fp2<C> x1 = f2_mul<C>(f2_conj<C>(Q.x), gamma_const<C>(1, 2));
        Fp2<C> y1 = f2_mul<C>(f2_conj<C>(Q.y), gamma_const<C>(1, 3));
        Fp2<C> x2 = f2_mul<C>(Q.x, gamma_const<C>(2, 2));
        Fp2<C> y2 = f2_neg<C>(f2_mul<C>(Q.y, gamma_const<C>(2, 3)));
        LineCoeffs<C> l = add_step_t<C, STEP_INL>(T, x1, y1);
        coop_write_line<C>(gb, j, l, P.x, P.y, valid, live, buf ? K::RL2 : K::RL);
        __syncthreads();
        buf ^= 1;
        l = add_step_t<C, STEP_INL>(T, x2, y2);
        coop_write_line<C>(gb, j, l, P.x, P.y, valid, live, buf ? K::RL2 : K::RL);
        __syncthreads();
        buf ^= 1;
      }
      __syncthreads();   // round boundary: the consumer has finished with both line buffers
    }
  } else {
    // ---------------- consumer: fold the lines into the shared accumulators
    Fp2<C> ft = j == 0 ? f2_one<C>() : f2_zero<C>();
    for (int r = 0; r < rounds; ++r) {
      Fp2<C> fj = j == 0 ? f2_one<C>() : f2_zero<C>();
      coop_publish<C>(gb + K::RB, j, fj, live);
      int buf = 0;
#pragma unroll 1
      for (int i = 1; i < C::LOOP_LEN; ++i) {
        __syncthreads();
        fj = coop_sqr<C, true>(gb, j);
        coop_publish<C>(gb + K::RB, j, fj, live);
        fj = coop_apply_lines<C, true>(gb, j, live, buf ? K::RL2 : K::RL);
        buf ^= 1;
        if (C::LOOP_NAF[i] != 0) {
          __syncthreads();
          fj = coop_apply_lines<C, true>(gb, j, live, buf ? K::RL2 : K::RL);
          buf ^= 1;
        }
      }
      if constexpr (C::CURVE_ID == 0) {
        __syncthreads();
        fj = coop_apply_lines<C, true>(gb, j, live, buf ? K::RL2 : K::RL);
        buf ^= 1;
        __syncthreads();
        fj = coop_apply_lines<C, true>(gb, j, live, buf ? K::RL2 : K::RL);
        buf ^= 1;
      } else {
        if (j & 1) fj = f2_neg<C>(fj);
        coop_publish<C>(gb + K::RB, j, fj, live);
      }
      ft = coop_mul<C, true>(gb, j, ft, live, K::RL);
      __syncthreads();   // round boundary
    }
    if (live && G < groups_total) out[G * 6 + j] = ft;
  }
}

// ---- fixed-argument lines of the generator --------------------------------------------------------------
// The (-sigma, g2) pair of every verification has Q = GetG2(), a curve constant (curves/altbn128.go:427-429,
// curves/bls12_381.go:279-281): its point steps do not depend on the input, only the scaling of the line
// by P = -sigma does.  k_gen_lines walks the Miller loop of g2 once per context and stores the unscaled
// coefficients of every step.
template <class C>
__global__ void k_gen_lines(LineCoeffs<C>* table, int* nsteps) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Fp2<C> qx = f2_load<C>(C::G2), qy = f2_load<C>(C::G2 + 2 * C::L);
  G2Proj<C> T = {qx, qy, f2_one<C>()};
  const Fp2<C> nyq = f2_neg<C>(qy);
  int s = 0;
  for (int i = 1; i < C::LOOP_LEN; ++i) {
    table[s++] = dbl_step<C>(T);
    const int d = C::LOOP_NAF[i];
    if (d != 0) table[s++] = add_step<C>(T, qx, d > 0 ? qy : nyq);
  }
  if constexpr (C::CURVE_ID == 0) {
    Fp2<C> x1 = f2_mul<C>(f2_conj<C>(qx), gamma_const<C>(1, 2));
    Fp2<C> y1 = f2_mul<C>(f2_conj<C>(qy), gamma_const<C>(1, 3));
    Fp2<C> x2 = f2_mul<C>(qx, gamma_const<C>(2, 2));
    Fp2<C> y2 = f2_neg<C>(f2_mul<C>(qy, gamma_const<C>(2, 3)));
    table[s++] = add_step<C>(T, x1, y1);
    table[s++] = add_step<C>(T, x2, y2);
  }
  *nsteps = s;
}

// ---- 64 pairings per block, 256 VGPRs (two waves per SIMD, 1024 blocks = exactly one 2^16 batch) ----------
// Same producer/consumer scheme as k_miller_ab, but the producer wave uses all 64 lanes (lane l feeds line
// slot l/10 of group l%10, so groups 0..3 fold seven lines and the others six plus a constant 1), the
// (-sigma, g2) pair needs no point steps (k_gen_lines table, scaled by lane 0 of block 0), and nothing
// spills: the point-step temporaries fit the 256-register budget.
template <class C, bool R28 = false>
struct Coop64 {
  static constexpr int S2 = R28 ? R28_S2 : 2 * C::L;        // R28: ten 28-bit limbs per field element (coop_r28.hpp), 38.4 KB per block
  static constexpr int NENT = 18;            // per group and buffer: 3 line pairs x 5 coefficients + 1 single line x 3
  // BLS12-381: xi = 1+i costs two additions, so the accumulator region keeps the plain coefficients only and
  // the wrap-around factor is applied after the load; that is what lets four blocks share a CU's 160 KB.
  static constexpr bool XF = C::XI_RE == 1;
  static constexpr int RBN = XF ? 6 : 12;
  static constexpr int RB = 0, RL = RBN * S2, RL2 = (RBN + NENT) * S2;
  static constexpr int GROUP_DW = (RBN + 2 * NENT) * S2;
  static constexpr int BLOCK_BYTES = 10 * GROUP_DW * 4;
};

// Lanes 0..59 of the producer wave are 30 neighbour pairs (lanes 6g+2m, 6g+2m+1 -> pair m of group g): each pair
// multiplies its two lines into one 5-coefficient element (coop_write_line_pair).  Lanes 60..63 feed the
// single-line slot of groups 0..3, the (-sigma, g2) line goes to the single slot of group 4, groups 5..9 keep
// the constant 1 there.  The consumer folds 3 five-term elements + 1 three-term line per step.
// DBG (development only, BGLS_AB64_DBG): 1 = producer work only, 2 = consumer work only -- wrong results, used to
// time the two halves of the pipeline separately.
// R28 (alt-bn128): the consumer works on 28-bit limbs (coop_r28.hpp); the producer converts what it stores.
template <class C, int DBG = 0, bool R28 = false>
__global__ void __launch_bounds__(128, 2) k_miller_ab64(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at,
                                                        const LineCoeffs<C>* gen_lines, Fp2<C>* out, uint32_t* flags, unsigned swap_mask) {
  typedef Coop64<C, R28> K;
  // The two waves of a block land on different SIMDs and every CU hosts four blocks: if wave 0 were the producer
  // everywhere, two SIMDs of a CU would carry two producers and the other two would carry two consumers, and the kernel
  // would run at the pace of the heavier role.  Blocks selected by swap_mask exchange the roles, so each SIMD carries
  // one producer and one consumer.
  const int wave = (int)(threadIdx.x >> 6) ^ ((blockIdx.x & swap_mask) ? 1 : 0);
  const int lane = threadIdx.x & 63;
  if (wave == 0) {
    // ---------------- producer: 64 pairings, one per lane
    const size_t idx = (size_t)blockIdx.x * 64 + lane;
    const bool paired = lane < 60;
    const int tg = paired ? lane / 6 : lane - 60;
    const int j = paired ? lane % 6 : 0;
    const int tgb = tg * K::GROUP_DW;
    Aff<F2<C>> Q;
    Aff<F1<C>> P;
    bool valid = idx < n;
    if (valid) {
      bool ok = g2_from_bytes<C>(Q, g2s + idx * 4 * C::FP_BYTES);
      ok = ok && aff_on_curve<F2<C>>(Q);
      if (!ok) atomicOr(flags, FLAG_ENC);
      P = g1s[idx];
      valid = !P.inf && !Q.inf;
    }
    if (!valid) {
      Q.x = f2_load<C>(C::G2);
      Q.y = f2_load<C>(C::G2 + 2 * C::L);
      P.x = fp_load<C>(C::G1X);
      P.y = fp_load<C>(C::G1Y);
    }
    const bool sig_lane = sig_at >= 0 && blockIdx.x == 0 && lane == 0;
    Aff<F1<C>> S;
    bool sig_valid = false;
    if (sig_lane) {
      S = g1s[sig_at];
      sig_valid = !S.inf;
    }
    // single-line slots without an owner hold the constant 1 in both buffers
    if (lane >= 4 && lane < 10) {
      for (int b = 0; b < 2; ++b) {
        const LReg r = {lane * K::GROUP_DW + (b ? K::RL2 : K::RL), K::NENT};
        st_entry<C, R28>(r, 15, f2_one<C>());
        st_entry<C, R28>(r, 16, f2_zero<C>());
        st_entry<C, R28>(r, 17, f2_zero<C>());
      }
    }
    G2Proj<C> T = {Q.x, Q.y, f2_one<C>()};
    int buf = 0, step = 0;
    auto publish = [&](LineCapture<C>& cap) {
      if constexpr (DBG == 2) { ++step; __syncthreads(); buf ^= 1; return; }
      if (!valid) { cap.e[0] = f2_one<C>(); cap.e[1] = f2_zero<C>(); cap.e[2] = f2_zero<C>(); }
      const LReg r = {tgb + (buf ? K::RL2 : K::RL), K::NENT};
      if (paired) {
        coop_write_line_pair<C, R28>(r, j, cap.e);
      } else {
        st_entry<C, R28>(r, 15, cap.e[0]);
        st_entry<C, R28>(r, 16, cap.e[1]);
        st_entry<C, R28>(r, 17, cap.e[2]);
      }
      if (sig_lane && sig_valid) {
        const LineCoeffs<C> l = gen_lines[step];
        LineEmitter<C, R28> em{LReg{4 * K::GROUP_DW + (buf ? K::RL2 : K::RL), K::NENT}, 5, S.x, S.y, true, true};   // entries 15..17
        em(0, l.c0);
        em(1, l.c1);
        em(2, l.c2);
      }
      ++step;
      wave_sync();
      if constexpr (DBG != 3) __syncthreads();
      buf ^= 1;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      {
        LineCapture<C> cap{{}, P.x, P.y};
        if constexpr (DBG != 2) dbl_step_emit<C>(T, cap);
        publish(cap);
      }
      const int d = C::LOOP_NAF[i];
      if (d != 0) {
        LineCapture<C> cap{{}, P.x, P.y};
        if constexpr (DBG != 2) add_step_emit<C>(T, Q.x, d > 0 ? Q.y : f2_neg<C>(Q.y), cap);
        publish(cap);
      }
    }
    if constexpr (C::CURVE_ID == 0) {
      {
        Fp2<C> x1 = f2_mul<C>(f2_conj<C>(Q.x), gamma_const<C>(1, 2));
        Fp2<C> y1 = f2_mul<C>(f2_conj<C>(Q.y), gamma_const<C>(1, 3));
        LineCapture<C> cap{{}, P.x, P.y};
        if constexpr (DBG != 2) add_step_emit<C>(T, x1, y1, cap);
        publish(cap);
      }
      {
        Fp2<C> x2 = f2_mul<C>(Q.x, gamma_const<C>(2, 2));
        Fp2<C> y2 = f2_neg<C>(f2_mul<C>(Q.y, gamma_const<C>(2, 3)));
        LineCapture<C> cap{{}, P.x, P.y};
        if constexpr (DBG != 2) add_step_emit<C>(T, x2, y2, cap);
        publish(cap);
      }
    }
  } else {
    // ---------------- consumer: 10 groups x 6 lanes; per step 3 line pairs + 1 single line
    const bool live = lane < 60;
    const int g = live ? lane / 6 : 9;
    const int j = live ? lane % 6 : lane - 60;
    const int gb = g * K::GROUP_DW;
    if constexpr (R28) {
      // ---- 28-bit-limb consumer: same schedule, every dot product a pile of carry-free column accumulations
      static_assert(C::CURVE_ID == 0 && C::TWIST_D, "alt-bn128 only");
      const int rbo = gb + K::RB;
      F28x2 fj;
      {
        const F28 one = r28_load<C>(C::R28_ONE);
#pragma unroll
        for (int q = 0; q < 10; ++q) { fj.c0.v[q] = j == 0 ? one.v[q] : 0u; fj.c1.v[q] = 0u; }
      }
      coop_publish28<C>(rbo, j, fj, live);
      int buf = 0;
      auto fold = [&]() {
        if constexpr (DBG == 1) { buf ^= 1; return; }
        const int rlo = gb + (buf ? K::RL2 : K::RL);
#pragma unroll 1
        for (int m = 0; m < 3; ++m) {
          fj = coop_dot28<C, 5>(rlo, 5 * m, rbo, j, COOP_SH_D5);
          coop_publish28<C>(rbo, j, fj, live);
        }
        fj = coop_dot28<C, 3>(rlo, 15, rbo, j, COOP_SH_D);
        coop_publish28<C>(rbo, j, fj, live);
        buf ^= 1;
      };
#pragma unroll 1
      for (int i = 1; i < C::LOOP_LEN; ++i) {
        if constexpr (DBG != 3) __syncthreads();
        if constexpr (DBG != 1) {
          fj = coop_sqr_sym28<C>(rbo, j);
          coop_publish28<C>(rbo, j, fj, live);
        }
        fold();
        if (C::LOOP_NAF[i] != 0) {
          if constexpr (DBG != 3) __syncthreads();
          fold();
        }
      }
      if constexpr (DBG != 3) __syncthreads();
      fold();
      if constexpr (DBG != 3) __syncthreads();
      fold();
      if (live) out[((size_t)blockIdx.x * 10 + g) * 6 + j] = from_r28<C>(fj);
      return;
    }
    const LReg rb = {gb + K::RB, 12};
    Fp2<C> fj = j == 0 ? f2_one<C>() : f2_zero<C>();
    coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
    int buf = 0;
    auto fold = [&]() {
      if constexpr (DBG == 1) { buf ^= 1; return; }
      const LReg rl = {gb + (buf ? K::RL2 : K::RL), K::NENT};
#pragma unroll 1
      for (int m = 0; m < 3; ++m) {
        fj = coop_dot_inl<C, 5, K::XF>(rl, 5 * m, 1, rb, j, C::TWIST_D ? COOP_SH_D5 : COOP_SH_M5);
        coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
      }
      fj = coop_dot_inl<C, 3, K::XF>(rl, 15, 1, rb, j, C::TWIST_D ? COOP_SH_D : COOP_SH_M);
      coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
      buf ^= 1;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      if constexpr (DBG != 3) __syncthreads();
      if constexpr (DBG != 1) {
        fj = coop_sqr_sym_inl<C, K::XF>(rb, j);
        coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
      }
      fold();
      if (C::LOOP_NAF[i] != 0) {
        if constexpr (DBG != 3) __syncthreads();
        fold();
      }
    }
    if constexpr (C::CURVE_ID == 0) {
      if constexpr (DBG != 3) __syncthreads();
      fold();
      if constexpr (DBG != 3) __syncthreads();
      fold();
    } else {
      if (j & 1) fj = f2_neg<C>(fj);                      // x < 0: f^(p^6), w -> -w
    }
    if (live) out[((size_t)blockIdx.x * 10 + g) * 6 + j] = fj;
  }
}

// ---- throughput shape of the alt-bn128 Miller kernel: 60 pairings per block, single lines, 28-bit-limb consumer ------
// With the consumer on 28-bit limbs (coop_r28.hpp) the producer wave is the slower half of k_miller_ab64, and a fifth of
// its step is the product of neighbouring lines.  Here the producer only steps its points and stores its own line
// (converted to 28-bit limbs); the consumer, which has the slack, folds the six three-term lines of its group itself --
// the same number of products as three five-term elements plus a single line, two more reductions.  Six lines per group
// and buffer is exactly the LDS footprint of k_miller_ab64 (38.4 KB), so four blocks still share a CU; lanes 60..63 of the
// producer idle and the signature pair moves to the epilogue kernel (as on BLS12-381).  A 2^16 batch is 1093 blocks: one
// more than fit at once, which is irrelevant while launches overlap and a second, nearly empty round when they do not --
// hence only in throughput mode (bgls_set_throughput_mode).
template <class C, int DBG = 0>    // DBG 1 / 2: producer / consumer work only (timing, wrong results)
__global__ void __launch_bounds__(128, 2) k_miller_s60(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, Fp2<C>* out, uint32_t* flags) {
  static_assert(C::CURVE_ID == 0 && C::TWIST_D, "alt-bn128 only");
  typedef Coop64<C, true> K;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  if (wave == 0) {
    // ---------------- producer: 60 pairings, one per lane; lane 6g + j feeds line j of group g
    const bool owner = lane < 60;
    const size_t idx = (size_t)blockIdx.x * 60 + lane;
    const int tg = owner ? lane / 6 : 0;
    const int j = owner ? lane % 6 : 0;
    const int tgb = tg * K::GROUP_DW;
    Aff<F2<C>> Q;
    Aff<F1<C>> P;
    bool valid = owner && idx < n;
    if (valid) {
      bool ok = g2_from_bytes<C>(Q, g2s + idx * 4 * C::FP_BYTES);
      ok = ok && aff_on_curve<F2<C>>(Q);
      if (!ok) atomicOr(flags, FLAG_ENC);
      P = g1s[idx];
      valid = !P.inf && !Q.inf;
    }
    if (!valid) {
      Q.x = f2_load<C>(C::G2);
      Q.y = f2_load<C>(C::G2 + 2 * C::L);
      P.x = fp_load<C>(C::G1X);
      P.y = fp_load<C>(C::G1Y);
    }
    G2Proj<C> T = {Q.x, Q.y, f2_one<C>()};
    int buf = 0;
    auto done = [&]() {
      wave_sync();
      __syncthreads();
      buf ^= 1;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      if constexpr (DBG != 2) dbl_step_emit<C>(T, LineEmitter<C, true>{LReg{tgb + (buf ? K::RL2 : K::RL), K::NENT}, j, P.x, P.y, valid, owner});
      done();
      const int d = C::LOOP_NAF[i];
      if (d != 0) {
        if constexpr (DBG != 2) add_step_emit<C>(T, Q.x, d > 0 ? Q.y : f2_neg<C>(Q.y), LineEmitter<C, true>{LReg{tgb + (buf ? K::RL2 : K::RL), K::NENT}, j, P.x, P.y, valid, owner});
        done();
      }
    }
    {
      Fp2<C> x1 = f2_mul<C>(f2_conj<C>(Q.x), gamma_const<C>(1, 2));
      Fp2<C> y1 = f2_mul<C>(f2_conj<C>(Q.y), gamma_const<C>(1, 3));
      if constexpr (DBG != 2) add_step_emit<C>(T, x1, y1, LineEmitter<C, true>{LReg{tgb + (buf ? K::RL2 : K::RL), K::NENT}, j, P.x, P.y, valid, owner});
      done();
      Fp2<C> x2 = f2_mul<C>(Q.x, gamma_const<C>(2, 2));
      Fp2<C> y2 = f2_neg<C>(f2_mul<C>(Q.y, gamma_const<C>(2, 3)));
      if constexpr (DBG != 2) add_step_emit<C>(T, x2, y2, LineEmitter<C, true>{LReg{tgb + (buf ? K::RL2 : K::RL), K::NENT}, j, P.x, P.y, valid, owner});
      done();
    }
  } else {
    // ---------------- consumer: 10 groups x 6 lanes, six three-term lines per step
    const bool live = lane < 60;
    const int g = live ? lane / 6 : 9;
    const int j = live ? lane % 6 : lane - 60;
    const int gb = g * K::GROUP_DW;
    const int rbo = gb + K::RB;
    F28x2 fj;
    {
      const F28 one = r28_load<C>(C::R28_ONE);
#pragma unroll
      for (int q = 0; q < 10; ++q) { fj.c0.v[q] = j == 0 ? one.v[q] : 0u; fj.c1.v[q] = 0u; }
    }
    coop_publish28<C>(rbo, j, fj, live);
    int buf = 0;
    auto fold = [&]() {
      if constexpr (DBG == 1) { buf ^= 1; return; }
      const int rlo = gb + (buf ? K::RL2 : K::RL);
#pragma unroll 1
      for (int m = 0; m < 6; ++m) {
        fj = coop_dot28_k3<C>(rlo, 3 * m, rbo, j, COOP_SH_D);
        coop_publish28<C>(rbo, j, fj, live);
      }
      buf ^= 1;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      __syncthreads();
      if constexpr (DBG != 1) {
        fj = coop_sqr_sym28<C>(rbo, j);
        coop_publish28<C>(rbo, j, fj, live);
      }
      fold();
      if (C::LOOP_NAF[i] != 0) {
        __syncthreads();
        fold();
      }
    }
    __syncthreads();
    fold();
    __syncthreads();
    fold();
    if (live) out[((size_t)blockIdx.x * 10 + g) * 6 + j] = from_r28<C>(fj);
  }
}

// out[G] = prod in[G*R .. min(count, (G+1)*R))   (w-basis Fp12 arrays)
template <class C>
__global__ void __launch_bounds__(64) k_reduce_coop(const Fp2<C>* in, size_t count, int R, Fp2<C>* out) {
  typedef Coop<C> K;
  CoopLane<C> ln;
  const int j = ln.j, gb = ln.gb;
  const bool live = ln.live;
  const size_t G = (size_t)blockIdx.x * K::GROUPS + ln.g;
  const size_t lo = G * (size_t)R;
  const bool has = lo < count;
  const size_t hi = lo + R < count ? lo + R : count;
  Fp2<C> acc = has ? in[lo * 6 + j] : (j == 0 ? f2_one<C>() : f2_zero<C>());
#pragma unroll 1
  for (int t = 1; t < R; ++t) {                           // uniform trip count; missing operands are 1
    coop_publish<C>(gb + K::RB, j, acc, live);
    const size_t k = lo + t;
    Fp2<C> x = (has && k < hi) ? in[k * 6 + j] : (j == 0 ? f2_one<C>() : f2_zero<C>());
    acc = coop_mul<C>(gb, j, x, live);
  }
  if (live && has) out[G * 6 + j] = acc;
}

template <class C>
__global__ void k_w_to_bytes(const Fp2<C>* in, uint8_t* out) {
  const int t = threadIdx.x;
  if (blockIdx.x != 0 || t >= 6) return;
  const int order[6] = {5, 3, 1, 4, 2, 0};                 // h.a2 h.a1 h.a0 g.a2 g.a1 g.a0
  Fp2<C> e = in[order[t]];
  fp_to_be<C>(out + (2 * t) * C::FP_BYTES, fp_from_mont<C>(e.c1));
  fp_to_be<C>(out + (2 * t + 1) * C::FP_BYTES, fp_from_mont<C>(e.c0));
}

// ---- cooperative final exponentiation: one group of six lanes ----
template <class C>
__device__ __forceinline__ Fp2<C> coop_frob(const Fp2<C>& e, int j, int k) {     // coefficient of f^(p^k)
  Fp2<C> x = (k & 1) ? f2_conj<C>(e) : e;
  return f2_mul<C>(x, gamma_const<C>(k, j));
}
template <class C>
__device__ __forceinline__ Fp2<C> coop_conj(const Fp2<C>& e, int j) { return (j & 1) ? f2_neg<C>(e) : e; }

// a * b for distributed values held in registers
template <class C>
__device__ __forceinline__ Fp2<C> cmul(const CoopLane<C>& ln, const Fp2<C>& a, const Fp2<C>& b) {
  coop_publish<C>(ln.gb + Coop<C>::RB, ln.j, a, ln.live);
  return coop_mul<C>(ln.gb, ln.j, b, ln.live);
}
template <class C>
__device__ __forceinline__ Fp2<C> csqr(const CoopLane<C>& ln, const Fp2<C>& a) {
  coop_publish<C>(ln.gb + Coop<C>::RB, ln.j, a, ln.live);
  Fp2<C> r = coop_sqr<C>(ln.gb, ln.j);
  wave_sync();
  return r;
}
// a^e, public exponent, top bit set
template <class C>
__device__ __noinline__ Fp2<C> cpow(const CoopLane<C>& ln, const Fp2<C>& a, const u32* e, int nbits) {
  Fp2<C> r = a;
  for (int i = nbits - 2; i >= 0; --i) {
    r = csqr<C>(ln, r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = cmul<C>(ln, r, a);
  }
  return r;
}
// inverse through lane 0 of the group (thread-local tower inversion)
template <class C>
__device__ __noinline__ Fp2<C> cinv(const CoopLane<C>& ln, const Fp2<C>& a) {
  typedef Coop<C> K;
  coop_publish<C>(ln.gb + K::RB, ln.j, a, ln.live);
  if (ln.live && ln.j == 0) {
    Fp2<C> e[6];
    for (int k = 0; k < 6; ++k) e[k] = lds_ld<C>(reg_rb<C>(ln.gb), 2 * k);
    Fp12<C> f = {{e[0], e[2], e[4]}, {e[1], e[3], e[5]}};
    Fp12<C> fi = f12_inv<C>(f);
    const Fp2<C> o[6] = {fi.g.a0, fi.h.a0, fi.g.a1, fi.h.a1, fi.g.a2, fi.h.a2};
    for (int k = 0; k < 6; ++k) lds_st<C>(reg_rl<C>(ln.gb, K::RL), k, o[k]);
  }
  wave_sync();
  Fp2<C> r = lds_ld<C>(reg_rl<C>(ln.gb, K::RL), ln.j);
  wave_sync();
  return r;
}

template <class C>
__device__ __noinline__ Fp2<C> coop_final_exp(const CoopLane<C>& ln, Fp2<C> f) {
  const int j = ln.j;
  // easy part
  Fp2<C> t = cmul<C>(ln, coop_conj<C>(f, j), cinv<C>(ln, f));
  f = cmul<C>(ln, coop_frob<C>(t, j, 2), t);
  if constexpr (C::CURVE_ID == 0) {
    Fp2<C> ft1 = cpow<C>(ln, f, C::U_ABS, C::U_BITS);
    Fp2<C> ft2 = cpow<C>(ln, ft1, C::U_ABS, C::U_BITS);
    Fp2<C> ft3 = cpow<C>(ln, ft2, C::U_ABS, C::U_BITS);
    Fp2<C> y0 = cmul<C>(ln, cmul<C>(ln, coop_frob<C>(f, j, 1), coop_frob<C>(f, j, 2)), coop_frob<C>(f, j, 3));
    Fp2<C> y1 = coop_conj<C>(f, j);
    Fp2<C> y2 = coop_frob<C>(ft2, j, 2);
    Fp2<C> y3 = coop_conj<C>(coop_frob<C>(ft1, j, 1), j);
    Fp2<C> y4 = coop_conj<C>(cmul<C>(ln, ft1, coop_frob<C>(ft2, j, 1)), j);
    Fp2<C> y5 = coop_conj<C>(ft2, j);
    Fp2<C> y6 = coop_conj<C>(cmul<C>(ln, ft3, coop_frob<C>(ft3, j, 1)), j);
    Fp2<C> t0 = cmul<C>(ln, cmul<C>(ln, csqr<C>(ln, y6), y4), y5);
    Fp2<C> t1 = cmul<C>(ln, cmul<C>(ln, y3, y5), t0);
    t0 = cmul<C>(ln, t0, y2);
    t1 = csqr<C>(ln, cmul<C>(ln, csqr<C>(ln, t1), t0));
    t0 = cmul<C>(ln, t1, y1);
    t1 = cmul<C>(ln, t1, y0);
    t0 = csqr<C>(ln, t0);
    return cmul<C>(ln, t1, t0);
  } else {
    Fp2<C> a = cpow<C>(ln, f, C::COFACTOR, C::COFACTOR_BITS);
    Fp2<C> ax = coop_conj<C>(cpow<C>(ln, a, C::U_ABS, C::U_BITS), j);
    Fp2<C> b = cmul<C>(ln, ax, coop_frob<C>(a, j, 1));
    Fp2<C> bx = coop_conj<C>(cpow<C>(ln, b, C::U_ABS, C::U_BITS), j);
    Fp2<C> bxx = coop_conj<C>(cpow<C>(ln, bx, C::U_ABS, C::U_BITS), j);
    Fp2<C> d = cmul<C>(ln, cmul<C>(ln, bxx, coop_frob<C>(b, j, 2)), coop_conj<C>(b, j));
    return cmul<C>(ln, d, f);
  }
}

// product of `count` serialised partials -> final exponentiation -> GT bytes + verdict (one wave, group 0 meaningful)
template <class C>
__global__ void __launch_bounds__(64) k_final_coop(const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out,
                                                   uint32_t* verdict, uint32_t* flags) {
  CoopLane<C> ln;
  const int j = ln.j;
  const int order_pos[6] = {5, 2, 4, 1, 3, 0};             // byte slot of w-coefficient j (inverse of k_w_to_bytes order)
  Fp2<C> acc = j == 0 ? f2_one<C>() : f2_zero<C>();
  for (size_t k = 0; k < count; ++k) {
    const uint8_t* b = partials + k * 12 * C::FP_BYTES + (size_t)(2 * order_pos[j]) * C::FP_BYTES;
    Fp<C> im = fp_from_be<C>(b), re = fp_from_be<C>(b + C::FP_BYTES);
    if (fp_geq_p<C>(im) || fp_geq_p<C>(re)) atomicOr(flags, FLAG_ENC);
    Fp2<C> x = {fp_to_mont<C>(re), fp_to_mont<C>(im)};
    acc = (k == 0) ? x : cmul<C>(ln, acc, x);
  }
  if (do_final_exp) acc = coop_final_exp<C>(ln, acc);
  const bool is_one = j == 0 ? f2_eq<C>(acc, f2_one<C>()) : f2_is_zero<C>(acc);
  const unsigned long long ball = __ballot(is_one);
  if (ln.lane < 6) {
    if (gt_out) {
      uint8_t* o = gt_out + (size_t)(2 * order_pos[j]) * C::FP_BYTES;
      fp_to_be<C>(o, fp_from_mont<C>(acc.c1));
      fp_to_be<C>(o + C::FP_BYTES, fp_from_mont<C>(acc.c0));
    }
    if (ln.lane == 0) verdict[0] = ((ball & 0x3Full) == 0x3Full) ? 1u : 0u;
  }
}

// product of `count` serialised partials -> final exponentiation on 36 lanes (finalexp.hpp)
template <class C>
__global__ void __launch_bounds__(64) k_final36(const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out,
                                                uint32_t* verdict, uint32_t* flags) {
  typedef FE<C> E;
  const int lane = threadIdx.x;
  const int order_pos[6] = {5, 2, 4, 1, 3, 0};
  for (size_t k = 0; k < count; ++k) {
    if (lane < 6) {
      const uint8_t* b = partials + k * 12 * C::FP_BYTES + (size_t)(2 * order_pos[lane]) * C::FP_BYTES;
      Fp<C> im = fp_from_be<C>(b), re = fp_from_be<C>(b + C::FP_BYTES);
      if (fp_geq_p<C>(im) || fp_geq_p<C>(re)) atomicOr(flags, FLAG_ENC);
      fe_put<C>(k == 0 ? FE_F : FE_X, lane, Fp2<C>{fp_to_mont<C>(re), fp_to_mont<C>(im)});
    }
    wave_sync();
    if (k > 0) fe_mul<C>(FE_F, FE_F, FE_X);
  }
  if (do_final_exp) fe_final_exp<C>();
  bool is_one = true;
  if (lane < 6) {
    Fp2<C> v = lds_load_f2<C>(E::coef(FE_F, lane, 0));
    is_one = lane == 0 ? f2_eq<C>(v, f2_one<C>()) : f2_is_zero<C>(v);
    if (gt_out) {
      uint8_t* o = gt_out + (size_t)(2 * order_pos[lane]) * C::FP_BYTES;
      fp_to_be<C>(o, fp_from_mont<C>(v.c1));
      fp_to_be<C>(o + C::FP_BYTES, fp_from_mont<C>(v.c0));
    }
  }
  const unsigned long long ball = __ballot(is_one);
  if (lane == 0) verdict[0] = (ball == ~0ull) ? 1u : 0u;
}

// BLS12-381 epilogue of the cofactor-in-GT verification path: out = miller(-sigma, g2) * rest^h, h = the G1
// cofactor, where rest = prod_i miller(S_i, pk_i) over the UNCLEARED hash points (k_bls_combine<true>).  Both
// factors are serial chains with no batch dimension, so they run side by side on the two waves of one block
// with the 36-lane arithmetic of finalexp.hpp: wave 0 raises rest to h (general squarings -- rest is not
// unitary before the final exponentiation), wave 1 walks the fixed-argument lines of g2 (k_gen_lines table)
// for the signature pair.  The result is an ordinary partial product: the final exponentiation maps it to the
// same GT element as the reference's prod e(H(m_i), pk_i) * e(-sigma, g2).
template <class C>
__global__ void __launch_bounds__(128) k_cofactor_epilogue(const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines,
                                                           uint8_t* out) {
  typedef FE<C> E;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  enum { S_BASE = 0, S_ACC = 1, S_SIG = 8, S_LINE = 9 };
  if (wave == 0) {
    if (lane < 6) {
      const Fp2<C> v = rest[lane];
      fe_put<C>(S_BASE, lane, v);
      fe_put<C>(S_ACC, lane, v);
    }
    wave_sync();
    if constexpr (C::CURVE_ID == 1) {                       // alt-bn128 has cofactor 1: the epilogue only folds the signature pair in
      for (int i = C::COFACTOR_BITS - 2; i >= 0; --i) {
        fe_mul<C>(S_ACC, S_ACC, S_ACC);
        if ((C::COFACTOR[i >> 5] >> (i & 31)) & 1u) fe_mul<C>(S_ACC, S_ACC, S_BASE);
      }
    }
  } else {
    if (lane < 6) {
      fe_put<C>(S_SIG, lane, lane == 0 ? f2_one<C>() : f2_zero<C>());
      fe_put<C>(S_LINE, lane, f2_zero<C>());
    }
    wave_sync();
    const bool have = sig != nullptr && !sig->inf;          // uniform
    if (have) {
      const Fp<C> xP = sig->x, yP = sig->y;
      int step = 0;
      auto apply = [&]() {
        if (lane < 3) {
          const LineCoeffs<C> l = gen_lines[step];
          // D-type: c0 yP + c1 xP w + c2 w^3     M-type: c2 + c1 xP w^2 + c0 yP w^3   (coop.hpp COOP_SH_D / COOP_SH_M)
          const Fp2<C> e = lane == 0 ? (C::TWIST_D ? f2_muls<C>(l.c0, yP) : l.c2)
                         : lane == 1 ? f2_muls<C>(l.c1, xP)
                                     : (C::TWIST_D ? l.c2 : f2_muls<C>(l.c0, yP));
          const int k = lane == 0 ? 0 : lane == 1 ? (C::TWIST_D ? 1 : 2) : 3;
          fe_put<C>(S_LINE, k, e);
        }
        ++step;
        wave_sync();
        fe_mul<C>(S_SIG, S_SIG, S_LINE);
      };
      for (int i = 1; i < C::LOOP_LEN; ++i) {
        if (i > 1) fe_mul<C>(S_SIG, S_SIG, S_SIG);
        apply();
        if (C::LOOP_NAF[i] != 0) apply();
      }
      if constexpr (C::CURVE_ID == 0) {
        apply();
        apply();
      } else {
        fe_conj<C>(S_SIG, S_SIG);                            // x < 0
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    fe_mul<C>(S_ACC, S_ACC, S_SIG);
    if (lane < 6) {
      const int order_pos[6] = {5, 2, 4, 1, 3, 0};
      const Fp2<C> v = lds_load_f2<C>(E::coef(S_ACC, lane, 0));
      uint8_t* o = out + (size_t)(2 * order_pos[lane]) * C::FP_BYTES;
      fp_to_be<C>(o, fp_from_mont<C>(v.c1));
      fp_to_be<C>(o + C::FP_BYTES, fp_from_mont<C>(v.c0));
    }
  }
}

// ======================================================================= host side
namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[256];
  if (e != hipSuccess)
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(buf, sizeof buf, "%s", what);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                         \
  do {                                                       \
    hipError_t e_ = (expr);                                  \
    if (e_ != hipSuccess) return fail(BGLS_ERR_HIP, #expr, e_); \
  } while (0)

struct Ctx {
  std::mutex mu;
  int device = 0;
  bool ready = false;
  hipStream_t stream = nullptr;
  std::vector<std::pair<void*, size_t>> ws;  // cached device workspaces by slot
  void* gen_lines[2] = {nullptr, nullptr};   // k_gen_lines tables, one per curve, built on first use
  uint32_t* h_res = nullptr;                 // pinned host words {verdict, final-stage flags, caller flags} of the verification in flight
  bool res_pending = false;
  hipStream_t res_stream = nullptr;
  // optional per-stage timing with HIP events on the launch stream (bench.py roofline leg)
  bool prof = false;
  struct Pending { hipEvent_t a, b; int stage; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;
  double stage_ms[8] = {0};
  unsigned long long stage_cnt[8] = {0};
  hipEvent_t ev() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void collect() {
    for (auto& p : pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { stage_ms[p.stage] += ms; stage_cnt[p.stage] += 1; }
      ev_pool.push_back(p.a);
      ev_pool.push_back(p.b);
    }
    pending.clear();
  }

  int ensure() {
    if (ready) return 0;
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) return fail(BGLS_ERR_NO_DEVICE, "no HIP device available", e);
    if (device >= cnt) return fail(BGLS_ERR_NO_DEVICE, "device index out of range");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreate(&stream));
    HIPCHK(hipHostMalloc((void**)&h_res, 64));
    ws.assign(40, {nullptr, 0});   // >= WS_NUM
    ready = true;
    return 0;
  }
  int get(int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (ws[slot].second < bytes) {
      if (ws[slot].first) HIPCHK(hipFree(ws[slot].first));
      ws[slot] = {nullptr, 0};
      size_t cap = bytes + bytes / 4;
      HIPCHK(hipMalloc(&ws[slot].first, cap));
      ws[slot].second = cap;
    }
    *out = ws[slot].first;
    return 0;
  }
};

// Contexts: each owns a stream, its workspaces and its stage timers.  A host thread works on the context it selected
// (bgls_select_context, default 0); two contexts let one thread keep two verifications in flight, so the serial,
// latency-bound stages of one (hashing rounds, reduction tail, final exponentiation) overlap the other's Miller launch.
constexpr int NCTX = 8;
thread_local int g_sel = 0;
Ctx* ctx_all() {
  static Ctx c[NCTX];
  return c;
}
Ctx& ctx() {
  Ctx* all = ctx_all();
  Ctx& c = all[g_sel];
  if (g_sel != 0 && !c.ready) c.device = all[0].device;
  return c;
}

// Throughput mode (bgls_set_throughput_mode / BGLS_THROUGHPUT=1): alt-bn128 batches use k_miller_s60, whose launches are
// meant to overlap with their neighbours (several verifications in flight).
std::atomic<int> g_throughput{-1};
bool throughput_mode() {
  int v = g_throughput.load();
  if (v < 0) {
    const char* e = getenv("BGLS_THROUGHPUT");
    v = (e && e[0] == '1') ? 1 : 0;
    g_throughput.store(v);
  }
  return v == 1;
}

// BGLS_R28=1: alt-bn128 Miller kernel with the 28-bit-limb consumer (coop_r28.hpp)
bool r28_mode() {
  static const bool v = [] { const char* e = getenv("BGLS_R28"); return e && e[0] == '1'; }();
  return v;
}

// role-swap mask of k_miller_ab64 (see the kernel); BGLS_AB64_SWAP overrides it for A/B runs
unsigned ab64_swap() {
  static const unsigned v = [] {
    const char* e = getenv("BGLS_AB64_SWAP");
    return e ? (unsigned)strtoul(e, nullptr, 0) : 0u;
  }();
  return v;
}

// BGLS_KERNELS=v1 selects the round-1 thread-per-pairing kernels (kept for A/B measurements);
// default is the wave-cooperative path (coop.hpp).
// BGLS_MILLER=coop1 / ab forces the single-wave / producer-consumer cooperative kernel; default: by batch size.
int miller_mode() {
  static const int v = [] {
    const char* e = getenv("BGLS_MILLER");
    if (e && !strcmp(e, "coop1")) return 1;
    if (e && !strcmp(e, "ab")) return 2;
    return 0;
  }();
  return v;
}
// BGLS_FINAL=6 selects the 6-lane final exponentiation; default is the 36-lane one.
int final_mode() {
  static const int v = [] {
    const char* e = getenv("BGLS_FINAL");
    return (e && !strcmp(e, "6")) ? 6 : 36;
  }();
  return v;
}
// BGLS_H2C=rounds selects the exponentiation-tested compacting rounds; default: Legendre-symbol tests.
int h2c_mode() {
  static const int v = [] {
    const char* e = getenv("BGLS_H2C");
    return (e && !strcmp(e, "rounds")) ? 1 : 0;
  }();
  return v;
}
bool use_coop() {
  static const bool v = [] {
    const char* e = getenv("BGLS_KERNELS");
    return !(e && !strcmp(e, "v1"));
  }();
  return v;
}

enum { ST_DUP = 0, ST_H2C, ST_MILLER, ST_REDUCE, ST_FINAL, ST_SUM, ST_NUM };
const char* const STAGE_NAMES[ST_NUM] = {"dup_check", "h2c", "miller", "reduce", "final_exp", "sum_points"};

struct Scope {  // brackets the launches of one stage with events when profiling is on
  Ctx& c; hipStream_t st; int stage; hipEvent_t a = nullptr;
  Scope(Ctx& c_, hipStream_t st_, int stage_) : c(c_), st(st_), stage(stage_) {
    if (c.prof) { a = c.ev(); (void)hipEventRecord(a, st); }
  }
  ~Scope() {
    if (c.prof && a) { hipEvent_t b = c.ev(); (void)hipEventRecord(b, st); c.pending.push_back({a, b, stage}); }
  }
};

// workspace slots
enum { WS_G1S = 0, WS_F_A, WS_F_B, WS_FLAGS, WS_TABLE, WS_IN_A, WS_IN_B, WS_IN_C, WS_IN_D, WS_OUT, WS_JAC_A, WS_JAC_B, WS_PART, WS_TMP, WS_TMP2, WS_H2C_LIST, WS_H2C_CNT, WS_H2C_PTS, WS_H2C_KIND, WS_HAE_ROOT, WS_HAE_T, WS_HAE_KEYS, WS_HAE_APK, WS_HAE_SIGN, WS_FLAGS2, WS_NUM };

inline unsigned nblk(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

template <class C>
struct Engine {
  typedef F1<C> G1F;
  typedef F2<C> G2F;
  static constexpr size_t FB = C::FP_BYTES, G1B = 2 * FB, G2B = 4 * FB, GTB = 12 * FB;

  // d_flags: device u32 (already zeroed by caller).  Writes the product of the n (+1) Miller
  // values to d_partial (GT bytes, no final exponentiation).
  static int miller_product(Ctx& c, hipStream_t st, const uint8_t* d_sig, const uint8_t* d_keys, MsgView mv, size_t n,
                            int check_dups, uint8_t* d_partial, uint32_t* d_flags) {
    const size_t total = n + (d_sig ? 1 : 0);
    const bool raw = bls_raw(n);          // BLS12-381: uncleared hash points, cofactor applied once in GT
    void *g1s, *fa, *fb;
    int rc;
    if ((rc = c.get(WS_G1S, (total + 1) * sizeof(Aff<G1F>), &g1s))) return rc;
    if ((rc = c.get(WS_F_A, (total + 1) * sizeof(Fp12<C>), &fa))) return rc;
    if ((rc = c.get(WS_F_B, (total / 4 + 2) * sizeof(Fp12<C>), &fb))) return rc;
    if (check_dups && n > 1) {
      uint32_t cap = 1;
      while (cap < 2 * n) cap <<= 1;
      void* tab;
      if ((rc = c.get(WS_TABLE, (size_t)cap * 4, &tab))) return rc;
      Scope sc(c, st, ST_DUP);
      HIPCHK(hipMemsetAsync(tab, 0, (size_t)cap * 4, st));
      k_dup_check<<<nblk(n, 256), 256, 0, st>>>(mv, n, (uint32_t*)tab, cap - 1, d_flags);
    }
    if (n) {
      Scope sc(c, st, ST_H2C);
      int rc2 = hash_to_g1(c, st, mv, n, (Aff<G1F>*)g1s, d_flags, raw);
      if (rc2) return rc2;
    }
    if (d_sig) k_g1_parse<C><<<1, 64, 0, st>>>(d_sig, 1, 1, (Aff<G1F>*)g1s + n, d_flags);
    if (total == 0) {
      HIPCHK(hipMemsetAsync(d_partial, 0, GTB, st));
      uint8_t one = 1;
      HIPCHK(hipMemcpyAsync(d_partial + GTB - 1, &one, 1, hipMemcpyHostToDevice, st));
      return 0;
    }
    if (raw) return miller_coop(c, st, (const Aff<G1F>*)g1s, d_keys, n, -1LL, d_partial, d_flags, true, d_sig ? (const Aff<G1F>*)g1s + n : nullptr);
    if (use_coop())
      return miller_coop(c, st, (const Aff<G1F>*)g1s, d_keys, total, d_sig ? (long long)n : -1LL, d_partial, d_flags);
    {
      Scope sc(c, st, ST_MILLER);
      k_miller<C><<<nblk(total, 64), 64, 0, st>>>((const Aff<G1F>*)g1s, d_keys, total, d_sig ? (long long)n : -1LL,
                                                  (Fp12<C>*)fa, d_flags);
    }
    Scope sc(c, st, ST_REDUCE);
    return reduce_to_bytes(st, (Fp12<C>*)fa, (Fp12<C>*)fb, total, d_partial);
  }

  // raw (BLS12-381 only, see bls_raw): points before cofactor clearing, for the cofactor-in-GT verification path
  static bool bls_raw(size_t n) {
    return C::CURVE_ID == 1 && use_coop() && n > 0 && (n >= 256 || h2c_mode() == 0) && !getenv("BGLS_G1_COFACTOR");
  }
  static int hash_to_g1(Ctx& c, hipStream_t st, MsgView mv, size_t n, Aff<G1F>* out, uint32_t* d_flags, bool raw = false) {
    if constexpr (C::CURVE_ID == 0) {
      if (use_coop() && n < 256 && h2c_mode() == 0) {
        k_h2c_bn_jacobi<<<nblk(n, 64), 64, 0, st>>>(mv, n, out, d_flags);
        HIPCHK(hipGetLastError());
        return 0;
      }
      if (use_coop() && n >= 256) {
        void *lists, *cnts;
        int rc;
        if ((rc = c.get(WS_H2C_LIST, 2 * n * 4, &lists))) return rc;
        if ((rc = c.get(WS_H2C_CNT, 64, &cnts))) return rc;
        HIPCHK(hipMemsetAsync(cnts, 0, 64, st));
        uint32_t* L0 = (uint32_t*)lists;
        uint32_t* L1 = L0 + n;
        uint32_t* cn = (uint32_t*)cnts;
        auto grid = [&](double expect_items, int lpm) {
          double lanes = expect_items * lpm * 2.0 + 512.0;
          size_t b = (size_t)(lanes / 64.0) + 1;
          return (unsigned)(b > 8192 ? 8192 : b);
        };
        const double N = (double)n;
        // (lanes per message, first counter): 1@0, 4@1, 32@5, then 64 lanes per message up to counter 255.
        // Each round costs one try of latency, so the schedule is short: after three rounds a message is
        // still unfinished with probability 2^-37.  Default test = Legendre symbol (JAC), square root once
        // at the end; BGLS_H2C=rounds tests by exponentiation as the reference does.
#define BGLS_ROUNDS(JAC)                                                                                                        \
        k_h2c_bn_round<1, JAC><<<nblk(n, 64), 64, 0, st>>>(mv, n, nullptr, nullptr, 0, L0, cn + 1, 0, out, d_flags);           \
        k_h2c_bn_round<4, JAC><<<grid(N / 2, 4), 64, 0, st>>>(mv, n, L0, cn + 1, 1, L1, cn + 2, 0, out, d_flags);              \
        k_h2c_bn_round<32, JAC><<<grid(N / 32, 32), 64, 0, st>>>(mv, n, L1, cn + 2, 5, L0, cn + 3, 0, out, d_flags);           \
        k_h2c_bn_round<64, JAC><<<8, 64, 0, st>>>(mv, n, L0, cn + 3, 37, L1, cn + 4, 0, out, d_flags);                         \
        k_h2c_bn_round<64, JAC><<<8, 64, 0, st>>>(mv, n, L1, cn + 4, 101, L0, cn + 5, 0, out, d_flags);                        \
        k_h2c_bn_round<64, JAC><<<8, 64, 0, st>>>(mv, n, L0, cn + 5, 165, L1, cn + 6, 0, out, d_flags);                        \
        k_h2c_bn_round<64, JAC><<<8, 64, 0, st>>>(mv, n, L1, cn + 6, 229, L0, cn + 7, 1, out, d_flags);
        if (h2c_mode() == 0) {
          BGLS_ROUNDS(true)
          k_h2c_bn_finish<<<nblk(n, 64), 64, 0, st>>>(mv, n, out);
        } else {
          BGLS_ROUNDS(false)
        }
#undef BGLS_ROUNDS
        HIPCHK(hipGetLastError());
        return 0;
      }
    }
    if constexpr (C::CURVE_ID == 1) {
      if (use_coop() && (n >= 256 || h2c_mode() == 0)) {
        void *lists, *cnts, *pts, *kinds;
        int rc;
        const size_t items = 2 * n;
        if ((rc = c.get(WS_H2C_LIST, 2 * items * 4, &lists))) return rc;
        if ((rc = c.get(WS_H2C_CNT, 64, &cnts))) return rc;
        if ((rc = c.get(WS_H2C_PTS, items * sizeof(Aff<G1F>), &pts))) return rc;
        if ((rc = c.get(WS_H2C_KIND, items * 4, &kinds))) return rc;
        HIPCHK(hipMemsetAsync(cnts, 0, 64, st));
        uint32_t* L0 = (uint32_t*)lists;
        uint32_t* L1 = L0 + items;
        uint32_t* cn = (uint32_t*)cnts;
        auto grid = [&](size_t expect) {
          size_t b = expect / 64 + 8;
          return (unsigned)(b > 8192 ? 8192 : b);
        };
        if (h2c_mode() == 0) {
          k_bls_sw_jacobi<<<nblk(items, 64), 64, 0, st>>>(mv, items, (Aff<G1F>*)pts, (uint32_t*)kinds);
          if (raw) k_bls_combine<true><<<nblk(n, 64), 64, 0, st>>>(n, (const Aff<G1F>*)pts, (const uint32_t*)kinds, out);
        else k_bls_combine<false><<<nblk(n, 64), 64, 0, st>>>(n, (const Aff<G1F>*)pts, (const uint32_t*)kinds, out);
          HIPCHK(hipGetLastError());
          return 0;
        }
        k_bls_sw<0><<<grid(items), 64, 0, st>>>(mv, items, nullptr, nullptr, L0, cn + 1, (Aff<G1F>*)pts, (uint32_t*)kinds);
        k_bls_sw<1><<<grid(items / 2 + items / 8), 64, 0, st>>>(mv, items, L0, cn + 1, L1, cn + 2, (Aff<G1F>*)pts, (uint32_t*)kinds);
        k_bls_sw<2><<<grid(items / 4 + items / 8), 64, 0, st>>>(mv, items, L1, cn + 2, L0, cn + 3, (Aff<G1F>*)pts, (uint32_t*)kinds);
        if (raw) k_bls_combine<true><<<nblk(n, 64), 64, 0, st>>>(n, (const Aff<G1F>*)pts, (const uint32_t*)kinds, out);
        else k_bls_combine<false><<<nblk(n, 64), 64, 0, st>>>(n, (const Aff<G1F>*)pts, (const uint32_t*)kinds, out);
        HIPCHK(hipGetLastError());
        return 0;
      }
    }
    k_h2c<C><<<nblk(n, 64), 64, 0, st>>>(mv, n, out, d_flags);
    HIPCHK(hipGetLastError());
    return 0;
  }

  // wave-cooperative Miller product: groups of 6 lanes share one accumulator (coop.hpp)
  static int ensure_gen_lines(Ctx& c, hipStream_t st) {
    if (c.gen_lines[C::CURVE_ID]) return 0;
    void *tab, *cnt;
    int rc;
    HIPCHK(hipMalloc(&tab, 160 * sizeof(LineCoeffs<C>)));
    if ((rc = c.get(WS_OUT, 16, &cnt))) return rc;
    k_gen_lines<C><<<1, 64, 0, st>>>((LineCoeffs<C>*)tab, (int*)cnt);
    HIPCHK(hipGetLastError());
    c.gen_lines[C::CURVE_ID] = tab;
    return 0;
  }
  // serialise the reduced product; with `cofactor` (BLS12-381 raw hash points) raise it to h and fold the signature pair in
  static int emit_partial(Ctx& c, hipStream_t st, const Fp2<C>* w, bool cofactor, const Aff<G1F>* sig, uint8_t* d_partial) {
    if (!cofactor) {
      k_w_to_bytes<C><<<1, 64, 0, st>>>(w, d_partial);
    } else {
      int rc;
      if ((rc = ensure_gen_lines(c, st))) return rc;
      k_cofactor_epilogue<C><<<1, 128, FE<C>::LDS_BYTES2, st>>>(w, sig, (const LineCoeffs<C>*)c.gen_lines[C::CURVE_ID], d_partial);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }

  static int miller_coop(Ctx& c, hipStream_t st, const Aff<G1F>* g1s, const uint8_t* g2s, size_t total, long long gen_at,
                         uint8_t* d_partial, uint32_t* d_flags, bool cofactor = false, const Aff<G1F>* cof_sig = nullptr) {
    typedef Coop<C> K;
    // 64 pairings per block / 256 registers: one 2^16 batch is exactly 1024 resident blocks
    {
      const size_t npairs = gen_at >= 0 ? total - 1 : total;
      const size_t nb64 = (npairs + 63) / 64;
      // Larger batches run as consecutive launches of 1024 blocks (2^16 pairings each): measured per 2^16 pairings,
      // alt-bn128 6.2 ms either way, BLS12-381 11.0 ms against 15.3 ms for the single-wave kernel (one wave per SIMD there).
      // BGLS_AB64_MAX_BLOCKS=1024 restores the earlier rule (single-wave kernel above 2^16) for A/B runs.
      if constexpr (C::CURVE_ID == 0) {
        if (throughput_mode() && miller_mode() == 0 && npairs >= 1 && (gen_at < 0 || gen_at == (long long)npairs)) {
          int rc;
          void *pa, *pb;
          const size_t nb60 = (npairs + 59) / 60;
          const size_t groups60 = nb60 * 10;
          if ((rc = c.get(WS_F_A, (groups60 + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
          if ((rc = c.get(WS_F_B, (groups60 / 4 + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
          {
            Scope sc(c, st, ST_MILLER);
            for (size_t blk0 = 0; blk0 < nb60; blk0 += 8192) {            // grid size stays comfortably inside 32 bits
              const size_t nblocks = nb60 - blk0 < 8192 ? nb60 - blk0 : 8192;
              const size_t p0 = blk0 * 60;
              const char* dbg = getenv("BGLS_AB64_DBG");
              if (dbg && dbg[0] == '1')
                k_miller_s60<C, 1><<<(unsigned)nblocks, 128, Coop64<C, true>::BLOCK_BYTES, st>>>(g1s + p0, g2s + p0 * G2B, npairs - p0, (Fp2<C>*)pa + blk0 * 60, d_flags);
              else if (dbg && dbg[0] == '2')
                k_miller_s60<C, 2><<<(unsigned)nblocks, 128, Coop64<C, true>::BLOCK_BYTES, st>>>(g1s + p0, g2s + p0 * G2B, npairs - p0, (Fp2<C>*)pa + blk0 * 60, d_flags);
              else
                k_miller_s60<C><<<(unsigned)nblocks, 128, Coop64<C, true>::BLOCK_BYTES, st>>>(g1s + p0, g2s + p0 * G2B, npairs - p0, (Fp2<C>*)pa + blk0 * 60, d_flags);
            }
          }
          Scope sc(c, st, ST_REDUCE);
          Fp2<C>*a = (Fp2<C>*)pa, *b = (Fp2<C>*)pb;
          size_t cnt = groups60;
          while (cnt > 1) {
            size_t nout = (cnt + 3) / 4;
            k_reduce_coop<C><<<nblk(nout, K::GROUPS), 64, K::WAVE_BYTES, st>>>(a, cnt, 4, b);
            Fp2<C>* t = a;
            a = b;
            b = t;
            cnt = nout;
          }
          // the (-sigma, g2) pair is folded in by the epilogue kernel (generator lines, 36-lane arithmetic)
          return emit_partial(c, st, a, gen_at >= 0, gen_at >= 0 ? g1s + gen_at : nullptr, d_partial);
        }
      }
      static const size_t max_blocks = [] { const char* e = getenv("BGLS_AB64_MAX_BLOCKS"); return e ? (size_t)strtoull(e, nullptr, 0) : (size_t)1 << 40; }();
      if (miller_mode() == 0 && !getenv("BGLS_NO_AB64") && nb64 >= 1 && nb64 <= max_blocks && (gen_at < 0 || gen_at == (long long)npairs)) {
        int rc;
        if ((rc = ensure_gen_lines(c, st))) return rc;
        void *pa, *pb;
        const size_t groups64 = nb64 * 10;
        if ((rc = c.get(WS_F_A, (groups64 + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
        if ((rc = c.get(WS_F_B, (groups64 / 4 + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
        {
          Scope sc(c, st, ST_MILLER);
          const char* dbg = getenv("BGLS_AB64_DBG");
          const int dbgm = dbg ? dbg[0] - '0' : 0;
          for (size_t blk0 = 0; blk0 < nb64; blk0 += 1024) {
            const size_t nblocks = nb64 - blk0 < 1024 ? nb64 - blk0 : 1024;
            const size_t p0 = blk0 * 64;                                   // first pairing of this launch
            const size_t np = npairs - p0 < nblocks * 64 ? npairs - p0 : nblocks * 64;
            const Aff<G1F>* g1c = g1s + p0;
            const uint8_t* g2c = g2s + p0 * G2B;
            const long long sig_at = (blk0 == 0 && gen_at >= 0) ? gen_at : -1LL;   // block 0 of the first launch scales the generator lines
            Fp2<C>* outc = (Fp2<C>*)pa + blk0 * 10 * 6;
            const LineCoeffs<C>* gl = (const LineCoeffs<C>*)c.gen_lines[C::CURVE_ID];
            if (dbgm == 1 && !(C::CURVE_ID == 0 && r28_mode()))
              k_miller_ab64<C, 1><<<(unsigned)nblocks, 128, Coop64<C>::BLOCK_BYTES, st>>>(g1c, g2c, np, sig_at, gl, outc, d_flags, ab64_swap());
            else if (dbgm == 2 && !(C::CURVE_ID == 0 && r28_mode()))
              k_miller_ab64<C, 2><<<(unsigned)nblocks, 128, Coop64<C>::BLOCK_BYTES, st>>>(g1c, g2c, np, sig_at, gl, outc, d_flags, ab64_swap());
            else if (dbgm == 3 && C::CURVE_ID == 0)
              k_miller_ab64<BN254, 3><<<(unsigned)nblocks, 128, Coop64<BN254>::BLOCK_BYTES, st>>>((const Aff<F1<BN254>>*)g1c, g2c, np, sig_at, (const LineCoeffs<BN254>*)gl, (Fp2<BN254>*)outc, d_flags, ab64_swap());
            else if (C::CURVE_ID == 0 && r28_mode() && dbgm == 1)
              k_miller_ab64<BN254, 1, true><<<(unsigned)nblocks, 128, Coop64<BN254, true>::BLOCK_BYTES, st>>>((const Aff<F1<BN254>>*)g1c, g2c, np, sig_at, (const LineCoeffs<BN254>*)gl, (Fp2<BN254>*)outc, d_flags, ab64_swap());
            else if (C::CURVE_ID == 0 && r28_mode() && dbgm == 2)
              k_miller_ab64<BN254, 2, true><<<(unsigned)nblocks, 128, Coop64<BN254, true>::BLOCK_BYTES, st>>>((const Aff<F1<BN254>>*)g1c, g2c, np, sig_at, (const LineCoeffs<BN254>*)gl, (Fp2<BN254>*)outc, d_flags, ab64_swap());
            else if (C::CURVE_ID == 0 && r28_mode())
              k_miller_ab64<BN254, 0, true><<<(unsigned)nblocks, 128, Coop64<BN254, true>::BLOCK_BYTES, st>>>((const Aff<F1<BN254>>*)g1c, g2c, np, sig_at, (const LineCoeffs<BN254>*)gl, (Fp2<BN254>*)outc, d_flags, ab64_swap());
            else
              k_miller_ab64<C><<<(unsigned)nblocks, 128, Coop64<C>::BLOCK_BYTES, st>>>(g1c, g2c, np, sig_at, gl, outc, d_flags, ab64_swap());
          }
        }
        Scope sc(c, st, ST_REDUCE);
        Fp2<C>*a = (Fp2<C>*)pa, *b = (Fp2<C>*)pb;
        size_t cnt = groups64;
        while (cnt > 1) {
          size_t nout = (cnt + 3) / 4;
          k_reduce_coop<C><<<nblk(nout, K::GROUPS), 64, K::WAVE_BYTES, st>>>(a, cnt, 4, b);
          Fp2<C>* t = a;
          a = b;
          b = t;
          cnt = nout;
        }
        return emit_partial(c, st, a, cofactor, cof_sig, d_partial);
      }
    }
    const size_t max_groups = 256 * 8 * K::GROUPS;                 // 8 resident waves per CU
    size_t groups = (total + 5) / 6;
    if (groups > max_groups) groups = max_groups;
    const int rounds = (int)((total + groups * 6 - 1) / (groups * 6));
    void *pa, *pb;
    int rc;
    if ((rc = c.get(WS_F_A, (groups + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
    if ((rc = c.get(WS_F_B, (groups / 4 + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
    {
      Scope sc(c, st, ST_MILLER);
      // producer/consumer pairs double the wave count: worth it while all blocks stay resident
      // (5 blocks per CU by LDS); larger batches saturate the chip with the single-wave form.
      const bool ab = miller_mode() == 2 || (miller_mode() == 0 && nblk(groups, K::GROUPS) <= 1280u);
      if (!ab)
        k_miller_coop<C><<<nblk(groups, K::GROUPS), 64, K::WAVE_BYTES, st>>>(g1s, g2s, total, gen_at, rounds, groups, (Fp2<C>*)pa, d_flags);
      else
        if (!getenv("BGLS_STEP_CALLS"))   // in-place point steps by default; BGLS_STEP_CALLS=1 keeps the out-of-line variant
          k_miller_ab<C, true><<<nblk(groups, K::GROUPS), 128, K::BLOCK_BYTES_AB, st>>>(g1s, g2s, total, gen_at, rounds, groups, (Fp2<C>*)pa, d_flags);
        else
          k_miller_ab<C, false><<<nblk(groups, K::GROUPS), 128, K::BLOCK_BYTES_AB, st>>>(g1s, g2s, total, gen_at, rounds, groups, (Fp2<C>*)pa, d_flags);
    }
    Scope sc(c, st, ST_REDUCE);
    Fp2<C>*a = (Fp2<C>*)pa, *b = (Fp2<C>*)pb;
    size_t cnt = groups;
    const int R = 4;                       // each pass costs R-1 dependent products of latency: keep the tree shallow per pass
    while (cnt > 1) {
      size_t nout = (cnt + R - 1) / R;
      k_reduce_coop<C><<<nblk(nout, K::GROUPS), 64, K::WAVE_BYTES, st>>>(a, cnt, R, b);
      Fp2<C>* t = a;
      a = b;
      b = t;
      cnt = nout;
    }
    return emit_partial(c, st, a, cofactor, cof_sig, d_partial);
  }

  static int reduce_to_bytes(hipStream_t st, Fp12<C>* a, Fp12<C>* b, size_t cnt, uint8_t* d_out) {
    const int R = 8;
    while (cnt > 1) {
      size_t nout = (cnt + R - 1) / R;
      k_f12_reduce<C><<<nblk(nout, 64), 64, 0, st>>>(a, cnt, R, b);
      Fp12<C>* t = a;
      a = b;
      b = t;
      cnt = nout;
    }
    k_f12_to_bytes<C><<<1, 64, 0, st>>>(a, d_out);
    HIPCHK(hipGetLastError());
    return 0;
  }

  // enqueue product-of-partials + final exponentiation + compare; the verdict lands in the context's pinned words
  static int finalize_submit(Ctx& c, hipStream_t st, const uint8_t* d_partials, size_t count, int do_final_exp, const uint32_t* d_flags_in,
                             uint8_t* h_gt_out) {
    void *tmp, *fl;
    int rc;
    if (c.res_pending) return fail(BGLS_ERR_ARG, "a verification is already in flight on this context (collect it first)");
    if ((rc = c.get(WS_TMP, GTB + 16, &tmp))) return rc;
    if ((rc = c.get(WS_OUT, 16, &fl))) return rc;
    uint8_t* d_gt = (uint8_t*)tmp;
    uint32_t* d_verdict = (uint32_t*)(d_gt + GTB);
    uint32_t* d_fl2 = (uint32_t*)fl;
    HIPCHK(hipMemsetAsync(d_fl2, 0, 4, st));
    {
      Scope sc(c, st, ST_FINAL);
      if (use_coop() && final_mode() == 36)
        k_final36<C><<<1, 64, FE<C>::LDS_BYTES, st>>>(d_partials, count, do_final_exp, d_gt, d_verdict, d_fl2);
      else if (use_coop())
        k_final_coop<C><<<1, 64, Coop<C>::WAVE_BYTES, st>>>(d_partials, count, do_final_exp, d_gt, d_verdict, d_fl2);
      else
        k_final<C><<<1, 64, 0, st>>>(d_partials, count, do_final_exp, d_gt, d_verdict, d_fl2);
    }
    HIPCHK(hipGetLastError());
    c.h_res[0] = c.h_res[1] = c.h_res[2] = 0;
    HIPCHK(hipMemcpyAsync(&c.h_res[0], d_verdict, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&c.h_res[1], d_fl2, 4, hipMemcpyDeviceToHost, st));
    if (d_flags_in) HIPCHK(hipMemcpyAsync(&c.h_res[2], d_flags_in, 4, hipMemcpyDeviceToHost, st));
    if (h_gt_out) HIPCHK(hipMemcpyAsync(h_gt_out, d_gt, GTB, hipMemcpyDeviceToHost, st));
    c.res_pending = true;
    c.res_stream = st;
    return 0;
  }
  // wait for the verification in flight; returns 1/0 or <0
  static int finalize_collect(Ctx& c) {
    if (!c.res_pending) return fail(BGLS_ERR_ARG, "no verification in flight on this context");
    c.res_pending = false;
    HIPCHK(hipStreamSynchronize(c.res_stream));
    c.collect();
    uint32_t f = c.h_res[1] | c.h_res[2];
    if (f & FLAG_ENC) return fail(BGLS_ERR_ENCODING, "non-canonical coordinate or point not on curve");
    if (f & FLAG_HASH) return fail(BGLS_ERR_HASH, "try-and-increment exhausted");
    if (f & FLAG_DUP) return 0;
    return c.h_res[0] ? 1 : 0;
  }
  // returns 1/0 or <0; optionally copies the GT bytes out
  static int finalize(Ctx& c, hipStream_t st, const uint8_t* d_partials, size_t count, int do_final_exp, const uint32_t* d_flags_in,
                      uint8_t* h_gt_out) {
    int rc;
    if ((rc = finalize_submit(c, st, d_partials, count, do_final_exp, d_flags_in, h_gt_out))) return rc;
    return finalize_collect(c);
  }

  template <class F, int PTB>
  static int sum_points(Ctx& c, hipStream_t st, const uint8_t* d_pts, size_t n, uint8_t* d_out, uint32_t* d_flags) {
    if (n == 0) {
      HIPCHK(hipMemsetAsync(d_out, 0, PTB, st));
      return 0;
    }
    // Fan-in per pass.  A pass costs (fan-in - 1) dependent additions of latency, so once the partial sums no longer
    // fill the chip (two resident waves per SIMD at these register counts = 131 072 threads) the tree narrows by 2
    // per pass instead of 16: log2 passes of ONE addition each instead of a few passes of 15.  BGLS_SUM_R forces one
    // fan-in everywhere (16 = the earlier shape) for A/B runs.
    static const int forced = [] { const char* e = getenv("BGLS_SUM_R"); return e ? atoi(e) : 0; }();
    auto fan = [&](size_t items) {
      if (forced >= 2) return forced;
      size_t r = (items + 131071) / 131072;
      return (int)(r < 2 ? 2 : r > 16 ? 16 : r);
    };
    void *ja, *jb;
    int rc;
    Scope sc(c, st, ST_SUM);
    const int R1 = fan(n);
    size_t n1 = (n + R1 - 1) / R1;
    if ((rc = c.get(WS_JAC_A, (n1 + 1) * sizeof(Jac<F>), &ja))) return rc;
    if ((rc = c.get(WS_JAC_B, (n1 / 2 + 2) * sizeof(Jac<F>), &jb))) return rc;
    k_sum_first<F, PTB><<<nblk(n1, 64), 64, 0, st>>>(d_pts, n, R1, (Jac<F>*)ja, d_flags);
    Jac<F>*a = (Jac<F>*)ja, *b = (Jac<F>*)jb;
    size_t cnt = n1;
    while (cnt > 1) {
      const int R = fan(cnt);
      size_t nout = (cnt + R - 1) / R;
      k_sum_next<F><<<nblk(nout, 64), 64, 0, st>>>(a, cnt, R, b);
      Jac<F>* t = a;
      a = b;
      b = t;
      cnt = nout;
    }
    k_jac_to_bytes<F><<<1, 64, 0, st>>>(a, 1, d_out, PTB);
    HIPCHK(hipGetLastError());
    return 0;
  }
};

int flags_to_rc(uint32_t f) {
  if (f & FLAG_ENC) return fail(BGLS_ERR_ENCODING, "non-canonical coordinate or point not on curve");
  if (f & FLAG_HASH) return fail(BGLS_ERR_HASH, "try-and-increment exhausted");
  return 0;
}

#define DISPATCH(curve, CALL)                                    \
  do {                                                           \
    if ((curve) == BGLS_CURVE_ALTBN128) {                        \
      typedef BN254 CV;                                          \
      return CALL;                                               \
    } else if ((curve) == BGLS_CURVE_BLS12_381) {                \
      typedef BLS381 CV;                                         \
      return CALL;                                               \
    }                                                            \
    return fail(BGLS_ERR_ARG, "unknown curve id");               \
  } while (0)

template <class C>
int verify_aggregate_t(const uint8_t* sig, const uint8_t* keys, const uint8_t* blob, const uint64_t* off, size_t n, int allow_dups) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = n ? off[n] : 0;
  void *d_sig, *d_keys, *d_blob, *d_off, *d_flags, *d_part;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, mv, n, !allow_dups, (uint8_t*)d_part,
                              (uint32_t*)d_flags)))
    return rc;
  return E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int verify_multi_dev_t(Ctx& c, hipStream_t st, const uint8_t* d_sig, const uint8_t* d_keys, size_t n, const uint8_t* d_msg,
                       size_t msg_len, bool submit_only = false) {
  typedef Engine<C> E;
  int rc;
  void *d_flags, *d_g2s, *d_g1s, *fa, *fb, *d_part;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_TMP2, 2 * E::G2B, &d_g2s))) return rc;
  if ((rc = c.get(WS_G1S, 4 * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_F_A, 4 * sizeof(Fp12<C>), &fa))) return rc;
  if ((rc = c.get(WS_F_B, 4 * sizeof(Fp12<C>), &fb))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  // apk = sum(keys)  (AggregatePoints)
  if ((rc = E::template sum_points<F2<C>, (int)E::G2B>(c, st, d_keys, n, (uint8_t*)d_g2s, (uint32_t*)d_flags))) return rc;
  // h = -H(msg); pairs (h, apk), (sig, g2)
  MsgView mv = {d_msg, nullptr, msg_len, msg_len};
  Aff<F1<C>>* g1s = (Aff<F1<C>>*)d_g1s;
  if (use_coop()) {
    // one message through the batch hashing path, then the two-pairing product on the cooperative Miller kernel with
    // the (sig, g2) pair on the pre-computed generator lines
    if ((rc = E::hash_to_g1(c, st, mv, 1, g1s + 2, (uint32_t*)d_flags))) return rc;
    k_g1_to_bytes<C><<<1, 64, 0, st>>>(g1s + 2, 1, (uint8_t*)d_part);           // scratch: H(m) bytes
    k_g1_parse<C><<<1, 64, 0, st>>>((const uint8_t*)d_part, 1, 1, g1s, (uint32_t*)d_flags);  // -H(m)
    k_g1_parse<C><<<1, 64, 0, st>>>(d_sig, 1, 0, g1s + 1, (uint32_t*)d_flags);               // sig
    if ((rc = E::miller_coop(c, st, g1s, (const uint8_t*)d_g2s, 2, 1LL, (uint8_t*)d_part, (uint32_t*)d_flags))) return rc;
  } else {
    k_h2c<C><<<1, 64, 0, st>>>(mv, 1, g1s + 2, (uint32_t*)d_flags);
    k_g1_to_bytes<C><<<1, 64, 0, st>>>(g1s + 2, 1, (uint8_t*)d_part);           // scratch: H(m) bytes
    k_g1_parse<C><<<1, 64, 0, st>>>((const uint8_t*)d_part, 1, 1, g1s, (uint32_t*)d_flags);  // -H(m)
    k_g1_parse<C><<<1, 64, 0, st>>>(d_sig, 1, 0, g1s + 1, (uint32_t*)d_flags);               // sig
    k_miller<C><<<1, 64, 0, st>>>(g1s, (const uint8_t*)d_g2s, 2, 1LL, (Fp12<C>*)fa, (uint32_t*)d_flags);
    if ((rc = E::reduce_to_bytes(st, (Fp12<C>*)fa, (Fp12<C>*)fb, 2, (uint8_t*)d_part))) return rc;
  }
  if (submit_only) return E::finalize_submit(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
  return E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int verify_multi_t(const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  void *d_sig, *d_keys, *d_msg;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, msg_len, &d_msg))) return rc;
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (msg_len) HIPCHK(hipMemcpyAsync(d_msg, msg, msg_len, hipMemcpyHostToDevice, st));
  return verify_multi_dev_t<C>(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, n, (const uint8_t*)d_msg, msg_len);
}

template <class C>
int pairing_product_t(const uint8_t* g1s, const uint8_t* g2s, size_t n, uint8_t* gt_out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  void *d_g1b, *d_g2b, *d_g1s, *fa, *fb, *d_flags, *d_part;
  if ((rc = c.get(WS_IN_A, n * E::G1B, &d_g1b))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_g2b))) return rc;
  if ((rc = c.get(WS_G1S, (n + 1) * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_F_A, (n + 1) * sizeof(Fp12<C>), &fa))) return rc;
  if ((rc = c.get(WS_F_B, (n / 4 + 2) * sizeof(Fp12<C>), &fb))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (n == 0) {
    memset(gt_out, 0, E::GTB);
    gt_out[E::GTB - 1] = 1;
    return 0;
  }
  HIPCHK(hipMemcpyAsync(d_g1b, g1s, n * E::G1B, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_g2b, g2s, n * E::G2B, hipMemcpyHostToDevice, st));
  k_g1_parse<C><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_g1b, n, 0, (Aff<F1<C>>*)d_g1s, (uint32_t*)d_flags);
  k_miller<C><<<nblk(n, 64), 64, 0, st>>>((const Aff<F1<C>>*)d_g1s, (const uint8_t*)d_g2b, n, -1LL, (Fp12<C>*)fa, (uint32_t*)d_flags);
  if ((rc = E::reduce_to_bytes(st, (Fp12<C>*)fa, (Fp12<C>*)fb, n, (uint8_t*)d_part))) return rc;
  rc = E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, gt_out);
  return rc < 0 ? rc : 0;
}

template <class C>
int hash_to_g1_t(const uint8_t* blob, const uint64_t* off, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = off[n];
  void *d_blob, *d_off, *d_g1s, *d_out, *d_flags;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_G1S, n * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_IN_A, n * E::G1B, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::hash_to_g1(c, st, mv, n, (Aff<F1<C>>*)d_g1s, (uint32_t*)d_flags))) return rc;
  k_g1_to_bytes<C><<<nblk(n, 64), 64, 0, st>>>((const Aff<F1<C>>*)d_g1s, n, (uint8_t*)d_out);
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * E::G1B, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int aggregate_points_t(int group, const uint8_t* pts, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_in, *d_out, *d_flags;
  if ((rc = c.get(WS_IN_B, n * PB, &d_in))) return rc;
  if ((rc = c.get(WS_OUT, PB, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (n) HIPCHK(hipMemcpyAsync(d_in, pts, n * PB, hipMemcpyHostToDevice, st));
  if (group == BGLS_G1)
    rc = E::template sum_points<F1<C>, (int)E::G1B>(c, st, (const uint8_t*)d_in, n, (uint8_t*)d_out, (uint32_t*)d_flags);
  else
    rc = E::template sum_points<F2<C>, (int)E::G2B>(c, st, (const uint8_t*)d_in, n, (uint8_t*)d_out, (uint32_t*)d_flags);
  if (rc) return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int scale_points_t(int group, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_in, *d_sc, *d_sg, *d_out, *d_flags;
  if ((rc = c.get(WS_IN_B, n * PB, &d_in))) return rc;
  if ((rc = c.get(WS_IN_C, n * 32, &d_sc))) return rc;
  if ((rc = c.get(WS_IN_D, n, &d_sg))) return rc;
  if ((rc = c.get(WS_IN_A, n * PB, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_in, pts, n * PB, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, st));
  if (signs) HIPCHK(hipMemcpyAsync(d_sg, signs, n, hipMemcpyHostToDevice, st));
  const uint8_t* sg = signs ? (const uint8_t*)d_sg : nullptr;
  if (group == BGLS_G1)
    k_scale<F1<C>, (int)E::G1B><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_in, (const uint8_t*)d_sc, sg, n, (uint8_t*)d_out,
                                                             (uint32_t*)d_flags);
  else
    k_scale<F2<C>, (int)E::G2B><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_in, (const uint8_t*)d_sc, sg, n, (uint8_t*)d_out,
                                                             (uint32_t*)d_flags);
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int point_check_t(int group, const uint8_t* a) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_in, *d_flags;
  if ((rc = c.get(WS_IN_B, PB, &d_in))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_in, a, PB, hipMemcpyHostToDevice, st));
  if (group == BGLS_G1)
    k_check<F1<C>, (int)E::G1B><<<1, 64, 0, st>>>((const uint8_t*)d_in, 1, (uint32_t*)d_flags);
  else
    k_check<F2<C>, (int)E::G2B><<<1, 64, 0, st>>>((const uint8_t*)d_in, 1, (uint32_t*)d_flags);
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return f ? 0 : 1;
}

template <class C>
int generator_t(int group, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void* d_out;
  if ((rc = c.get(WS_OUT, PB, &d_out))) return rc;
  k_generator<C><<<1, 64, 0, st>>>(group, (uint8_t*)d_out);
  HIPCHK(hipMemcpyAsync(out, d_out, PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

template <class C>
int gt_mul_t(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  void* d_in;
  if ((rc = c.get(WS_IN_A, 2 * E::GTB, &d_in))) return rc;
  HIPCHK(hipMemcpyAsync(d_in, a, E::GTB, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((uint8_t*)d_in + E::GTB, b, E::GTB, hipMemcpyHostToDevice, st));
  rc = E::finalize(c, st, (const uint8_t*)d_in, 2, 0, nullptr, out);
  return rc < 0 ? rc : 0;
}

template <class C>
int miller_product_dev_t(const void* d_sig, const void* d_keys, const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n,
                         int check_dups, void* d_partial, void* d_flags, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  MsgView mv = {(const uint8_t*)d_msgs, nullptr, msg_len, msg_stride};
  return E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, mv, n, check_dups, (uint8_t*)d_partial,
                           (uint32_t*)d_flags);
}

// containsDuplicateMessage (bgls/bgls.go:139-150) over device-resident fixed-stride messages: exact byte comparison
int duplicate_scan_dev(const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n, void* d_flags, void* stream) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  if (n < 2) return 0;
  MsgView mv = {(const uint8_t*)d_msgs, nullptr, msg_len, msg_stride};
  uint32_t cap = 1;
  while (cap < 2 * n) cap <<= 1;
  void* tab;
  if ((rc = c.get(WS_TABLE, (size_t)cap * 4, &tab))) return rc;
  Scope sc(c, st, ST_DUP);
  HIPCHK(hipMemsetAsync(tab, 0, (size_t)cap * 4, st));
  k_dup_check<<<nblk(n, 256), 256, 0, st>>>(mv, n, (uint32_t*)tab, cap - 1, (uint32_t*)d_flags);
  HIPCHK(hipGetLastError());
  return 0;
}

template <class C>
int final_verify_dev_t(const void* d_partials, size_t count, const void* d_flags, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  return E::finalize(c, st, (const uint8_t*)d_partials, count, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int final_verify_submit_dev_t(const void* d_partials, size_t count, const void* d_flags, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  return E::finalize_submit(c, st, (const uint8_t*)d_partials, count, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int aggregate_points_dev_t(int group, const void* d_pts, size_t n, void* d_out, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  void* d_flags;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (group == BGLS_G1)
    rc = E::template sum_points<F1<C>, (int)E::G1B>(c, st, (const uint8_t*)d_pts, n, (uint8_t*)d_out, (uint32_t*)d_flags);
  else
    rc = E::template sum_points<F2<C>, (int)E::G2B>(c, st, (const uint8_t*)d_pts, n, (uint8_t*)d_out, (uint32_t*)d_flags);
  if (rc) return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int verify_multi_dev_entry_t(const void* d_sig, const void* d_keys, size_t n, const void* d_msg, size_t msg_len, void* stream,
                             bool submit_only = false) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  return verify_multi_dev_t<C>(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, n, (const uint8_t*)d_msg, msg_len, submit_only);
}

// ---- hashed aggregation exponents / weighted sums: host flows -------------------------------------------------
// Root digest of BLAKE2Xb (hashes.hpp has the device-side tables; these are the host's own copies).  One sequential
// compression chain over all key bytes -- by construction not parallel -- computed while the keys travel to the device.
namespace host_blake2 {
const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                        0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
const uint8_t SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
inline uint64_t ror(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
inline void compress(uint64_t h[8], const uint8_t* block, uint64_t t, bool last) {
  uint64_t m[16], v[16];
  memcpy(m, block, 128);                                     // little-endian host
  for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
  v[12] ^= t;
  if (last) v[14] = ~v[14];
#define BGLS_G(a, b, c, d, x, y)                                                    \
  v[a] += v[b] + (x); v[d] = ror(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 24); \
  v[a] += v[b] + (y); v[d] = ror(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 63);
  for (int r = 0; r < 12; ++r) {
    const uint8_t* s = SIGMA[r];
    BGLS_G(0, 4, 8, 12, m[s[0]], m[s[1]]) BGLS_G(1, 5, 9, 13, m[s[2]], m[s[3]])
    BGLS_G(2, 6, 10, 14, m[s[4]], m[s[5]]) BGLS_G(3, 7, 11, 15, m[s[6]], m[s[7]])
    BGLS_G(0, 5, 10, 15, m[s[8]], m[s[9]]) BGLS_G(1, 6, 11, 12, m[s[10]], m[s[11]])
    BGLS_G(2, 7, 8, 13, m[s[12]], m[s[13]]) BGLS_G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
#undef BGLS_G
  for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
// h <- BLAKE2Xb root of data[0..len) for an XOF of xof_len bytes (x/crypto/blake2b/blake2x.go Reset + Write + finalize)
void xb_root(const uint8_t* data, size_t len, uint32_t xof_len, uint64_t h[8]) {
  for (int i = 0; i < 8; ++i) h[i] = IV[i];
  h[0] ^= 0x01010040ull;
  h[1] ^= (uint64_t)xof_len << 32;
  size_t off = 0;
  while (len - off > 128) {
    compress(h, data + off, (uint64_t)off + 128, false);
    off += 128;
  }
  uint8_t lastb[128];
  memset(lastb, 0, 128);
  if (len > off) memcpy(lastb, data + off, len - off);
  compress(h, lastb, (uint64_t)len, true);
}
}  // namespace host_blake2

// d_t (WS_HAE_T) <- the n 16-byte exponents of hashPubKeysToExponents (blsHAE.go:80-93) for the keys' wire bytes
template <class C>
int hae_exponents_dev(Ctx& c, hipStream_t st, const uint8_t* h_keys, size_t n, void** d_t) {
  typedef Engine<C> E;
  if (n >= (1ull << 28)) return fail(BGLS_ERR_ARG, "XOF length 16 n must fit a uint32 (blsHAE.go:81)");
  int rc;
  void* d_root;
  if ((rc = c.get(WS_HAE_ROOT, 64, &d_root))) return rc;
  if ((rc = c.get(WS_HAE_T, n * 16, d_t))) return rc;
  if (n == 0) return 0;
  uint64_t root[8];
  const uint32_t xof_len = (uint32_t)(16 * n);
  host_blake2::xb_root(h_keys, n * E::G2B, xof_len, root);
  HIPCHK(hipMemcpyAsync(d_root, root, 64, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));                          // root[] is a stack buffer
  k_blake2x_expand<<<nblk((xof_len + 63) / 64, 64), 64, 0, st>>>((const u64*)d_root, xof_len, (uint8_t*)*d_t);
  HIPCHK(hipGetLastError());
  return 0;
}

template <class C>
int hae_exponents_t(const uint8_t* keys, size_t n, uint8_t* t_out) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  void* d_t;
  if ((rc = hae_exponents_dev<C>(c, c.stream, keys, n, &d_t))) return rc;
  if (n) HIPCHK(hipMemcpyAsync(t_out, d_t, n * 16, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipStreamSynchronize(c.stream));
  return 0;
}

// d_out (affine bytes) <- sum_i w_i P_i over device-resident points and 16-byte weights
template <class C, class F, int PTB>
int weighted_sum_dev(Ctx& c, hipStream_t st, const uint8_t* d_pts, const uint8_t* d_w16, const uint8_t* d_signs, size_t n,
                     uint8_t* d_out, uint32_t* d_flags) {
  if (n == 0) {
    HIPCHK(hipMemsetAsync(d_out, 0, PTB, st));
    return 0;
  }
  void *ja, *jb;
  int rc;
  Scope sc(c, st, ST_SUM);
  if ((rc = c.get(WS_JAC_A, (n + 1) * sizeof(Jac<F>), &ja))) return rc;
  if ((rc = c.get(WS_JAC_B, (n / 2 + 2) * sizeof(Jac<F>), &jb))) return rc;
  k_wsum_first<F, PTB><<<nblk(n, 64), 64, 0, st>>>(d_pts, d_w16, d_signs, n, 1, (Jac<F>*)ja, d_flags);
  Jac<F>*a = (Jac<F>*)ja, *b = (Jac<F>*)jb;
  size_t cnt = n;
  while (cnt > 1) {
    size_t r16 = (cnt + 131071) / 131072;                 // same fan-in rule as Engine::sum_points
    const int R = (int)(r16 < 2 ? 2 : r16 > 16 ? 16 : r16);
    size_t nout = (cnt + R - 1) / R;
    k_sum_next<F><<<nblk(nout, 64), 64, 0, st>>>(a, cnt, R, b);
    Jac<F>* t = a;
    a = b;
    b = t;
    cnt = nout;
  }
  k_jac_to_bytes<F><<<1, 64, 0, st>>>(a, 1, d_out, PTB);
  HIPCHK(hipGetLastError());
  return 0;
}

// verify_multi with apk = sum w_i pk_i: VerifyMultiSignatureWithHAE (blsHAE.go:56-58; weights hashed from the keys) when
// mult == nullptr, the core of KoskVerifyMultiSignatureWithMultiplicity (blsKosk.go:137-150) otherwise.
template <class C>
int verify_multi_weighted_t(const uint8_t* sig, const uint8_t* keys, const int64_t* mult, size_t n, const uint8_t* msg, size_t msg_len) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  void *d_sig, *d_keys, *d_msg, *d_apk, *d_fl2, *d_t = nullptr, *d_sg = nullptr;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, msg_len, &d_msg))) return rc;
  if ((rc = c.get(WS_HAE_APK, E::G2B, &d_apk))) return rc;
  if ((rc = c.get(WS_FLAGS2, 16, &d_fl2))) return rc;
  HIPCHK(hipMemsetAsync(d_fl2, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (msg_len) HIPCHK(hipMemcpyAsync(d_msg, msg, msg_len, hipMemcpyHostToDevice, st));
  if (!mult) {
    if ((rc = hae_exponents_dev<C>(c, st, keys, n, &d_t))) return rc;
  } else {
    std::vector<uint8_t> w(n * 16, 0), sg(n, 0);
    for (size_t i = 0; i < n; ++i) {
      const int64_t m = mult[i];
      uint64_t mag = m < 0 ? (uint64_t)0 - (uint64_t)m : (uint64_t)m;
      sg[i] = m < 0 ? 1 : 0;
      for (int b = 0; b < 8; ++b) w[i * 16 + 15 - b] = (uint8_t)(mag >> (8 * b));
    }
    if ((rc = c.get(WS_HAE_T, n * 16, &d_t))) return rc;
    if ((rc = c.get(WS_HAE_SIGN, n, &d_sg))) return rc;
    if (n) {
      HIPCHK(hipMemcpyAsync(d_t, w.data(), n * 16, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_sg, sg.data(), n, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));                      // w, sg are locals
    }
  }
  if ((rc = weighted_sum_dev<C, F2<C>, (int)E::G2B>(c, st, (const uint8_t*)d_keys, (const uint8_t*)d_t, (const uint8_t*)d_sg, n,
                                                   (uint8_t*)d_apk, (uint32_t*)d_fl2)))
    return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(&f, d_fl2, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if ((rc = flags_to_rc(f))) return rc;
  return verify_multi_dev_t<C>(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_apk, 1, (const uint8_t*)d_msg, msg_len);
}

// VerifyAggregateSignatureWithHAE (blsHAE.go:49-53): keys scaled by their exponents, then verifyAggSig with duplicates allowed
template <class C>
int verify_aggregate_hae_t(const uint8_t* sig, const uint8_t* keys, const uint8_t* blob, const uint64_t* off, size_t n) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = n ? off[n] : 0;
  void *d_sig, *d_keys, *d_blob, *d_off, *d_flags, *d_part, *d_scaled, *d_t;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  if ((rc = c.get(WS_HAE_KEYS, n * E::G2B, &d_scaled))) return rc;
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if ((rc = hae_exponents_dev<C>(c, st, keys, n, &d_t))) return rc;
  if (n) {
    k_scale<F2<C>, (int)E::G2B><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_keys, (const uint8_t*)d_t, nullptr, n, (uint8_t*)d_scaled,
                                                             (uint32_t*)d_flags, 16);
    HIPCHK(hipGetLastError());
  }
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_scaled, mv, n, 0, (uint8_t*)d_part, (uint32_t*)d_flags)))
    return rc;
  return E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
}

// AggregateSignaturesWithHAE (blsHAE.go:39-46): sum_i t_i sigma_i
template <class C>
int aggregate_signatures_hae_t(const uint8_t* sigs, const uint8_t* keys, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  void *d_sigs, *d_out, *d_flags, *d_t;
  if ((rc = c.get(WS_IN_B, n * E::G1B, &d_sigs))) return rc;
  if ((rc = c.get(WS_OUT, E::G1B, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (n) HIPCHK(hipMemcpyAsync(d_sigs, sigs, n * E::G1B, hipMemcpyHostToDevice, st));
  if ((rc = hae_exponents_dev<C>(c, st, keys, n, &d_t))) return rc;
  if ((rc = weighted_sum_dev<C, F1<C>, (int)E::G1B>(c, st, (const uint8_t*)d_sigs, (const uint8_t*)d_t, nullptr, n, (uint8_t*)d_out,
                                                   (uint32_t*)d_flags)))
    return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, E::G1B, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

// Marshal / Unmarshal* compressed branch over a batch (alt-bn128 only: BLS12-381's compressed layout belongs to the
// un-vendored dis2/bls12 and is unpinned, curves/bls12_381.go:55,60,116,121)
int wire_points(int curve, int group, bool compress, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
  if (curve != BGLS_CURVE_ALTBN128) return fail(BGLS_ERR_ARG, "compressed point formats are defined for alt-bn128 only");
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  const size_t CB = group == BGLS_G1 ? 32 : 64, UB = 2 * CB;
  const size_t in_b = compress ? UB : CB, out_b = compress ? CB : UB;
  void *d_in, *d_out, *d_ok, *d_flags;
  if ((rc = c.get(WS_IN_B, n * in_b, &d_in))) return rc;
  if ((rc = c.get(WS_IN_A, n * out_b, &d_out))) return rc;
  if ((rc = c.get(WS_IN_D, n, &d_ok))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_in, in, n * in_b, hipMemcpyHostToDevice, st));
  {
    Scope sc(c, st, ST_SUM);
    if (compress) {
      if (group == BGLS_G1) k_compress_bn<BGLS_G1><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_in, n, (uint8_t*)d_out, (uint32_t*)d_flags);
      else k_compress_bn<BGLS_G2><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_in, n, (uint8_t*)d_out, (uint32_t*)d_flags);
    } else {
      if (group == BGLS_G1) k_decompress_bn<BGLS_G1><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_in, n, (uint8_t*)d_out, (uint8_t*)d_ok);
      else k_decompress_bn<BGLS_G2><<<nblk(n, 64), 64, 0, st>>>((const uint8_t*)d_in, n, (uint8_t*)d_out, (uint8_t*)d_ok);
    }
  }
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * out_b, hipMemcpyDeviceToHost, st));
  if (!compress) HIPCHK(hipMemcpyAsync(ok, d_ok, n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  c.collect();
  return flags_to_rc(f);
}

// LoadPublicKey over a batch (bgls/bgls.go:40-43): out[i] = sk_i * g2 (group = BGLS_G2) or sk_i * g1
template <class C>
int scale_generator_t(int group, const uint8_t* sks, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_sc, *d_out;
  if ((rc = c.get(WS_IN_C, n * 32, &d_sc))) return rc;
  if ((rc = c.get(WS_IN_A, n * PB, &d_out))) return rc;
  HIPCHK(hipMemcpyAsync(d_sc, sks, n * 32, hipMemcpyHostToDevice, st));
  if (group == BGLS_G1)
    k_scale_aff<C, F1<C>, (int)E::G1B><<<nblk(n, 64), 64, 0, st>>>(nullptr, (const uint8_t*)d_sc, n, (uint8_t*)d_out);
  else
    k_scale_aff<C, F2<C>, (int)E::G2B><<<nblk(n, 64), 64, 0, st>>>(nullptr, (const uint8_t*)d_sc, n, (uint8_t*)d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d_out, n * PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// Sign over a batch (bgls/bgls.go:46-56): out[i] = sk_i * HashToG1(msg_i); the hash points never leave the device
template <class C>
int sign_batch_t(const uint8_t* sks, const uint8_t* blob, const uint64_t* off, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = off[n];
  void *d_blob, *d_off, *d_g1s, *d_out, *d_flags, *d_sc;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_G1S, n * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_IN_A, n * E::G1B, &d_out))) return rc;
  if ((rc = c.get(WS_IN_B, n * 32, &d_sc))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_sc, sks, n * 32, hipMemcpyHostToDevice, st));
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::hash_to_g1(c, st, mv, n, (Aff<F1<C>>*)d_g1s, (uint32_t*)d_flags))) return rc;
  k_scale_aff<C, F1<C>, (int)E::G1B><<<nblk(n, 64), 64, 0, st>>>((const Aff<F1<C>>*)d_g1s, (const uint8_t*)d_sc, n, (uint8_t*)d_out);
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * E::G1B, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

bool group_ok(int g) { return g == BGLS_G1 || g == BGLS_G2; }

}  // namespace

// ======================================================================= C ABI
extern "C" {

int bgls_abi_version(void) { return 1; }

const char* bgls_last_error(void) { return g_err.c_str(); }

int bgls_init(int device) {
  Ctx* all = ctx_all();
  for (int i = 0; i < NCTX; ++i) {
    std::lock_guard<std::mutex> lk(all[i].mu);
    if (all[i].ready && all[i].device != device) return fail(BGLS_ERR_ARG, "bgls_init: context already bound to another device");
    if (!all[i].ready) all[i].device = device;
  }
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  return c.ensure();
}

size_t bgls_fp_size(int curve) { return curve == BGLS_CURVE_ALTBN128 ? 32 : curve == BGLS_CURVE_BLS12_381 ? 48 : 0; }
size_t bgls_g1_size(int curve) { return 2 * bgls_fp_size(curve); }
size_t bgls_g2_size(int curve) { return 4 * bgls_fp_size(curve); }
size_t bgls_gt_size(int curve) { return 12 * bgls_fp_size(curve); }

int bgls_verify_aggregate(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob, const uint64_t* msg_off,
                          size_t n, int allow_duplicates) {
  if (!sig || !msg_off || (n && !keys)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_aggregate_t<CV>(sig, keys, msg_blob, msg_off, n, allow_duplicates));
}

int bgls_verify_multi(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len) {
  if (!sig || (n && !keys) || (msg_len && !msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_t<CV>(sig, keys, n, msg, msg_len));
}

int bgls_pairing_product(int curve, const uint8_t* g1s, const uint8_t* g2s, size_t n, uint8_t* gt_out) {
  if (!gt_out || (n && (!g1s || !g2s))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, pairing_product_t<CV>(g1s, g2s, n, gt_out));
}

int bgls_hash_to_g1(int curve, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n, uint8_t* g1_out) {
  if (n && (!msg_off || !g1_out)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, hash_to_g1_t<CV>(msg_blob, msg_off, n, g1_out));
}

int bgls_aggregate_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* out) {
  if (!group_ok(group) || !out || (n && !pts)) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, aggregate_points_t<CV>(group, pts, n, out));
}

int bgls_scale_points(int curve, int group, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n,
                      uint8_t* out) {
  if (!group_ok(group) || (n && (!pts || !scalars || !out))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, scale_points_t<CV>(group, pts, scalars, signs, n, out));
}

int bgls_point_add(int curve, int group, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (!group_ok(group) || !a || !b || !out) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  size_t pb = group == BGLS_G1 ? bgls_g1_size(curve) : bgls_g2_size(curve);
  if (!pb) return fail(BGLS_ERR_ARG, "unknown curve id");
  std::vector<uint8_t> two(2 * pb);
  memcpy(two.data(), a, pb);
  memcpy(two.data() + pb, b, pb);
  return bgls_aggregate_points(curve, group, two.data(), 2, out);
}

int bgls_point_check(int curve, int group, const uint8_t* a) {
  if (!group_ok(group) || !a) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, point_check_t<CV>(group, a));
}

int bgls_generator(int curve, int group, uint8_t* out) {
  if (!group_ok(group) || !out) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, generator_t<CV>(group, out));
}

int bgls_pair(int curve, const uint8_t* g1, const uint8_t* g2, uint8_t* gt_out) {
  return bgls_pairing_product(curve, g1, g2, 1, gt_out);
}

int bgls_gt_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (!a || !b || !out) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, gt_mul_t<CV>(a, b, out));
}

int bgls_gt_identity(int curve, uint8_t* out) {
  size_t n = bgls_gt_size(curve);
  if (!n || !out) return fail(BGLS_ERR_ARG, "unknown curve id or NULL argument");
  memset(out, 0, n);
  out[n - 1] = 1;
  return 0;
}

int bgls_profile_enable(int on) {
  Ctx* all = ctx_all();
  for (int k = 0; k < NCTX; ++k) {
    Ctx& c = all[k];
    std::lock_guard<std::mutex> lk(c.mu);
    c.prof = on != 0;
    for (int i = 0; i < 8; ++i) { c.stage_ms[i] = 0; c.stage_cnt[i] = 0; }
  }
  return 0;
}

int bgls_profile_get(const char* stage, double* total_ms, unsigned long long* launches) {
  if (!stage || !total_ms || !launches) return fail(BGLS_ERR_ARG, "NULL argument");
  for (int i = 0; i < ST_NUM; ++i)
    if (!strcmp(stage, STAGE_NAMES[i])) {
      *total_ms = 0;
      *launches = 0;
      Ctx* all = ctx_all();
      for (int k = 0; k < NCTX; ++k) {                      // summed over the contexts
        std::lock_guard<std::mutex> lk(all[k].mu);
        *total_ms += all[k].stage_ms[i];
        *launches += all[k].stage_cnt[i];
      }
      return 0;
    }
  return fail(BGLS_ERR_ARG, "unknown stage name");
}

int bgls_probe_mad_peak(double* mac_per_s) {
  if (!mac_per_s) return fail(BGLS_ERR_ARG, "NULL argument");
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.ensure())) return rc;
  hipStream_t st = c.stream;
  void* sink;
  if ((rc = c.get(WS_OUT, 16, &sink))) return rc;
  const int iters = 4096, blocks = 256 * 8, threads = 256;
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  double best = 0;
  for (int rep = 0; rep < 4; ++rep) {
    HIPCHK(hipEventRecord(a, st));
    k_mad_probe<<<blocks, threads, 0, st>>>(12345u + rep, iters, (uint64_t*)sink);
    HIPCHK(hipEventRecord(b, st));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    double macs = (double)blocks * threads * iters * 16.0;
    double rate = macs / (ms * 1e-3);
    if (rep > 0 && rate > best) best = rate;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *mac_per_s = best;
  return 0;
}

int bgls_miller_product_dev(int curve, const void* d_sig, const void* d_keys, const void* d_msgs, size_t msg_len,
                            size_t msg_stride, size_t n, int check_duplicates, void* d_partial_out, void* d_flags,
                            void* stream) {
  if (!d_partial_out || !d_flags || (n && (!d_keys || (!d_msgs && msg_len)))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, miller_product_dev_t<CV>(d_sig, d_keys, d_msgs, msg_len, msg_stride, n, check_duplicates, d_partial_out,
                                           d_flags, stream));
}

int bgls_set_throughput_mode(int on) {
  g_throughput.store(on ? 1 : 0);
  return 0;
}

int bgls_select_context(int index) {
  if (index < 0 || index >= NCTX) return fail(BGLS_ERR_ARG, "context index out of range");
  g_sel = index;
  return 0;
}

int bgls_final_verify_submit_dev(int curve, const void* d_partials, size_t count, const void* d_flags, void* stream) {
  if (!d_partials || !count) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, final_verify_submit_dev_t<CV>(d_partials, count, d_flags, stream));
}

int bgls_final_verify_collect(int curve) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  DISPATCH(curve, Engine<CV>::finalize_collect(c));
}

int bgls_duplicate_scan_dev(const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n, void* d_flags, void* stream) {
  if (!d_flags || (n && !d_msgs && msg_len)) return fail(BGLS_ERR_ARG, "NULL argument");
  if (n >= (1ull << 30)) return fail(BGLS_ERR_ARG, "too many messages for one scan");
  return duplicate_scan_dev(d_msgs, msg_len, msg_stride, n, d_flags, stream);
}

int bgls_final_verify_dev(int curve, const void* d_partials, size_t count, const void* d_flags, void* stream) {
  if (!d_partials || !count) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, final_verify_dev_t<CV>(d_partials, count, d_flags, stream));
}

int bgls_aggregate_points_dev(int curve, int group, const void* d_pts, size_t n, void* d_out, void* stream) {
  if (!group_ok(group) || !d_out || (n && !d_pts)) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, aggregate_points_dev_t<CV>(group, d_pts, n, d_out, stream));
}

int bgls_verify_multi_dev(int curve, const void* d_sig, const void* d_keys, size_t n, const void* d_msg, size_t msg_len,
                          void* stream) {
  if (!d_sig || (n && !d_keys) || (msg_len && !d_msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_dev_entry_t<CV>(d_sig, d_keys, n, d_msg, msg_len, stream));
}

int bgls_verify_multi_submit_dev(int curve, const void* d_sig, const void* d_keys, size_t n, const void* d_msg, size_t msg_len,
                                 void* stream) {
  if (!d_sig || (n && !d_keys) || (msg_len && !d_msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_dev_entry_t<CV>(d_sig, d_keys, n, d_msg, msg_len, stream, true));
}

/* ---- hashed aggregation exponents (bgls/blsHAE.go) and multiplicities (bgls/blsKosk.go:137-150) ---- */
int bgls_hae_exponents(int curve, const uint8_t* keys, size_t n, uint8_t* t_out) {
  if (n && (!keys || !t_out)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, hae_exponents_t<CV>(keys, n, t_out));
}

int bgls_aggregate_signatures_hae(int curve, const uint8_t* sigs, const uint8_t* keys, size_t n, uint8_t* out) {
  if (!out || (n && (!sigs || !keys))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, aggregate_signatures_hae_t<CV>(sigs, keys, n, out));
}

int bgls_verify_multi_hae(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len) {
  if (!sig || (n && !keys) || (msg_len && !msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_weighted_t<CV>(sig, keys, nullptr, n, msg, msg_len));
}

int bgls_verify_aggregate_hae(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob, const uint64_t* msg_off,
                              size_t n) {
  if (!sig || !msg_off || (n && !keys)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_aggregate_hae_t<CV>(sig, keys, msg_blob, msg_off, n));
}

int bgls_verify_multi_multiplicity(int curve, const uint8_t* sig, const uint8_t* keys, const int64_t* multiplicity, size_t n,
                                   const uint8_t* msg, size_t msg_len) {
  if (!sig || (n && !keys) || (msg_len && !msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  if (!multiplicity) DISPATCH(curve, verify_multi_t<CV>(sig, keys, n, msg, msg_len));
  DISPATCH(curve, verify_multi_weighted_t<CV>(sig, keys, multiplicity, n, msg, msg_len));
}

/* ---- compressed wire formats (alt-bn128; curves/altbn128.go:81-89,203-221,296-376) ---- */
int bgls_compress_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* out) {
  if (!group_ok(group) || (n && (!pts || !out))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  return wire_points(curve, group, true, pts, n, out, nullptr);
}

int bgls_decompress_points(int curve, int group, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
  if (!group_ok(group) || (n && (!in || !out || !ok))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  return wire_points(curve, group, false, in, n, out, ok);
}

/* ---- batch key generation / signing (bgls/bgls.go:40-56) ---- */
int bgls_scale_generator(int curve, int group, const uint8_t* scalars, size_t n, uint8_t* out) {
  if (!group_ok(group) || (n && (!scalars || !out))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, scale_generator_t<CV>(group, scalars, n, out));
}

int bgls_sign_batch(int curve, const uint8_t* sks, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n, uint8_t* sigs_out) {
  if (!msg_off || (n && (!sks || !sigs_out))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, sign_batch_t<CV>(sks, msg_blob, msg_off, n, sigs_out));
}

}  // extern "C"
