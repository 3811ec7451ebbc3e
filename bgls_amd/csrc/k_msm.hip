// Windowed / bucketed scalar multiplications (north_star "windowed scalar multiplication"):
//
//  * weighted sums  sum_i w_i P_i  with 128-bit weights by the bucket method -- getAggregatePubKey (bgls/blsHAE.go:74-77 =
//    AggregatePoints(ScalePoints(keys, t)), curves/curve.go:73-121,190-214) and AggregateSignaturesWithHAE
//    (bgls/blsHAE.go:39-46) without one scalar multiplication per point;
//  * fixed-base multiples of the group generators from a resident table of window multiples -- LoadPublicKey
//    (bgls/bgls.go:40-43: GetG2().Mul(sk)) over a batch;
//  * in-place scaling of G1 points by 128-bit weights (VerifyAggregateSignatureWithHAE, bgls/blsHAE.go:49-53: the
//    exponent moves from the key to the hash point, e(t H, pk) = e(H, t pk)).
//
// Bucket method: the weight of point i is cut into W digits of c bits; digit d of window j sends the point to bucket
// (j, d).  A counting sort by bucket (histogram with atomics, one-block prefix sum, scatter of the point indices) turns
// the n W (point, window) pairs into bucket lists; a thread per bucket adds its points (mixed additions on resident
// Montgomery affine points); sum_d d B_(j,d) is formed per chunk of K digits with running sums, the chunk's offset is
// applied with c doublings, chunks are added per window by one wave each, and a last wave doubles window j c j times
// and adds the windows up.  The order inside a bucket depends on the atomics; the sum does not (exact group law, the
// exceptional cases of the addition formulas are handled, the result leaves as canonical affine bytes).
#include "dev_common.hpp"
#include "wire.hpp"
#include "launch.hpp"
#include "points_inl.hpp"
#include "jac_coop.hpp"
#include "../../include/bgls_hip.h"

using namespace bgls;

namespace {

// 16-byte big-endian weight -> (lo, hi)
__device__ __forceinline__ void load_w16(const uint8_t* w, u64& lo, u64& hi) {
  u64 h = 0, l = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    h = (h << 8) | w[k];
    l = (l << 8) | w[8 + k];
  }
  lo = l;
  hi = h;
}
__device__ __forceinline__ u32 digit_of(u64 lo, u64 hi, int bit, u32 mask) {
  u64 v;
  if (bit >= 64) v = hi >> (bit - 64);
  else v = (lo >> bit) | (bit ? hi << (64 - bit) : 0ull);
  return (u32)v & mask;
}

// parse + check the points once (sign applied), count the bucket populations
template <class F, int PT_BYTES>
__global__ void __launch_bounds__(64) k_msm_parse(const uint8_t* pts, const uint8_t* w16, const uint8_t* signs, size_t n, int c, int W,
                                                  Aff<F>* aff, u32* cnt, u32* flags) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p;
  bool ok = aff_from_bytes<F>(p, pts + i * PT_BYTES);
  ok = ok && aff_on_curve<F>(p);
  if (!ok) atomicOr(flags, FLAG_ENC);
  if (signs && signs[i] == 1) p = aff_neg<F>(p);
  aff[i] = p;
  if (p.inf) return;
  u64 lo, hi;
  load_w16(w16 + i * 16, lo, hi);
  const u32 mask = (1u << c) - 1u;
  for (int j = 0; j < W; ++j) {
    const u32 d = digit_of(lo, hi, j * c, mask);
    if (d) atomicAdd(&cnt[((u32)j << c) + d], 1u);
  }
}

// exclusive prefix sum of the NB bucket populations (one block): start[b], start[NB] = total; cnt[b] becomes the bucket's
// write cursor; meta = {largest bucket, total}
__global__ void __launch_bounds__(1024) k_msm_scan(u32* cnt, u32 NB, u32* start, u32* meta) {
  __shared__ u32 part[1024], mx[1024];
  const u32 tid = threadIdx.x;
  const u32 per = (NB + 1023u) / 1024u;
  const u32 lo = tid * per < NB ? tid * per : NB, hi = lo + per < NB ? lo + per : NB;
  u32 s = 0, m = 0;
  for (u32 b = lo; b < hi; ++b) {
    const u32 v = cnt[b];
    s += v;
    m = v > m ? v : m;
  }
  part[tid] = s;
  mx[tid] = m;
  __syncthreads();
  if (tid == 0) {
    u32 run = 0, mm = 0;
    for (int t = 0; t < 1024; ++t) {
      const u32 v = part[t];
      part[t] = run;
      run += v;
      mm = mx[t] > mm ? mx[t] : mm;
    }
    meta[0] = mm;
    meta[1] = run;
    start[NB] = run;
  }
  __syncthreads();
  u32 run = part[tid];
  for (u32 b = lo; b < hi; ++b) {
    const u32 v = cnt[b];
    start[b] = run;
    cnt[b] = run;
    run += v;
  }
}

template <class F>
__global__ void __launch_bounds__(64) k_msm_scatter(const Aff<F>* aff, const uint8_t* w16, size_t n, int c, int W, u32* cursor, u32* list) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (aff[i].inf) return;
  u64 lo, hi;
  load_w16(w16 + i * 16, lo, hi);
  const u32 mask = (1u << c) - 1u;
  for (int j = 0; j < W; ++j) {
    const u32 d = digit_of(lo, hi, j * c, mask);
    if (d) list[atomicAdd(&cursor[((u32)j << c) + d], 1u)] = (u32)i;
  }
}

// Partial bucket sums: thread t adds every S-th point listed for bucket t / S, starting at t % S (S = 1: the whole
// bucket; the zero digit's bucket stays empty = infinity).  OCC = waves per SIMD the register budget is sized for: 1 when
// the launch has about one wave per SIMD anyway (no spills: a lone wave cannot hide scratch latency), 2 above that.
template <class F, int OCC>
__global__ void __launch_bounds__(64, OCC) k_msm_buckets(const Aff<F>* aff, const u32* list, const u32* start, u32 nthreads, int S, Jac<F>* out) {
  const u32 t = blockIdx.x * 64u + threadIdx.x;
  if (t >= nthreads) return;
  const u32 b = t / (u32)S, seg = t % (u32)S;
  const u32 lo = start[b] + seg, hi = start[b + 1];
  Jac<F> acc = jac_inf<F>();
#pragma unroll 1
  for (u32 k = lo; k < hi; k += (u32)S) acc = jac_madd_inl<F>(acc, aff[list[k]]);
  out[t] = acc;
}

// The three kernels below are chains of dependent additions on a few waves.  Their additions go through ONE out-of-line
// copy per field of the expanded formulas (a call costs a few microseconds of a 30-50 microsecond addition; a copy per
// call site multiplies the build time of this unit by five).
template <class F>
__device__ __noinline__ Jac<F> jac_add_site(const Jac<F>& p, const Jac<F>& q) { return jac_add_inl<F>(p, q); }
template <class F>
__device__ __noinline__ Jac<F> jac_dbl_site(const Jac<F>& p) { return jac_dbl_inl<F>(p); }

// chunk t = K consecutive digits d0 .. d0+K-1 of one window (flat bucket index t K: 2^c is a multiple of K), each bucket
// given as S partial sums:  out[t] = sum_k (d0 + k) B[d0 + k] = (running sums over k >= 1) + d0 (sum of the chunk)
template <class F>
__global__ void __launch_bounds__(64) k_msm_chunks(const Jac<F>* parts, int c, int K, int S, u32 nchunks, Jac<F>* out) {
  const u32 t = blockIdx.x * 64u + threadIdx.x;
  if (t >= nchunks) return;
  const u32 base = t * (u32)K;
  const u32 d0 = base & ((1u << c) - 1u);
  Jac<F> run = jac_inf<F>(), acc = jac_inf<F>();
#pragma unroll 1
  for (int k = K - 1; k >= 0; --k) {
    const Jac<F>* pb = parts + (size_t)(base + k) * S;
#pragma unroll 1
    for (int q = 0; q < S; ++q) run = jac_add_site<F>(run, pb[q]);
    if (k) acc = jac_add_site<F>(acc, run);
  }
  if (d0 && !jac_is_inf<F>(run)) {
    Jac<F> m = jac_inf<F>();
#pragma unroll 1
    for (int bit = c - 1; bit >= 0; --bit) {
      m = jac_dbl_site<F>(m);
      if ((d0 >> bit) & 1u) m = jac_add_site<F>(m, run);
    }
    acc = jac_add_site<F>(acc, m);
  }
  out[t] = acc;
}

// S_j = sum of the chunks of window j (one wave per window)
template <class F>
__global__ void __launch_bounds__(64) k_msm_windows(const Jac<F>* chunks, u32 NQ, Jac<F>* out) {
  const u32 j = blockIdx.x, lane = threadIdx.x;
  Jac<F> acc = jac_inf<F>();
#pragma unroll 1
  for (u32 q = lane; q < NQ; q += 64) acc = jac_add_site<F>(acc, chunks[j * NQ + q]);
#pragma unroll 1
  for (int off = 32; off >= 1; off >>= 1) {
    const Jac<F> o = jac_shfl_down<F>(acc, off);
    acc = jac_add_site<F>(acc, o);
  }
  if (lane == 0) out[j] = acc;
}

// sum_j 2^(c j) S_j (one wave; W <= 32)
template <class F>
__global__ void __launch_bounds__(64) k_msm_final(const Jac<F>* wins, int W, int c, Jac<F>* out) {
  const int lane = threadIdx.x;
  Jac<F> acc = lane < W ? wins[lane] : jac_inf<F>();
  const int nd = lane < W ? c * lane : 0;
#pragma unroll 1
  for (int k = 0; k < nd; ++k) acc = jac_dbl_site<F>(acc);
#pragma unroll 1
  for (int off = 32; off >= 1; off >>= 1) {
    const Jac<F> o = jac_shfl_down<F>(acc, off);
    acc = jac_add_site<F>(acc, o);
  }
  if (lane == 0) out[0] = acc;
}

// G2: window j doubled c j times, one wave per window (jac_coop.hpp: ~7 us a doubling instead of 23)
template <class C>
__global__ void __launch_bounds__(64) k_msm_shift(Jac<F2<C>>* wins, int c) {
  const int j = blockIdx.x;
  Jac<F2<C>> acc = wins[j];
  const CoopF2<C> k(0);
#pragma unroll 1
  for (int i = 0; i < c * j; ++i) acc = coop_jac_dbl<C>(k, acc);
  if (threadIdx.x == 0) wins[j] = acc;
}

// ---- fixed-base: table[j * 255 + d - 1] = d 2^(8 j) G  (j < 32 byte positions, d = 1 .. 255), Montgomery affine
constexpr int FB_ROW = 255, FB_WINDOWS = 32;

template <class C, class F>
__device__ __forceinline__ Aff<F> generator_of() {
  if constexpr (F::NFP == 1) return Aff<F>{fp_load<C>(C::G1X), fp_load<C>(C::G1Y), false};
  else return Aff<F>{f2_load<C>(C::G2), f2_load<C>(C::G2 + 2 * C::L), false};
}

template <class C, class F>
__global__ void __launch_bounds__(64) k_fb_build(Aff<F>* table) {
  const u32 t = blockIdx.x * 64u + threadIdx.x;
  if (t >= (u32)(FB_ROW * FB_WINDOWS)) return;
  const u32 j = t / FB_ROW, d = t % FB_ROW + 1;
  u32 k[8];
  for (int q = 0; q < 8; ++q) k[q] = 0;
  k[j >> 2] = d << (8 * (j & 3));
  table[t] = jac_to_aff<F>(jac_mul<F>(generator_of<C, F>(), k, 8 * (int)j + 8));
}

// out[i] = k_i G: one mixed addition per non-zero byte of the 32-byte big-endian scalar
template <class F, int PT_BYTES>
__global__ void __launch_bounds__(64, 2) k_fb_scale(const Aff<F>* table, const uint8_t* scalars, size_t n, uint8_t* out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac<F> acc = jac_inf<F>();
#pragma unroll 1
  for (int j = 0; j < FB_WINDOWS; ++j) {
    const u32 d = scalars[i * 32 + 31 - j];
    if (d) acc = jac_madd_inl<F>(acc, table[j * FB_ROW + d - 1]);
  }
  aff_to_bytes<F>(out + i * PT_BYTES, jac_to_aff<F>(acc));
}

// pts[i] <- w_i pts[i] (G1, 16-byte big-endian weights)
template <class C>
__global__ void __launch_bounds__(64) k_scale_g1_inplace(Aff<F1<C>>* pts, const uint8_t* w16, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[4];
  int top = -1;
  for (int j = 0; j < 4; ++j) {
    const uint8_t* q = w16 + i * 16 + 4 * (3 - j);
    k[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
  for (int j = 3; j >= 0 && top < 0; --j)
    if (k[j]) top = j * 32 + (31 - __clz(k[j]));
  pts[i] = jac_to_aff<F1<C>>(jac_mul_w4<F1<C>>(pts[i], k, top + 1));
}

}  // namespace

namespace bgls {
namespace kl {

// Window width: about eight points per bucket, between 4 and 13 bits (13: 10 windows x 8192 buckets = one thread per
// bucket fills the chip; wider windows only add empty buckets at the batch sizes a uint32 XOF length allows).
MsmPlan msm_plan(size_t n) {
  int lg = 0;
  while (((size_t)2 << lg) <= n) ++lg;                  // floor(log2 n)
  int c = lg - 3;
  c = c < 4 ? 4 : c > 13 ? 13 : c;
  MsmPlan p;
  p.c = c;
  p.W = (128 + c - 1) / c;
  p.K = 8;
  p.NB = (uint32_t)p.W << c;
  p.NQ = (1u << c) / (uint32_t)p.K;
  p.NCH = p.NB / (uint32_t)p.K;
  // threads per bucket: 8 to 16 points per thread.  Many short work items keep the rounds of a launch that does not fit
  // the chip at once short (a 2^20-point sum is 10 240 waves of 16 additions, not 1 280 waves of 128).
  const size_t mean = n >> c;
  p.S = 1;
  while (p.S < 16 && (size_t)p.S * 2 * 8 <= mean) p.S *= 2;
  return p;
}

template <class C>
size_t msm_aff_bytes(int group) { return group == BGLS_G1 ? sizeof(Aff<F1<C>>) : sizeof(Aff<F2<C>>); }

template <class C>
void msm_parse(hipStream_t st, int group, const uint8_t* pts, const uint8_t* w16, const uint8_t* signs, size_t n, const MsmPlan& p, void* aff,
               uint32_t* cnt, uint32_t* flags) {
  if (group == BGLS_G1) k_msm_parse<F1<C>, 2 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, w16, signs, n, p.c, p.W, (Aff<F1<C>>*)aff, cnt, flags);
  else k_msm_parse<F2<C>, 4 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, w16, signs, n, p.c, p.W, (Aff<F2<C>>*)aff, cnt, flags);
}

void msm_scan(hipStream_t st, uint32_t* cnt, uint32_t NB, uint32_t* start, uint32_t* meta) { k_msm_scan<<<1, 1024, 0, st>>>(cnt, NB, start, meta); }

template <class C>
void msm_scatter(hipStream_t st, int group, const void* aff, const uint8_t* w16, size_t n, const MsmPlan& p, uint32_t* cursor, uint32_t* list) {
  if (group == BGLS_G1) k_msm_scatter<F1<C>><<<nblk(n, 64), 64, 0, st>>>((const Aff<F1<C>>*)aff, w16, n, p.c, p.W, cursor, list);
  else k_msm_scatter<F2<C>><<<nblk(n, 64), 64, 0, st>>>((const Aff<F2<C>>*)aff, w16, n, p.c, p.W, cursor, list);
}

template <class F>
static void msm_buckets_f(hipStream_t st, const void* aff, const uint32_t* list, const uint32_t* start, const MsmPlan& p, void* buckets) {
  const uint32_t nthreads = p.NB * (uint32_t)p.S;
  if ((size_t)nthreads > 64 * 1024)                       // more than one wave per SIMD: the two-wave register budget
    k_msm_buckets<F, 2><<<nblk(nthreads, 64), 64, 0, st>>>((const Aff<F>*)aff, list, start, nthreads, p.S, (Jac<F>*)buckets);
  else
    k_msm_buckets<F, 1><<<nblk(nthreads, 64), 64, 0, st>>>((const Aff<F>*)aff, list, start, nthreads, p.S, (Jac<F>*)buckets);
}
size_t msm_tail_points(const MsmPlan& p) { return (size_t)p.NCH + p.NCH / 2 + 2 * (size_t)p.W + 4; }

// G1: chunk sums, one wave per window, one wave for the final doublings and the sum over windows
template <class F>
static void msm_tail_f(hipStream_t st, const void* folded, const MsmPlan& p, void* scratch, void** result) {
  Jac<F>* chunks = (Jac<F>*)scratch;
  Jac<F>* wins = chunks + p.NCH;
  Jac<F>* out = wins + p.W;
  k_msm_chunks<F><<<nblk(p.NCH, 64), 64, 0, st>>>((const Jac<F>*)folded, p.c, p.K, 1, p.NCH, chunks);
  k_msm_windows<F><<<p.W, 64, 0, st>>>(chunks, p.NQ, wins);
  k_msm_final<F><<<1, 64, 0, st>>>(wins, p.W, p.c, out);
  *result = out;
}
// G2: the chunk sums are added per window by a pairwise tree of wave-per-addition launches (NQ is a power of two: the
// halving keeps the windows apart), every window is doubled into place by its own wave, and the windows are added up
template <class C>
static void msm_tail_g2(hipStream_t st, const void* folded, const MsmPlan& p, void* scratch, void** result) {
  typedef F2<C> F;
  Jac<F>* a = (Jac<F>*)scratch;
  Jac<F>* b = a + p.NCH;
  k_msm_chunks<F><<<nblk(p.NCH, 64), 64, 0, st>>>((const Jac<F>*)folded, p.c, p.K, 1, p.NCH, a);
  size_t cnt = p.NCH;
  while (cnt > (size_t)p.W) {
    sum_coop<C>(st, a, cnt, b);
    cnt /= 2;
    std::swap(a, b);
  }
  k_msm_shift<C><<<p.W, 64, CoopF2<C>::WAVE_DW * 4, st>>>(a, p.c);
  while (cnt > 1) {
    sum_coop<C>(st, a, cnt, b);
    cnt = (cnt + 1) / 2;
    std::swap(a, b);
  }
  *result = a;
}
// partial bucket sums: NB * S Jacobian points, the S partials of a bucket next to each other
template <class C>
void msm_buckets(hipStream_t st, int group, const void* aff, const uint32_t* list, const uint32_t* start, const MsmPlan& p, void* parts) {
  if (group == BGLS_G1) msm_buckets_f<F1<C>>(st, aff, list, start, p, parts);
  else msm_buckets_f<F2<C>>(st, aff, list, start, p, parts);
}
// buckets (one Jacobian point each) -> the weighted sum
template <class C>
void msm_tail(hipStream_t st, int group, const void* buckets, const MsmPlan& p, void* scratch, void** result) {
  if (group == BGLS_G1) msm_tail_f<F1<C>>(st, buckets, p, scratch, result);
  else msm_tail_g2<C>(st, buckets, p, scratch, result);
}

template <class C>
size_t fb_table_bytes(int group) { return (size_t)FB_ROW * FB_WINDOWS * msm_aff_bytes<C>(group); }
template <class C>
void fb_build(hipStream_t st, int group, void* table) {
  const unsigned nb = nblk((size_t)FB_ROW * FB_WINDOWS, 64);
  if (group == BGLS_G1) k_fb_build<C, F1<C>><<<nb, 64, 0, st>>>((Aff<F1<C>>*)table);
  else k_fb_build<C, F2<C>><<<nb, 64, 0, st>>>((Aff<F2<C>>*)table);
}
template <class C>
void fb_scale(hipStream_t st, int group, const void* table, const uint8_t* scalars, size_t n, uint8_t* out) {
  if (group == BGLS_G1) k_fb_scale<F1<C>, 2 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>((const Aff<F1<C>>*)table, scalars, n, out);
  else k_fb_scale<F2<C>, 4 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>((const Aff<F2<C>>*)table, scalars, n, out);
}
template <class C>
void scale_g1_inplace(hipStream_t st, Aff<F1<C>>* pts, const uint8_t* w16, size_t n) {
  k_scale_g1_inplace<C><<<nblk(n, 64), 64, 0, st>>>(pts, w16, n);
}

#define BGLS_INST(C)                                                                                                                        \
  template size_t msm_aff_bytes<C>(int);                                                                                                    \
  template void msm_parse<C>(hipStream_t, int, const uint8_t*, const uint8_t*, const uint8_t*, size_t, const MsmPlan&, void*, uint32_t*,    \
                             uint32_t*);                                                                                                    \
  template void msm_scatter<C>(hipStream_t, int, const void*, const uint8_t*, size_t, const MsmPlan&, uint32_t*, uint32_t*);                \
  template void msm_buckets<C>(hipStream_t, int, const void*, const uint32_t*, const uint32_t*, const MsmPlan&, void*);                     \
  template void msm_tail<C>(hipStream_t, int, const void*, const MsmPlan&, void*, void**);                                                  \
  template size_t fb_table_bytes<C>(int);                                                                                                   \
  template void fb_build<C>(hipStream_t, int, void*);                                                                                       \
  template void fb_scale<C>(hipStream_t, int, const void*, const uint8_t*, size_t, uint8_t*);                                               \
  template void scale_g1_inplace<C>(hipStream_t, Aff<F1<C>>*, const uint8_t*, size_t);
BGLS_INST(BN254)
BGLS_INST(BLS381)
#undef BGLS_INST

}  // namespace kl
}  // namespace bgls
