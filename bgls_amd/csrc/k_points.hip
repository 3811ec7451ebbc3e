// Point kernels: parsing / serialisation, AggregatePoints, ScalePoints, compressed wire formats, weighted sums,
// batch key generation / signing, validity checks, and the multiplier-peak probe.
#include "dev_common.hpp"
#include "wire.hpp"
#include "launch.hpp"
#include "points_inl.hpp"
#include "jac_coop.hpp"
#include "../../include/bgls_hip.h"

using namespace bgls;

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

// ---- point sums ----
// PARSED = false: pts are wire-format bytes (parsed and checked here); true: resident Montgomery affine points of a
// key-set handle (validated when the handle was made).
template <class C, class F, int PT_BYTES, bool PARSED>
__global__ void __launch_bounds__(64, 2) k_sum_main(const uint8_t* pts, size_t n, Jac<F>* out, uint32_t* flags) {
  const size_t T = (size_t)gridDim.x * 64;
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  const bool aligned = (reinterpret_cast<uintptr_t>(pts) & 15) == 0;       // uniform
  Jac<F> acc = jac_inf<F>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = t; k < n; k += T) {
    Aff<F> p;
    if constexpr (PARSED) {
      p = reinterpret_cast<const Aff<F>*>(pts)[k];
    } else {
      bool ok = aligned ? aff_from_bytes16<F, C>(p, pts + k * PT_BYTES) : aff_from_bytes<F>(p, pts + k * PT_BYTES);
      ok = ok && aff_on_curve_inl<F>(p);
      bad = bad || !ok;
    }
    acc = jac_madd_inl<F>(acc, p);
  }
  if (bad) atomicOr(flags, FLAG_ENC);
  out[t] = acc;
}

// B sums in one launch (the key sets of KoskVerifyBatchMultiSignature, bgls/blsKosk.go:126-133): set b = points off[b] .. off[b+1],
// P partials per set (P a power of two, a multiple of 64: the pairwise / 64-to-1 levels above then never straddle two
// sets), block = 64 consecutive partials of one set.
template <class C, class F, int PT_BYTES>
__global__ void __launch_bounds__(64, 2) k_sumseg_main(const uint8_t* pts, const uint64_t* off, unsigned P, Jac<F>* out, uint32_t* flags) {
  const unsigned per = P / 64;
  const size_t b = blockIdx.x / per;
  const size_t t = (size_t)(blockIdx.x % per) * 64 + threadIdx.x;
  const size_t lo = off[b], hi = off[b + 1];
  const bool aligned = (reinterpret_cast<uintptr_t>(pts) & 15) == 0;       // uniform
  Jac<F> acc = jac_inf<F>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = lo + t; k < hi; k += P) {
    Aff<F> p;
    bool ok = aligned ? aff_from_bytes16<F, C>(p, pts + k * PT_BYTES) : aff_from_bytes<F>(p, pts + k * PT_BYTES);
    ok = ok && aff_on_curve_inl<F>(p);
    bad = bad || !ok;
    acc = jac_madd_inl<F>(acc, p);
  }
  if (bad) atomicOr(flags, FLAG_ENC);
  out[b * P + t] = acc;
}

// one partial per wave from up to 64 Jacobian partials per wave (the upper levels of the tree: few elements, latency-bound)
template <class F>
__global__ void __launch_bounds__(64) k_sum_wave(const Jac<F>* in, size_t n, Jac<F>* out) {      // lone waves: full register budget, no spills
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  Jac<F> acc = t < n ? in[t] : jac_inf<F>();
#pragma unroll 1
  for (int off = 32; off >= 1; off >>= 1) {
    const Jac<F> o = jac_shfl_down<F>(acc, off);
    acc = jac_add_inl<F>(acc, o);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// out[t] = in[2t] + in[2t+1]: one addition per thread.  The upper levels of the tree are latency-bound whatever their
// shape; halving with half as many threads per launch keeps their MACHINE time small (a wave-shuffle tree runs every
// level on all lanes), so the tail of one verification leaves the SIMDs to the main pass of the next.
// OCC: waves per SIMD the register budget is sized for (1 once the launch is down to a wave per SIMD or less: a lone
// wave cannot hide the latency of spilled registers).
template <class F, int OCC>
__global__ void __launch_bounds__(64, OCC) k_sum_pair(const Jac<F>* in, size_t n, Jac<F>* out) {
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  const size_t lo = 2 * t;
  if (lo >= n) return;
  Jac<F> acc = in[lo];
  if (lo + 1 < n) acc = jac_add_inl<F>(acc, in[lo + 1]);
  out[t] = acc;
}
// out[b] = in[2b] + in[2b+1], one WAVE per addition (jac_coop.hpp): the levels of a G2 tree with a few thousand additions
// or fewer, where a thread-per-addition launch is a set of lone waves at 45-50 us per addition
template <class C>
__global__ void __launch_bounds__(64) k_sum_coop(const Jac<F2<C>>* in, size_t n, Jac<F2<C>>* out) {
  const size_t lo = 2 * (size_t)blockIdx.x;
  Jac<F2<C>> acc = in[lo];
  if (lo + 1 < n) {
    const CoopF2<C> k(0);
    acc = coop_jac_add<C>(k, acc, in[lo + 1]);
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
// fan-in R, out-of-line additions: used by the weighted sums (few elements per thread)
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
  Jac<F> r = jac_mul_w4<F>(p, k, top + 1);
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
    good = good && g2_in_subgroup<C>(p);                   // UnmarshalG2 rejects twist points outside G2 (upstream G2.Unmarshal)
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

// ---- weighted key sums (hashed aggregation exponents, bgls/blsHAE.go; multiplicities, bgls/blsKosk.go:137-150) ----
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
    acc = jac_add<F>(acc, jac_mul_w4<F>(p, k, top + 1));
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
  aff_to_bytes<F>(out + i * PT_BYTES, jac_to_aff<F>(jac_mul_w4<F>(p, k, top + 1)));
}

// validity of n points: canonical coordinates, on the curve and in the order-r subgroup (G2 on both curves, G1 on
// BLS12-381: what the reference checks when a Point is constructed, see g2_in_subgroup / g1_in_subgroup).  ok == nullptr: failures only set flags; else ok[i] = 1 / 0.
template <class C, class F, int PT_BYTES>
__global__ void __launch_bounds__(64) k_check(const uint8_t* pts, size_t n, uint32_t* flags, uint8_t* ok_out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F> p;
  bool ok = aff_from_bytes<F>(p, pts + i * PT_BYTES);
  ok = ok && aff_on_curve<F>(p);
  uint32_t f = ok ? 0u : FLAG_ENC;
  if constexpr (F::NFP == 2) {
    if (ok && !g2_in_subgroup<C>(p)) { ok = false; f = FLAG_SUBGROUP; }
  } else {
    if (ok && !g1_in_subgroup<C>(p)) { ok = false; f = FLAG_SUBGROUP; }     // BLS12-381 only: alt-bn128's G1 has cofactor 1
  }
  if (f) atomicOr(flags, f);
  if (ok_out) ok_out[i] = ok ? 1 : 0;
}

// key-set upload: wire bytes -> resident Montgomery affine points; invalid keys set FLAG_ENC / FLAG_SUBGROUP
template <class C>
__global__ void __launch_bounds__(64) k_g2_parse(const uint8_t* in, size_t n, int check_subgroup, Aff<F2<C>>* out, uint32_t* flags) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff<F2<C>> p;
  bool ok = g2_from_bytes<C>(p, in + i * 4 * C::FP_BYTES);
  ok = ok && aff_on_curve<F2<C>>(p);
  if (!ok) atomicOr(flags, FLAG_ENC);
  else if (check_subgroup && !g2_in_subgroup<C>(p)) atomicOr(flags, FLAG_SUBGROUP);
  out[i] = p;
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

// ======================================================================= launchers
namespace bgls {
namespace kl {

template <class C>
void g1_to_bytes(hipStream_t st, const Aff<F1<C>>* in, size_t n, uint8_t* out) {
  k_g1_to_bytes<C><<<nblk(n, 64), 64, 0, st>>>(in, n, out);
}
template <class C>
void g1_parse(hipStream_t st, const uint8_t* in, size_t n, int negate, Aff<F1<C>>* out, uint32_t* flags) {
  k_g1_parse<C><<<nblk(n, 64), 64, 0, st>>>(in, n, negate, out, flags);
}

template <class C>
void sum_main(hipStream_t st, int group, bool parsed, const uint8_t* pts, size_t n, unsigned waves, void* out, uint32_t* flags) {
  if (group == BGLS_G1) {
    if (parsed) k_sum_main<C, F1<C>, 2 * C::FP_BYTES, true><<<waves, 64, 0, st>>>(pts, n, (Jac<F1<C>>*)out, flags);
    else k_sum_main<C, F1<C>, 2 * C::FP_BYTES, false><<<waves, 64, 0, st>>>(pts, n, (Jac<F1<C>>*)out, flags);
  } else {
    if (parsed) k_sum_main<C, F2<C>, 4 * C::FP_BYTES, true><<<waves, 64, 0, st>>>(pts, n, (Jac<F2<C>>*)out, flags);
    else k_sum_main<C, F2<C>, 4 * C::FP_BYTES, false><<<waves, 64, 0, st>>>(pts, n, (Jac<F2<C>>*)out, flags);
  }
}
template <class C>
void sumseg_main(hipStream_t st, int group, const uint8_t* pts, const uint64_t* off, size_t nsets, unsigned P, void* out, uint32_t* flags) {
  const unsigned blocks = (unsigned)(nsets * (P / 64));
  if (group == BGLS_G1) k_sumseg_main<C, F1<C>, 2 * C::FP_BYTES><<<blocks, 64, 0, st>>>(pts, off, P, (Jac<F1<C>>*)out, flags);
  else k_sumseg_main<C, F2<C>, 4 * C::FP_BYTES><<<blocks, 64, 0, st>>>(pts, off, P, (Jac<F2<C>>*)out, flags);
}
template <class C>
void sum_pair(hipStream_t st, int group, const void* in, size_t n, void* out) {
  const size_t nout = (n + 1) / 2;
  const bool lone = nout <= 64 * 1024;                    // at most one wave per SIMD
  if (group == BGLS_G1) {
    if (lone) k_sum_pair<F1<C>, 1><<<nblk(nout, 64), 64, 0, st>>>((const Jac<F1<C>>*)in, n, (Jac<F1<C>>*)out);
    else k_sum_pair<F1<C>, 2><<<nblk(nout, 64), 64, 0, st>>>((const Jac<F1<C>>*)in, n, (Jac<F1<C>>*)out);
  } else {
    if (lone) k_sum_pair<F2<C>, 1><<<nblk(nout, 64), 64, 0, st>>>((const Jac<F2<C>>*)in, n, (Jac<F2<C>>*)out);
    else k_sum_pair<F2<C>, 2><<<nblk(nout, 64), 64, 0, st>>>((const Jac<F2<C>>*)in, n, (Jac<F2<C>>*)out);
  }
}
template <class C>
void sum_coop(hipStream_t st, const void* in, size_t n, void* out) {
  k_sum_coop<C><<<(unsigned)((n + 1) / 2), 64, CoopF2<C>::WAVE_DW * 4, st>>>((const Jac<F2<C>>*)in, n, (Jac<F2<C>>*)out);
}
template <class C>
void sum_wave(hipStream_t st, int group, const void* in, size_t n, void* out) {
  if (group == BGLS_G1) k_sum_wave<F1<C>><<<nblk(n, 64), 64, 0, st>>>((const Jac<F1<C>>*)in, n, (Jac<F1<C>>*)out);
  else k_sum_wave<F2<C>><<<nblk(n, 64), 64, 0, st>>>((const Jac<F2<C>>*)in, n, (Jac<F2<C>>*)out);
}
template <class C>
void sum_next(hipStream_t st, int group, const void* in, size_t n, int R, void* out) {
  const size_t nout = (n + R - 1) / R;
  if (group == BGLS_G1) k_sum_next<F1<C>><<<nblk(nout, 64), 64, 0, st>>>((const Jac<F1<C>>*)in, n, R, (Jac<F1<C>>*)out);
  else k_sum_next<F2<C>><<<nblk(nout, 64), 64, 0, st>>>((const Jac<F2<C>>*)in, n, R, (Jac<F2<C>>*)out);
}
template <class C>
void jac_to_bytes(hipStream_t st, int group, const void* in, size_t n, uint8_t* out) {
  if (group == BGLS_G1) k_jac_to_bytes<F1<C>><<<nblk(n, 64), 64, 0, st>>>((const Jac<F1<C>>*)in, n, out, 2 * C::FP_BYTES);
  else k_jac_to_bytes<F2<C>><<<nblk(n, 64), 64, 0, st>>>((const Jac<F2<C>>*)in, n, out, 4 * C::FP_BYTES);
}
template <class C>
void wsum_first(hipStream_t st, int group, const uint8_t* pts, const uint8_t* w16, const uint8_t* signs, size_t n, void* out,
                uint32_t* flags) {
  if (group == BGLS_G1) k_wsum_first<F1<C>, 2 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, w16, signs, n, 1, (Jac<F1<C>>*)out, flags);
  else k_wsum_first<F2<C>, 4 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, w16, signs, n, 1, (Jac<F2<C>>*)out, flags);
}
template <class C>
void scale(hipStream_t st, int group, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n, uint8_t* out,
           uint32_t* flags, int sbytes) {
  if (group == BGLS_G1) k_scale<F1<C>, 2 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, scalars, signs, n, out, flags, sbytes);
  else k_scale<F2<C>, 4 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, scalars, signs, n, out, flags, sbytes);
}
template <class C>
void scale_aff(hipStream_t st, int group, const Aff<F1<C>>* g1_pts, const uint8_t* scalars, size_t n, uint8_t* out) {
  if (group == BGLS_G1) k_scale_aff<C, F1<C>, 2 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(g1_pts, scalars, n, out);
  else k_scale_aff<C, F2<C>, 4 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(nullptr, scalars, n, out);
}
template <class C>
void check(hipStream_t st, int group, const uint8_t* pts, size_t n, uint32_t* flags, uint8_t* ok) {
  if (group == BGLS_G1) k_check<C, F1<C>, 2 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, n, flags, ok);
  else k_check<C, F2<C>, 4 * C::FP_BYTES><<<nblk(n, 64), 64, 0, st>>>(pts, n, flags, ok);
}
template <class C>
void g2_parse(hipStream_t st, const uint8_t* in, size_t n, int check_subgroup, void* out, uint32_t* flags) {
  k_g2_parse<C><<<nblk(n, 64), 64, 0, st>>>(in, n, check_subgroup, (Aff<F2<C>>*)out, flags);
}
template <class C>
size_t g2_parsed_bytes() { return sizeof(Aff<F2<C>>); }
template <class C>
void generator(hipStream_t st, int group, uint8_t* out) {
  k_generator<C><<<1, 64, 0, st>>>(group, out);
}

void compress_bn(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint32_t* flags) {
  if (group == BGLS_G1) k_compress_bn<BGLS_G1><<<nblk(n, 64), 64, 0, st>>>(in, n, out, flags);
  else k_compress_bn<BGLS_G2><<<nblk(n, 64), 64, 0, st>>>(in, n, out, flags);
}
void decompress_bn(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
  if (group == BGLS_G1) k_decompress_bn<BGLS_G1><<<nblk(n, 64), 64, 0, st>>>(in, n, out, ok);
  else k_decompress_bn<BGLS_G2><<<nblk(n, 64), 64, 0, st>>>(in, n, out, ok);
}
void mad_probe(hipStream_t st, unsigned blocks, unsigned threads, uint32_t seed, int iters, uint64_t* sink) {
  k_mad_probe<<<blocks, threads, 0, st>>>(seed, iters, sink);
}

#define BGLS_INST(C)                                                                                                             \
  template void g1_to_bytes<C>(hipStream_t, const Aff<F1<C>>*, size_t, uint8_t*);                                                \
  template void g1_parse<C>(hipStream_t, const uint8_t*, size_t, int, Aff<F1<C>>*, uint32_t*);                                   \
  template void sum_main<C>(hipStream_t, int, bool, const uint8_t*, size_t, unsigned, void*, uint32_t*);                         \
  template void sumseg_main<C>(hipStream_t, int, const uint8_t*, const uint64_t*, size_t, unsigned, void*, uint32_t*);                     \
  template void sum_wave<C>(hipStream_t, int, const void*, size_t, void*);                                                       \
  template void sum_pair<C>(hipStream_t, int, const void*, size_t, void*);                                                       \
  template void sum_coop<C>(hipStream_t, const void*, size_t, void*);                                                            \
  template void sum_next<C>(hipStream_t, int, const void*, size_t, int, void*);                                                  \
  template void jac_to_bytes<C>(hipStream_t, int, const void*, size_t, uint8_t*);                                                \
  template void wsum_first<C>(hipStream_t, int, const uint8_t*, const uint8_t*, const uint8_t*, size_t, void*, uint32_t*);       \
  template void scale<C>(hipStream_t, int, const uint8_t*, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint32_t*, int);    \
  template void scale_aff<C>(hipStream_t, int, const Aff<F1<C>>*, const uint8_t*, size_t, uint8_t*);                             \
  template void check<C>(hipStream_t, int, const uint8_t*, size_t, uint32_t*, uint8_t*);                                                  \
  template void g2_parse<C>(hipStream_t, const uint8_t*, size_t, int, void*, uint32_t*);                                         \
  template size_t g2_parsed_bytes<C>();                                                                                          \
  template void generator<C>(hipStream_t, int, uint8_t*);
BGLS_INST(BN254)
BGLS_INST(BLS381)
#undef BGLS_INST

}  // namespace kl
}  // namespace bgls
