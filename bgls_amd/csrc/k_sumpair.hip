// G2 key sums (the main pass of AggregatePoints, curves/curve.go:73-121) on LANE PAIRS in the carry-free form: rx_jacpair.hpp.
//   k_sumpair_main      pair t adds keys t, t + T, t + 2T, ... (T = lane pairs in the launch) into its own Jacobian sum
//   k_sumpairseg_main   the same for nsets key sets in one launch (KoskVerifyBatchMultiSignature, bgls/blsKosk.go:126-133)
// The 32 sums of a block (one wave) are then added in the block, five levels through LDS (general Jacobian additions on the same
// lane pairs), so that a block leaves ONE partial: 3072 for 2^20 keys instead of 98 304, and the tree above them (k_sum_coop:
// one wave per addition) starts where it has few enough additions to be worth a wave each.  Partials leave in the library's
// 32-bit Montgomery Jacobian form (the even lane writes the real parts, the odd lane the imaginary parts).  Own translation unit: everything is
// expanded in place for this kernel's register budget (three waves per SIMD).
#include "dev_common.hpp"
#include "coop.hpp"
#include "rx_jacpair.hpp"
#include "launch.hpp"

namespace bgls {

template <class C, bool PARSED>
__device__ __forceinline__ bool sumpair_fetch(AffP<C>& q, const uint8_t* pts, size_t k, bool odd) {
  if constexpr (PARSED) {
    q = affp_from_mont<C>(reinterpret_cast<const Aff<F2<C>>*>(pts)[k], odd);
    return true;
  } else {
    const bool ok = affp_from_bytes<C>(q, pts + k * 4 * C::FP_BYTES, odd);
    return affp_on_curve<C>(q, odd) && ok;
  }
}

template <class C>
__device__ __forceinline__ void sumpair_store(Jac<F2<C>>* out, const JacP<C>& acc, bool odd) {
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out) + (odd ? 1 : 0);      // Jac = X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1
  if (acc.inf) {                                                  // jac_inf: (1, 1, 0)
    const Fp<C> one = odd ? fp_zero<C>() : fp_one<C>();
    o[0] = one;
    o[2] = one;
    o[4] = fp_zero<C>();
    return;
  }
  o[0] = sxp_to_mont<C>(acc.X);
  o[2] = sxp_to_mont<C>(acc.Y);
  o[4] = sxp_to_mont<C>(acc.Z);
}

// the block's 32 per-pair sums -> one (in pair 0).  LDS: limb-major columns of the 64 lanes (conflict-free), 3 NL dwords per lane.
template <class C>
__device__ __noinline__ JacP<C> sumpair_block_tree(JacP<C> acc, bool odd) {
  constexpr int N = C::RX_NL;
  __shared__ i32 cols[3 * N * 64];
  __shared__ int infs[64];
  const int lane = threadIdx.x & 63, pi = lane >> 1;
  auto put = [&]() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      cols[(0 * N + i) * 64 + lane] = acc.X.v[i];
      cols[(1 * N + i) * 64 + lane] = acc.Y.v[i];
      cols[(2 * N + i) * 64 + lane] = acc.Z.v[i];
    }
    infs[lane] = acc.inf ? 1 : 0;
  };
  put();
#pragma unroll 1
  for (int s = 16; s >= 1; s >>= 1) {
    wave_sync();
    if (pi < s) {
      const int src = lane + 2 * s;
      JacP<C> o;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        o.X.v[i] = cols[(0 * N + i) * 64 + src];
        o.Y.v[i] = cols[(1 * N + i) * 64 + src];
        o.Z.v[i] = cols[(2 * N + i) * 64 + src];
      }
      o.inf = infs[src] != 0;
      acc = jacp_add<C>(acc, o, odd);
      put();                       // slots below s are read again only after the next barrier
    }
  }
  return acc;
}

template <class C, bool PARSED>
__global__ void __launch_bounds__(64, 3) k_sumpair_main(const uint8_t* pts, size_t n, Jac<F2<C>>* out, uint32_t* flags) {
  const size_t T = (size_t)gridDim.x * 32;
  const size_t t = (size_t)blockIdx.x * 32 + (threadIdx.x >> 1);
  const bool odd = threadIdx.x & 1;
  JacP<C> acc = jacp_inf<C>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = t; k < n; k += T) {
    AffP<C> q;
    bad = !sumpair_fetch<C, PARSED>(q, pts, k, odd) || bad;
    acc = jacp_madd<C>(acc, q, odd);
  }
  if (bad && !odd) atomicOr(flags, FLAG_ENC);
  acc = sumpair_block_tree<C>(acc, odd);
  if (threadIdx.x < 2) sumpair_store<C>(out + blockIdx.x, acc, odd);
}

template <class C>
__global__ void __launch_bounds__(64, 3) k_sumpairseg_main(const uint8_t* pts, const uint64_t* off, unsigned P, Jac<F2<C>>* out, uint32_t* flags) {
  const unsigned per = P / 32;                 // blocks per set: P partials of one lane pair each
  const size_t b = blockIdx.x / per;
  const size_t t = (size_t)(blockIdx.x % per) * 32 + (threadIdx.x >> 1);
  const bool odd = threadIdx.x & 1;
  const size_t lo = off[b], hi = off[b + 1];
  JacP<C> acc = jacp_inf<C>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = lo + t; k < hi; k += P) {
    AffP<C> q;
    bad = !sumpair_fetch<C, false>(q, pts, k, odd) || bad;
    acc = jacp_madd<C>(acc, q, odd);
  }
  if (bad && !odd) atomicOr(flags, FLAG_ENC);
  acc = sumpair_block_tree<C>(acc, odd);
  if (threadIdx.x < 2) sumpair_store<C>(out + blockIdx.x, acc, odd);      // partials of set b: blocks b * per .. (b + 1) * per - 1
}

namespace kl {

// `pairs` = lane pairs in the launch (a multiple of 32); pairs / 32 Jacobian partial sums are written, one per block
template <class C>
void sumpair_main(hipStream_t st, bool parsed, const uint8_t* pts, size_t n, unsigned pairs, void* out, uint32_t* flags) {
  if (parsed) k_sumpair_main<C, true><<<pairs / 32, 64, 0, st>>>(pts, n, (Jac<F2<C>>*)out, flags);
  else k_sumpair_main<C, false><<<pairs / 32, 64, 0, st>>>(pts, n, (Jac<F2<C>>*)out, flags);
}
template <class C>
void sumpairseg_main(hipStream_t st, const uint8_t* pts, const uint64_t* off, size_t nsets, unsigned P, void* out, uint32_t* flags) {
  k_sumpairseg_main<C><<<(unsigned)(nsets * (P / 32)), 64, 0, st>>>(pts, off, P, (Jac<F2<C>>*)out, flags);
}
template void sumpair_main<BN254>(hipStream_t, bool, const uint8_t*, size_t, unsigned, void*, uint32_t*);
template void sumpair_main<BLS381>(hipStream_t, bool, const uint8_t*, size_t, unsigned, void*, uint32_t*);
template void sumpairseg_main<BN254>(hipStream_t, const uint8_t*, const uint64_t*, size_t, unsigned, void*, uint32_t*);
template void sumpairseg_main<BLS381>(hipStream_t, const uint8_t*, const uint64_t*, size_t, unsigned, void*, uint32_t*);

}  // namespace kl
}  // namespace bgls
