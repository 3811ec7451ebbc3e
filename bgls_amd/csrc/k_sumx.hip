// G2 key sums (the main pass of AggregatePoints, curves/curve.go:73-121) on carry-free 28-bit limbs: rx_jac.hpp.
//   k_sumx_main      thread t adds keys t, t + T, t + 2T, ... (T = threads in the launch) into its own Jacobian partial
//   k_sumxseg_main   the same for nsets key sets in one launch (KoskVerifyBatchMultiSignature, bgls/blsKosk.go:126-133)
// Partials leave in the library's 32-bit Montgomery Jacobian form: the tree above them (k_sum_pair / k_sum_coop / k_sum_wave)
// is unchanged.  Own translation unit: its helpers are compiled for this kernel's register budget.
#include "dev_common.hpp"
#include "rx_jac.hpp"
#include "launch.hpp"

namespace bgls {

template <class C, bool PARSED>
__device__ __forceinline__ bool sumx_fetch(AffX<C>& q, const uint8_t* pts, size_t k) {
  if constexpr (PARSED) {
    q = affx_from_mont<C>(reinterpret_cast<const Aff<F2<C>>*>(pts)[k]);
    return true;
  } else {
    const bool ok = affx_from_bytes<C>(q, pts + k * 4 * C::FP_BYTES);
    return ok && affx_on_curve<C>(q);
  }
}

template <class C, bool PARSED>
__global__ void __launch_bounds__(64, 2) k_sumx_main(const uint8_t* pts, size_t n, Jac<F2<C>>* out, uint32_t* flags) {
  const size_t T = (size_t)gridDim.x * 64;
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  JacX<C> acc = jacx_inf<C>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = t; k < n; k += T) {
    AffX<C> q;
    bad = !sumx_fetch<C, PARSED>(q, pts, k) || bad;
    acc = jacx_madd<C>(acc, q);
  }
  if (bad) atomicOr(flags, FLAG_ENC);
  out[t] = jacx_to_mont<C>(acc);
}

template <class C>
__global__ void __launch_bounds__(64, 2) k_sumxseg_main(const uint8_t* pts, const uint64_t* off, unsigned P, Jac<F2<C>>* out, uint32_t* flags) {
  const unsigned per = P / 64;
  const size_t b = blockIdx.x / per;
  const size_t t = (size_t)(blockIdx.x % per) * 64 + threadIdx.x;
  const size_t lo = off[b], hi = off[b + 1];
  JacX<C> acc = jacx_inf<C>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = lo + t; k < hi; k += P) {
    AffX<C> q;
    bad = !sumx_fetch<C, false>(q, pts, k) || bad;
    acc = jacx_madd<C>(acc, q);
  }
  if (bad) atomicOr(flags, FLAG_ENC);
  out[b * P + t] = jacx_to_mont<C>(acc);
}

namespace kl {

template <class C>
void sumx_main(hipStream_t st, bool parsed, const uint8_t* pts, size_t n, unsigned waves, void* out, uint32_t* flags) {
  if (parsed) k_sumx_main<C, true><<<waves, 64, 0, st>>>(pts, n, (Jac<F2<C>>*)out, flags);
  else k_sumx_main<C, false><<<waves, 64, 0, st>>>(pts, n, (Jac<F2<C>>*)out, flags);
}
template <class C>
void sumxseg_main(hipStream_t st, const uint8_t* pts, const uint64_t* off, size_t nsets, unsigned P, void* out, uint32_t* flags) {
  k_sumxseg_main<C><<<(unsigned)(nsets * (P / 64)), 64, 0, st>>>(pts, off, P, (Jac<F2<C>>*)out, flags);
}
template void sumx_main<BN254>(hipStream_t, bool, const uint8_t*, size_t, unsigned, void*, uint32_t*);
template void sumx_main<BLS381>(hipStream_t, bool, const uint8_t*, size_t, unsigned, void*, uint32_t*);
template void sumxseg_main<BN254>(hipStream_t, const uint8_t*, const uint64_t*, size_t, unsigned, void*, uint32_t*);
template void sumxseg_main<BLS381>(hipStream_t, const uint8_t*, const uint64_t*, size_t, unsigned, void*, uint32_t*);

}  // namespace kl
}  // namespace bgls
