// Development microbenchmark: one Fp inversion on a LONE wave -- fp_inv_euclid (binary Euclid, 32-bit limbs) against a^(p - 2) on the
// carry-free limbs (rx_pow.hpp) and fp_inv_ds (batched division steps: the library's fp_inv since round 4).  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibgls_amd/csrc -Iinclude tools/mb_inv.hip -o tools/mb_inv.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "dev_common.hpp"
#include "rx_pow.hpp"
using namespace bgls;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class C, int KIND>
__global__ void __launch_bounds__(64) k_inv(u32* io, int reps) {
  __shared__ i32 tab[4 * C::RX_NL * 64];
  Fp<C> a;
  for (int k = 0; k < C::L; ++k) a.v[k] = io[(KIND >= 3 ? 0 : threadIdx.x) * C::L + k];
  a.v[C::L - 1] &= 0x0FFFFFFFu;
  a = fp_to_mont<C>(a);
  for (int r = 0; r < reps; ++r) {
    if constexpr (KIND == 0) {
      a = fp_inv_euclid<C>(a);
    } else if constexpr (KIND == 2) {
      a = fp_inv_ds<C>(a);
    } else if constexpr (KIND == 3) {            // as the final exponentiation calls it: six lanes, the same value
      if (threadIdx.x < 6) a = fp_inv_ds<C>(a);
    } else if constexpr (KIND == 4) {
      if (threadIdx.x < 6) a = fp_inv_euclid<C>(a);
    } else {
      constexpr int N = C::RX_NL;
      const int lane = threadIdx.x & 63;
      auto ld = [&](int e, int i) { return tab[(e * N + i) * 64 + lane]; };
      auto st = [&](int e, int i, i32 v) { tab[(e * N + i) * 64 + lane] = v; };
      auto word = [&](int k) { return C::EXP_INV[k]; };
      const Sx<C, SX_T> rr = sx_pow_sw<C, 3, 32 * C::L>(ux_to_sx<C>(to_ux<C>(a)), word, ld, st);
      Ux<C> u;
      for (int i = 0; i < N; ++i) u.v[i] = (u32)rr.v[i];
      a = from_ux<C>(u);
    }
  }
  for (int k = 0; k < C::L; ++k) io[threadIdx.x * C::L + k] = a.v[k];
}
template <class C, int KIND>
static int run(const char* name) {
  u32* d; CHK(hipMalloc(&d, 64 * C::L * 4));
  u32 h[64 * 12];
  for (int i = 0; i < 64 * C::L; ++i) h[i] = 0x9e3779b9u * (i + 7) + 12345u;
  CHK(hipMemcpy(d, h, 64 * C::L * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  k_inv<C, KIND><<<1, 64>>>(d, 2); CHK(hipDeviceSynchronize());
  const int reps = 10;
  CHK(hipEventRecord(a)); k_inv<C, KIND><<<1, 64>>>(d, reps); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  u32 o[12]; CHK(hipMemcpy(o, d, C::L * 4, hipMemcpyDeviceToHost));
  printf("%-40s %.1f us per inversion (one wave)   [%08x]\n", name, ms * 1000.0 / reps, o[0]);
  return 0;
}
int main() {

  run<BN254, 0>("alt-bn128 fp_inv (binary Euclid)");
  run<BN254, 1>("alt-bn128 a^(p-2), carry-free");
  run<BN254, 2>("alt-bn128 division steps");
  run<BN254, 4>("alt-bn128 Euclid, 6 lanes same value");
  run<BN254, 3>("alt-bn128 division steps, 6 lanes same");
  run<BLS381, 0>("BLS12-381 fp_inv (binary Euclid)");
  run<BLS381, 1>("BLS12-381 a^(p-2), carry-free");
  run<BLS381, 2>("BLS12-381 division steps");
  run<BLS381, 4>("BLS12-381 Euclid, 6 lanes same value");
  run<BLS381, 3>("BLS12-381 division steps, 6 lanes same");
  return 0;
}
