// Development probe (not part of the library; verdict r5 item 5): does the int8 MFMA pipe run BESIDE the integer multiplier?
// The one dense contraction on this path is q x p of a Montgomery reduction (a batch of quotients times the constant modulus: byte-split q,
// a Toeplitz matrix of p's bytes, v_mfma_i32_16x16x64_i8).  Before building the layouts, price the idea:
//   (a) issue: a stream of v_mad_u64_u32 with one MFMA per 16 of them -- the ratio a reduction needs (81 multiplier instructions <-> 20 MFMAs per
//       wave of 64 elements, 5 column tiles x 4 batches of 16, next to the ~300 vector instructions the kernels issue per reduction) -- against the
//       same stream without the MFMAs and the MFMAs alone;
//   (b) the recombination is counted, not measured: it is plain vector work.
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb_mfma.hip -o tools/mb_mfma.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void __launch_bounds__(256) k_mix(int iters, uint64_t* sink) {
  uint64_t c[8];
  uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u;
  for (int j = 0; j < 8; ++j) c[j] = (uint64_t)j * 0x9e3779b97f4a7c15ull + a;
  v4i acc[4], fa = {(int)a, (int)b, (int)(a ^ b), (int)(a + b)}, fb = {(int)b, (int)a, (int)(a * 3), (int)(b * 5)};
  for (int j = 0; j < 4; ++j) acc[j] = (v4i){j, j + 1, j + 2, j + 3};
  const unsigned long long tc0 = __builtin_readcyclecounter(), tr0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (KIND != 2) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b) : "vcc");
      }
      if constexpr (KIND != 0) {
        asm volatile("" : "+v"(fa));                                   // opaque: the MFMAs stay in the loop
        acc[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, acc[r], 0, 0, 0);
      }
    }
  }
  const unsigned long long tc1 = __builtin_readcyclecounter(), tr1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { sink[1] = tc1 - tc0; sink[2] = tr1 - tr0; }
  uint64_t x = 0;
  for (int j = 0; j < 8; ++j) x ^= c[j];
  for (int j = 0; j < 4; ++j) x ^= (uint64_t)(uint32_t)(acc[j].x ^ acc[j].y ^ acc[j].z ^ acc[j].w);
  if (x == 0x1234567ull) sink[0] = x;
}

template <int KIND>
static int run(const char* what, int wps, uint64_t* sink) {
  const int iters = 4000;
  const unsigned blocks = 256 * wps;
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  k_mix<KIND><<<blocks, 256>>>(10, sink);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  k_mix<KIND><<<blocks, 256>>>(iters, sink);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  uint64_t h[3];
  CHK(hipMemcpy(h, sink, 24, hipMemcpyDeviceToHost));
  const double mhz = h[2] ? (double)h[1] / (double)h[2] * 100.0 : 0;
  const double ns_iter = (double)ms * 1e6 / ((double)iters * wps);     // ns per loop iteration (64 multiplier instructions and / or 4 MFMAs) per SIMD and wave slot
  printf("%d waves/SIMD  %-46s %8.2f ns per iteration and SIMD = %7.1f clocks at %.0f MHz\n", wps, what, ns_iter, ns_iter * mhz / 1000.0, mhz);
  return 0;
}

int main() {
  uint64_t* sink;
  CHK(hipMalloc(&sink, 32));
  for (int wps = 2; wps <= 4; ++wps) {
    if (run<0>("64 v_mad_u64_u32", wps, sink)) return 1;
    if (run<2>("4 v_mfma_i32_16x16x64_i8", wps, sink)) return 1;
    if (run<1>("64 v_mad_u64_u32 + 4 v_mfma_i32_16x16x64_i8", wps, sink)) return 1;
  }
  return 0;
}
