// Development micro-benchmark (not part of the library): Fp2 multiplication on alt-bn128 in two number
// representations.
//   A: the library's form (fp.hpp): 8 x 32-bit limbs, operand-scanning products with carry chains, Karatsuba
//      in double width, two Montgomery reductions.
//   B: 10 x 28-bit limbs, column accumulators in 64 bits (acc = a*b + acc, no carries between products),
//      schoolbook over Fp2 with the subtraction folded in as a "fat" negation, radix-2^28 Montgomery reduction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb_radix.hip -o /tmp/mb_radix && /tmp/mb_radix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../bgls_amd/csrc/tower.hpp"
#include "mb_radix_consts.h"
using namespace bgls;

struct F28 { uint32_t v[10]; };
struct F28x2 { F28 c0, c1; };
static constexpr uint32_t M28 = (1u << 28) - 1;

__device__ __forceinline__ void acc_prod(uint64_t (&c)[20], const F28& a, const F28& b) {
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int j = 0; j < 10; ++j) c[i + j] = (uint64_t)a.v[i] * b.v[j] + c[i + j];
}
__device__ __forceinline__ F28 redc28(uint64_t (&c)[20]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t m = ((uint32_t)c[i] * NP28) & M28;
#pragma unroll
    for (int j = 0; j < 10; ++j) c[i + j] = (uint64_t)m * P28[j] + c[i + j];
    c[i + 1] += c[i] >> 28;
  }
  F28 r;
#pragma unroll
  for (int k = 10; k < 19; ++k) {
    r.v[k - 10] = (uint32_t)c[k] & M28;
    c[k + 1] += c[k] >> 28;
  }
  r.v[9] = (uint32_t)c[19];
  return r;
}
__device__ __forceinline__ F28x2 f2mul28(const F28x2& a, const F28x2& b) {
  F28 nb1;
#pragma unroll
  for (int i = 0; i < 10; ++i) nb1.v[i] = FAT28[i] - b.c1.v[i];
  F28x2 r;
  {
    uint64_t c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
    acc_prod(c, a.c0, b.c0);
    acc_prod(c, a.c1, nb1);
    r.c0 = redc28(c);
  }
  {
    uint64_t c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
    acc_prod(c, a.c0, b.c1);
    acc_prod(c, a.c1, b.c0);
    r.c1 = redc28(c);
  }
  return r;
}

template <int CH>
__global__ void __launch_bounds__(64) mb_a(Fp2<BN254>* io, int iters, size_t n) {
  const size_t t = blockIdx.x * 64 + threadIdx.x;
  Fp2<BN254> x[CH], y = io[n + t];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = io[(t + c * 7) % n];
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = f2_mul_inl<BN254>(x[c], y);
  Fp2<BN254> s = x[0];
#pragma unroll
  for (int c = 1; c < CH; ++c) s = f2_add<BN254>(s, x[c]);
  io[t] = s;
}
template <int CH>
__global__ void __launch_bounds__(64) mb_b(F28x2* io, int iters, size_t n) {
  const size_t t = blockIdx.x * 64 + threadIdx.x;
  F28x2 x[CH], y = io[n + t];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = io[(t + c * 7) % n];
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = f2mul28(x[c], y);
  F28x2 s = x[0];
#pragma unroll
  for (int c = 1; c < CH; ++c)
#pragma unroll
    for (int k = 0; k < 10; ++k) { s.c0.v[k] += x[c].c0.v[k]; s.c1.v[k] += x[c].c1.v[k]; }
  io[t] = s;
}

template <class F> float run(F launch, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < reps; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
  const size_t n = 1 << 20;
  // ---- correctness sample for B: one product, printed for an offline check (x*y*2^-280 mod p per the schoolbook formula)
  {
    F28x2* d; hipMalloc(&d, 2 * n * sizeof(F28x2));
    F28x2* h = (F28x2*)malloc(2 * n * sizeof(F28x2));
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < 2 * n; ++i)
      for (int k = 0; k < 10; ++k) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].c0.v[k] = (uint32_t)s & (k == 9 ? 1u : M28);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].c1.v[k] = (uint32_t)s & (k == 9 ? 1u : M28);
      }
    hipMemcpy(d, h, 2 * n * sizeof(F28x2), hipMemcpyHostToDevice);
    F28x2 x0 = h[0], y0 = h[n];
    mb_b<1><<<1, 64>>>(d, 1, n);
    F28x2 r; hipMemcpy(&r, d, sizeof r, hipMemcpyDeviceToHost);
    auto pr = [](const char* nm, const F28& f) { printf("%s", nm); for (int k = 0; k < 10; ++k) printf(" %x", f.v[k]); printf("\n"); };
    pr("X0", x0.c0); pr("X1", x0.c1); pr("Y0", y0.c0); pr("Y1", y0.c1); pr("R0", r.c0); pr("R1", r.c1);
    hipMemcpy(d, h, 2 * n * sizeof(F28x2), hipMemcpyHostToDevice);
    const int iters = 256;
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
      float t1 = run([&] { mb_b<1><<<blocks, 64>>>(d, iters, n); }, 3);
      float t2 = run([&] { mb_b<2><<<blocks, 64>>>(d, iters, n); }, 3);
      printf("B radix-2^28   blocks=%5d  1 chain: %8.3f ms = %6.2f G f2mul/s | 2 chains: %8.3f ms = %6.2f G f2mul/s\n", blocks, t1,
             (double)blocks * 64 * iters / t1 / 1e6, t2, (double)blocks * 64 * iters * 2 / t2 / 1e6);
    }
    hipFree(d); free(h);
  }
  {
    Fp2<BN254>* d; hipMalloc(&d, 2 * n * sizeof(Fp2<BN254>)); hipMemset(d, 1, 2 * n * sizeof(Fp2<BN254>));
    const int iters = 256;
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
      float t1 = run([&] { mb_a<1><<<blocks, 64>>>(d, iters, n); }, 3);
      float t2 = run([&] { mb_a<2><<<blocks, 64>>>(d, iters, n); }, 3);
      printf("A 32-bit limbs blocks=%5d  1 chain: %8.3f ms = %6.2f G f2mul/s | 2 chains: %8.3f ms = %6.2f G f2mul/s\n", blocks, t1,
             (double)blocks * 64 * iters / t1 / 1e6, t2, (double)blocks * 64 * iters * 2 / t2 / 1e6);
    }
    hipFree(d);
  }
  return 0;
}
