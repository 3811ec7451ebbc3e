// Micro-benchmarks of the Miller-loop building blocks (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o /tmp/microbench && /tmp/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../bgls_amd/csrc/coop_r28.hpp"
using namespace bgls;

template <class C>
__global__ void __launch_bounds__(64) mb_coop(Fp2<C>* io, int iters) {       // sqr + 6 line folds per iteration
  typedef Coop<C> K;
  int lane = threadIdx.x; bool live = lane < 60; int g = live ? lane / 6 : 9, j = live ? lane % 6 : lane - 60; int gb = g * K::GROUP_DW;
  Fp2<C> fj = io[blockIdx.x * 64 + lane];
  coop_publish<C>(gb + K::RB, j, fj, live);
  if (live) for (int t = 0; t < 3; ++t) lds_st<C>(reg_rl<C>(gb, K::RL), j * 3 + t, fj);
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    fj = coop_sqr<C>(gb, j);
    coop_publish<C>(gb + K::RB, j, fj, live);
    fj = coop_apply_lines<C>(gb, j, live);
  }
  io[blockIdx.x * 64 + lane] = fj;
}
template <class C, int MODE>
__global__ void __launch_bounds__(64) mb_step(Fp2<C>* io, int iters) {       // doubling step per lane
  G2Proj<C> T = {io[threadIdx.x], io[threadIdx.x + 64], io[threadIdx.x + 128]};
  Fp2<C> acc = f2_zero<C>();
  for (int i = 0; i < iters; ++i) {
    LineCoeffs<C> l = MODE == 0 ? dbl_step<C>(T) : dbl_step_inl<C>(T);
    acc = f2_add<C>(acc, f2_add<C>(l.c0, f2_add<C>(l.c1, l.c2)));
  }
  io[blockIdx.x * 64 + threadIdx.x] = f2_add<C>(acc, T.X);
}
template <class C>
__global__ void __launch_bounds__(64) mb_fpmul(Fp<C>* io, int iters) {      // inline mont mul chain
  Fp<C> a = io[threadIdx.x], b = io[threadIdx.x + 64];
  for (int i = 0; i < iters; ++i) { u32 t[2 * C::L]; mul_wide<C>(t, a.v, b.v); a = redc<C>(t); }
  io[blockIdx.x * 64 + threadIdx.x] = a;
}

template <class F> float run(F launch, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a); for (int r = 0; r < reps; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

template <class C> void suite(const char* name) {
  void* buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20);
  const int iters = 64;
  for (int blocks : {1, 256, 1024, 2048, 4096, 8192}) {
    float t1 = run([&] { mb_coop<C><<<blocks, 64, Coop<C>::WAVE_BYTES>>>((Fp2<C>*)buf, iters); }, 3);
    float t2 = run([&] { mb_step<C, 0><<<blocks, 64>>>((Fp2<C>*)buf, iters); }, 3);
    float t3 = run([&] { mb_step<C, 1><<<blocks, 64>>>((Fp2<C>*)buf, iters); }, 3);
    float t4 = run([&] { mb_fpmul<C><<<blocks, 64>>>((Fp<C>*)buf, 4096); }, 3);
    double macs = (double)blocks * 64 * 4096 * (2.0 * C::L * C::L + C::L);
    printf("%s blocks=%5d  coop(sqr+6 lines)x64: %8.3f ms | dbl_step(calls)x64: %8.3f ms | dbl_step(inline)x64: %8.3f ms | fpmul x4096: %8.3f ms = %.2f TMAC/s\n",
           name, blocks, t1, t2, t3, t4, macs / t4 / 1e9);
  }
  hipFree(buf);
}
int main() { suite<BN254>("bn "); suite<BLS381>("bls"); return 0; }
