// Quick development timing of k_miller_x60 alone (not part of the library): both curves, the shipped instantiation only.
//   mb_x60q_<NP>_<curve>.bin <mode> <reps> <n> [<n> ...]   per-launch HIP-event times (min / median / max) of `reps` back-to-back launches
// build (one binary per block form and curve, they compile in parallel):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibgls_amd/csrc -Iinclude -DMB_NP=64 -DMB_ONLY_BN tools/mb_x60q.hip -o tools/mb_x60q_64_bn.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <vector>
#include "miller_x.hpp"

using namespace bgls;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class C, int NP>
static void time_x(const char* name, size_t n, int rot, int reps) {
  typedef MX<C, NP> K;
  const size_t nb = (n + NP - 1) / NP;
  Aff<F1<C>>* g1s; uint8_t* g2s; Fp2<C>* out; uint32_t* flags; u32* park;
  CHK(hipMalloc(&g1s, n * sizeof(Aff<F1<C>>)));
  CHK(hipMalloc(&g2s, n * 4 * C::FP_BYTES));
  CHK(hipMalloc(&out, nb * 60 * sizeof(Fp2<C>)));
  CHK(hipMalloc(&flags, 4));
  CHK(hipMalloc(&park, K::park_bytes(nb)));
  std::vector<uint8_t> h2(n * 4 * C::FP_BYTES);
  for (size_t i = 0; i < h2.size(); ++i) h2[i] = (uint8_t)((i * 2654435761u) >> 13);
  for (size_t i = 0; i < n * 4; ++i) h2[i * C::FP_BYTES] = 0x01;
  std::vector<Aff<F1<C>>> h1(n);
  memset(h1.data(), 0, n * sizeof(Aff<F1<C>>));
  for (size_t i = 0; i < n; ++i) { for (int k = 0; k < C::L; ++k) { h1[i].x.v[k] = (u32)(i * 97 + k * 13 + 5); h1[i].y.v[k] = (u32)(i * 31 + k * 7 + 3); } h1[i].x.v[C::L - 1] = 1; h1[i].y.v[C::L - 1] = 2; }
  CHK(hipMemcpy(g2s, h2.data(), h2.size(), hipMemcpyHostToDevice));
  CHK(hipMemcpy(g1s, h1.data(), n * sizeof(Aff<F1<C>>), hipMemcpyHostToDevice));
  CHK(hipFuncSetAttribute((const void*)k_miller_x60<C, 0, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, K::BLOCK_BYTES));
  std::vector<hipEvent_t> ev(reps + 1);
  for (auto& e : ev) CHK(hipEventCreate(&e));
  for (int w = 0; w < 2; ++w) k_miller_x60<C, 0, NP><<<(unsigned)nb, K::THREADS, K::BLOCK_BYTES>>>(g1s, g2s, n, out, flags, park, rot, nullptr);
  CHK(hipDeviceSynchronize());
  std::vector<float> ms(reps);
  const char* gap = getenv("MB_GAP_US");                  // idle time before every launch (a lone verification starts on an idle GPU)
  if (gap) {
    for (int r = 0; r < reps; ++r) {
      CHK(hipDeviceSynchronize());
      usleep(atoi(gap));
      CHK(hipEventRecord(ev[0]));
      k_miller_x60<C, 0, NP><<<(unsigned)nb, K::THREADS, K::BLOCK_BYTES>>>(g1s, g2s, n, out, flags, park, rot, nullptr);
      CHK(hipEventRecord(ev[1]));
      CHK(hipDeviceSynchronize());
      CHK(hipEventElapsedTime(&ms[r], ev[0], ev[1]));
    }
  } else {
    CHK(hipEventRecord(ev[0]));
    for (int r = 0; r < reps; ++r) {
      k_miller_x60<C, 0, NP><<<(unsigned)nb, K::THREADS, K::BLOCK_BYTES>>>(g1s, g2s, n, out, flags, park, rot, nullptr);
      CHK(hipEventRecord(ev[r + 1]));
    }
    CHK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) CHK(hipEventElapsedTime(&ms[r], ev[r], ev[r + 1]));
  }
  std::sort(ms.begin(), ms.end());
  printf("%-6s NP=%d n=%zu blocks=%zu mode=%d lds=%d B  per launch: min %.3f  median %.3f  max %.3f ms   (%.2f M pairings/s at the median)\n", name, NP, n, nb, rot, K::BLOCK_BYTES,
         ms[0], ms[reps / 2], ms[reps - 1], n / ms[reps / 2] / 1e3);
  CHK(hipFree(g1s)); CHK(hipFree(g2s)); CHK(hipFree(out)); CHK(hipFree(flags)); CHK(hipFree(park));
}

#ifndef MB_NP
#define MB_NP 60
#endif
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int reps = argc > 2 ? atoi(argv[2]) : 7;
  for (int a = 3; a < (argc > 3 ? argc : 4); ++a) {
    const size_t n = argc > 3 ? (size_t)atol(argv[a]) : 61440;
#ifndef MB_ONLY_BLS
    time_x<BN254, MB_NP>("BN254", n, mode, reps);
#endif
#ifndef MB_ONLY_BN
    time_x<BLS381, MB_NP>("BLS381", n, mode, reps);
#endif
  }
  return 0;
}
