// Development tool (not part of the library): where and when do the blocks of ONE lone k_miller_x60 launch run?
// A third to a half of the lone 1024-block launches of an alt-bn128 2^16 batch take 1.5x the cycles of the others; this
// records, per wave, HW_ID / XCC_ID and start / end times (s_memrealtime, 100 MHz) and prints, per launch, the duration, the
// number of blocks that started late and the per-CU residency.
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibgls_amd/csrc -Iinclude tools/mb_lone.hip -o tools/mb_lone.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <map>
#include <algorithm>
#include <unistd.h>
#include "miller_x.hpp"

using namespace bgls;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// stand-ins for what precedes the Miller launch in a verification: many one-wave blocks with (or without) static LDS, then a
// one-block kernel; `pre` selects: 0 nothing, 1 LDS blocks + one-block kernel, 2 blocks without LDS + one-block kernel,
// 3 LDS blocks only, 4 = 1 followed by a 30 us one-wave spin
template <int LDSW>
__global__ void __launch_bounds__(64) k_pre(uint32_t* sink, int spin) {
  __shared__ uint32_t tab[LDSW > 0 ? LDSW : 1];
  const unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t v = threadIdx.x;
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { tab[threadIdx.x % (LDSW > 0 ? LDSW : 1)] = v; v = v * 3 + tab[(threadIdx.x + 1) % (LDSW > 0 ? LDSW : 1)]; }
  if (v == 0x12345678u) sink[0] = v;
}

__global__ void __launch_bounds__(64) k_priv(uint32_t* sink, int v) {        // a kernel with a private segment (dynamically indexed private array)
  volatile uint32_t priv[64];
  priv[threadIdx.x & 63] = v + threadIdx.x;
  if (priv[(threadIdx.x * 7 + v) & 63] == 0xFFFFFFFFu) sink[0] = 1;
}

template <class C, int NP>
static void run(size_t n, int rot, int launches, int pre) {
  typedef MX<C, NP> K;
  const size_t nb = (n + NP - 1) / NP;
  Aff<F1<C>>* g1s; uint8_t* g2s; Fp2<C>* out; uint32_t* flags; u32* park; u32* rec;
  CHK(hipMalloc(&g1s, n * sizeof(Aff<F1<C>>)));
  CHK(hipMalloc(&g2s, n * 4 * C::FP_BYTES));
  CHK(hipMalloc(&out, nb * 60 * sizeof(Fp2<C>)));
  CHK(hipMalloc(&flags, 4));
  CHK(hipMalloc(&park, K::park_bytes(nb)));
  CHK(hipMalloc(&rec, nb * 3 * 8 * 4));
  std::vector<uint8_t> h2(n * 4 * C::FP_BYTES);
  for (size_t i = 0; i < h2.size(); ++i) h2[i] = (uint8_t)((i * 2654435761u) >> 13);
  for (size_t i = 0; i < n * 4; ++i) h2[i * C::FP_BYTES] = 0x01;
  std::vector<Aff<F1<C>>> h1(n);
  memset(h1.data(), 0, n * sizeof(Aff<F1<C>>));
  for (size_t i = 0; i < n; ++i) { for (int k = 0; k < C::L; ++k) { h1[i].x.v[k] = (u32)(i * 97 + k * 13 + 5); h1[i].y.v[k] = (u32)(i * 31 + k * 7 + 3); } h1[i].x.v[C::L - 1] = 1; h1[i].y.v[C::L - 1] = 2; }
  CHK(hipMemcpy(g2s, h2.data(), h2.size(), hipMemcpyHostToDevice));
  CHK(hipMemcpy(g1s, h1.data(), n * sizeof(Aff<F1<C>>), hipMemcpyHostToDevice));
  CHK(hipFuncSetAttribute((const void*)k_miller_x60<C, 3, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, K::BLOCK_BYTES));
  std::vector<u32> h(nb * 3 * 8);
  hipStream_t st = nullptr;
  hipEvent_t ea, eb;
  CHK(hipEventCreate(&ea)); CHK(hipEventCreate(&eb));
  if (pre >= 5 && pre <= 9) CHK(hipStreamCreate(&st));       // 5..9: on a created stream; 6, 7: bracketed by event records; 7, 8: + a kernel with a private segment before
  for (int l = 0; l < launches; ++l) {
    CHK(hipMemset(rec, 0, nb * 3 * 8 * 4));
    CHK(hipDeviceSynchronize());
    if (pre >= 10) usleep((pre - 10) * 1000);          // 1x: an idle gap of (pre - 10) ms before the launch
    if (pre >= 100) { usleep((pre - 100) * 1000); k_pre<0><<<1024, 192>>>(flags, 20000); }      // 1xx: the same gap, then a wake-up kernel on every CU
    if (pre == 1 || pre == 3 || pre == 4) k_pre<2560><<<1024, 64>>>(flags, 200000);
    if (pre == 2) k_pre<0><<<1024, 64>>>(flags, 200000);
    if (pre == 1 || pre == 2 || pre == 4) k_pre<0><<<1, 64>>>(flags, 10000);
    if (pre == 4) k_pre<0><<<1, 64>>>(flags, 60000);
    if (pre >= 7 && pre <= 9) { k_pre<2560><<<1024, 64, 0, st>>>(flags, 200000); k_priv<<<1024, 64, 0, st>>>(flags, l); k_priv<<<1, 64, 0, st>>>(flags, l); }
    if (pre == 6 || pre == 7) CHK(hipEventRecord(ea, st));
    k_miller_x60<C, 3, NP><<<(unsigned)nb, K::THREADS, K::BLOCK_BYTES, st>>>(g1s, g2s, n, out, flags, park, rot, rec);
    if (pre == 6 || pre == 7) CHK(hipEventRecord(eb, st));
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h.data(), rec, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    auto ts = [&](size_t w, int k) { return (unsigned long long)h[w * 8 + k] | ((unsigned long long)h[w * 8 + k + 1] << 32); };
    for (size_t w = 0; w < nb * 3; ++w) { t0 = std::min(t0, ts(w, 2)); t1 = std::max(t1, ts(w, 4)); }
    const double us = 1.0 / 100.0;       // 100 MHz
    int late = 0;
    std::map<u32, int> per_cu;           // blocks resident per CU at the start
    std::map<u32, std::vector<int>> cons_simd;
    double max_dur = 0, min_dur = 1e9;
    for (size_t b = 0; b < nb; ++b) {
      const double st = (ts(b * 3, 2) - t0) * us, en = (ts(b * 3, 4) - t0) * us;
      max_dur = std::max(max_dur, en - st); min_dur = std::min(min_dur, en - st);
      const u32 hw = h[b * 3 * 8], xcc = h[b * 3 * 8 + 1] & 15;
      const u32 cu = (xcc << 8) | ((hw >> 8) & 0xFF);
      if (st > 500.0) { ++late; if (late <= 0) printf("    late block %zu: start %.0f us end %.0f us  xcc %u cu_key %02x\n", b, st, en, xcc, (hw >> 8) & 0xFF); }
      else per_cu[cu]++;
      for (int w = 0; w < 3; ++w) if (h[(b * 3 + w) * 8 + 6] == 2) cons_simd[cu].push_back((h[(b * 3 + w) * 8] >> 4) & 3);
    }
    {   // slot occupancy: sum of block residencies / (4 slots x CUs x launch duration); and over the steady part (10 % .. 80 % of the launch)
      double tot = 0, steady = 0;
      const double T = (t1 - t0) * us, a = 0.1 * T, b = 0.8 * T;
      for (size_t blk = 0; blk < nb; ++blk) {
        const double st = (ts(blk * 3, 2) - t0) * us, en = (ts(blk * 3, 4) - t0) * us;
        tot += en - st;
        const double lo = st > a ? st : a, hi = en < b ? en : b;
        if (hi > lo) steady += hi - lo;
      }
      printf("  slot occupancy %.3f overall, %.3f in the steady part; blocks %zu\n", tot / (4.0 * 256 * T), steady / (4.0 * 256 * (b - a)), nb);
    }
    int hist[8] = {0};
    for (auto& kv : per_cu) hist[std::min(kv.second, 7)]++;
    int dup = 0;
    for (auto& kv : cons_simd) { int c[4] = {0, 0, 0, 0}; for (int s : kv.second) c[s]++; for (int s = 0; s < 4; ++s) if (c[s] > 1) ++dup; }
    if (0) printf("");
    printf("launch %2d: %.2f ms  late blocks %d  block time %.2f .. %.2f ms  CUs %zu  blocks/CU at start: 3:%d 4:%d 5:%d  SIMDs with >1 consumer: %d\n", l, (t1 - t0) * us / 1000.0, late,
           min_dur / 1000.0, max_dur / 1000.0, per_cu.size(), hist[3], hist[4], hist[5], dup);
  }
}

int main(int argc, char** argv) {
  const int curve = argc > 1 ? atoi(argv[1]) : 0, np = argc > 2 ? atoi(argv[2]) : 64, rot = argc > 3 ? atoi(argv[3]) : 8, launches = argc > 4 ? atoi(argv[4]) : 12;
  const int pre = argc > 5 ? atoi(argv[5]) : 0;
  const size_t n = (size_t)(argc > 6 ? atoi(argv[6]) : 1024) * np;
  if (curve == 0 && np == 64) run<BN254, 64>(n, rot, launches, pre);
  else if (curve == 0) run<BN254, 60>(n, rot, launches, pre);
  else if (np == 64) run<BLS381, 64>(n, rot, launches, pre);
  else run<BLS381, 60>(n, rot, launches, pre);
  return 0;
}
