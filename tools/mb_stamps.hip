// Development tool (not part of the library): who waits for whom inside k_miller_x60?
// Runs the kernel's DBG == 4 form (miller_x.hpp: s_memtime stamps at the arrival at and the release from the two block barriers of every
// line step, every sixteenth block) over a whole batch and writes the raw records to a file; tools/stamps_report.py turns them into
// the per-role histograms of profiles/r6/.  Also times the plain kernel, the consumer-only and the producer-only forms beside it, and
// the issue rate of v_mad_u64_u32 with ONE, two and three waves per SIMD.
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibgls_amd/csrc -Iinclude tools/mb_stamps.hip -o tools/mb_stamps.bin
// run:    tools/mb_stamps.bin <n pairings> <out prefix> [rot_mode]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "miller_x.hpp"

using namespace bgls;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class C>
struct Inputs {
  Aff<F1<C>>* g1s; uint8_t* g2s; Fp2<C>* out; uint32_t* flags; u32* park; size_t nb;
  Inputs(size_t n) {
    typedef MX<C, 60> K;
    nb = (n + 59) / 60;
    CHK(hipMalloc(&g1s, n * sizeof(Aff<F1<C>>)));
    CHK(hipMalloc(&g2s, n * 4 * C::FP_BYTES));
    CHK(hipMalloc(&out, nb * 60 * sizeof(Fp2<C>)));
    CHK(hipMalloc(&flags, 4));
    CHK(hipMalloc(&park, K::park_bytes(nb)));
    std::vector<uint8_t> h2(n * 4 * C::FP_BYTES);
    for (size_t i = 0; i < h2.size(); ++i) h2[i] = (uint8_t)((i * 2654435761u) >> 13);
    for (size_t i = 0; i < n * 4; ++i) h2[i * C::FP_BYTES] = 0x01;       // top byte small: below p
    std::vector<Aff<F1<C>>> h1(n);
    memset(h1.data(), 0, n * sizeof(Aff<F1<C>>));
    for (size_t i = 0; i < n; ++i) { for (int k = 0; k < C::L; ++k) { h1[i].x.v[k] = (u32)(i * 97 + k * 13 + 5); h1[i].y.v[k] = (u32)(i * 31 + k * 7 + 3); } h1[i].x.v[C::L - 1] = 1; h1[i].y.v[C::L - 1] = 2; }
    CHK(hipMemcpy(g2s, h2.data(), h2.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(g1s, h1.data(), n * sizeof(Aff<F1<C>>), hipMemcpyHostToDevice));
  }
  ~Inputs() { (void)hipFree(g1s); (void)hipFree(g2s); (void)hipFree(out); (void)hipFree(flags); (void)hipFree(park); }
};

template <class C, int DBG>
static double time_kernel(Inputs<C>& in, size_t n, int rot, int reps, u32* rec) {
  typedef MX<C, 60> K;
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  k_miller_x60<C, DBG, 60><<<(unsigned)in.nb, K::THREADS, K::BLOCK_BYTES>>>(in.g1s, in.g2s, n, in.out, in.flags, in.park, rot, rec);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) k_miller_x60<C, DBG, 60><<<(unsigned)in.nb, K::THREADS, K::BLOCK_BYTES>>>(in.g1s, in.g2s, n, in.out, in.flags, in.park, rot, rec);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

template <class C>
static void run(const char* name, size_t n, int rot, const std::string& prefix) {
  Inputs<C> in(n);
  const size_t nsamp = (in.nb + MX_STAMP_EVERY - 1) / MX_STAMP_EVERY;
  u32* rec;
  CHK(hipMalloc(&rec, nsamp * 3 * MX_STAMP_DW * 4));
  CHK(hipMemset(rec, 0, nsamp * 3 * MX_STAMP_DW * 4));
  const double whole = time_kernel<C, 0>(in, n, rot, 3, nullptr);
  printf("%s whole %.3f ms\n", name, whole); fflush(stdout);
  {   // the partial products of the plain kernel as one checksum: every build variant must print the same one
    std::vector<uint8_t> ho(in.nb * 60 * sizeof(Fp2<C>));
    CHK(hipMemcpy(ho.data(), in.out, ho.size(), hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;
    for (size_t i = 0; i < ho.size(); ++i) { hsh ^= ho[i]; hsh *= 1099511628211ull; }
    printf("%s partial products checksum %016llx\n", name, hsh);
  }
  const double prod = time_kernel<C, 1>(in, n, rot, 2, nullptr);
  const double cons = time_kernel<C, 2>(in, n, rot, 2, nullptr);
  const double stamped = time_kernel<C, 4>(in, n, rot, 1, rec);      // the records of the LAST launch stay
  printf("%s n=%zu mode=%d  whole %.3f ms   producers only %.3f ms   consumer only %.3f ms   with stamps %.3f ms\n", name, n, rot, whole, prod, cons, stamped);
  std::vector<u32> h(nsamp * 3 * MX_STAMP_DW);
  CHK(hipMemcpy(h.data(), rec, h.size() * 4, hipMemcpyDeviceToHost));
  const std::string fn = prefix + "_" + name + ".bin";
  FILE* f = fopen(fn.c_str(), "wb");
  if (f) {
    const u32 hdr[4] = {(u32)nsamp, (u32)MX_STAMP_DW, (u32)MX_STAMP_STEPS, (u32)MX_STAMP_EVERY};
    fwrite(hdr, 4, 4, f);
    fwrite(h.data(), 4, h.size(), f);
    fclose(f);
    printf("  %zu sampled blocks -> %s\n", nsamp, fn.c_str());
  }
  CHK(hipFree(rec));
}

// ---- issue rate of the multiplier instruction by waves per SIMD (one wave alone: can a lone wave keep the multiplier busy?)
template <int KIND>
__global__ void __launch_bounds__(256) k_issue(int iters, uint64_t* sink) {
  uint64_t c[8];
  uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u;
  for (int j = 0; j < 8; ++j) c[j] = (uint64_t)j * 0x9e3779b97f4a7c15ull + a;
  const unsigned long long tc0 = __builtin_readcyclecounter(), tr0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    for (int r = 0; r < 8; ++r) {
      if constexpr (KIND == 0)
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b) : "vcc");
      else if constexpr (KIND == 2)   // the kernel's mix: two multiplier instructions, one plain 32-bit instruction
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_add_u32 %9, %9, %8\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_and_b32 %9, %9, %8\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_add_u32 %9, %9, %8\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n v_and_b32 %9, %9, %8"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b) : "vcc");
      else      // a dependent pair: the second instruction accumulates into the first one's column (what a column of a pile sees NL instructions apart is here back to back)
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %6, vcc, %8, %9, %6"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b) : "vcc");
    }
  }
  const unsigned long long tc1 = __builtin_readcyclecounter(), tr1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) { sink[1] = tc1 - tc0; sink[2] = tr1 - tr0; }       // shader clocks and 100 MHz ticks of one wave's loop
  uint64_t x = 0;
  for (int j = 0; j < 8; ++j) x ^= c[j];
  if (x == 0x1234567ull) sink[0] = x;
}
static double g_probe_mhz = 0;
template <int KIND>
static double time_issue(int waves_per_simd, uint64_t* sink) {
  const int iters = 4000;
  const unsigned blocks = 256 * waves_per_simd;      // 256 threads = 4 waves per block: one per SIMD of a CU
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  k_issue<KIND><<<blocks, 256>>>(10, sink);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  k_issue<KIND><<<blocks, 256>>>(iters, sink);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  uint64_t h[3];
  CHK(hipMemcpy(h, sink, 24, hipMemcpyDeviceToHost));
  g_probe_mhz = h[2] ? (double)h[1] / (double)h[2] * 100.0 : 0;
  return (double)ms * 1e6 / ((double)iters * 64 * waves_per_simd);     // ns per wave-instruction per SIMD
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1048560;
  const std::string prefix = argc > 2 ? argv[2] : "stamps";
  const int rot = argc > 3 ? atoi(argv[3]) : 0;
  uint64_t* sink;
  CHK(hipMalloc(&sink, 32));
  for (int wps = 1; wps <= 4; ++wps) {
    const double t0 = time_issue<0>(wps, sink), mhz0 = g_probe_mhz, t1 = time_issue<1>(wps, sink);
    printf("v_mad_u64_u32, %d wave(s) per SIMD: independent columns %.3f ns per wave-instruction and SIMD = %.2f clocks at the %.0f MHz the probe runs at, dependent pairs %.3f ns\n", wps, t0,
           t0 * mhz0 / 1000.0, mhz0, t1);
    const double t2 = time_issue<2>(wps, sink), mhz2 = g_probe_mhz;
    printf("   mix of 8 multiplier + 4 plain instructions: %.3f ns per group of 12 = %.2f clocks at %.0f MHz\n", t2 * 8.0 / 8.0 * 1.0, t2 * mhz2 / 1000.0, mhz2);
  }
#ifndef MB_ONLY_BLS
  run<BN254W>("BN254W", n, rot, prefix);
#endif
#ifndef MB_ONLY_BN
  run<BLS381>("BLS381", n, rot, prefix);
#endif
  return 0;
}
