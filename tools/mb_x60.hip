// Development microbenchmarks for k_miller_x60 (not part of the library):
//   * issue cost of the instructions the kernel is made of, relative to v_mad_u64_u32
//   * the kernel with only one of its two roles doing work (DBG 1 / 2), next to the whole kernel
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibgls_amd/csrc -Iinclude tools/mb_x60.hip -o tools/mb_x60.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "miller_x.hpp"

using namespace bgls;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- instruction issue cost: 8 independent chains per lane, 64 instructions per loop iteration
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
template <int KIND>
__global__ void __launch_bounds__(256) k_issue(int iters, uint64_t* sink) {
  uint64_t c[8];
  uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u;
  for (int j = 0; j < 8; ++j) c[j] = (uint64_t)j * 0x9e3779b97f4a7c15ull + a;
  uint32_t w[8];
  for (int j = 0; j < 8; ++j) w[j] = a * (j + 3);
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
#define X(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if constexpr (KIND == 1) {
#define X(j) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if constexpr (KIND == 2) {       // one asm block of 8 (no s_nop between)
      for (int r = 0; r < 8; ++r)
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b) : "vcc");
    } else if constexpr (KIND == 3) {       // v_add_u32
#define X(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[j]) : "v"(a));
      REP64(X)
#undef X
    } else if constexpr (KIND == 4) {       // v_and_b32
#define X(j) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[j]) : "v"(a));
      REP64(X)
#undef X
    } else if constexpr (KIND == 5) {       // v_lshl_add_u64
#define X(j) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c[j]) : "v"(c[(j + 1) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 6) {       // v_ashrrev_i64
#define X(j) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(c[j]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 7) {       // v_mov_b32_dpp quad_perm
#define X(j) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(w[j]) : "v"(w[(j + 1) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 8) {       // v_mul_lo_u32
#define X(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[j]) : "v"(a));
      REP64(X)
#undef X
    } else if constexpr (KIND == 9) {       // v_cndmask_b32
#define X(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[j]) : "v"(a));
      REP64(X)
#undef X
    } else if constexpr (KIND == 10) {      // mad with a scalar constant factor
#define X(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c[j]) : "v"(a), "s"(0x0ffc123u + j) : "vcc");
      REP64(X)
#undef X
    } else if constexpr (KIND == 11) {      // mad followed by s_nop 0
#define X(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0" : "+v"(c[j]) : "v"(a), "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if constexpr (KIND == 12) {      // mad + one plain VALU alternating
#define X(j) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32 %1, %1, %2" : "+v"(c[j]), "+v"(w[j]) : "v"(a), "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if constexpr (KIND == 14) {      // v_cndmask_b32_e64 with an SGPR-pair mask
      const uint64_t msk = 0x5555555555555555ull;
#define X(j) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(w[j]) : "v"(a), "s"(msk));
      REP64(X)
#undef X
    } else if constexpr (KIND == 15) {      // v_bfi_b32
#define X(j) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(w[j]) : "v"(a), "v"(b));
      REP64(X)
#undef X
    } else if constexpr (KIND == 16) {      // v_xor_b32
#define X(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(w[j]) : "v"(a));
      REP64(X)
#undef X
    } else if constexpr (KIND == 17) {      // v_mov_b32_dpp quad_perm [0,0,2,2]
#define X(j) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "+v"(w[j]) : "v"(w[(j + 1) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 18) {      // v_add_u32 with DPP operand (dpp folded into the consumer instruction)
#define X(j) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(w[j]) : "v"(w[(j + 1) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 19) {      // v_sub_u32
#define X(j) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(w[j]) : "v"(a));
      REP64(X)
#undef X
    } else if constexpr (KIND == 20) {      // v_lshrrev_b32
#define X(j) asm volatile("v_lshrrev_b32 %0, 28, %0" : "+v"(w[j]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 21) {      // v_add3_u32
#define X(j) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(w[j]) : "v"(a), "v"(b));
      REP64(X)
#undef X
    } else if constexpr (KIND == 22) {      // v_and_or_b32
#define X(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(w[j]) : "v"(a), "v"(b));
      REP64(X)
#undef X
    } else if constexpr (KIND == 23) {      // v_mov_b32
#define X(j) asm volatile("v_mov_b32 %0, %1" : "+v"(w[j]) : "v"(w[(j + 1) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 24) {      // v_fma_f64 (round 5, variant D of the number-form question: is the FP64 pipe any faster than the integer multiplier?)
#define X(j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(c[j]) : "v"(c[(j + 1) & 7]), "v"(c[(j + 2) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 25) {      // 64-bit add as v_add_co_u32 + v_addc_co_u32 (counted as ONE operation of two instructions)
#define X(j) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(w[j]), "+v"(w[(j + 4) & 7]) : "v"(a), "v"(b) : "vcc");
      REP64(X)
#undef X
    } else if constexpr (KIND == 26) {      // one 52 x 52-bit limb product of the FP64 form: hi = fma(a, b, C1), lo = fma(a, b, C2 - hi), both bit patterns added into 64-bit integer columns
#define X(j) asm volatile("v_fma_f64 %0, %2, %3, %0\n v_fma_f64 %1, %2, %3, %1\n v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0" : "+v"(c[j]), "+v"(c[(j + 4) & 7]) : "v"(c[(j + 1) & 7]), "v"(c[(j + 2) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 27) {      // v_add_f64
#define X(j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(c[j]) : "v"(c[(j + 1) & 7]));
      REP64(X)
#undef X
    } else if constexpr (KIND == 13) {      // v_alignbit_b32
#define X(j) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(w[j]) : "v"(a));
      REP64(X)
#undef X
    }
  }
  uint64_t x = 0;
  for (int j = 0; j < 8; ++j) x ^= c[j] ^ w[j];
  if (x == 0x1234567ull) sink[0] = x;
}

template <int KIND>
static double time_issue(int waves_per_simd, uint64_t* sink) {
  const int iters = 2000;
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
  // ns per wave-instruction per SIMD
  return (double)ms * 1e6 / ((double)iters * 64 * waves_per_simd);
}

template <class C, int DBG, int NP = 60>
static double time_x60(size_t n, int rot, int reps) {
  typedef MX<C, NP> K;
  const size_t nb = (n + NP - 1) / NP;
  Aff<F1<C>>* g1s; uint8_t* g2s; Fp2<C>* out; uint32_t* flags; u32* park;
  CHK(hipMalloc(&g1s, n * sizeof(Aff<F1<C>>)));
  CHK(hipMalloc(&g2s, n * 4 * C::FP_BYTES));
  CHK(hipMalloc(&out, nb * 60 * sizeof(Fp2<C>)));
  CHK(hipMalloc(&flags, 4));
  CHK(hipMalloc(&park, K::park_bytes(nb)));
  // synthetic operands: arbitrary reduced field elements (not on the curve: the arithmetic does the same work)
  std::vector<uint8_t> h2(n * 4 * C::FP_BYTES);
  for (size_t i = 0; i < h2.size(); ++i) h2[i] = (uint8_t)((i * 2654435761u) >> 13);
  for (size_t i = 0; i < n * 4; ++i) h2[i * C::FP_BYTES] = 0x01;       // top byte small: below p
  std::vector<Aff<F1<C>>> h1(n);
  memset(h1.data(), 0, n * sizeof(Aff<F1<C>>));
  for (size_t i = 0; i < n; ++i) { for (int k = 0; k < C::L; ++k) { h1[i].x.v[k] = (u32)(i * 97 + k * 13 + 5); h1[i].y.v[k] = (u32)(i * 31 + k * 7 + 3); } h1[i].x.v[C::L - 1] = 1; h1[i].y.v[C::L - 1] = 2; }
  CHK(hipMemcpy(g2s, h2.data(), h2.size(), hipMemcpyHostToDevice));
  CHK(hipMemcpy(g1s, h1.data(), n * sizeof(Aff<F1<C>>), hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  k_miller_x60<C, DBG, NP><<<(unsigned)nb, K::THREADS, K::BLOCK_BYTES>>>(g1s, g2s, n, out, flags, park, rot, nullptr);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) k_miller_x60<C, DBG, NP><<<(unsigned)nb, K::THREADS, K::BLOCK_BYTES>>>(g1s, g2s, n, out, flags, park, rot, nullptr);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  CHK(hipFree(g1s)); CHK(hipFree(g2s)); CHK(hipFree(out)); CHK(hipFree(flags)); CHK(hipFree(park));
  return ms / reps;
}

// ---- where do the three waves of a block land?  Same launch shape and register allocation as k_miller_x60.
__global__ void __launch_bounds__(192, 3) k_where(uint32_t* rec, int spin) {
  extern __shared__ u32 lds[];
  asm volatile("v_mov_b32 v167, 0" ::: "v167");                     // forces the 168-register allocation
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { lds[threadIdx.x] = (u32)t0; }
  if ((threadIdx.x & 63) == 0) {
    uint32_t* r = rec + ((size_t)blockIdx.x * 3 + (threadIdx.x >> 6)) * 2;
    r[0] = hw;
    r[1] = xcc;
  }
}
static void where_test() {
  const unsigned nb = 1024;
  uint32_t* rec;
  CHK(hipMalloc(&rec, nb * 3 * 2 * 4));
  k_where<<<nb, 192, MX<BLS381>::BLOCK_BYTES>>>(rec, 400000);
  CHK(hipDeviceSynchronize());
  std::vector<uint32_t> h(nb * 3 * 2);
  CHK(hipMemcpy(h.data(), rec, h.size() * 4, hipMemcpyDeviceToHost));
  // role pattern per (xcc, se, sh, cu, simd): count waves by index within the block (0, 1 = producers, 2 = consumer at rot 0)
  int hist[4][4][4] = {};      // [waves 0][waves 1][waves 2] per SIMD -> number of SIMDs with that mix
  std::vector<int> cnt(8 * 16 * 16 * 4 * 3, 0);
  for (unsigned b = 0; b < nb; ++b)
    for (int w = 0; w < 3; ++w) {
      const uint32_t hw = h[(b * 3 + w) * 2], xcc = h[(b * 3 + w) * 2 + 1] & 15;
      const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      cnt[((((xcc * 16 + se * 2 + sh) * 16 + cu) * 4) + simd) * 3 + w]++;
      if (b < 8) printf("block %u wave %d: hw_id %08x xcc %u se %d sh %d cu %d simd %d wave_slot %d\n", b, w, hw, xcc, se, sh, cu, simd, hw & 15);
    }
  int simds = 0;
  for (size_t k = 0; k < cnt.size(); k += 3) {
    const int a = cnt[k], c1 = cnt[k + 1], c2 = cnt[k + 2];
    if (a + c1 + c2 == 0) continue;
    ++simds;
    hist[a < 4 ? a : 3][c1 < 4 ? c1 : 3][c2 < 4 ? c2 : 3]++;
  }
  printf("SIMDs hosting waves: %d\n", simds);
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c)
    if (hist[a][b][c]) printf("  SIMDs with %d x wave0, %d x wave1, %d x wave2: %d\n", a, b, c, hist[a][b][c]);
  CHK(hipFree(rec));
}

int main(int argc, char** argv) {
  uint64_t* sink;
  CHK(hipMalloc(&sink, 8));
  const char* what = argc > 1 ? argv[1] : "all";
  if (!strcmp(what, "all") || !strcmp(what, "issue")) {
    const char* names[] = {"v_mad_u64_u32 (vcc sink)", "v_mad_i64_i32", "v_mad_u64_u32 x8 per asm", "v_add_u32", "v_and_b32", "v_lshl_add_u64", "v_ashrrev_i64",
                           "v_mov_b32_dpp", "v_mul_lo_u32", "v_cndmask_b32", "v_mad_u64_u32 sgpr factor", "v_mad_u64_u32 + s_nop 0", "v_mad_u64_u32 + v_add_u32 (pair)", "v_alignbit_b32",
                           "v_cndmask_b32_e64 sgpr mask", "v_bfi_b32", "v_xor_b32", "v_mov_b32_dpp [0,0,2,2]", "v_add_u32_dpp", "v_sub_u32", "v_lshrrev_b32", "v_add3_u32", "v_and_or_b32", "v_mov_b32",
                           "v_fma_f64", "v_add_co_u32 + v_addc_co_u32 (64-bit add, 2 instr)", "FP64 limb product: 2 fma_f64 + 2 lshl_add_u64 (4 instr)", "v_add_f64"};
    for (int wps = 2; wps <= 3; ++wps) {
      double t[28];
      t[24] = time_issue<24>(wps, sink); t[25] = time_issue<25>(wps, sink); t[26] = time_issue<26>(wps, sink); t[27] = time_issue<27>(wps, sink);
      t[14] = time_issue<14>(wps, sink); t[15] = time_issue<15>(wps, sink); t[16] = time_issue<16>(wps, sink); t[17] = time_issue<17>(wps, sink);
      t[18] = time_issue<18>(wps, sink); t[19] = time_issue<19>(wps, sink); t[20] = time_issue<20>(wps, sink); t[21] = time_issue<21>(wps, sink);
      t[22] = time_issue<22>(wps, sink); t[23] = time_issue<23>(wps, sink);
      t[0] = time_issue<0>(wps, sink); t[1] = time_issue<1>(wps, sink); t[2] = time_issue<2>(wps, sink); t[3] = time_issue<3>(wps, sink);
      t[4] = time_issue<4>(wps, sink); t[5] = time_issue<5>(wps, sink); t[6] = time_issue<6>(wps, sink); t[7] = time_issue<7>(wps, sink);
      t[8] = time_issue<8>(wps, sink); t[9] = time_issue<9>(wps, sink); t[10] = time_issue<10>(wps, sink); t[11] = time_issue<11>(wps, sink);
      t[12] = time_issue<12>(wps, sink); t[13] = time_issue<13>(wps, sink);
      for (int k = 0; k < 28; ++k) printf("waves/SIMD %d  %-34s %.3f ns per wave-instruction per SIMD   (x%.2f of mad)\n", wps, names[k], t[k], t[k] / t[0]);
    }
  }
  if (!strcmp(what, "where")) where_test();
  if (!strcmp(what, "all") || !strcmp(what, "x60")) {
    const size_t n = argc > 2 ? (size_t)atol(argv[2]) : 61440;
    for (int a = 3; a < (argc > 3 ? argc : 4); ++a) {
      const int rot = argc > 3 ? atoi(argv[a]) : 0;
      printf("BLS381 n=%zu mode=%d  whole %.3f ms   producer only %.3f ms   consumer only %.3f ms\n", n, rot, time_x60<BLS381, 0>(n, rot, 5), time_x60<BLS381, 1>(n, rot, 3),
             time_x60<BLS381, 2>(n, rot, 3));
      printf("BN254  n=%zu mode=%d  whole %.3f ms   producer only %.3f ms   consumer only %.3f ms\n", n, rot, time_x60<BN254, 0>(n, rot, 5), time_x60<BN254, 1>(n, rot, 3),
             time_x60<BN254, 2>(n, rot, 3));
      printf("BN254W n=%zu mode=%d  whole %.3f ms   producer only %.3f ms   consumer only %.3f ms   (nine limbs of 29 bits)\n", n, rot, time_x60<BN254W, 0>(n, rot, 5),
             time_x60<BN254W, 1>(n, rot, 3), time_x60<BN254W, 2>(n, rot, 3));
    }
  }
  return 0;
}
