// Development micro-benchmark (not part of the library): Fp2 multiplication on alt-bn128 in two number
// representations.
//   A: the library's form (fp.hpp): 8 x 32-bit limbs, operand-scanning products with carry chains, Karatsuba
//      in double width, two Montgomery reductions.
//   B: 10 x 28-bit limbs, column accumulators in 64 bits (acc = a*b + acc, no carries between products),
//      schoolbook over Fp2 with the subtraction folded in as a "fat" negation, radix-2^28 Montgomery reduction.
//   C: A's limbs with the three Karatsuba products computed column-wise on the multiplier's own carry-out (inline asm,
//      three chains in lock step so that no hazard padding is needed); cross-checked against A on 16.7 M products.
//   Measured on MI355X (G Fp2 products/s, saturated): A 44.6, B 47.2, C 41.2 -- the carry-out form is bit-exact and
//   issues fewer instructions, but its in-place accumulator chains run slower than A's independent row products.
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

#if defined(__HIP_DEVICE_COMPILE__)
// ---- product scanning with the multiplier's own carry-out (device only) ----------------------------------------------
// v_mad_u64_u32 returns a*b + c64 together with its carry as a lane mask in an SGPR pair; the compiler never uses that
// operand (it re-derives carries with 64-bit compares), so the column form is spelled in inline asm: a column's products
// accumulate in a 64-bit register pair and the carry of each goes to a third word with one v_addc.  Per limb product:
// 1 multiplier + 1 add, against multiplier + add + re-zeroed addend + hazard nop in the operand-scanning form above.
// gfx940-family rule (LLVM GCNHazardRecognizer, hasVDecCoExecHazard): a VALU that reads an SGPR written by a VALU needs
// 2 wait states in between.  Three independent chains are therefore issued in lock step inside ONE asm statement -- the
// three multipliers, then the three carry adds -- so every carry add sits two instructions behind its multiplier and
// the carry masks never leave the statement (early-clobber scratch outputs: nothing for the hazard recogniser to pad).
// FIRST = the column's first product: the overflow word is written, not accumulated (saves zeroing it).
template <bool FIRST>
__device__ __forceinline__ void ps_step3(u64& acc0, u64& acc1, u64& acc2, u32& ov0, u32& ov1, u32& ov2, u32 a0, u32 b0, u32 a1, u32 b1,
                                         u32 a2, u32 b2) {
  u64 c0, c1, c2;
  if constexpr (FIRST)
    asm volatile(
        "v_mad_u64_u32 %0, %6, %9, %10, %0\n\t"
        "v_mad_u64_u32 %1, %7, %11, %12, %1\n\t"
        "v_mad_u64_u32 %2, %8, %13, %14, %2\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, 0, %6\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, 0, %7\n\t"
        "v_addc_co_u32_e64 %5, vcc, 0, 0, %8"
        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "=v"(ov0), "=v"(ov1), "=v"(ov2), "=&s"(c0), "=&s"(c1), "=&s"(c2)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2)
        : "vcc");
  else
    asm volatile(
        "v_mad_u64_u32 %0, %6, %9, %10, %0\n\t"
        "v_mad_u64_u32 %1, %7, %11, %12, %1\n\t"
        "v_mad_u64_u32 %2, %8, %13, %14, %2\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %7\n\t"
        "v_addc_co_u32_e64 %5, vcc, 0, %5, %8"
        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(ov0), "+v"(ov1), "+v"(ov2), "=&s"(c0), "=&s"(c1), "=&s"(c2)
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2)
        : "vcc");
}
// t[c] = a[c] * b[c], c = 0..2: three independent L-limb products, column by column
template <int L>
__device__ __forceinline__ void mul_wide_ps3(u32 (&t)[3][2 * L], const u32 (&a)[3][L], const u32 (&b)[3][L]) {
  u64 acc0 = 0, acc1 = 0, acc2 = 0;
  u32 ov0 = 0, ov1 = 0, ov2 = 0;
#pragma unroll
  for (int k = 0; k < 2 * L - 1; ++k) {
    bool first = true;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int j = k - i;
      if (j < 0 || j >= L) continue;
      if (first) ps_step3<true>(acc0, acc1, acc2, ov0, ov1, ov2, a[0][i], b[0][j], a[1][i], b[1][j], a[2][i], b[2][j]);
      else ps_step3<false>(acc0, acc1, acc2, ov0, ov1, ov2, a[0][i], b[0][j], a[1][i], b[1][j], a[2][i], b[2][j]);
      first = false;
    }
    t[0][k] = (u32)acc0;
    t[1][k] = (u32)acc1;
    t[2][k] = (u32)acc2;
    acc0 = (acc0 >> 32) | ((u64)ov0 << 32);
    acc1 = (acc1 >> 32) | ((u64)ov1 << 32);
    acc2 = (acc2 >> 32) | ((u64)ov2 << 32);
  }
  t[0][2 * L - 1] = (u32)acc0;
  t[1][2 * L - 1] = (u32)acc1;
  t[2][2 * L - 1] = (u32)acc2;
}
#else
template <int L>
__device__ void mul_wide_ps3(u32 (&t)[3][2 * L], const u32 (&a)[3][L], const u32 (&b)[3][L]);   // host pass of hipcc: declaration only
#endif

// C: the library's 8 x 32-bit limbs with product scanning on the multiplier's carry-out (mul_wide_ps3 above)
__device__ __forceinline__ Fp2<BN254> f2_mul_ps(const Fp2<BN254>& a, const Fp2<BN254>& b) {
  typedef BN254 C;
  constexpr int L = C::L, W = 2 * L;
  u32 A[3][L], B[3][L], T[3][W];
  Fp<C> sa = fp_add_nr<C>(a.c0, a.c1), sb = fp_add_nr<C>(b.c0, b.c1);
#pragma unroll
  for (int k = 0; k < L; ++k) { A[0][k] = a.c0.v[k]; B[0][k] = b.c0.v[k]; A[1][k] = a.c1.v[k]; B[1][k] = b.c1.v[k]; A[2][k] = sa.v[k]; B[2][k] = sb.v[k]; }
  mul_wide_ps3<L>(T, A, B);
  w_sub<W>(T[2], T[2], T[0]);
  w_sub<W>(T[2], T[2], T[1]);
  w_add<W>(T[0], T[0], C::P2W);
  w_sub<W>(T[0], T[0], T[1]);
  Fp2<C> r;
  r.c0 = redc<C>(T[0]);
  r.c1 = redc<C>(T[2]);
  return r;
}
template <int CH>
__global__ void __launch_bounds__(64) mb_c(Fp2<BN254>* io, int iters, size_t n) {
  const size_t t = blockIdx.x * 64 + threadIdx.x;
  Fp2<BN254> x[CH], y = io[n + t];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = io[(t + c * 7) % n];
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = f2_mul_ps(x[c], y);
  Fp2<BN254> s = x[0];
#pragma unroll
  for (int c = 1; c < CH; ++c) s = f2_add<BN254>(s, x[c]);
  io[t] = s;
}
// element-wise comparison of the two multiplications on random inputs: count of mismatching results
__global__ void __launch_bounds__(64) mb_check(const Fp2<BN254>* in, size_t n, unsigned long long* bad) {
  const size_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n) return;
  Fp2<BN254> a = in[t], b = in[n + t];
  for (int r = 0; r < 16; ++r) {
    Fp2<BN254> u = f2_mul_inl<BN254>(a, b), v = f2_mul_ps(a, b);
    if (!f2_eq<BN254>(u, v)) atomicAdd(bad, 1ull);
    a = f2_add<BN254>(u, b);
    b = f2_sub<BN254>(v, a);
  }
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

// ---- D: the consumer's unit of work -- a five-term Fp2 dot product with ONE reduction per output half ----
// A-form: per term three double-width products (Karatsuba over i) added into three double-width accumulators, then
// real = sum v0 - sum v1 + 6 p^2, imag = sum s - sum v0 - sum v1, two Montgomery reductions (as coop_dot_inl).
__device__ __forceinline__ Fp2<BN254> dot5_a(const Fp2<BN254> (&a)[5], const Fp2<BN254> (&b)[5]) {
  typedef BN254 C;
  constexpr int W = 2 * C::L;
  u32 v0[W], v1[W], s[W], tmp[W];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    Fp<C> sa = fp_add_nr<C>(a[t].c0, a[t].c1), sb = fp_add_nr<C>(b[t].c0, b[t].c1);
    if (t == 0) {
      mul_wide<C>(v0, a[t].c0.v, b[t].c0.v);
      mul_wide<C>(v1, a[t].c1.v, b[t].c1.v);
      mul_wide<C>(s, sa.v, sb.v);
    } else {
      mul_wide<C>(tmp, a[t].c0.v, b[t].c0.v); w_add<W>(v0, v0, tmp);
      mul_wide<C>(tmp, a[t].c1.v, b[t].c1.v); w_add<W>(v1, v1, tmp);
      mul_wide<C>(tmp, sa.v, sb.v); w_add<W>(s, s, tmp);
    }
  }
  w_sub<W>(s, s, v0);
  w_sub<W>(s, s, v1);
  w_add<W>(v0, v0, C::P2W6);
  w_sub<W>(v0, v0, v1);
  Fp2<C> r;
  r.c0 = redc_k<C, 3>(v0);
  r.c1 = redc_k<C, 3>(s);
  return r;
}
// B-form: all twenty limb-level products of the five terms go straight into one set of 64-bit column accumulators per
// output half (schoolbook over i, the subtraction as a "fat" negation): no carries, no wide additions, no fix-ups.
__device__ __forceinline__ F28x2 dot5_b(const F28x2 (&a)[5], const F28x2 (&b)[5]) {
  F28x2 r;
  {
    uint64_t c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      F28 nb1;
#pragma unroll
      for (int i = 0; i < 10; ++i) nb1.v[i] = FAT28[i] - b[t].c1.v[i];
      acc_prod(c, a[t].c0, b[t].c0);
      acc_prod(c, a[t].c1, nb1);
    }
    r.c0 = redc28(c);
  }
  {
    uint64_t c[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) c[k] = 0;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      acc_prod(c, a[t].c0, b[t].c1);
      acc_prod(c, a[t].c1, b[t].c0);
    }
    r.c1 = redc28(c);
  }
  return r;
}
__global__ void __launch_bounds__(64) mb_dot_a(Fp2<BN254>* io, int iters, size_t n) {
  const size_t t = blockIdx.x * 64 + threadIdx.x;
  Fp2<BN254> a[5], b[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { a[k] = io[(t + 3 * k) % n]; b[k] = io[n + (t + 5 * k) % n]; }
  for (int i = 0; i < iters; ++i) {
    Fp2<BN254> r = dot5_a(a, b);
    a[i % 5] = r;                       // keep a dependency so nothing is hoisted
  }
  io[t] = f2_add<BN254>(a[0], f2_add<BN254>(a[1], f2_add<BN254>(a[2], f2_add<BN254>(a[3], a[4]))));
}
__global__ void __launch_bounds__(64) mb_dot_b(F28x2* io, int iters, size_t n) {
  const size_t t = blockIdx.x * 64 + threadIdx.x;
  F28x2 a[5], b[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) { a[k] = io[(t + 3 * k) % n]; b[k] = io[n + (t + 5 * k) % n]; }
  for (int i = 0; i < iters; ++i) {
    F28x2 r = dot5_b(a, b);
    a[i % 5] = r;
  }
  F28x2 s = a[0];
#pragma unroll
  for (int c = 1; c < 5; ++c)
#pragma unroll
    for (int k = 0; k < 10; ++k) { s.c0.v[k] += a[c].c0.v[k]; s.c1.v[k] += a[c].c1.v[k]; }
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
    // random reduced field elements for the cross-check: limbs below the modulus' top limb
    {
      Fp2<BN254>* h = (Fp2<BN254>*)malloc(2 * n * sizeof(Fp2<BN254>));
      uint64_t s = 0x9e3779b97f4a7c15ull;
      for (size_t i = 0; i < 2 * n; ++i)
        for (int k = 0; k < 8; ++k) {
          s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].c0.v[k] = k == 7 ? (uint32_t)s & 0x1fffffffu : (uint32_t)s;
          s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].c1.v[k] = k == 7 ? (uint32_t)s & 0x1fffffffu : (uint32_t)s;
        }
      hipMemcpy(d, h, 2 * n * sizeof(Fp2<BN254>), hipMemcpyHostToDevice);
      unsigned long long* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
      mb_check<<<(unsigned)(n / 64), 64>>>(d, n, bad);
      unsigned long long hb = 0; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
      printf("C vs A cross-check: %llu mismatches in %zu products\n", hb, n * 16);
      free(h);
    }
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
      float t1 = run([&] { mb_c<1><<<blocks, 64>>>(d, iters, n); }, 3);
      float t2 = run([&] { mb_c<2><<<blocks, 64>>>(d, iters, n); }, 3);
      printf("C carry-out asm blocks=%5d  1 chain: %8.3f ms = %6.2f G f2mul/s | 2 chains: %8.3f ms = %6.2f G f2mul/s\n", blocks, t1,
             (double)blocks * 64 * iters / t1 / 1e6, t2, (double)blocks * 64 * iters * 2 / t2 / 1e6);
    }
    hipFree(d);
  }
  {
    const int iters = 100;            // multiple of 5: the a[i % 5] rotation is resolved at compile time after unrolling by 5
    Fp2<BN254>* da; hipMalloc(&da, 2 * n * sizeof(Fp2<BN254>)); hipMemset(da, 1, 2 * n * sizeof(Fp2<BN254>));
    F28x2* db; hipMalloc(&db, 2 * n * sizeof(F28x2)); hipMemset(db, 1, 2 * n * sizeof(F28x2));
    for (int blocks : {2048, 8192, 16384}) {
      float ta = run([&] { mb_dot_a<<<blocks, 64>>>(da, iters, n); }, 3);
      float tb = run([&] { mb_dot_b<<<blocks, 64>>>(db, iters, n); }, 3);
      printf("D five-term dot  blocks=%5d  A (32-bit limbs, lazy): %8.3f ms = %6.2f G dot/s | B (28-bit limbs, columns): %8.3f ms = %6.2f G dot/s  (B/A = %.2f)\n",
             blocks, ta, (double)blocks * 64 * iters / ta / 1e6, tb, (double)blocks * 64 * iters / tb / 1e6, ta / tb);
    }
    hipFree(da); hipFree(db);
  }
  return 0;
}
