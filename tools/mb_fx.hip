// Where does a cooperative Fp12 product of finalx.hpp spend its time?  One block of 128 threads, `reps` products in a row;
// s_memtime around the stages of an instrumented copy of fx_mul.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibgls_amd/csrc -Iinclude tools/mb_fx.hip -o tools/mb_fx.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "dev_common.hpp"
#include "finalx.hpp"
using namespace bgls;

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

template <class C>
__global__ void __launch_bounds__(128) k_probe(int reps, unsigned long long* out) {
  typedef FX<C> E;
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  if (tid < 6) {
    X2<C, SX_T> v = {sx_const<C>(C::RX_ONE), sx_const<C>(C::RX_R2)};
    fx_put<C>(FE_F, tid, v);
    fx_put<C>(FE_X, tid, v);
  }
  __syncthreads();
  unsigned long long tl = 0, tm = 0, ts = 0, tb1 = 0, t2 = 0, tb2 = 0, tall = 0;
  const unsigned long long t00 = now();
  for (int r = 0; r < reps; ++r) {
    const unsigned long long a0 = now();
    Sx<C, SX_T> p;
    X2<C, SX_T> x;
    Sx<C, SX_T> ya, yb;
    if (lane < 36) {
      const int j = lane / 6, t = lane % 6;
      int k = j - t;
      const int wrap = k < 0 ? 1 : 0;
      k += 6 * wrap;
      x = fx_ld2<C>(E::coef(FE_F, t, 0));
      const int yo = E::coef(FE_X, k, wrap);
      ya = fx_ld<C>(yo + (h ? E::HS : 0));
      yb = fx_ld<C>(yo + (h ? 0 : E::HS));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long a1 = now();
    if (lane < 36) {
      const i32 sg = h ? 0 : -1;
      const i32* const cols[2] = {ya.v, yb.v};
      p = sx_montr<C, 2, 2 * SX_T * SX_T>(cols, [&](int q, int i) { return q == 0 ? x.c0.v[i] : (x.c1.v[i] ^ sg) - sg; });
    }
    asm volatile("" ::: "memory");
    const unsigned long long a2 = now();
    if (lane < 36) fx_st<C>(E::SCR + lane * E::ES + h * E::HS, p);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long a3 = now();
    __syncthreads();
    const unsigned long long a4 = now();
    if (tid < 12) {
      const int j = tid >> 1, hh = tid & 1;
      const int o = E::SCR + (6 * j) * E::ES + hh * E::HS;
      const Sx<C, SX_T> t0 = fx_ld<C>(o), t1 = fx_ld<C>(o + E::ES), t2_ = fx_ld<C>(o + 2 * E::ES);
      const Sx<C, SX_T> t3 = fx_ld<C>(o + 3 * E::ES), t4 = fx_ld<C>(o + 4 * E::ES), t5 = fx_ld<C>(o + 5 * E::ES);
      const Sx<C, SX_T> mine = sx_norm<C>(sx_add<C>(sx_add<C>(sx_add<C>(t0, t1), sx_add<C>(t2_, t3)), sx_add<C>(t4, t5)));
      Sx<C, SX_T> other;
#pragma unroll
      for (int i = 0; i < C::RX_NL; ++i) other.v[i] = __builtin_amdgcn_update_dpp(0, mine.v[i], 0xB1, 0xF, 0xF, true);
      fx_st<C>(E::coef(FE_F, j, 0) + hh * E::HS, mine);
      fx_st<C>(E::coef(FE_F, j, 1) + hh * E::HS, fx_mulxi_half<C>(mine, other, hh == 1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long a5 = now();
    __syncthreads();
    const unsigned long long a6 = now();
    tl += a1 - a0; tm += a2 - a1; ts += a3 - a2; tb1 += a4 - a3; t2 += a5 - a4; tb2 += a6 - a5;
  }
  tall = now() - t00;
  if (tid == 0 || tid == 64) {
    unsigned long long* o = out + (tid ? 8 : 0);
    o[0] = tl; o[1] = tm; o[2] = ts; o[3] = tb1; o[4] = t2; o[5] = tb2; o[6] = tall;
  }
  // the real thing, timed as a whole
  __syncthreads();
  const unsigned long long b0 = now();
  for (int r = 0; r < reps; ++r) fx_mul<C>(FE_F, FE_F, FE_X);
  if (tid == 0) out[7] = now() - b0;
}

template <class C>
void run(const char* name) {
  unsigned long long* d;
  hipMalloc(&d, 16 * 8);
  const int reps = 200;
  for (int it = 0; it < 2; ++it) k_probe<C><<<1, 128, FX<C>::LDS_BYTES>>>(reps, d);
  unsigned long long h[16];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* n[] = {"operand loads", "montr (300/588 mads)", "store", "barrier 1", "stage 2 (sum, carry, xi)", "barrier 2", "loop total", "fx_mul (shipped)"};
  for (int w = 0; w < 2; ++w) {
    printf("%s wave %d:", name, w);
    for (int k = 0; k < (w ? 7 : 8); ++k) printf("  %s %.0f", n[k], (double)h[8 * w + k] / reps);
    printf("   [s_memtime ticks per product; 100 MHz]\n");
  }
}
int main() {
  run<BN254>("BN254");
  run<BLS381>("BLS381");
  return 0;
}
