// Miller-loop kernels (templates; instantiated per curve by k_miller_bn.hip / k_miller_bls.hip).
//
//   k_miller_ab64   fused producer/consumer block, 64 pairings, 32-bit limbs (both curves; the shape used when one
//                   verification has the machine to itself)
//   k_miller_s60    fused producer/consumer block, 60 pairings, 28-bit-limb consumer (alt-bn128, throughput mode)
// DBG template values other than 0 are timing variants with WRONG results; they are only instantiated in
// development builds (-DBGLS_DEV) and never reachable from the shipped library.
#pragma once
#include "dev_common.hpp"
#include "coop.hpp"
#include "coop_r28.hpp"

namespace bgls {

// ======================================================================= cooperative (v2) kernels
// One wave per block, 10 groups of 6 lanes; see coop.hpp.  Partial products are kept in the
// "w-basis" layout: 6 consecutive Fp2 per Fp12, coefficient j of w^j.
template <class C>
struct CoopLane {
  int lane, g, j, gb;
  bool live;
  __device__ __forceinline__ CoopLane() {
    lane = threadIdx.x;
    live = lane < 60;
    g = live ? lane / 6 : 9;
    j = live ? lane % 6 : lane - 60;
    gb = g * Coop<C>::GROUP_DW;
  }
};


// ---- the producer's fixed operands, parked in HBM ----------------------------------------------------------------------
// A producer lane needs its key Q (Fp2 x, y) only in the addition steps and its hash point P (x, y) only to scale the line
// coefficients, but as loop-invariant values they sit in registers through every Fp2 product and push the point-step
// temporaries out to the private stack (BLS12-381: 72 of 256 registers; 14.6 GB of spill traffic per launch, 870x the
// algorithmic bytes, profiles/r2 interim).  They are parked in a lane-interleaved workspace instead (element k of lane l
// of block b at ((b * QP_DW + k) * 64 + l)) and re-read through a laundered pointer where they are used: 96-288 bytes per
// step and lane from L2 instead of spills around every product.
template <class C>
struct QP {
  static constexpr int L = C::L;
  static constexpr int QX = 0, QY = 2 * L, PX = 4 * L, PY = 5 * L, DW = 6 * L;
  static __device__ __forceinline__ void st_fp(u32* base, int off, const Fp<C>& a) {
#pragma unroll
    for (int k = 0; k < L; ++k) base[(size_t)(off + k) * 64] = a.v[k];
  }
  static __device__ __forceinline__ Fp<C> ld_fp(const u32* base, int off) {
    Fp<C> r;
#pragma unroll
    for (int k = 0; k < L; ++k) r.v[k] = base[(size_t)(off + k) * 64];
    return r;
  }
  static __device__ __forceinline__ void park(u32* base, const Aff<F2<C>>& Q, const Aff<F1<C>>& P) {
    st_fp(base, QX, Q.x.c0); st_fp(base, QX + L, Q.x.c1);
    st_fp(base, QY, Q.y.c0); st_fp(base, QY + L, Q.y.c1);
    st_fp(base, PX, P.x); st_fp(base, PY, P.y);
  }
  static __device__ __forceinline__ const u32* launder(const u32* p) {       // opaque to the optimiser: no hoisting out of the step loop
    asm volatile("" : "+v"(p));
    return p;
  }
  static __device__ __forceinline__ Fp2<C> ld_f2(const u32* base, int off) { return {ld_fp(base, off), ld_fp(base, off + L)}; }
};

// ---- 64 pairings per block, 256 VGPRs (two waves per SIMD, 1024 blocks = exactly one 2^16 batch) ----------
// Same producer/consumer scheme as k_miller_ab, but the producer wave uses all 64 lanes (lane l feeds line
// slot l/10 of group l%10, so groups 0..3 fold seven lines and the others six plus a constant 1), the
// (-sigma, g2) pair needs no point steps (k_gen_lines table, scaled by lane 0 of block 0), and nothing
// spills: the point-step temporaries fit the 256-register budget.
template <class C, bool R28 = false>
struct Coop64 {
  static constexpr int S2 = R28 ? R28_S2 : 2 * C::L;        // R28: ten 28-bit limbs per field element (coop_r28.hpp), 38.4 KB per block
  static constexpr int NENT = 18;            // per group and buffer: 3 line pairs x 5 coefficients + 1 single line x 3
  // BLS12-381: xi = 1+i costs two additions, so the accumulator region keeps the plain coefficients only and
  // the wrap-around factor is applied after the load; that is what lets four blocks share a CU's 160 KB.
  static constexpr bool XF = C::XI_RE == 1;
  static constexpr int RBN = XF ? 6 : 12;
  static constexpr int RB = 0, RL = RBN * S2, RL2 = (RBN + NENT) * S2;
  static constexpr int GROUP_DW = (RBN + 2 * NENT) * S2;
  static constexpr int BLOCK_BYTES = 10 * GROUP_DW * 4;
};

// Lanes 0..59 of the producer wave are 30 neighbour pairs (lanes 6g+2m, 6g+2m+1 -> pair m of group g): each pair
// multiplies its two lines into one 5-coefficient element (coop_write_line_pair).  Lanes 60..63 feed the
// single-line slot of groups 0..3, the (-sigma, g2) line goes to the single slot of group 4, groups 5..9 keep
// the constant 1 there.  The consumer folds 3 five-term elements + 1 three-term line per step.
// DBG (development only, BGLS_AB64_DBG): 1 = producer work only, 2 = consumer work only -- wrong results, used to
// time the two halves of the pipeline separately.
// R28 (alt-bn128): the consumer works on 28-bit limbs (coop_r28.hpp); the producer converts what it stores.
template <class C, int DBG = 0, bool R28 = false>
__global__ void __launch_bounds__(128, 2) k_miller_ab64(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at,
                                                        const LineCoeffs<C>* gen_lines, Fp2<C>* out, uint32_t* flags, unsigned swap_mask, u32* qp) {
  typedef Coop64<C, R28> K;
  // The two waves of a block land on different SIMDs and every CU hosts four blocks: if wave 0 were the producer
  // everywhere, two SIMDs of a CU would carry two producers and the other two would carry two consumers, and the kernel
  // would run at the pace of the heavier role.  Blocks selected by swap_mask exchange the roles, so each SIMD carries
  // one producer and one consumer.
  const int wave = (int)(threadIdx.x >> 6) ^ ((blockIdx.x & swap_mask) ? 1 : 0);
  const int lane = threadIdx.x & 63;
  if (wave == 0) {
    // ---------------- producer: 64 pairings, one per lane
    const size_t idx = (size_t)blockIdx.x * 64 + lane;
    const bool paired = lane < 60;
    const int tg = paired ? lane / 6 : lane - 60;
    const int j = paired ? lane % 6 : 0;
    const int tgb = tg * K::GROUP_DW;
    Aff<F2<C>> Q;
    Aff<F1<C>> P;
    bool valid = idx < n;
    if (valid) {
      bool ok = g2_from_bytes<C>(Q, g2s + idx * 4 * C::FP_BYTES);
      ok = ok && aff_on_curve<F2<C>>(Q);
      if (!ok) atomicOr(flags, FLAG_ENC);
      P = g1s[idx];
      valid = !P.inf && !Q.inf;
    }
    if (!valid) {
      Q.x = f2_load<C>(C::G2);
      Q.y = f2_load<C>(C::G2 + 2 * C::L);
      P.x = fp_load<C>(C::G1X);
      P.y = fp_load<C>(C::G1Y);
    }
    const bool sig_lane = sig_at >= 0 && blockIdx.x == 0 && lane == 0;
    Aff<F1<C>> S;
    bool sig_valid = false;
    if (sig_lane) {
      S = g1s[sig_at];
      sig_valid = !S.inf;
    }
    // single-line slots without an owner hold the constant 1 in both buffers
    if (lane >= 4 && lane < 10) {
      for (int b = 0; b < 2; ++b) {
        const LReg r = {lane * K::GROUP_DW + (b ? K::RL2 : K::RL), K::NENT};
        st_entry<C, R28>(r, 15, f2_one<C>());
        st_entry<C, R28>(r, 16, f2_zero<C>());
        st_entry<C, R28>(r, 17, f2_zero<C>());
      }
    }
    G2Proj<C> T = {Q.x, Q.y, f2_one<C>()};
    u32* const myqp = qp + (size_t)blockIdx.x * QP<C>::DW * 64 + lane;
    QP<C>::park(myqp, Q, P);
    int buf = 0, step = 0;
    auto publish = [&](LineCapture<C>& cap) __attribute__((always_inline)) {
      if constexpr (DBG == 2) { ++step; __syncthreads(); buf ^= 1; return; }
      if (!valid) { cap.e[0] = f2_one<C>(); cap.e[1] = f2_zero<C>(); cap.e[2] = f2_zero<C>(); }
      const LReg r = {tgb + (buf ? K::RL2 : K::RL), K::NENT};
      if (paired) {
        coop_write_line_pair<C, R28>(r, j, cap.e);
      } else {
        st_entry<C, R28>(r, 15, cap.e[0]);
        st_entry<C, R28>(r, 16, cap.e[1]);
        st_entry<C, R28>(r, 17, cap.e[2]);
      }
      if (sig_lane && sig_valid) {
        const LineCoeffs<C> l = gen_lines[step];
        LineEmitter<C, R28> em{LReg{4 * K::GROUP_DW + (buf ? K::RL2 : K::RL), K::NENT}, 5, S.x, S.y, true, true};   // entries 15..17
        em(0, l.c0);
        em(1, l.c1);
        em(2, l.c2);
      }
      ++step;
      wave_sync();
      if constexpr (DBG != 3) __syncthreads();
      buf ^= 1;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      {
        const u32* q = QP<C>::launder(myqp);
        const Fp<C> px = QP<C>::ld_fp(q, QP<C>::PX), py = QP<C>::ld_fp(q, QP<C>::PY);
        LineCapture<C> cap{{}, px, py};
        if constexpr (DBG != 2) dbl_step_emit<C>(T, cap);
        publish(cap);
      }
      const int d = C::LOOP_NAF[i];
      if (d != 0) {
        const u32* q = QP<C>::launder(myqp);
        const Fp<C> px = QP<C>::ld_fp(q, QP<C>::PX), py = QP<C>::ld_fp(q, QP<C>::PY);
        const Fp2<C> qx = QP<C>::ld_f2(q, QP<C>::QX), qy = QP<C>::ld_f2(q, QP<C>::QY);
        LineCapture<C> cap{{}, px, py};
        if constexpr (DBG != 2) add_step_emit<C>(T, qx, d > 0 ? qy : f2_neg<C>(qy), cap);
        publish(cap);
      }
    }
    if constexpr (C::CURVE_ID == 0) {
      {
        const u32* q = QP<C>::launder(myqp);
        const Fp<C> px = QP<C>::ld_fp(q, QP<C>::PX), py = QP<C>::ld_fp(q, QP<C>::PY);
        Fp2<C> x1 = f2_mul<C>(f2_conj<C>(QP<C>::ld_f2(q, QP<C>::QX)), gamma_const<C>(1, 2));
        Fp2<C> y1 = f2_mul<C>(f2_conj<C>(QP<C>::ld_f2(q, QP<C>::QY)), gamma_const<C>(1, 3));
        LineCapture<C> cap{{}, px, py};
        if constexpr (DBG != 2) add_step_emit<C>(T, x1, y1, cap);
        publish(cap);
      }
      {
        const u32* q = QP<C>::launder(myqp);
        const Fp<C> px = QP<C>::ld_fp(q, QP<C>::PX), py = QP<C>::ld_fp(q, QP<C>::PY);
        Fp2<C> x2 = f2_mul<C>(QP<C>::ld_f2(q, QP<C>::QX), gamma_const<C>(2, 2));
        Fp2<C> y2 = f2_neg<C>(f2_mul<C>(QP<C>::ld_f2(q, QP<C>::QY), gamma_const<C>(2, 3)));
        LineCapture<C> cap{{}, px, py};
        if constexpr (DBG != 2) add_step_emit<C>(T, x2, y2, cap);
        publish(cap);
      }
    }
    if (DBG == 0 && valid && f2_is_zero<C>(T.Z)) atomicOr(flags, FLAG_DEGENERATE);     // a degenerate point step leaves Z = 0 (miller_x.hpp)
  } else {
    // ---------------- consumer: 10 groups x 6 lanes; per step 3 line pairs + 1 single line
    const bool live = lane < 60;
    const int g = live ? lane / 6 : 9;
    const int j = live ? lane % 6 : lane - 60;
    const int gb = g * K::GROUP_DW;
    if constexpr (R28) {
      // ---- 28-bit-limb consumer: same schedule, every dot product a pile of carry-free column accumulations
      static_assert(C::CURVE_ID == 0 && C::TWIST_D, "alt-bn128 only");
      const int rbo = gb + K::RB;
      F28x2 fj;
      {
        const F28 one = r28_load<C>(C::R28_ONE);
#pragma unroll
        for (int q = 0; q < 10; ++q) { fj.c0.v[q] = j == 0 ? one.v[q] : 0u; fj.c1.v[q] = 0u; }
      }
      coop_publish28<C>(rbo, j, fj, live);
      int buf = 0;
      auto fold = [&]() {
        if constexpr (DBG == 1) { buf ^= 1; return; }
        const int rlo = gb + (buf ? K::RL2 : K::RL);
#pragma unroll 1
        for (int m = 0; m < 3; ++m) {
          fj = coop_dot28<C, 5>(rlo, 5 * m, rbo, j, COOP_SH_D5);
          coop_publish28<C>(rbo, j, fj, live);
        }
        fj = coop_dot28<C, 3>(rlo, 15, rbo, j, COOP_SH_D);
        coop_publish28<C>(rbo, j, fj, live);
        buf ^= 1;
      };
#pragma unroll 1
      for (int i = 1; i < C::LOOP_LEN; ++i) {
        if constexpr (DBG != 3) __syncthreads();
        if constexpr (DBG != 1) {
          fj = coop_sqr_sym28<C>(rbo, j);
          coop_publish28<C>(rbo, j, fj, live);
        }
        fold();
        if (C::LOOP_NAF[i] != 0) {
          if constexpr (DBG != 3) __syncthreads();
          fold();
        }
      }
      if constexpr (DBG != 3) __syncthreads();
      fold();
      if constexpr (DBG != 3) __syncthreads();
      fold();
      if (live) out[((size_t)blockIdx.x * 10 + g) * 6 + j] = from_r28<C>(fj);
      return;
    }
    const LReg rb = {gb + K::RB, 12};
    Fp2<C> fj = j == 0 ? f2_one<C>() : f2_zero<C>();
    coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
    int buf = 0;
    auto fold = [&]() {
      if constexpr (DBG == 1) { buf ^= 1; return; }
      const LReg rl = {gb + (buf ? K::RL2 : K::RL), K::NENT};
#pragma unroll 1
      for (int m = 0; m < 3; ++m) {
        fj = coop_dot_inl<C, 5, K::XF>(rl, 5 * m, 1, rb, j, C::TWIST_D ? COOP_SH_D5 : COOP_SH_M5);
        coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
      }
      fj = coop_dot_inl<C, 3, K::XF>(rl, 15, 1, rb, j, C::TWIST_D ? COOP_SH_D : COOP_SH_M);
      coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
      buf ^= 1;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      if constexpr (DBG != 3) __syncthreads();
      if constexpr (DBG != 1) {
        fj = coop_sqr_sym_inl<C, K::XF>(rb, j);
        coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
      }
      fold();
      if (C::LOOP_NAF[i] != 0) {
        if constexpr (DBG != 3) __syncthreads();
        fold();
      }
    }
    if constexpr (C::CURVE_ID == 0) {
      if constexpr (DBG != 3) __syncthreads();
      fold();
      if constexpr (DBG != 3) __syncthreads();
      fold();
    } else {
      if (j & 1) fj = f2_neg<C>(fj);                      // x < 0: f^(p^6), w -> -w
    }
    if (live) out[((size_t)blockIdx.x * 10 + g) * 6 + j] = fj;
  }
}

// Step count and line layout of a Miller loop's line table (the prepared key sets of prepared.hpp keep one per key; the decoupled
// k_lines / k_fold kernels that first used it -- the Miller loop as two kernels joined through HBM, rounds 2-4: never faster than
// the fused kernels -- were removed in round 5).
template <class C, bool R28>
struct LineTab {
  static constexpr int S2 = R28 ? R28_S2 : 2 * C::L;
  static constexpr int LINE_DW = 3 * S2;
  // number of line steps of the Miller loop (doublings + additions [+ 2 Frobenius steps on alt-bn128])
  static constexpr int nsteps() {
    int s = 0;
    for (int i = 1; i < C::LOOP_LEN; ++i) s += 1 + (C::LOOP_NAF[i] != 0 ? 1 : 0);
    return s + (C::CURVE_ID == 0 ? 2 : 0);
  }
  static constexpr int NSTEPS = nsteps();
};

}  // namespace bgls
