// Latency form of the Miller loop on the carry-free limbs: ONE pairing per block (the two-pairing tail of verifyMultiSignature,
// bgls/bgls.go:59-70,89-92; the reference's own n = 64 benchmark shape).  Same roles, hand-over and results as k_miller_lat
// (k_tail.inc), which it replaces: wave 1 computes the G2 point step as 3 (doubling) / 4 (addition) rounds of independent Fp2
// products, one product per LANE PAIR (rx_pair.hpp: the even lane holds the real parts, the odd lane the imaginary parts; two
// limb products and one reduction per lane) with the operands exchanged through LDS; wave 0 keeps the Fp12 accumulator on the
// 36-lane arithmetic of finalx.hpp (fx_mul1) and folds the line of step s while wave 1 computes step s + 1.  On a lone wave
// the rows of a carry-free product are independent multiplier instructions, where the 32-bit form walks a carry chain: a
// round costs ~1 500 clocks instead of ~6 000.  Block `n` (when sig_at >= 0) is the (-sigma, g2) pair on the pre-computed
// generator lines.  out: one w-basis Fp12 (6 Fp2, the library's 32-bit Montgomery form) per block.
#include <stdlib.h>
#include "dev_common.hpp"
#include "pairing.hpp"
#include "finalx.hpp"
#include "rx_jacpair.hpp"
#include "constants_latx_gen.hpp"
#include "launch.hpp"
#include "launch_tail.hpp"

namespace bgls {

#ifdef LATX_DBG
// development only (tools/exp/latx_steps.sh): 100 MHz time stamps of the hand-overs, [block 0 | the signature block][producer before /
// after the barrier, consumer before / after][event]
__device__ unsigned long long g_latx_t[2][4][160];
__device__ unsigned long long g_latx_c[2][4][160];    // the same events on the shader clock (s_memtime)
#define LATX_T(kind, idx) do { if ((threadIdx.x & 63) == 0 && (idx) < 160) { g_latx_t[is_sig ? 1 : 0][kind][idx] = wall_clock64(); g_latx_c[is_sig ? 1 : 0][kind][idx] = clock64(); } } while (0)
__device__ unsigned long long g_latx_r[12][160];      // shader clock inside the general block's point steps
#define LATX_R(k) do { if (!is_sig && (threadIdx.x & 63) == 0 && step < 160) g_latx_r[k][step] = clock64(); } while (0)
#else
#define LATX_T(kind, idx) do { } while (0)
#define LATX_R(k) do { } while (0)
#endif

// one block = one pairing: pair `blk` of the batch (blk == n with sig_at >= 0: the (-sigma, g2) pair); the six w-basis coefficients
// of the Miller value go to out[blk * 6 ..]
// AW = waves of the accumulator: 1 (round 3: the one-wave product fx_mul1, ~3.5 us on alt-bn128) or 2 (round 4: fx_mul2w, the
// two-wave product with LDS hand-overs instead of block barriers, ~1.6 us); the point steps run on wave AW.
template <class C, int AW>
__device__ __forceinline__ void miller_latx_block(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at,
                                                  const LineCoeffs<C>* gen_lines, Fp2<C>* out, uint32_t* flags, size_t blk) {
  typedef FX<C> E;
  constexpr int ES = E::ES, HS = E::HS, N = C::RX_NL;
  enum { S_F = 0, S_L0 = 2, S_L1 = 3, S_PA = 4 };                      // accumulator, two line buffers, producer scratch (slots 4..7: 48 Fp2)
  enum { PX = 0, PY, PZ, PZT, PZU, PQX, PQY, PT0, PT1,                // the point (Zt = beta Z, Zu = xi beta Z, see the steps), Q, (xP, 0), (yP, 0)
         PB, PC, PS, PJ, PM, PE, PW, PV, PG, PE2,                       // doubling: round 1 (consecutive, in product order), round 2
         PU0, PU1, PD, PC2, PLT, PZL, PXL, PZH, PLY, PZTL, PV0, PV1, PZUL,   // addition: round 1, round 2 (consecutive, in product order)
         PN0, PNEND = PN0 + 5 };                                        // addition: round 3
  static_assert(PNEND <= 48, "producer scratch fits four slots");
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_sig = sig_at >= 0 && blk == n;
  const int pbase = E::coef(S_PA, 0, 0);
  // (no static LDS in a kernel of this file: it would sit in front of the dynamic array and shift every slot off its 16-byte
  // alignment -- the 16-byte LDS accesses of fx_ld / fx_st then cost a third more: measured on k_finalx, round 4.  The word lives
  // behind fx_mul2w's hand-over words.)
  extern __shared__ u32 lds_words[];
  volatile int& s_valid = *reinterpret_cast<volatile int*>(lds_words + E::LDS_DW + 4);
  const bool odd = lane & 1;
  const int q = lane >> 1;                                              // the lane pair's product index within a round
  // own half of scratch entry e / store it (values are tight: reductions' outputs or carried sums)
  auto LD = [&](int e) { return fx_ld<C>(pbase + e * ES + (odd ? HS : 0)); };
  auto ST = [&](int e, const Sx<C, SX_T>& v, bool active) {
    if (active) fx_st<C>(pbase + e * ES + (odd ? HS : 0), v);
  };
  // own half of line coefficient `which` into line buffer `buf`.  The line is the product's FIRST factor (fx_mul*: term t of
  // coefficient j is a_t b_(j-t), only b is read as a xi multiple where the index wraps), so no xi multiple of it is formed.
  // D-type  k = 0: c0 yP, 1: c1 xP, 3: c2;   M-type  0: c2, 2: c1 xP, 3: c0 yP
  auto put_line = [&](int buf, int which, const Sx<C, SX_T>& v, bool active) {
    const int k = which == 0 ? (C::TWIST_D ? 0 : 3) : which == 1 ? (C::TWIST_D ? 1 : 2) : (C::TWIST_D ? 3 : 0);
    if (active) fx_st<C>(E::coef(buf ? S_L1 : S_L0, k, 0) + (odd ? HS : 0), v);
  };
  if (wave == 0) {
    if (AW == 2 && lane == 63) fx_mul2w_init<C>();
    if (lane < 6) {
      const Sx<C, SX_T> zero = ux_to_sx<C>(ux_zero<C>());
      const X2<C, SX_T> z2 = {zero, zero}, one = {sx_const<C>(C::RX_ONE), zero};
      fx_put<C>(S_F, lane, lane == 0 ? one : z2);
      fx_put<C>(S_L0, lane, z2);
      fx_put<C>(S_L1, lane, z2);
    }
  } else if (wave == AW && lane < 2) {
    // lane pair 0 parses this block's pair (block n: -sigma against the generator)
    Aff<F1<C>> P = g1s[is_sig ? (size_t)sig_at : blk];
    AffP<C> Q;
    bool valid;
    if (is_sig) {
      Q.x = ux_to_sx<C>(to_ux<C>(fp_load<C>(C::G2 + (odd ? C::L : 0))));
      Q.y = ux_to_sx<C>(to_ux<C>(fp_load<C>(C::G2 + 2 * C::L + (odd ? C::L : 0))));
      Q.inf = false;
    } else {
      bool ok = affp_from_bytes<C>(Q, g2s + blk * 4 * C::FP_BYTES, odd);
      ok = affp_on_curve<C>(Q, odd) && ok;
      if (!ok && !odd) atomicOr(flags, FLAG_ENC);
    }
    valid = !P.inf && !Q.inf;
    if (!valid) {
      Q.x = ux_to_sx<C>(to_ux<C>(fp_load<C>(C::G2 + (odd ? C::L : 0))));
      Q.y = ux_to_sx<C>(to_ux<C>(fp_load<C>(C::G2 + 2 * C::L + (odd ? C::L : 0))));
      P.x = fp_load<C>(C::G1X);
      P.y = fp_load<C>(C::G1Y);
    }
    const Sx<C, SX_T> zero = ux_to_sx<C>(ux_zero<C>());
    ST(PX, Q.x, true); ST(PY, Q.y, true); ST(PZ, pair_one<C>(odd), true);
    ST(PZT, sx_const<C>(odd ? LatxK<C>::BETA_IM : LatxK<C>::BETA_RE), true);
    ST(PZU, sx_const<C>(odd ? LatxK<C>::XIBETA_IM : LatxK<C>::XIBETA_RE), true);
    ST(PQX, Q.x, true); ST(PQY, Q.y, true);
    ST(PT0, odd ? zero : ux_to_sx<C>(to_ux<C>(P.x)), true);             // (xP, 0), (yP, 0): Fp2 operands of the scaling products
    ST(PT1, odd ? zero : ux_to_sx<C>(to_ux<C>(P.y)), true);
    if (!odd) s_valid = valid ? 1 : 0;
  }
  __syncthreads();
  const bool valid = s_valid != 0;
  if (wave == AW) {
    // ------------------------------------------------ producer: point steps on lane pairs, or generator lines
    const Sx<C, SX_T> xP = LD(PT0), yP = LD(PT1);                        // own halves of (xP, 0), (yP, 0)
    Sx<C, SX_T> xPs, yPs;                                                // the same scalars on both lanes (the odd lane holds zero)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      xPs.v[i] = xP.v[i] + pair_swap1(xP.v[i]);
      yPs.v[i] = yP.v[i] + pair_swap1(yP.v[i]);
    }
    int buf = 0, step = 0;
    auto publish = [&]() {                                    // line of this step is in the buffer: hand it over
      LATX_T(0, step);
      ++step;
      __syncthreads();
      LATX_T(1, step - 1);
      buf ^= 1;
    };
    auto F = [&](const auto& v) { return sx_normf<C>(v); };                      // any bound below 128 -> almost tight
    auto T = [&](const Sx<C, SX_T>& v) { return sx_as<SX_F, C>(v); };
    auto gen_step = [&]() {                                   // (-sigma, g2): scale the pre-computed coefficients
      const LineCoeffs<C> l = gen_lines[step];
      const Fp2<C> cq = q == 0 ? l.c0 : q == 1 ? l.c1 : l.c2;
      const Sx<C, SX_T> own = ux_to_sx<C>(to_ux<C>(odd ? cq.c1 : cq.c0));
      const Sx<C, SX_T> s = sx_select<C>(q == 0, yPs, xPs);
      const Sx<C, SX_T> r = pair_muls<C>(own, s);
      put_line(buf, q == 0 ? 0 : q == 1 ? 1 : 2, q == 2 ? own : r, q < 3);
      wave_sync();
      publish();
    };
    // The point steps are those of rx_pair.hpp / pairing.hpp rearranged for a LONE wave, where what counts is the length of the
    // instruction stream (a round of independent products costs the same whether 3 or 12 lane pairs work in it, and every lane
    // executes whatever any lane needs): the same values mod p, hence the same Miller value.
    //  - the doubling step's E = 3 b' Z^2 was a product by a constant in a round of its own between Z^2 and the products that
    //    need E.  3 b' = xi beta^2 on both curves (constants_latx_gen.hpp), so with Zt = beta Z and Zu = xi beta Z carried beside
    //    Z, E = Zu Zt is one of the products of the first round; Zt' = B (2 Y Zt), Zu' = B (2 Y Zu).  Two rounds instead of 3.
    //  - the addition step's X', Y', Z' are written out as sums of products of values two rounds deep
    //    (X' = D (D - 2 X la) + (Z la) C, Y' = (la th)(3 X la - D) - (Z th) C - D (la Y), Z' = (Z la) D, with D = la^2,
    //    C = th^2), each product on its own lane pair and the sums formed where they are read.  Three rounds instead of 4.
    //  - the factors of a round's products are small integer combinations of scratch entries (B - 3 E, (B + 3 E) / 2, S - B - C,
    //    ..): instead of every lane forming every combination and selecting its own, a lane forms ONE combination
    //    c0 e0 + c1 e1 + c2 e2 with its own entries and coefficients (lin3).
    // own half of c0 [e0] + c1 [e1] + c2 [e2], halved mod p when `half` (sx_half); |c0| + |c1| + |c2| <= 5
    auto lin3 = [&](int e0, int c0, int e1, int c1, int e2, int c2, bool half) {
      const Sx<C, SX_T> v0 = LD(e0), v1 = LD(e1), v2 = LD(e2);
      i32 t[N];
#pragma unroll
      for (int i = 0; i < N; ++i) t[i] = c0 * v0.v[i] + c1 * v1.v[i] + c2 * v2.v[i];
      const i32 hm = half ? -(t[0] & 1) : 0, sh = half ? 1 : 0;
#pragma unroll
      for (int i = 0; i < N; ++i) t[i] += (i32)C::RX_P[i] & hm;
      Sx<C, 96> r;
#pragma unroll
      for (int i = 0; i < N; ++i) r.v[i] = (t[i] >> sh) + (i + 1 < N ? (t[i + 1] & sh) << 27 : 0);
      return r;
    };
    auto dbl_step = [&]() {
      LATX_R(0);
      {   // round 1: B = Y^2, C = Z^2, S = (Y+Z)^2, J = X^2, M = X Y, E = Zu Zt, W = Y Zt, V = Y Zu
        const int ia = q == 0 ? PY : q == 1 ? PZ : q == 2 ? PY : q == 3 ? PX : q == 4 ? PX : q == 5 ? PZU : PY;
        const int ib = q == 2 ? PZ : q == 4 ? PY : (q == 5 || q == 6) ? PZT : q == 7 ? PZU : ia;
        const Sx<C, SX_T> A1 = LD(ia), A2 = LD(ib);
        const Sx<C, SX_F> sum = F(sx_add<C>(A1, A2));
        const Sx<C, SX_F> a = sx_select<C>(q == 2, sum, T(A1)), b = sx_select<C>(q == 2, sum, T(A2));
        LATX_R(1);
        const Sx<C, SX_T> r = pair_mul<C>(a, b, odd);
        LATX_R(2);
        ST(PB + q, r, q < 8);                                 // PB, PC, PS, PJ, PM, PE, PW, PV are consecutive
        wave_sync();
      }
      LATX_R(3);
      {   // round 2: Z' = B H, Zt' = B 2W, Zu' = B 2V, line c1 xP = 3J xP, line c0 yP = -H yP   (H = S - B - C = 2 Y Z),
          //          X' = (M/2)(B - 3E), G^2 with G = (B + 3E)/2, E^2
        const bool one = q < 3 || q == 5 || q == 7;           // a is one entry as it stands
        const int a0 = q < 3 ? PB : q == 3 ? PJ : q == 4 ? PS : q == 5 ? PM : q == 6 ? PB : PE;
        const Sx<C, SX_F> a = F(lin3(a0, q == 3 ? 3 : q == 4 ? -1 : 1, q == 4 ? PB : PE, one ? 0 : q == 3 ? 0 : q == 4 ? 1 : 3, PC, q == 4 ? 1 : 0,
                                     q == 5 || q == 6));
        const int b0 = q == 0 ? PS : q == 1 ? PW : q == 2 ? PV : q == 3 ? PT0 : q == 4 ? PT1 : PB;
        const Sx<C, SX_F> bb = F(lin3(b0, (q == 1 || q == 2) ? 2 : 1, q == 0 ? PB : PE, q == 0 ? -1 : q == 5 ? -3 : 0, PC, q == 0 ? -1 : 0, false));
        const Sx<C, SX_F> b = sx_select<C>(q >= 6, a, bb);
        LATX_R(4);
        const Sx<C, SX_T> r = pair_mul<C>(a, b, odd);
        LATX_R(5);
        // nothing in this round reads X, Z, Zt, Zu: the new coordinates go straight to their places
        ST(q == 0 ? PZ : q == 1 ? PZT : q == 2 ? PZU : q == 5 ? PX : q == 6 ? PG : PE2, r, q < 3 || (q >= 5 && q < 8));
        put_line(buf, q == 3 ? 1 : 0, r, q == 3 || q == 4);
        wave_sync();
      }
      LATX_R(6);
      {   // Y' = G^2 - 3 E^2, line c2 = E - B
        const Sx<C, SX_T> v = sx_norm<C>(lin3(q == 0 ? PG : PE, 1, q == 0 ? PE2 : PB, q == 0 ? -3 : -1, PB, 0, false));
        ST(PY, v, q == 0);
        put_line(buf, 2, v, q == 1);
      }
      wave_sync();
      LATX_R(7);
      publish();
    };
    auto add_step = [&](const Sx<C, SX_T>& xq, const Sx<C, SX_T>& yq) {
      LATX_R(0);
      {   // round 1: yq Z, xq Z
        ST(q == 0 ? PU0 : PU1, pair_mul<C>(q == 0 ? yq : xq, LD(PZ), odd), q < 2);
        wave_sync();
      }
      LATX_R(1);
      const Sx<C, SX_F> th = F(sx_sub<C>(LD(PY), LD(PU0))), la = F(sx_sub<C>(LD(PX), LD(PU1)));
      {   // round 2: D = la^2, C = th^2, la th, la Z, la X, th Z, la Y, la Zt, th xq, la yq, la Zu, line c0 yP = la yP, line c1 xP = -th xP
        const Sx<C, SX_F> a = (q == 1 || q == 5 || q == 8) ? th : q == 12 ? sx_neg<C>(th) : la;
        const int ib = q == 3 ? PZ : q == 4 ? PX : q == 5 ? PZ : q == 6 ? PY : q == 7 ? PZT : q == 10 ? PZU : q == 11 ? PT1 : PT0;
        const Sx<C, SX_F> b = q == 0 ? la : q < 3 ? th : q == 8 ? T(xq) : q == 9 ? T(yq) : T(LD(ib));
        const Sx<C, SX_T> r = pair_mul<C>(a, b, odd);
        ST(PD + q, r, q < 11);                                // PD, PC2, PLT, PZL, PXL, PZH, PLY, PZTL, PV0, PV1, PZUL are consecutive
        put_line(buf, q == 11 ? 0 : 1, r, q == 11 || q == 12);
        wave_sync();
      }
      LATX_R(2);
      {   // round 3: D (D - 2 XL), ZL C, LT (3 XL - D), ZH C, D LY, Z' = ZL D, Zt' = ZTL D, Zu' = ZUL D
        const int ia = q == 0 ? PD : q == 1 ? PZL : q == 2 ? PLT : q == 3 ? PZH : q == 4 ? PD : q == 5 ? PZL : q == 6 ? PZTL : PZUL;
        const int b0 = (q == 1 || q == 3) ? PC2 : q == 2 ? PXL : q == 4 ? PLY : PD;
        const Sx<C, SX_F> b = F(lin3(b0, q == 2 ? 3 : 1, q == 0 ? PXL : PD, q == 0 ? -2 : q == 2 ? -1 : 0, PD, 0, false));
        const Sx<C, SX_T> r = pair_mul<C>(T(LD(ia)), b, odd);
        ST(q < 5 ? PN0 + q : q == 5 ? PZ : q == 6 ? PZT : PZU, r, q < 8);
        wave_sync();
      }
      LATX_R(3);
      {   // X' = N0 + N1, Y' = N2 - N3 - N4, line c2 = th xq - la yq
        const Sx<C, SX_T> v = sx_norm<C>(lin3(q == 0 ? PN0 : q == 1 ? PN0 + 2 : PV0, 1, q == 0 ? PN0 + 1 : q == 1 ? PN0 + 3 : PV1, q == 0 ? 1 : -1,
                                              PN0 + 4, q == 1 ? -1 : 0, false));
        ST(q == 0 ? PX : PY, v, q < 2);
        put_line(buf, 2, v, q == 2);
      }
      wave_sync();
      LATX_R(4);
      publish();
    };
    const Sx<C, SX_T> qx = LD(PQX), qy = LD(PQY);
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      if (is_sig) gen_step(); else dbl_step();
      const int d = C::LOOP_NAF[i];
      if (d != 0) {
        if (is_sig) gen_step(); else add_step(qx, d > 0 ? qy : sx_neg<C>(qy));
      }
    }
    if constexpr (C::CURVE_ID == 0) {
      if (is_sig) {
        gen_step();
        gen_step();
      } else {
        // Q1 = pi(Q) = (conj(x) g12, conj(y) g13), -Q2 = -pi^2(Q) = (x g22, -y g23) on the twist (pairing.hpp miller_loop)
        constexpr int N2 = 2 * N;
        const Sx<C, SX_T> cx = sx_select<C>(odd, sx_neg<C>(qx), qx), cy = sx_select<C>(odd, sx_neg<C>(qy), qy);
        const Sx<C, SX_T> x1 = pair_mul_const<C>(cx, C::RX_GAMMA + 0 * N2, C::RX_GAMMA + 0 * N2 + N, odd);
        const Sx<C, SX_T> y1 = pair_mul_const<C>(cy, C::RX_GAMMA + 1 * N2, C::RX_GAMMA + 1 * N2 + N, odd);
        add_step(x1, y1);
        const Sx<C, SX_T> x2 = pair_mul_const<C>(qx, C::RX_GAMMA + 2 * N2, C::RX_GAMMA + 2 * N2 + N, odd);
        const Sx<C, SX_T> y2 = sx_neg<C>(pair_mul_const<C>(qy, C::RX_GAMMA + 3 * N2, C::RX_GAMMA + 3 * N2 + N, odd));
        add_step(x2, y2);
      }
    }
    if (!is_sig && valid && lane < 2) {                       // a degenerate point step leaves Z = 0 (miller_x.hpp)
      const bool z_own = sx_is_zero_mod_p<C>(LD(PZ));
      const bool z_zero = z_own && pair_swap1(z_own ? 1 : 0) != 0;
      if (z_zero && !odd) atomicOr(flags, FLAG_DEGENERATE);
    }
  } else {
    // ------------------------------------------------ consumer: f <- f^2 * line, one Fp12 on 36 lanes
    int buf = 0;
    u32 epoch = 0;
    auto mul = [&](int dst, int a, int b) {
      if constexpr (AW == 2) fx_mul2w<C>(dst, a, b, ++epoch);
      else fx_mul1<C>(dst, a, b);
    };
    int nfold = 0;
    (void)nfold;
    auto fold = [&]() {
      if (wave == 0) LATX_T(2, nfold);
      __syncthreads();                                        // the line of this step is complete
      if (wave == 0) LATX_T(3, nfold);
      ++nfold;
      if (valid) mul(S_F, buf ? S_L1 : S_L0, S_F);
      buf ^= 1;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      if (i > 1 && valid) mul(S_F, S_F, S_F);
      fold();
      if (C::LOOP_NAF[i] != 0) fold();
    }
    if constexpr (C::CURVE_ID == 0) {
      fold();
      fold();
    } else {
      if (wave == 0 && lane < 24) {                           // x < 0: conjugate (w -> -w)
        const int k = lane >> 2, xi = (lane >> 1) & 1, h = lane & 1;
        Sx<C, SX_T> v = fx_ld<C>(E::coef(S_F, k, xi) + h * HS);
        if (k & 1) v = sx_neg<C>(v);
        fx_st<C>(E::coef(S_F, k, xi) + h * HS, v);
      }
      wave_sync();
    }
    if (wave == 0 && lane < 6) {
      const X2<C, SX_T> x = fx_ld2<C>(E::coef(S_F, lane, 0));
      out[blk * 6 + lane] = Fp2<C>{sx_to_mont<C>(x.c0), sx_to_mont<C>(x.c1)};
    }
  }
}

template <class C, int AW>
__global__ void __launch_bounds__(64 * (AW + 1)) k_miller_latx(const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at,
                                                               const LineCoeffs<C>* gen_lines, Fp2<C>* out, uint32_t* flags) {
  miller_latx_block<C, AW>(g1s, g2s, n, sig_at, gen_lines, out, flags, blockIdx.x);
}

// ---- epilogue of a batch verification on the same arithmetic (replaces k_cofactor_epilogue of k_tail.inc): partial product =
// miller(-sigma, g2) * rest^h, h = the G1 cofactor (BLS12-381 raw hash points; alt-bn128: h = 1).  Both
// factors are serial chains with no batch dimension: they run side by side as the two blocks of ONE launch -- block 0 walks the
// generator's pre-computed lines for the signature pair (the latency-form Miller block above), block 1 raises rest to h on the
// one-wave 36-lane product (general squarings: rest is not unitary before the final exponentiation) -- and a second, one-wave
// launch multiplies the two and serialises the result.  tmp: 12 Fp2 (w-basis, the library's Montgomery form).
// first_role: the role of block 0.  A launch of two blocks (first_role 0) runs both chains side by side; a launch of ONE block runs
// the signature pair alone (first_role 0) -- it depends on nothing but sigma, so a verification with the machine to itself walks
// it on the context's side stream while the messages are hashed -- or rest^h alone (first_role 1).
template <class C, int AW>
__global__ void __launch_bounds__(AW == 2 ? 256 : 128) k_epilogue_ax(const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines, Fp2<C>* tmp,
                                                                     uint32_t* flags, int first_role) {
  typedef FX<C> E;
  if ((int)blockIdx.x + first_role == 0) {
    if (threadIdx.x >= 64 * (AW + 1)) return;               // the Miller block is AW + 1 waves
    if (sig != nullptr) {
      miller_latx_block<C, AW>(sig, nullptr, 0, 0, gen_lines, tmp, flags, 0);  // (-sigma, g2): pair "n" of an empty batch
    } else if (threadIdx.x < 6) {
      tmp[threadIdx.x] = threadIdx.x == 0 ? f2_one<C>() : f2_zero<C>();
    }
    return;
  }
  if constexpr (AW == 2) {
    // round 4b: rest^h on FOUR waves, right to left -- the squarings on one pair of waves, the products the cofactor's bits select
    // on the other (finalx.hpp fx_pow / fx_mul_pair): 126 products deep instead of 125 squarings and 60 products one after the other
    const int lane = threadIdx.x;
    enum { S_BASE = 0, S_ACC = 1, S_SQ = 2 };
    if (lane < 6) {
      const Fp2<C> v = rest[lane];
      fx_put<C>(S_BASE, lane, X2<C, SX_T>{sx_from_mont<C>(v.c0), sx_from_mont<C>(v.c1)});
    }
    __syncthreads();
    int res = S_BASE;
    if constexpr (C::CURVE_ID == 1) {
      fx_pow<C>(S_ACC, S_BASE, C::COFACTOR, C::COFACTOR_BITS, S_SQ);
      res = S_ACC;
    }
    if (lane < 6) {
      const X2<C, SX_T> x = fx_ld2<C>(E::coef(res, lane, 0));
      tmp[6 + lane] = Fp2<C>{sx_to_mont<C>(x.c0), sx_to_mont<C>(x.c1)};
    }
    return;
  }
  if (threadIdx.x >= 64 * AW) return;                       // rest^h on the accumulator's waves (a wave that has ended does not count at a barrier)
  const int lane = threadIdx.x;
  enum { S_BASE = 0, S_ACC = 1 };
  if (lane < 6) {
    const Fp2<C> v = rest[lane];
    const X2<C, SX_T> x = {sx_from_mont<C>(v.c0), sx_from_mont<C>(v.c1)};
    fx_put<C>(S_BASE, lane, x);
    fx_put<C>(S_ACC, lane, x);
  }
  if (AW == 2 && lane == 63) fx_mul2w_init<C>();
  if constexpr (AW == 2) __syncthreads(); else wave_sync();
  if constexpr (C::CURVE_ID == 1) {
    u32 epoch = 0;
    auto mul = [&](int dst, int a, int b) {
      if constexpr (AW == 2) fx_mul2w<C>(dst, a, b, ++epoch);
      else fx_mul1<C>(dst, a, b);
    };
    for (int i = C::COFACTOR_BITS - 2; i >= 0; --i) {
      mul(S_ACC, S_ACC, S_ACC);
      if ((C::COFACTOR[i >> 5] >> (i & 31)) & 1u) mul(S_ACC, S_ACC, S_BASE);
    }
  }
  if (lane < 6) {
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(S_ACC, lane, 0));
    tmp[6 + lane] = Fp2<C>{sx_to_mont<C>(x.c0), sx_to_mont<C>(x.c1)};
  }
}
template <class C>
__global__ void __launch_bounds__(64) k_epilogue_bx(const Fp2<C>* tmp, uint8_t* out) {
  typedef FX<C> E;
  const int lane = threadIdx.x;
  if (lane < 6) {
    const Fp2<C> a = tmp[lane], b = tmp[6 + lane];
    fx_put<C>(0, lane, X2<C, SX_T>{sx_from_mont<C>(a.c0), sx_from_mont<C>(a.c1)});
    fx_put<C>(1, lane, X2<C, SX_T>{sx_from_mont<C>(b.c0), sx_from_mont<C>(b.c1)});
  }
  wave_sync();
  fx_mul1<C>(0, 0, 1);
  if (lane < 6) {
    const int order_pos[6] = {5, 2, 4, 1, 3, 0};
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(0, lane, 0));
    uint8_t* o = out + (size_t)(2 * order_pos[lane]) * C::FP_BYTES;
    fp_to_be<C>(o, fp_from_mont<C>(sx_to_mont<C>(x.c1)));
    fp_to_be<C>(o + C::FP_BYTES, fp_from_mont<C>(sx_to_mont<C>(x.c0)));
  }
}

// out[G] = prod in[G R .. min(count, (G + 1) R))  (w-basis Fp12 arrays in the library's Montgomery form) on the two-wave 36-lane
// product of finalx.hpp (fx_mul): the passes of the reduce stage where few products are left and each pass is a chain of
// dependent ones.  k_reduce_coop (six lanes per product, 32-bit limbs: ~11 us per alt-bn128 product, ~22 us per BLS12-381 one)
// stays for the wide first passes of big batches, where lanes matter and latency does not.
template <class C>
__global__ void __launch_bounds__(128) k_reduce_fx(const Fp2<C>* in, size_t count, int R, Fp2<C>* out) {
  typedef FX<C> E;
  const int tid = threadIdx.x;
  const size_t G = blockIdx.x, lo = G * (size_t)R;
  const int nin = (int)((lo + R < count ? lo + R : count) - lo);          // 1 .. R <= 12 operands, one LDS slot each
  // all operands are fetched and converted side by side (one coefficient per lane), then the products are a bare chain
  for (int idx = tid; idx < 6 * nin; idx += 128) {
    const int part = idx / 6, coeff = idx % 6;
    const Fp2<C> v = in[(lo + part) * 6 + coeff];
    fx_put<C>(part, coeff, X2<C, SX_T>{sx_from_mont<C>(v.c0), sx_from_mont<C>(v.c1)});
  }
  __syncthreads();
  for (int k = 1; k < nin; ++k) fx_mul<C>(0, 0, k);
  if (tid < 6) {
    const X2<C, SX_T> x = fx_ld2<C>(E::coef(0, tid, 0));
    out[G * 6 + tid] = Fp2<C>{sx_to_mont<C>(x.c0), sx_to_mont<C>(x.c1)};
  }
}

namespace kl {
template <class C>
void reduce_fx(hipStream_t st, const Fp2<C>* in, size_t count, int R, Fp2<C>* out) {
  const size_t nout = (count + R - 1) / R;
  k_reduce_fx<C><<<(unsigned)nout, 128, FX<C>::LDS_BYTES, st>>>(in, count, R, out);
}
template void reduce_fx<BN254>(hipStream_t, const Fp2<BN254>*, size_t, int, Fp2<BN254>*);
template void reduce_fx<BLS381>(hipStream_t, const Fp2<BLS381>*, size_t, int, Fp2<BLS381>*);
static_assert(FX<BN254>::LDS_BYTES_PAIR >= FX<BN254>::LDS_BYTES + FX2W_EXTRA_BYTES, "the pair scratch covers the hand-over words");
template <class C>
static void launch_epilogue_ax(hipStream_t st, unsigned blocks, const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines, Fp2<C>* tmp, uint32_t* flags,
                               int first_role) {
  k_epilogue_ax<C, 2><<<blocks, 256, FX<C>::LDS_BYTES_PAIR, st>>>(rest, sig, gen_lines, tmp, flags, first_role);
}
template <class C>
void cofactor_epiloguex(hipStream_t st, const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines, Fp2<C>* tmp, uint8_t* out,
                        uint32_t* flags) {
  launch_epilogue_ax<C>(st, 2, rest, sig, gen_lines, tmp, flags, 0);
  k_epilogue_bx<C><<<1, 64, FX<C>::LDS_BYTES, st>>>(tmp, out);
}
// the two chains as separate launches: part 1 = the signature pair only (tmp[0..5]), part 2 = rest^h (tmp[6..11]) and the product
template <class C>
void cofactor_epiloguex_part(hipStream_t st, int part, const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines, Fp2<C>* tmp, uint8_t* out) {
  if (part == 1) {
    launch_epilogue_ax<C>(st, 1, rest, sig, gen_lines, tmp, nullptr, 0);
  } else {
    launch_epilogue_ax<C>(st, 1, rest, sig, gen_lines, tmp, nullptr, 1);
    k_epilogue_bx<C><<<1, 64, FX<C>::LDS_BYTES, st>>>(tmp, out);
  }
}
template void cofactor_epiloguex_part<BN254>(hipStream_t, int, const Fp2<BN254>*, const Aff<F1<BN254>>*, const LineCoeffs<BN254>*, Fp2<BN254>*, uint8_t*);
template void cofactor_epiloguex_part<BLS381>(hipStream_t, int, const Fp2<BLS381>*, const Aff<F1<BLS381>>*, const LineCoeffs<BLS381>*, Fp2<BLS381>*, uint8_t*);
template void cofactor_epiloguex<BN254>(hipStream_t, const Fp2<BN254>*, const Aff<F1<BN254>>*, const LineCoeffs<BN254>*, Fp2<BN254>*, uint8_t*, uint32_t*);
template void cofactor_epiloguex<BLS381>(hipStream_t, const Fp2<BLS381>*, const Aff<F1<BLS381>>*, const LineCoeffs<BLS381>*, Fp2<BLS381>*, uint8_t*, uint32_t*);
template <class C>
void miller_latx(hipStream_t st, const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at, const LineCoeffs<C>* gen_lines, Fp2<C>* out,
                 uint32_t* flags) {
  const unsigned blocks = (unsigned)(n + (sig_at >= 0 ? 1 : 0));
  k_miller_latx<C, 2><<<blocks, 192, FX<C>::LDS_BYTES + FX2W_EXTRA_BYTES, st>>>(g1s, g2s, n, sig_at, gen_lines, out, flags);
}
template void miller_latx<BN254>(hipStream_t, const Aff<F1<BN254>>*, const uint8_t*, size_t, long long, const LineCoeffs<BN254>*, Fp2<BN254>*, uint32_t*);
template void miller_latx<BLS381>(hipStream_t, const Aff<F1<BLS381>>*, const uint8_t*, size_t, long long, const LineCoeffs<BLS381>*, Fp2<BLS381>*, uint32_t*);
}  // namespace kl
}  // namespace bgls
#ifdef LATX_DBG
extern "C" int bgls_dbg_latx_dump(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out + 4 * 4 * 160, HIP_SYMBOL(bgls::g_latx_r), sizeof(bgls::g_latx_r));
  (void)hipMemcpyFromSymbol(out + 2 * 4 * 160, HIP_SYMBOL(bgls::g_latx_c), sizeof(bgls::g_latx_c));
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bgls::g_latx_t), sizeof(bgls::g_latx_t));
}
#endif
