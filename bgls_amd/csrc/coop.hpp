// Wave-cooperative Fp12 arithmetic: one Fp12 value is spread over a GROUP of 6 lanes, lane j
// holding the Fp2 coefficient e_j of  f = sum_j e_j w^j  (w^6 = xi).  A wavefront carries 10
// groups (60 lanes; lanes 60..63 shadow group 9 and never store).
//
// Why: in the product of pairings  prod_i e(P_i, Q_i)  the Miller accumulator can be SHARED:
//      f <- f^2 * l_1 * l_2 * ... * l_6      (one squaring for six pairings)
// gives exactly prod_i f_i because field arithmetic is exact and commutative -- this is the
// reference's PairingProduct value (curves/curve.go:125-170) with 5/6 of the Fp12 squarings
// removed.  Each lane runs the G2 point step of ITS pairing in registers (embarrassingly
// parallel), publishes the three line coefficients to LDS, and then the six lanes cooperate on
// the Fp12 updates: every lane produces ONE output coefficient as a short dot product
//      c_j = sum_t A_t * B_{(j - s_t) mod 6} * (xi if s_t > j)
// accumulated in double width (Karatsuba per term, 2 Montgomery reductions per coefficient).
// Operands come from LDS: region RB holds each coefficient twice (plain and pre-multiplied by
// xi) so that the wrap-around factor is an address choice, not a branch; region RL holds the
// 6 x 3 line coefficients (or, for a general product, the plain coefficients of the other factor).
//
// LDS per group: 30 Fp2 = 1920 B (alt-bn128) / 2880 B (BLS12-381); per wave 19.2 / 28.8 KB.
#pragma once
#include "pairing.hpp"

namespace bgls {

template <class C>
struct Coop {
  static constexpr int L = C::L;
  static constexpr int S2 = 2 * C::L;        // dwords per Fp2
  static constexpr int RB = 0;               // [6][2] Fp2: coefficient k -> {e_k, xi*e_k}
  static constexpr int RL = 12 * S2;         // [6][3] Fp2 line coefficients / [6] plain operand
  static constexpr int PAD = 0;
  static constexpr int GROUP_DW = 30 * S2 + PAD;
  static constexpr int GROUPS = 10;
  static constexpr int WAVE_BYTES = GROUPS * GROUP_DW * 4;
  static constexpr int LAZY_K = C::CURVE_ID == 0 ? 3 : 2;   // 12 p^2 < LAZY_K * p * 2^(32L)
  // producer/consumer variant: a second line buffer so the point-step wave can run one step ahead
  static constexpr int RL2 = 30 * S2;
  static constexpr int GROUP_DW_AB = 48 * S2 + PAD;
  static constexpr int BLOCK_BYTES_AB = GROUPS * GROUP_DW_AB * 4;
};

// Ordering point for LDS traffic between the lanes of ONE wave: the DS unit executes a wave's
// instructions in order, so only the compiler has to be kept from moving accesses across it.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __constant__ const int COOP_SH6[6] = {0, 1, 2, 3, 4, 5};
__device__ __constant__ const int COOP_SH_D[3] = {0, 1, 3};   // D-type line: e0 + e1 w + e3 w^3
__device__ __constant__ const int COOP_SH_M[3] = {0, 2, 3};   // M-type line: e0 + e2 w^2 + e3 w^3
// product of TWO lines (p + q w^a + r w^3)(p' + q' w^a + r' w^3), a = 1 (D) or 2 (M): five coefficients
//   [pp' + xi rr', pq'+qp', qq', pr'+rp', qr'+rq'] at these powers of w
__device__ __constant__ const int COOP_SH_D5[5] = {0, 1, 2, 3, 4};
__device__ __constant__ const int COOP_SH_M5[5] = {0, 2, 4, 3, 5};

// A region is an array of Fp2 entries, entry-major (each entry 2L contiguous dwords, read with
// ds_read_b128).  A chunk-major variant (the c-th 16-byte chunk of all entries contiguous) was
// measured: it removes no time at 2^16 and costs 45 % at 2^20, so the plain layout stays.
struct LReg {
  int base, nent;
};
template <class C>
__device__ __forceinline__ Fp2<C> lds_load_f2(int off);
template <class C>
__device__ __forceinline__ void lds_store_f2(int off, const Fp2<C>& a);
template <class C>
__device__ __forceinline__ Fp2<C> lds_ld(LReg r, int e) {
  return lds_load_f2<C>(r.base + e * 2 * C::L);
}
template <class C>
__device__ __forceinline__ void lds_st(LReg r, int e, const Fp2<C>& x) {
  lds_store_f2<C>(r.base + e * 2 * C::L, x);
}
template <class C>
__device__ __forceinline__ LReg reg_rb(int gb) { return {gb + Coop<C>::RB, 12}; }
template <class C>
__device__ __forceinline__ LReg reg_rl(int gb, int rl) { return {gb + rl, 18}; }

template <class C>
__device__ __forceinline__ Fp2<C> lds_load_f2(int off) {
  extern __shared__ u32 lds[];
  Fp2<C> r;
  const uint4* p = reinterpret_cast<const uint4*>(lds + off);
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) {
    uint4 v = p[k];
    r.c0.v[4 * k] = v.x; r.c0.v[4 * k + 1] = v.y; r.c0.v[4 * k + 2] = v.z; r.c0.v[4 * k + 3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) {
    uint4 v = p[C::L / 4 + k];
    r.c1.v[4 * k] = v.x; r.c1.v[4 * k + 1] = v.y; r.c1.v[4 * k + 2] = v.z; r.c1.v[4 * k + 3] = v.w;
  }
  return r;
}

template <class C>
__device__ __forceinline__ void lds_store_f2(int off, const Fp2<C>& a) {
  extern __shared__ u32 lds[];
  uint4* p = reinterpret_cast<uint4*>(lds + off);
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) p[k] = make_uint4(a.c0.v[4 * k], a.c0.v[4 * k + 1], a.c0.v[4 * k + 2], a.c0.v[4 * k + 3]);
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k)
    p[C::L / 4 + k] = make_uint4(a.c1.v[4 * k], a.c1.v[4 * k + 1], a.c1.v[4 * k + 2], a.c1.v[4 * k + 3]);
}

// c_j = sum_{t<NT} A[t] * B[(j - sh[t]) mod 6] * xi^[sh[t] > j]
//   A[t] = entry a_e0 + t*a_es of region ra;  B = entries {2k, 2k+1} = {e_k, xi e_k} of region rb.
template <class C, int NT, bool XF = false>
__device__ __forceinline__ Fp2<C> coop_dot_inl(LReg ra, int a_e0, int a_es, LReg rb, int j, const int* sh) {
  constexpr int W = 2 * C::L;
  u32 v0[W], v1[W], s[W];
#pragma unroll
  for (int k = 0; k < W; ++k) v0[k] = v1[k] = s[k] = 0;
#pragma unroll 1
  for (int t = 0; t < NT; ++t) {
    const int sht = sh[t];
    int k = j - sht;
    const int wrap = k < 0 ? 1 : 0;
    k += 6 * wrap;
    Fp2<C> a = lds_ld<C>(ra, a_e0 + t * a_es);
    Fp2<C> b = lds_ld<C>(rb, XF ? k : 2 * k + wrap);
    if constexpr (XF) b = f2_select<C>(wrap != 0, f2_mulxi<C>(b), b);   // region holds plain coefficients only
    u32 tmp[W];
    mul_wide<C>(tmp, a.c0.v, b.c0.v);
    w_add<W>(v0, v0, tmp);
    mul_wide<C>(tmp, a.c1.v, b.c1.v);
    w_add<W>(v1, v1, tmp);
    Fp<C> sa = fp_add_nr<C>(a.c0, a.c1);
    Fp<C> sb = fp_add_nr<C>(b.c0, b.c1);
    mul_wide<C>(tmp, sa.v, sb.v);
    w_add<W>(s, s, tmp);
  }
  w_sub<W>(s, s, v0);
  w_sub<W>(s, s, v1);                                   // sum (a0 b1 + a1 b0)         < 2 NT p^2
  if constexpr (NT > 3) w_add<W>(v0, v0, C::P2W6); else w_add<W>(v0, v0, C::P2W3);
  w_sub<W>(v0, v0, v1);                                 // sum (a0 b0 - a1 b1) + NT p^2 in (0, 2 NT p^2)
  Fp2<C> r;
  r.c0 = redc_k<C, Coop<C>::LAZY_K>(v0);
  r.c1 = redc_k<C, Coop<C>::LAZY_K>(s);
  return r;
}

template <class C, int NT>
__device__ __noinline__ Fp2<C> coop_dot(LReg ra, int a_e0, int a_es, LReg rb, int j, const int* sh) {
  return coop_dot_inl<C, NT>(ra, a_e0, a_es, rb, j, sh);
}

// f^2 with the symmetric terms merged: c_j = sum over i <= k, i + k = j (mod 6) of (2 if i != k) a_i a_k xi^[i+k >= 6]:
// 4 products per lane instead of 6 (odd j have only 3; the 4th slot is masked out).  Table entry per term:
// bits 0-2 = i (7 = unused slot), bits 3-5 = k, bit 6 = wrap, bit 7 = doubled.
__device__ __constant__ const unsigned COOP_SQ_TAB[6] = {0x5be2e900u, 0xffe3ea88u, 0x64eb0990u, 0xffec9198u, 0x6d1299a0u, 0xff9aa1a8u};
template <class C, bool XF = false>
__device__ __forceinline__ Fp2<C> coop_sqr_sym_inl(LReg rb, int j) {
  constexpr int W = 2 * C::L;
  u32 v0[W], v1[W], s[W];
#pragma unroll
  for (int k = 0; k < W; ++k) v0[k] = v1[k] = s[k] = 0;
  const unsigned row = COOP_SQ_TAB[j];
#pragma unroll 1
  for (int t = 0; t < 4; ++t) {
    const unsigned e = (row >> (8 * t)) & 0xFFu;
    const int i = e & 7u, k = (e >> 3) & 7u;
    const bool used = i != 7;
    Fp2<C> a = lds_ld<C>(rb, (XF ? 1 : 2) * (used ? i : 0));
    Fp2<C> b = lds_ld<C>(rb, used ? (XF ? k : 2 * k + (int)((e >> 6) & 1u)) : 0);
    if constexpr (XF) b = f2_select<C>(((e >> 6) & 1u) != 0, f2_mulxi<C>(b), b);
    b = f2_select<C>((e >> 7) & 1u, f2_dbl<C>(b), b);
    a = f2_select<C>(used, a, f2_zero<C>());
    u32 tmp[W];
    mul_wide<C>(tmp, a.c0.v, b.c0.v);
    w_add<W>(v0, v0, tmp);
    mul_wide<C>(tmp, a.c1.v, b.c1.v);
    w_add<W>(v1, v1, tmp);
    Fp<C> sa = fp_add_nr<C>(a.c0, a.c1);
    Fp<C> sb = fp_add_nr<C>(b.c0, b.c1);
    mul_wide<C>(tmp, sa.v, sb.v);
    w_add<W>(s, s, tmp);
  }
  w_sub<W>(s, s, v0);
  w_sub<W>(s, s, v1);
  w_add<W>(v0, v0, C::P2W6);
  w_sub<W>(v0, v0, v1);
  Fp2<C> r;
  r.c0 = redc_k<C, Coop<C>::LAZY_K>(v0);
  r.c1 = redc_k<C, Coop<C>::LAZY_K>(s);
  return r;
}

// publish this lane's coefficient (plain and xi-multiplied) into the group's RB region
template <class C, bool XF = false>
__device__ __forceinline__ void coop_publish(int rb_off, int j, const Fp2<C>& v, bool live) {
  if (live) {
    const LReg rb = {rb_off, 12};
    if constexpr (XF) {
      lds_st<C>(rb, j, v);
    } else {
      lds_st<C>(rb, 2 * j, v);
      lds_st<C>(rb, 2 * j + 1, f2_mulxi<C>(v));
    }
  }
  wave_sync();
}

// f <- f * g for two distributed values: g's coefficient of this lane is `gj`, f lives in RB.
template <class C, bool INL = false>
__device__ __forceinline__ Fp2<C> coop_mul(int gb, int j, const Fp2<C>& gj, bool live, int rl = Coop<C>::RL) {
  if (live) lds_st<C>(reg_rl<C>(gb, rl), j, gj);
  wave_sync();
  Fp2<C> r;
  if constexpr (INL) r = coop_dot_inl<C, 6>(reg_rl<C>(gb, rl), 0, 1, reg_rb<C>(gb), j, COOP_SH6);
  else r = coop_dot<C, 6>(reg_rl<C>(gb, rl), 0, 1, reg_rb<C>(gb), j, COOP_SH6);
  wave_sync();
  return r;
}

template <class C>
__device__ __noinline__ Fp2<C> coop_sqr_sym(LReg rb, int j) {
  return coop_sqr_sym_inl<C>(rb, j);
}
// f <- f^2, f in RB
template <class C, bool INL = false>
__device__ __forceinline__ Fp2<C> coop_sqr(int gb, int j) {
  if constexpr (INL) return coop_sqr_sym_inl<C>(reg_rb<C>(gb), j);
  else return coop_sqr_sym<C>(reg_rb<C>(gb), j);
}

// f <- f * line_m, line coefficients at RL[m][0..2]
template <class C, bool INL = false>
__device__ __forceinline__ Fp2<C> coop_mul_line(int gb, int j, int m, int rl = Coop<C>::RL) {
  if constexpr (INL) return coop_dot_inl<C, 3>(reg_rl<C>(gb, rl), 3 * m, 1, reg_rb<C>(gb), j, C::TWIST_D ? COOP_SH_D : COOP_SH_M);
  else return coop_dot<C, 3>(reg_rl<C>(gb, rl), 3 * m, 1, reg_rb<C>(gb), j, C::TWIST_D ? COOP_SH_D : COOP_SH_M);
}

// write this lane's (scaled) line into RL[j]; an inactive pairing contributes the constant 1
template <class C>
__device__ __forceinline__ void coop_write_line(int gb, int j, const LineCoeffs<C>& l, const Fp<C>& xP, const Fp<C>& yP, bool valid,
                                                bool live, int rl = Coop<C>::RL) {
  Fp2<C> e0, e1, e2;
  if constexpr (C::TWIST_D) {
    e0 = f2_muls<C>(l.c0, yP); e1 = f2_muls<C>(l.c1, xP); e2 = l.c2;
  } else {
    e0 = l.c2; e1 = f2_muls<C>(l.c1, xP); e2 = f2_muls<C>(l.c0, yP);
  }
  if (!valid) { e0 = f2_one<C>(); e1 = f2_zero<C>(); e2 = f2_zero<C>(); }
  if (live) {
    const LReg r = reg_rl<C>(gb, rl);
    lds_st<C>(r, 3 * j, e0);
    lds_st<C>(r, 3 * j + 1, e1);
    lds_st<C>(r, 3 * j + 2, e2);
  }
  wave_sync();
}

// Emitter for dbl_step_emit / add_step_emit: scales LineCoeffs slot `which` by yP / xP and stores it as the
// matching entry of RL[j] (D-type: c0 yP, c1 xP, c2;  M-type: c2, c1 xP, c0 yP).
template <class C, bool R28>
__device__ __forceinline__ void st_entry(LReg r, int e, const Fp2<C>& x);   // coop_r28.hpp: plain store or converted to 28-bit limbs

template <class C, bool R28 = false>
struct LineEmitter {
  LReg r;
  int j;
  const Fp<C>& xP;
  const Fp<C>& yP;
  bool valid, live;
  __device__ __forceinline__ void operator()(int which, const Fp2<C>& v) const {
    Fp2<C> e;
    int entry;
    if (which == 0) {
      e = f2ms<C, true>(v, yP);
      entry = C::TWIST_D ? 0 : 2;
    } else if (which == 1) {
      e = f2ms<C, true>(v, xP);
      entry = 1;
    } else {
      e = v;
      entry = C::TWIST_D ? 2 : 0;
    }
    if (!valid) e = (entry == 0) ? f2_one<C>() : f2_zero<C>();
    if (live) st_entry<C, R28>(r, 3 * j + entry, e);
  }
};

// ---- pair-of-lines form: the producer wave multiplies the lines of two neighbouring pairings (lanes 2m,
// 2m+1 of a group) into one 5-coefficient element, so the consumer folds three 5-term elements instead of
// six 3-term lines.  Same total work, moved to the producer -- pays off only where the producer has the
// registers for it (k_miller_ab64; in the 168-register kernel it was a loss).
template <class C>
struct LineCapture {          // emitter that keeps the scaled line (p, q, r) in registers
  Fp2<C> e[3];
  const Fp<C>& xP;
  const Fp<C>& yP;
  __device__ __forceinline__ void operator()(int which, const Fp2<C>& v) {
    if (which == 0) e[C::TWIST_D ? 0 : 2] = f2ms<C, true>(v, yP);
    else if (which == 1) e[1] = f2ms<C, true>(v, xP);
    else e[C::TWIST_D ? 2 : 0] = v;
  }
};
template <class C>
__device__ __forceinline__ Fp2<C> f2_shfl_xor1(const Fp2<C>& a) {
  Fp2<C> r;
#pragma unroll
  for (int k = 0; k < C::L; ++k) {
    r.c0.v[k] = __shfl_xor(a.c0.v[k], 1);
    r.c1.v[k] = __shfl_xor(a.c1.v[k], 1);
  }
  return r;
}
// lanes (2m, 2m+1) hold lines A (even lane) and B (odd lane); writes the five coefficients of A*B into entries
// 5m .. 5m+4 of region rl.  Uniform instruction stream: operand selects by lane parity.
template <class C, bool R28 = false>
__device__ __forceinline__ void coop_write_line_pair(LReg rl, int j, const Fp2<C> (&own)[3]) {
  // a = even lane's line, b = odd lane's line (index 2 = the w^3 coefficient)
  //   even lane: P00 = a0 b0, P11 = a1 b1, K01 = (a0+a1)(b0+b1);   odd lane: P22 = a2 b2, K02 = (a0+a2)(b0+b2), K12 = (a1+a2)(b1+b2)
  // Each lane forms its left operand from its own line and receives exactly the matching right operand from
  // its neighbour, one Fp2 at a time (keeps the live set small).
  const bool odd = j & 1;
  // what the neighbour needs from me: it has the opposite parity
  Fp2<C> p1 = f2_mul_inl<C>(f2_select<C>(odd, own[2], own[0]),
                            f2_shfl_xor1<C>(f2_select<C>(odd, own[0], own[2])));            // even: P00   odd: P22
  Fp2<C> p2 = f2_mul_inl<C>(f2_select<C>(odd, f2_add<C>(own[0], own[2]), own[1]),
                            f2_shfl_xor1<C>(f2_select<C>(odd, own[1], f2_add<C>(own[0], own[2]))));   // even: P11   odd: K02
  Fp2<C> p3 = f2_mul_inl<C>(f2_add<C>(own[1], f2_select<C>(odd, own[2], own[0])),
                            f2_shfl_xor1<C>(f2_add<C>(own[1], f2_select<C>(odd, own[0], own[2]))));  // even: K01   odd: K12
  Fp2<C> q1 = f2_shfl_xor1<C>(p1), q2 = f2_shfl_xor1<C>(p2);
  const int m = j >> 1;
  if (!odd) {       // c0 = P00 + xi P22, c1 = K01 - P00 - P11, c2 = P11
    st_entry<C, R28>(rl, 5 * m + 0, f2_add<C>(p1, f2_mulxi<C>(q1)));
    st_entry<C, R28>(rl, 5 * m + 1, f2_sub<C>(f2_sub<C>(p3, p1), p2));
    st_entry<C, R28>(rl, 5 * m + 2, p2);
  } else {          // c3 = K02 - P00 - P22, c4 = K12 - P11 - P22   (q1 = P00, q2 = P11 from the even lane)
    st_entry<C, R28>(rl, 5 * m + 3, f2_sub<C>(f2_sub<C>(p2, q1), p1));
    st_entry<C, R28>(rl, 5 * m + 4, f2_sub<C>(f2_sub<C>(p3, q2), p1));
  }
}

// fold the six published lines into f (f in RB on entry and on exit; returns this lane's coefficient)
template <class C, bool INL = false>
__device__ __forceinline__ Fp2<C> coop_apply_lines(int gb, int j, bool live, int rl = Coop<C>::RL) {
  Fp2<C> fj;
#pragma unroll 1
  for (int m = 0; m < 6; ++m) {
    fj = coop_mul_line<C, INL>(gb, j, m, rl);
    coop_publish<C>(gb + Coop<C>::RB, j, fj, live);
  }
  return fj;
}

}  // namespace bgls
