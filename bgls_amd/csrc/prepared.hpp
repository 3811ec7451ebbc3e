// Prepared key sets: the Miller loop against keys whose line functions were computed when the key set was uploaded.
//
// The optimal-ate line of step s for the pair (P, Q) is  l = c0 yP + c1 xP w + c2 w^3  (D-type twist, alt-bn128) or
// l = c2 + c1 xP w^2 + c0 yP w^3  (M-type, BLS12-381), where c0, c1, c2 in Fp2 depend on the key Q only (pairing.hpp).
// A key set that is verified many times (a validator set) pays the G2 point steps once: k_prepare walks them per key and
// stores, per step, the two ratios  A = c0 / c2,  B = c1 / c2  -- the line divided by its P-free coefficient, an element
// of Fp2, a proper subfield of Fp12, so the final exponentiation maps the normalised product to the SAME GT element
// (tests compare the final GT bytes with the unprepared path).  A verification then scales A by yP and B by xP and folds
//      D-type:  f <- f * (A yP + B xP w + w^3)          M-type:  f <- f * (1 + B xP w^2 + A yP w^3)
// which is two Fp2 products per lane and line plus one shifted copy of f, instead of three products, and no point steps:
// k_fold_prep is the whole Miller stage.  The same shared-accumulator structure as k_fold (10 groups x 6 lanes per wave,
// NG pairings per squaring, no block-level synchronisation), the ratios streamed from HBM through a two-slot LDS ring.
//
// Degenerate steps (c2 = 0: the running point T satisfies y_T^2 = 3 b' Z_T^2 or the chord through T and Q is degenerate)
// cannot occur for honestly generated keys except with probability ~2^-254 per step; a key that hits one is reported by
// the upload (BGLS_ERR_ENCODING) instead of being prepared wrongly.
#pragma once
#include "miller_kernels.hpp"

namespace bgls {

template <class C, bool R28>
struct Prep {
  static constexpr int HALF_DW = R28 ? 12 : C::L;            // one Fp of a ratio: 10 limbs padded to 12 (r28) or L limbs
  static constexpr int LINE_DW = 4 * HALF_DW;                // A.c0 A.c1 B.c0 B.c1
  static constexpr int P_DW = R28 ? 32 : 2 * C::L + 4;       // x, y (padded halves), skip word
  static constexpr int P_SKIP = R28 ? 24 : 2 * C::L;
  static constexpr int NSTEPS = LineTab<C, false>::NSTEPS;
  static constexpr int TMP_DW = 8 * C::L;                    // per (step, key) scratch of k_prepare: c0, c1, c2, prefix product
};

// emitter keeping the raw (unscaled) coefficients of a step
template <class C>
struct RawCapture {
  Fp2<C> c[3];
  __device__ __forceinline__ void operator()(int which, const Fp2<C>& v) { c[which] = v; }
};

template <class C>
__device__ __forceinline__ void st_f2_strided(u32* p, const Fp2<C>& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) q[k] = make_uint4(a.c0.v[4 * k], a.c0.v[4 * k + 1], a.c0.v[4 * k + 2], a.c0.v[4 * k + 3]);
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) q[C::L / 4 + k] = make_uint4(a.c1.v[4 * k], a.c1.v[4 * k + 1], a.c1.v[4 * k + 2], a.c1.v[4 * k + 3]);
}
template <class C>
__device__ __forceinline__ Fp2<C> ld_f2_strided(const u32* p) {
  Fp2<C> r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) { const uint4 v = q[k]; r.c0.v[4 * k] = v.x; r.c0.v[4 * k + 1] = v.y; r.c0.v[4 * k + 2] = v.z; r.c0.v[4 * k + 3] = v.w; }
#pragma unroll
  for (int k = 0; k < C::L / 4; ++k) { const uint4 v = q[C::L / 4 + k]; r.c1.v[4 * k] = v.x; r.c1.v[4 * k + 1] = v.y; r.c1.v[4 * k + 2] = v.z; r.c1.v[4 * k + 3] = v.w; }
  return r;
}

// one Fp of a ratio in the table's form
template <class C, bool R28>
__device__ __forceinline__ void st_half(u32* p, const Fp<C>& a) {
  if constexpr (R28) {
    const F28 r = to_r28<C>(a);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
    q[2] = make_uint4(r.v[8], r.v[9], 0u, 0u);
  } else {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int k = 0; k < C::L / 4; ++k) q[k] = make_uint4(a.v[4 * k], a.v[4 * k + 1], a.v[4 * k + 2], a.v[4 * k + 3]);
  }
}

// thread per key i < n_pad (keys i >= n: padding up to whole fold groups).  keys: resident Montgomery affine points.
// table[(s * n_pad + i) * LINE_DW ...] <- (A, B) of step s;  kinf[i] = 1 for infinite / padding keys (their lines are 1);
// tmp: TMP_DW dwords per (step, key) of scratch, same indexing.
template <class C, bool R28>
__global__ void __launch_bounds__(64) k_prepare(const Aff<F2<C>>* keys, size_t n, size_t n_pad, size_t i0, size_t count, u32* table, uint8_t* kinf,
                                                u32* tmp, uint32_t* flags) {
  typedef Prep<C, R28> T;
  const size_t li = (size_t)blockIdx.x * 64 + threadIdx.x;      // keys i0 .. i0 + count of the set; tmp is indexed within the chunk
  if (li >= count) return;
  const size_t i = i0 + li;
  Aff<F2<C>> Q;
  bool inf = true;
  if (i < n) {
    Q = keys[i];
    inf = Q.inf;
  }
  kinf[i] = inf ? 1 : 0;
  if (inf) {                                                  // lines of an infinite key are the constant 1: skipped by the fold
#pragma unroll 1
    for (int s = 0; s < T::NSTEPS; ++s) {
      uint4* q = reinterpret_cast<uint4*>(table + ((size_t)s * n_pad + i) * T::LINE_DW);
      for (int k = 0; k < T::LINE_DW / 4; ++k) q[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  const size_t step_tmp = count * T::TMP_DW;
  u32* t = tmp + li * T::TMP_DW;
  G2Proj<C> R = {Q.x, Q.y, f2_one<C>()};
  auto put = [&](RawCapture<C>& cap) {
    st_f2_strided<C>(t, cap.c[0]);
    st_f2_strided<C>(t + 2 * C::L, cap.c[1]);
    st_f2_strided<C>(t + 4 * C::L, cap.c[2]);
    t += step_tmp;
  };
#pragma unroll 1
  for (int k = 1; k < C::LOOP_LEN; ++k) {
    {
      RawCapture<C> cap;
      dbl_step_emit<C>(R, cap);
      put(cap);
    }
    const int d = C::LOOP_NAF[k];
    if (d != 0) {
      RawCapture<C> cap;
      add_step_emit<C>(R, Q.x, d > 0 ? Q.y : f2_neg<C>(Q.y), cap);
      put(cap);
    }
  }
  if constexpr (C::CURVE_ID == 0) {
    {
      Fp2<C> x1 = f2_mul<C>(f2_conj<C>(Q.x), gamma_const<C>(1, 2));
      Fp2<C> y1 = f2_mul<C>(f2_conj<C>(Q.y), gamma_const<C>(1, 3));
      RawCapture<C> cap;
      add_step_emit<C>(R, x1, y1, cap);
      put(cap);
    }
    {
      Fp2<C> x2 = f2_mul<C>(Q.x, gamma_const<C>(2, 2));
      Fp2<C> y2 = f2_neg<C>(f2_mul<C>(Q.y, gamma_const<C>(2, 3)));
      RawCapture<C> cap;
      add_step_emit<C>(R, x2, y2, cap);
      put(cap);
    }
  }
  // Montgomery's trick over the steps: prefix products of the c2's, one inversion, then 1 / c2_s going backwards
  t = tmp + li * T::TMP_DW;
  Fp2<C> acc = f2_one<C>();
  bool degenerate = false;
#pragma unroll 1
  for (int s = 0; s < T::NSTEPS; ++s) {
    Fp2<C> c2 = ld_f2_strided<C>(t + 4 * C::L);
    if (f2_is_zero<C>(c2)) { degenerate = true; c2 = f2_one<C>(); st_f2_strided<C>(t + 4 * C::L, c2); }
    st_f2_strided<C>(t + 6 * C::L, acc);                      // product of c2_0 .. c2_{s-1}
    acc = f2_mul<C>(acc, c2);
    t += step_tmp;
  }
  if (degenerate) atomicOr(flags, FLAG_DEGENERATE);
  Fp2<C> inv = f2_inv<C>(acc);
#pragma unroll 1
  for (int s = T::NSTEPS - 1; s >= 0; --s) {
    t -= step_tmp;
    const Fp2<C> c2 = ld_f2_strided<C>(t + 4 * C::L);
    const Fp2<C> ic = f2_mul<C>(inv, ld_f2_strided<C>(t + 6 * C::L));      // 1 / c2_s
    inv = f2_mul<C>(inv, c2);
    const Fp2<C> A = f2_mul<C>(ld_f2_strided<C>(t), ic);
    const Fp2<C> B = f2_mul<C>(ld_f2_strided<C>(t + 2 * C::L), ic);
    u32* o = table + ((size_t)s * n_pad + i) * T::LINE_DW;
    st_half<C, R28>(o, A.c0);
    st_half<C, R28>(o + T::HALF_DW, A.c1);
    st_half<C, R28>(o + 2 * T::HALF_DW, B.c0);
    st_half<C, R28>(o + 3 * T::HALF_DW, B.c1);
  }
}

// per pairing i < n_pad: the hash point in the fold's form + the skip word (infinite hash point, infinite or padding key)
template <class C, bool R28>
__global__ void k_prep_points(const Aff<F1<C>>* g1s, const uint8_t* kinf, size_t n, size_t n_pad, u32* ptab) {
  typedef Prep<C, R28> T;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  u32* o = ptab + i * T::P_DW;
  bool skip = true;
  Fp<C> x = fp_zero<C>(), y = fp_zero<C>();
  if (i < n) {
    const Aff<F1<C>> P = g1s[i];
    skip = P.inf || kinf[i] != 0;
    x = P.x;
    y = P.y;
  }
  st_half<C, R28>(o, x);
  st_half<C, R28>(o + T::HALF_DW, y);
  o[T::P_SKIP] = skip ? 1u : 0u;
}

// ---- r28 helpers of the prepared fold -----------------------------------------------------------------------------------
// value back below ~4 p: a tight (limbs < 2^28), top limb < 2^12.  q = floor(top * 21 / 64) never exceeds a / p
// (p > 3.03 * 2^252, 64 / 21 = 3.0476), so a - q p >= 0; what is left has top limb <= 3 + (top - 3.0476 q) < 8.
template <class C>
__device__ __forceinline__ F28 r28_reduce_small(const F28& a) {
  const u32 q = (a.v[9] * 21u) >> 6;
  F28 z;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int64_t t = (int64_t)a.v[i] - (int64_t)((u64)q * C::R28_P[i]) + carry;
    if (i < 9) {
      z.v[i] = (u32)((u64)t & R28_MASK);
      carry = t >> 28;
    } else {
      z.v[i] = (u32)t;
    }
  }
  return z;
}
__device__ __forceinline__ F28 lds_ld28h(int off) {           // one Fp (10 limbs) at a 8-byte aligned dword offset
  extern __shared__ u32 lds[];
  F28 r;
  const uint2* p = reinterpret_cast<const uint2*>(lds + off);
#pragma unroll
  for (int k = 0; k < 5; ++k) { const uint2 v = p[k]; r.v[2 * k] = v.x; r.v[2 * k + 1] = v.y; }
  return r;
}
__device__ __forceinline__ void lds_st28h(int off, const F28& a) {
  extern __shared__ u32 lds[];
  uint2* p = reinterpret_cast<uint2*>(lds + off);
#pragma unroll
  for (int k = 0; k < 5; ++k) p[k] = make_uint2(a.v[2 * k], a.v[2 * k + 1]);
}

// LDS of one prepared-fold wave: per group the accumulator region (12 entries, {e_k, xi e_k}) + two line slots of two
// scaled entries each
template <class C, bool R28>
struct PrepLds {
  static constexpr int S2 = R28 ? R28_S2 : 2 * C::L;
  static constexpr bool XF = !R28 && C::XI_RE == 1;
  static constexpr int RBN = XF ? 6 : 12;
  static constexpr int RB = 0, RL = RBN * S2;
  static constexpr int GROUP_DW = (RBN + 2 * 2) * S2;
  static constexpr int WAVE_BYTES = 10 * GROUP_DW * 4;
};

// KARA (r28 form only): three-pile Karatsuba dot products at two waves per SIMD (the shipped form), or the four-pile
// schoolbook form that fits the 168 registers of three waves per SIMD -- measured at 2^20 alt-bn128 pairings: 39.4 ms
// against 65.2 ms (the extra wave does not pay for a third more multiplications and 259 spilled registers)
template <class C, bool R28, bool KARA = true>
__global__ void __launch_bounds__(64, KARA ? 2 : 3) k_fold_prep(const u32* table, const u32* ptab, size_t n_pad, int ng, Fp2<C>* out) {
  typedef Prep<C, R28> T;
  typedef PrepLds<C, R28> K;
  extern __shared__ u32 lds[];
  const int lane = threadIdx.x;
  const bool live = lane < 60;
  const int g = live ? lane / 6 : 9;
  const int j = live ? lane % 6 : lane - 60;
  const int gb = g * K::GROUP_DW;
  const size_t G = (size_t)blockIdx.x * 10 + g;                 // pairings [G * ng, (G + 1) * ng)
  const size_t step_dw = n_pad * T::LINE_DW;
  // lanes 0..3 of a group scale one half each: j = 0, 1 -> A.c0, A.c1 (by yP), j = 2, 3 -> B.c0, B.c1 (by xP)
  const bool scaler = live && j < 4;
  const int hsel = j & 3;
  const u32* lsrc = table + (G * (size_t)ng) * T::LINE_DW + hsel * T::HALF_DW;
  const u32* psrc = ptab + (G * (size_t)ng) * T::P_DW + (hsel < 2 ? T::HALF_DW : 0);      // yP for A, xP for B
  const u32* ssrc = ptab + (G * (size_t)ng) * T::P_DW + T::P_SKIP;
  constexpr int NV = T::HALF_DW / 4;
  uint4 pl[NV], pp[NV];
  u32 pskip = 0;
  auto prefetch = [&](int s2, int m2) {
    const u32* a = lsrc + (size_t)s2 * step_dw + (size_t)m2 * T::LINE_DW;
    const u32* b = psrc + (size_t)m2 * T::P_DW;
    if (scaler) {
#pragma unroll
      for (int q = 0; q < NV; ++q) { pl[q] = reinterpret_cast<const uint4*>(a)[q]; pp[q] = reinterpret_cast<const uint4*>(b)[q]; }
    }
    pskip = ssrc[(size_t)m2 * T::P_DW];
  };
  int s = 0, slot = 0;
  prefetch(0, 0);
  if constexpr (R28) {
    const int rbo = gb + K::RB;
    F28x2 fj;
    {
      const F28 one = r28_load<C>(C::R28_ONE);
#pragma unroll
      for (int q = 0; q < 10; ++q) { fj.c0.v[q] = j == 0 ? one.v[q] : 0u; fj.c1.v[q] = 0u; }
    }
    coop_publish28<C>(rbo, j, fj, live);
    auto fold_step = [&]() {
#pragma unroll 1
      for (int m = 0; m < ng; ++m) {
        // scale the prefetched halves into the slot, fetch the next line
        const u32 skip = pskip;
        if (scaler) {
          F28 a, b;
#pragma unroll
          for (int q = 0; q < 10; ++q) { a.v[q] = reinterpret_cast<const u32*>(pl)[q]; b.v[q] = reinterpret_cast<const u32*>(pp)[q]; }
          u64 col[20];
#pragma unroll
          for (int q = 0; q < 20; ++q) col[q] = 0;
          r28_acc(col, a, b);
          lds_st28h(gb + K::RL + slot * 2 * R28_S2 + (hsel >> 1) * R28_S2 + (hsel & 1) * 10, r28_redc<C>(col));
        }
        wave_sync();
        {
          int s2 = s, m2 = m + 1;
          if (m2 == ng) { m2 = 0; ++s2; }
          if (s2 < T::NSTEPS) prefetch(s2, m2);
        }
        // c_j = e0 f_j + e1 f_{j-1} + f_{j-3}  (D-type; wrap-around factors by address)
        const int rlo = gb + K::RL + slot * 2 * R28_S2;
        F28x2 r;
        if constexpr (KARA) {
          u64 v0[20], v1[20], ss[20];
#pragma unroll
          for (int q = 0; q < 20; ++q) v0[q] = v1[q] = ss[q] = 0;
#pragma unroll 1
          for (int t = 0; t < 2; ++t) {
            int k = j - t;
            const int wrap = k < 0 ? 1 : 0;
            k += 6 * wrap;
            const F28x2 a = lds_ld28(rlo + t * R28_S2);
            const F28x2 b = lds_ld28(rbo + (2 * k + wrap) * R28_S2);
            r28_kara_term(v0, v1, ss, a, b);
          }
          r = r28_kara_finish<C>(v0, v1, ss);
        } else {
          u64 cr[20], ci[20];
#pragma unroll
          for (int q = 0; q < 20; ++q) cr[q] = ci[q] = 0;
#pragma unroll 1
          for (int t = 0; t < 2; ++t) {
            int k = j - t;
            const int wrap = k < 0 ? 1 : 0;
            k += 6 * wrap;
            const F28x2 a = lds_ld28(rlo + t * R28_S2);
            const F28x2 b = lds_ld28(rbo + (2 * k + wrap) * R28_S2);
            r28_acc(cr, a.c0, b.c0);
            r28_acc(cr, a.c1, r28_fatneg<C>(b.c1));
            r28_acc(ci, a.c0, b.c1);
            r28_acc(ci, a.c1, b.c0);
          }
          r.c0 = r28_redc<C>(cr);
          r.c1 = r28_redc<C>(ci);
        }
        {
          int k = j - 3;
          const int wrap = k < 0 ? 1 : 0;
          k += 6 * wrap;
          const F28x2 u = lds_ld28(rbo + (2 * k + wrap) * R28_S2);
#pragma unroll
          for (int q = 0; q < 10; ++q) { r.c0.v[q] += u.c0.v[q]; r.c1.v[q] += u.c1.v[q]; }
        }
        r.c0 = r28_reduce_small<C>(r28_norm(r.c0));
        r.c1 = r28_reduce_small<C>(r28_norm(r.c1));
        if (!skip) fj = r;                                     // a skipped pairing contributes the constant line 1
        coop_publish28<C>(rbo, j, fj, live);
        slot ^= 1;
      }
      ++s;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      fj = coop_sqr_sym28<C>(rbo, j);
      coop_publish28<C>(rbo, j, fj, live);
      fold_step();
      if (C::LOOP_NAF[i] != 0) fold_step();
    }
    fold_step();
    fold_step();
    if (live) out[G * 6 + j] = from_r28<C>(fj);
  } else {
    const LReg rb = {gb + K::RB, 12};
    Fp2<C> fj = j == 0 ? f2_one<C>() : f2_zero<C>();
    coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
    auto fold_step = [&]() {
#pragma unroll 1
      for (int m = 0; m < ng; ++m) {
        const u32 skip = pskip;
        if (scaler) {
          Fp<C> a, b;
#pragma unroll
          for (int q = 0; q < C::L; ++q) { a.v[q] = reinterpret_cast<const u32*>(pl)[q]; b.v[q] = reinterpret_cast<const u32*>(pp)[q]; }
          const Fp<C> r = fp_mul_inl<C>(a, b);
          // M-type: entry 0 = B xP (w^2), entry 1 = A yP (w^3);  D-type: entry 0 = A yP (w^0), entry 1 = B xP (w^1)
          const int e = C::TWIST_D ? (hsel >> 1) : 1 - (hsel >> 1);
          u32* dst = lds + gb + K::RL + slot * 2 * K::S2 + e * K::S2 + (hsel & 1) * C::L;
#pragma unroll
          for (int q = 0; q < C::L / 4; ++q) reinterpret_cast<uint4*>(dst)[q] = make_uint4(r.v[4 * q], r.v[4 * q + 1], r.v[4 * q + 2], r.v[4 * q + 3]);
        }
        wave_sync();
        {
          int s2 = s, m2 = m + 1;
          if (m2 == ng) { m2 = 0; ++s2; }
          if (s2 < T::NSTEPS) prefetch(s2, m2);
        }
        const LReg rl = {gb + K::RL + slot * 2 * K::S2, 2};
        Fp2<C> r;
        if constexpr (C::TWIST_D) {
          r = coop_dot_inl<C, 2, K::XF>(rl, 0, 1, rb, j, COOP_SH_D);           // shifts {0, 1}
          int k = j - 3;
          const int wrap = k < 0 ? 1 : 0;
          k += 6 * wrap;
          Fp2<C> u = lds_ld<C>(rb, K::XF ? k : 2 * k + wrap);
          if constexpr (K::XF) u = f2_select<C>(wrap != 0, f2_mulxi<C>(u), u);
          r = f2_add<C>(r, u);
        } else {
          r = coop_dot_inl<C, 2, K::XF>(rl, 0, 1, rb, j, COOP_SH_M + 1);       // shifts {2, 3}
          r = f2_add<C>(r, fj);
        }
        if (!skip) fj = r;
        coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
        slot ^= 1;
      }
      ++s;
    };
#pragma unroll 1
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      fj = coop_sqr_sym_inl<C, K::XF>(rb, j);
      coop_publish<C, K::XF>(gb + K::RB, j, fj, live);
      fold_step();
      if (C::LOOP_NAF[i] != 0) fold_step();
    }
    if constexpr (C::CURVE_ID == 0) {
      fold_step();
      fold_step();
    } else {
      if (j & 1) fj = f2_neg<C>(fj);                      // x < 0: f^(p^6), w -> -w
    }
    if (live) out[G * 6 + j] = fj;
  }
}

}  // namespace bgls
