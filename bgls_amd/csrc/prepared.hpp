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
// k_fold_prep is the whole Miller stage.  The same shared-accumulator structure as the Miller kernel's consumer (10 groups x 6 lanes per
// wave, NG pairings per squaring, no block-level synchronisation), the ratios streamed from HBM through a two-slot LDS ring.
//
// Round 6: the fold runs on the carry-free limbs of rx.hpp -- alt-bn128 on nine limbs of 29 bits, BLS12-381 on fourteen of 28 -- with the
// Miller kernel's own consumer pieces (miller_x.hpp: two-pile Karatsuba dot products, the two-pile / symmetric squaring, the xi copies of
// e_3..e_5 only) at THREE waves per SIMD.  Rounds 2-5 ran it on the round-2 helpers (ten 28-bit limbs with three piles at two waves per SIMD
// on alt-bn128, 32-bit Montgomery limbs on BLS12-381): 0.41 / 0.40 of the multiplier peak.  The table holds the ratios in the fold's limb form
// (tight, below 2 p), 12 / 16 dwords per half.
//
// Degenerate steps (c2 = 0: the running point T satisfies y_T^2 = 3 b' Z_T^2 or the chord through T and Q is degenerate)
// cannot occur for honestly generated keys except with probability ~2^-254 per step; a key that hits one is reported by
// the upload (BGLS_ERR_ENCODING) instead of being prepared wrongly.
#pragma once
#include "miller_kernels.hpp"
#include "miller_x.hpp"

namespace bgls {

// table formats (global memory); X = the fold's number form
template <class C>
struct Prep {
  typedef typename MxForm<C>::type X;
  static constexpr int HALF_DW = (X::RX_NL + 3) & ~3;         // one Fp of a ratio: 9 / 14 limbs padded to whole 16-byte words (12 / 16 dwords)
  static constexpr int LINE_DW = 4 * HALF_DW;                // A.c0 A.c1 B.c0 B.c1
  static constexpr int P_SKIP = 2 * HALF_DW;                 // x, y (padded halves), skip word
  static constexpr int P_DW = 2 * HALF_DW + (HALF_DW == 12 ? 8 : 4);      // 32 / 36: entries stay 16-byte aligned
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

// one Fp (the library's Montgomery form) into the table's form: the fold's carry-free limbs, tight, value below 2 p
template <class C>
__device__ __forceinline__ void st_half(u32* p, const Fp<C>& a) {
  typedef typename MxForm<C>::type X;
  constexpr int HD = Prep<C>::HALF_DW;
  Fp<X> ax;
#pragma unroll
  for (int i = 0; i < C::L; ++i) ax.v[i] = a.v[i];
  const Ux<X> u = to_ux<X>(ax);
  u32 w[HD];
#pragma unroll
  for (int i = 0; i < HD; ++i) w[i] = i < X::RX_NL ? u.v[i] : 0u;
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int k = 0; k < HD / 4; ++k) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
// thread per key i < n_pad (keys i >= n: padding up to whole fold groups).  keys: resident Montgomery affine points.
// table[(s * n_pad + i) * LINE_DW ...] <- (A, B) of step s;  kinf[i] = 1 for infinite / padding keys (their lines are 1);
// tmp: TMP_DW dwords per (step, key) of scratch, same indexing.
template <class C>
__global__ void __launch_bounds__(64) k_prepare(const Aff<F2<C>>* keys, size_t n, size_t n_pad, size_t i0, size_t count, u32* table, uint8_t* kinf,
                                                u32* tmp, uint32_t* flags) {
  typedef Prep<C> T;
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
    st_half<C>(o, A.c0);
    st_half<C>(o + T::HALF_DW, A.c1);
    st_half<C>(o + 2 * T::HALF_DW, B.c0);
    st_half<C>(o + 3 * T::HALF_DW, B.c1);
  }
}

// per pairing i < n_pad: the hash point in the fold's form + the skip word (infinite hash point, infinite or padding key)
template <class C>
__global__ void k_prep_points(const Aff<F1<C>>* g1s, const uint8_t* kinf, size_t n, size_t n_pad, u32* ptab) {
  typedef Prep<C> T;
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
  st_half<C>(o, x);
  st_half<C>(o + T::HALF_DW, y);
  o[T::P_SKIP] = skip ? 1u : 0u;
}

// LDS of one prepared-fold wave, per group (dwords): the accumulator where the Miller kernel's consumer has it (plain e_k, the xi copies of
// e_3..e_5 only) and two line slots of two scaled entries each in the room the unused xi copies leave.
//   alt-bn128 (entries of 24): plain e_k at 24 k [0, 144), line entry e = 2 slot + t at 144 + 24 e [144, 240), xi e_k at 176 + 24 k [248, 320);
//                              stride 332: 13.3 KB per wave, eleven waves per CU (LDS is handed out in 2 KB steps)
//   BLS12-381 (entries of 28): plain e_k at 56 k, xi e_k at 28 + 56 (k - 3), line entries at 196, 252, 308, 336; stride 364: 14.6 KB per wave, eight waves per CU (two per SIMD: registers)
template <class X>
struct PrepX {
  static constexpr bool PACKED = MX<X>::PACKED;
  static constexpr int HS = MX<X>::HS, ES = MX<X>::ES;
  static constexpr int GROUP_DW = PACKED ? 364 : 332;
  static constexpr int WAVE_BYTES = 10 * GROUP_DW * 4;
  static __device__ __forceinline__ int acc_off(int k, int wrap) {
    if constexpr (PACKED) return wrap ? 28 + 56 * (k - 3) : 56 * k;
    else return 24 * k + 176 * wrap;
  }
  static __device__ __forceinline__ int line_off(int e) {
    if constexpr (PACKED) return e < 3 ? 196 + 56 * e : 336;
    else return 144 + 24 * e;
  }
  static_assert(!PACKED || ES == 28, "BLS12-381 entries");
  static_assert(PACKED || ES == 24, "alt-bn128 entries");
};

// one wave = 10 groups x 6 lanes (lane = 10 j + g, as the Miller kernel's consumer), ng pairings per group and squaring.
// Waves per SIMD: three on alt-bn128 (168 registers, 7 spilled); BLS12-381's two 14-limb piles, its accumulator coefficient and the prefetched
// line leave 243 registers spilled at 168, so it runs two waves per SIMD at 256.
template <class C>
constexpr int prep_fold_waves() { return C::CURVE_ID == 0 ? 3 : 2; }
template <class C>
__global__ void __launch_bounds__(64, prep_fold_waves<C>()) k_fold_prep(const u32* table, const u32* ptab, size_t n_pad, int ng, Fp2<C>* out) {
  typedef typename MxForm<C>::type X;
  typedef Prep<C> T;
  typedef PrepX<X> K;
  constexpr int NL = X::RX_NL;
  const int lane = threadIdx.x;
  const bool live = lane < 60;
  const int cl = live ? lane : lane - 12;                      // lanes 60..63 shadow lanes 48..51
  const int g = cl % 10;
  const int j = cl / 10;
  const int gb = g * K::GROUP_DW;
  const size_t G = (size_t)blockIdx.x * 10 + g;                 // pairings [G * ng, (G + 1) * ng)
  const size_t step_dw = n_pad * T::LINE_DW;
  // lanes j = 0..3 of a group scale one half each: j = 0, 1 -> A.c0, A.c1 (by yP), j = 2, 3 -> B.c0, B.c1 (by xP)
  const bool scaler = live && j < 4;
  const int hsel = j & 3;
  const u32* lsrc = table + (G * (size_t)ng) * T::LINE_DW + hsel * T::HALF_DW;
  const u32* psrc = ptab + (G * (size_t)ng) * T::P_DW + (hsel < 2 ? T::HALF_DW : 0);      // yP for A, xP for B
  const u32* ssrc = ptab + (G * (size_t)ng) * T::P_DW + T::P_SKIP;
  constexpr int NV = T::HALF_DW / 4;
  uint4 pl[NV], pp[NV];
  u32 pskip = 0;
  auto prefetch = [&](int s2, int m2) __attribute__((always_inline)) {
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
  Ux2<X> fj;
  {
    const Ux<X> one = ux_load<X>(X::RX_ONE);
#pragma unroll
    for (int k = 0; k < NL; ++k) { fj.c0.v[k] = j == 0 ? one.v[k] : 0u; fj.c1.v[k] = 0u; }
  }
  mxk_publish<X, K, true>(gb, j, fj, live);
  unsigned sq_d = 0, sq_p = 0;
  if constexpr (!rx_lazy<X>) mx_sq_split(COOP_SQ_TAB[j], sq_d, sq_p);
  // the powers of w the two scaled entries sit at: D-type l = A yP + B xP w + w^3 -> {0, 1} (the unit coefficient at w^3);
  // M-type l = 1 + B xP w^2 + A yP w^3 -> {2, 3} (the unit coefficient at w^0)
  auto fold_step = [&]() __attribute__((always_inline)) {
#pragma unroll 1
    for (int m = 0; m < ng; ++m) {
      const u32 skip = pskip;
      if (scaler) {               // scale the prefetched half into the slot
        Ux<X> a, b;
#pragma unroll
        for (int q = 0; q < NL; ++q) { a.v[q] = reinterpret_cast<const u32*>(pl)[q]; b.v[q] = reinterpret_cast<const u32*>(pp)[q]; }
        u64 col[2 * NL];
        ux_acc_new<X>(col, a, b);
        const Ux<X> r = ux_redc<X>(col);
        // M-type: entry 0 = B xP (w^2), entry 1 = A yP (w^3);  D-type: entry 0 = A yP (w^0), entry 1 = B xP (w^1)
        const int e = X::TWIST_D ? (hsel >> 1) : 1 - (hsel >> 1);
        mx_st_half<X, K::PACKED>(gb + K::line_off(2 * slot + e) + (hsel & 1) * K::HS, (hsel & 1) != 0, r);
      }
      wave_sync();
      {                           // fetch the next line while this one is folded
        int s2 = s, m2 = m + 1;
        if (m2 == ng) { m2 = 0; ++s2; }
        if (s2 < T::NSTEPS) prefetch(s2, m2);
      }
      // (ux_dot_k2q, every fetch one stage ahead, was tried here too: 63 / 92 spilled registers next to the prefetched line, 35.9 -> 32.0 M pairs/s)
      Ux2<X> r = ux_dot_k2p<X, 2, (NL <= 10)>(
          [&](int t, int h) { return mx_ld_half<X, K::PACKED>(gb + K::line_off(2 * slot + t) + h * K::HS, h != 0); },
          [&](int t, int h) {
            int k = j - (X::TWIST_D ? t : t + 2);
            const int wrap = k < 0 ? 1 : 0;
            k += 6 * wrap;
            return mx_ld_half<X, K::PACKED>(gb + K::acc_off(k, wrap) + h * K::HS, h != 0);
          });
      // + the unit coefficient's share: f_(j-3) (xi f_(j+3) where it wraps) on the D-type twist, f_j itself on the M-type
      Ux2<X> u;
      if constexpr (X::TWIST_D) {
        int k = j - 3;
        const int wrap = k < 0 ? 1 : 0;
        k += 6 * wrap;
        u.c0 = mx_ld_half<X, K::PACKED>(gb + K::acc_off(k, wrap), false);
        u.c1 = mx_ld_half<X, K::PACKED>(gb + K::acc_off(k, wrap) + K::HS, true);
      } else {
        u = fj;
      }
#pragma unroll
      for (int q = 0; q < NL; ++q) { r.c0.v[q] += u.c0.v[q]; r.c1.v[q] += u.c1.v[q]; }
      // a reduction's output (below 1.3 p) plus a published value (below 2.31 p; the D-type twist's xi copies below 3.001 p): under 4.31 p.  ONE
      // conditional subtraction of 2 p brings it below 2.31 p again -- all that the next xi multiple (below 3 p) and the piles (below 4 p) ask for
      r = ux_quasi<X, 1, 1>(r);
      if (!skip) fj = r;                                     // a skipped pairing contributes the constant line 1
      mxk_publish<X, K, true>(gb, j, fj, live);
      slot ^= 1;
    }
    ++s;
  };
#pragma unroll 1
  for (int i = 1; i < C::LOOP_LEN; ++i) {
    if (i > 1) {                                             // f = 1 before the first step
      if constexpr (rx_lazy<X>) fj = mxk_sqr<X, K>(gb, j);
      else fj = mxk_sqr3<X, K>(gb, sq_d, sq_p);
      mxk_publish<X, K, true>(gb, j, fj, live);
    }
    fold_step();
    if (C::LOOP_NAF[i] != 0) fold_step();
  }
  if constexpr (C::CURVE_ID == 0) {
    fold_step();
    fold_step();
  }
  if (live) {
    Fp2<X> r = {from_ux_inl<X>(fj.c0), from_ux_inl<X>(fj.c1)};
    if constexpr (C::CURVE_ID != 0) {
      if (j & 1) r = f2_neg<X>(r);                          // x < 0: f^(p^6), w -> -w
    }
    Fp2<C> o;
#pragma unroll
    for (int q = 0; q < C::L; ++q) { o.c0.v[q] = r.c0.v[q]; o.c1.v[q] = r.c1.v[q]; }
    out[G * 6 + j] = o;
  }
}

}  // namespace bgls
