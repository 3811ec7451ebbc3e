// One G2 Jacobian addition / doubling per WAVE, for the places where only a few additions are left and each is on the
// critical path: the top of the key-sum tree of verifyMultiSignature (curves/curve.go:73-121), the window sums and the
// final doublings of the bucket method (k_msm.hip).  A thread-per-addition kernel runs those as lone waves, 16 dependent
// Fp2 products = 45-50 us per addition; here the products of one dependency level run side by side:
//
//   * every lane holds the SAME operands (replicated in registers) and computes the cheap glue (additions, selects)
//     redundantly, so an addition composes with the next one without any layout change;
//   * product q of a level sits on lanes 4q .. 4q+2: lane 4q+k multiplies piece k of the Karatsuba triple (a0 b0, a1 b1,
//     (a0+a1)(b0+b1)) in double width, the triple is exchanged with lane shuffles, lane 4q reduces the real part, lane 4q+1
//     the imaginary part (one instruction stream: operand selects, no divergent branches), both store their half to a
//     per-wave LDS scratch from which every lane reads the level's results back.
//
// An addition is five levels (5 + 4 + 3 + 2 + 2 products), a doubling three (3 + 3 + 1): one wide product and one
// reduction deep each.  Exceptional inputs are exact: infinity is a select, equal x takes the doubling or returns infinity.
#pragma once
#include "coop.hpp"
#include "curve.hpp"

namespace bgls {

template <class C>
struct CoopF2 {
  static constexpr int L = C::L, W = 2 * C::L, S2 = 2 * C::L;
  static constexpr int SLOTS = 8;
  static constexpr int WAVE_DW = SLOTS * S2;              // LDS dwords per wave
  int base, q, piece;
  __device__ __forceinline__ explicit CoopF2(int lds_base) : base(lds_base) {
    const int lane = threadIdx.x & 63;
    q = lane >> 2;
    piece = lane & 3;
  }
  // this lane's half (piece 0: real, piece 1: imaginary) of product q of the level
  __device__ __forceinline__ Fp<C> half(const Fp2<C>& a, const Fp2<C>& b) const {
    const Fp<C> x = piece == 0 ? a.c0 : piece == 1 ? a.c1 : fp_add_nr<C>(a.c0, a.c1);
    const Fp<C> y = piece == 0 ? b.c0 : piece == 1 ? b.c1 : fp_add_nr<C>(b.c0, b.c1);
    u32 w[W], v0[W], v1[W], sw[W], tmp[W];
    mul_wide<C>(w, x.v, y.v);
    const int first = (threadIdx.x & 63) & ~3;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      v0[k] = __shfl(w[k], first);
      v1[k] = __shfl(w[k], first + 1);
      sw[k] = __shfl(w[k], first + 2);
    }
    // uniform form R = X + K - V:  real: v0 + 6 p^2 - v1      imaginary: s + 0 - (v0 + v1)
    const u32 m0 = piece == 0 ? 0xFFFFFFFFu : 0u;
    u32 X[W], V[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      X[k] = (v0[k] & m0) | (sw[k] & ~m0);
      tmp[k] = v0[k] & ~m0;
    }
    w_add<W>(V, v1, tmp);
#pragma unroll
    for (int k = 0; k < W; ++k) tmp[k] = C::P2W6[k] & m0;
    w_add<W>(X, X, tmp);
    w_sub<W>(X, X, V);
    return redc_k<C, Coop<C>::LAZY_K>(X);
  }
  // one level: lanes of quad q multiply (a, b) -- every lane passes the operands of ITS quad -- and the first n results
  // become readable through get(0 .. n-1)
  __device__ __forceinline__ void level(const Fp2<C>& a, const Fp2<C>& b, int n) const {
    extern __shared__ u32 lds[];
    const Fp<C> r = half(a, b);
    if (q < n && piece < 2) {
      u32* o = lds + base + q * S2 + piece * L;
#pragma unroll
      for (int k = 0; k < L; ++k) o[k] = r.v[k];
    }
    wave_sync();
  }
  __device__ __forceinline__ Fp2<C> get(int i) const { return lds_load_f2<C>(base + i * S2); }
};

template <class C>
__device__ __forceinline__ Jac<F2<C>> coop_jac_dbl(const CoopF2<C>& k, const Jac<F2<C>>& p) {      // dbl-2009-l
  const int q = k.q;
  k.level(q == 0 ? p.X : p.Y, q == 0 ? p.X : q == 1 ? p.Y : p.Z, 3);                  // A = X^2, B = Y^2, Y Z
  const Fp2<C> A = k.get(0), B = k.get(1), YZ = k.get(2);
  const Fp2<C> E = f2_add<C>(f2_dbl<C>(A), A);
  const Fp2<C> XB = f2_add<C>(p.X, B);
  k.level(q == 0 ? B : q == 1 ? XB : E, q == 0 ? B : q == 1 ? XB : E, 3);             // C = B^2, (X + B)^2, F = E^2
  const Fp2<C> Cc = k.get(0);
  const Fp2<C> D = f2_dbl<C>(f2_sub<C>(f2_sub<C>(k.get(1), A), Cc));
  Jac<F2<C>> r;
  r.X = f2_sub<C>(k.get(2), f2_dbl<C>(D));
  k.level(E, f2_sub<C>(D, r.X), 1);
  r.Y = f2_sub<C>(k.get(0), f2_dbl<C>(f2_dbl<C>(f2_dbl<C>(Cc))));
  r.Z = f2_dbl<C>(YZ);
  return r;
}

template <class C>
__device__ __forceinline__ Jac<F2<C>> coop_jac_add(const CoopF2<C>& k, const Jac<F2<C>>& p, const Jac<F2<C>>& o) {   // add-2007-bl
  typedef F2<C> F;
  const int q = k.q;
  const bool pinf = jac_is_inf<F>(p), oinf = jac_is_inf<F>(o);
  const Fp2<C> zs = f2_add<C>(p.Z, o.Z);
  // Z1^2, Z2^2, (Z1 + Z2)^2, Y1 Z2, Y2 Z1
  k.level(q == 0 ? p.Z : q == 1 ? o.Z : q == 2 ? zs : q == 3 ? p.Y : o.Y, q == 0 ? p.Z : q == 1 ? o.Z : q == 2 ? zs : q == 3 ? o.Z : p.Z, 5);
  const Fp2<C> Z1Z1 = k.get(0), Z2Z2 = k.get(1), ZZ = k.get(2), t1 = k.get(3), t2 = k.get(4);
  // U1 = X1 Z2Z2, U2 = X2 Z1Z1, S1 = Y1 Z2 Z2Z2, S2 = Y2 Z1 Z1Z1
  k.level(q == 0 ? p.X : q == 1 ? o.X : q == 2 ? t1 : t2, (q == 0 || q == 2) ? Z2Z2 : Z1Z1, 4);
  const Fp2<C> U1 = k.get(0), U2 = k.get(1), S1 = k.get(2), S2v = k.get(3);
  const Fp2<C> H = f2_sub<C>(U2, U1);
  const Fp2<C> rr0 = f2_sub<C>(S2v, S1);
  if (f2_is_zero<C>(H) && !pinf && !oinf) {                       // same x (uniform across the wave): P = Q or P = -Q
    if (f2_is_zero<C>(rr0)) return coop_jac_dbl<C>(k, p);
    return jac_inf<F>();
  }
  const Fp2<C> rr = f2_dbl<C>(rr0), H2 = f2_dbl<C>(H);
  const Fp2<C> zmix = f2_sub<C>(f2_sub<C>(ZZ, Z1Z1), Z2Z2);
  k.level(q == 0 ? H2 : q == 1 ? rr : zmix, q == 0 ? H2 : q == 1 ? rr : H, 3);        // I = (2H)^2, rr^2, Z3
  const Fp2<C> I = k.get(0), R2 = k.get(1);
  Jac<F> r;
  r.Z = k.get(2);
  k.level(q == 0 ? H : U1, I, 2);                                                       // J = H I, V = U1 I
  const Fp2<C> J = k.get(0), V = k.get(1);
  r.X = f2_sub<C>(f2_sub<C>(R2, J), f2_dbl<C>(V));
  k.level(q == 0 ? rr : S1, q == 0 ? f2_sub<C>(V, r.X) : J, 2);                         // rr (V - X3), S1 J
  r.Y = f2_sub<C>(k.get(0), f2_dbl<C>(k.get(1)));
  r.X = f2_select<C>(oinf, p.X, f2_select<C>(pinf, o.X, r.X));
  r.Y = f2_select<C>(oinf, p.Y, f2_select<C>(pinf, o.Y, r.Y));
  r.Z = f2_select<C>(oinf, p.Z, f2_select<C>(pinf, o.Z, r.Z));
  return r;
}

}  // namespace bgls
