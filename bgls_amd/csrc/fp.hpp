// Prime-field arithmetic for alt-bn128 (8 x u32) and BLS12-381 (12 x u32), Montgomery form.
//
// This is the bottom of the HIP hot path: every kernel in kernels.hip is built from these
// routines.  The reference has no field arithmetic of its own -- it calls into
// bn256/cloudflare and dis2/bls12 (curves/altbn128.go:11, curves/bls12_381.go:11) -- so the
// algorithms here are written from the maths: operand-scanning products on v_mad_u64_u32
// (32x32+64 -> 64), separate Montgomery reduction so Fp2 can reduce lazily.
//
// All functions are plain inline C++ over fixed-size limb arrays so that hipcc keeps operands
// in VGPRs after full unrolling.  The same header is compiled for the host ONLY by the unit
// tests (tests/host_harness.cpp) to diff each routine against the oracle without a GPU; the
// shipped library never executes it on the CPU.
#pragma once
#include <stdint.h>
#include "constants_gen.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BGLS_HD __host__ __device__ __forceinline__
#define BGLS_FN __host__ __device__ __noinline__
#else
#define BGLS_HD inline __attribute__((always_inline))
#define BGLS_FN __attribute__((noinline))
#endif

namespace bgls {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

BGLS_HD u32 addc(u32 a, u32 b, u32& c) {
  u32 co;
  u32 r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
}
BGLS_HD u32 subb(u32 a, u32 b, u32& c) {
  u32 co;
  u32 r = __builtin_subc(a, b, c, &co);
  c = co;
  return r;
}

template <class C>
struct Fp {
  u32 v[C::L];
};

// ---------------------------------------------------------------- basic
template <class C>
BGLS_HD Fp<C> fp_zero() {
  Fp<C> r;
#pragma unroll
  for (int j = 0; j < C::L; ++j) r.v[j] = 0;
  return r;
}

template <class C>
BGLS_HD Fp<C> fp_load(const u32* p) {
  Fp<C> r;
#pragma unroll
  for (int j = 0; j < C::L; ++j) r.v[j] = p[j];
  return r;
}

template <class C>
BGLS_HD Fp<C> fp_one() {
  return fp_load<C>(C::ONE);
}

template <class C>
BGLS_HD bool fp_is_zero(const Fp<C>& a) {
  u32 o = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) o |= a.v[j];
  return o == 0;
}

template <class C>
BGLS_HD bool fp_eq(const Fp<C>& a, const Fp<C>& b) {
  u32 o = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) o |= a.v[j] ^ b.v[j];
  return o == 0;
}

// r = c ? a : b
template <class C>
BGLS_HD Fp<C> fp_select(bool c, const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
#pragma unroll
  for (int j = 0; j < C::L; ++j) r.v[j] = c ? a.v[j] : b.v[j];
  return r;
}

// a >= p ?  (plain integer compare)
template <class C>
BGLS_HD bool fp_geq_p(const Fp<C>& a) {
  u32 bw = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) (void)subb(a.v[j], C::P[j], bw);
  return bw == 0;
}

// a + b without reduction (caller guarantees a + b < 2^(32L); true for a, b < p)
template <class C>
BGLS_HD Fp<C> fp_add_nr(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> r;
  u32 c = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) r.v[j] = addc(a.v[j], b.v[j], c);
  return r;
}

// conditional subtract: a in [0, 2p) -> [0, p)
template <class C>
BGLS_HD Fp<C> fp_reduce_once(const Fp<C>& a) {
  Fp<C> d;
  u32 bw = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) d.v[j] = subb(a.v[j], C::P[j], bw);
  return fp_select<C>(bw != 0, a, d);
}

template <class C>
BGLS_HD Fp<C> fp_add(const Fp<C>& a, const Fp<C>& b) {
  return fp_reduce_once<C>(fp_add_nr<C>(a, b));
}

template <class C>
BGLS_HD Fp<C> fp_sub(const Fp<C>& a, const Fp<C>& b) {
  Fp<C> d;
  u32 bw = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) d.v[j] = subb(a.v[j], b.v[j], bw);
  u32 mask = 0u - bw;
  u32 c = 0;
  Fp<C> r;
#pragma unroll
  for (int j = 0; j < C::L; ++j) r.v[j] = addc(d.v[j], C::P[j] & mask, c);
  return r;
}

template <class C>
BGLS_HD Fp<C> fp_neg(const Fp<C>& a) {
  Fp<C> d;
  u32 bw = 0;
  u32 nz = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) {
    d.v[j] = subb(C::P[j], a.v[j], bw);
    nz |= a.v[j];
  }
  u32 mask = nz ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int j = 0; j < C::L; ++j) d.v[j] &= mask;
  return d;
}

template <class C>
BGLS_HD Fp<C> fp_dbl(const Fp<C>& a) {
  return fp_add<C>(a, a);
}

// a / 2 mod p without a multiplication: (a + (a odd ? p : 0)) >> 1.  Valid on Montgomery residues as well (halving
// commutes with the factor R).  a + p < 2^(32 L) for both moduli (254 and 381 bits), so the sum needs no extra limb.
template <class C>
BGLS_HD Fp<C> fp_half(const Fp<C>& a) {
  const u32 mask = 0u - (a.v[0] & 1u);
  Fp<C> t;
  u32 c = 0;
#pragma unroll
  for (int j = 0; j < C::L; ++j) t.v[j] = addc(a.v[j], C::P[j] & mask, c);
  Fp<C> r;
#pragma unroll
  for (int j = 0; j < C::L; ++j) r.v[j] = (t.v[j] >> 1) | (j + 1 < C::L ? (t.v[j + 1] << 31) : 0u);
  return r;
}

template <class C>
BGLS_HD Fp<C> fp_mul3(const Fp<C>& a) {
  return fp_add<C>(fp_dbl<C>(a), a);
}

// ---------------------------------------------------------------- wide (2L-limb) helpers
template <int N>
BGLS_HD void w_add(u32 (&r)[N], const u32 (&a)[N], const u32 (&b)[N]) {
  u32 c = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) r[j] = addc(a[j], b[j], c);
}
template <int N>
BGLS_HD void w_sub(u32 (&r)[N], const u32 (&a)[N], const u32 (&b)[N]) {
  u32 c = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) r[j] = subb(a[j], b[j], c);
}

// t[0..2L) = a * b   (operand scanning; one v_mad_u64_u32 + one v_addc per limb product)
template <class C>
BGLS_HD void mul_wide(u32 (&t)[2 * C::L], const u32 (&a)[C::L], const u32 (&b)[C::L]) {
  constexpr int L = C::L;
  {
    u64 P[L];
#pragma unroll
    for (int j = 0; j < L; ++j) P[j] = (u64)a[j] * b[0];
    t[0] = (u32)P[0];
    u32 c = 0;
#pragma unroll
    for (int j = 1; j < L; ++j) t[j] = addc((u32)P[j], (u32)(P[j - 1] >> 32), c);
    t[L] = (u32)(P[L - 1] >> 32) + c;
  }
#pragma unroll
  for (int i = 1; i < L; ++i) {
    u64 P[L];
#pragma unroll
    for (int j = 0; j < L; ++j) P[j] = (u64)a[j] * b[i] + t[i + j];
    t[i] = (u32)P[0];
    u32 c = 0;
#pragma unroll
    for (int j = 1; j < L; ++j) t[i + j] = addc((u32)P[j], (u32)(P[j - 1] >> 32), c);
    t[i + L] = (u32)(P[L - 1] >> 32) + c;
  }
}


// t = a^2: off-diagonal products once, doubled, plus the diagonal.
template <class C>
BGLS_HD void sqr_wide(u32 (&t)[2 * C::L], const u32 (&a)[C::L]) {
  constexpr int L = C::L;
#pragma unroll
  for (int k = 0; k < 2 * L; ++k) t[k] = 0;
  // off-diagonal: rows i, columns j > i
#pragma unroll
  for (int i = 0; i < L - 1; ++i) {
    u32 hi = 0, c = 0;
#pragma unroll
    for (int j = i + 1; j < L; ++j) {
      u64 P = (u64)a[j] * a[i] + t[i + j];
      t[i + j] = addc((u32)P, hi, c);
      hi = (u32)(P >> 32);
    }
    t[i + L] = hi + c;
  }
  // double
  {
    u32 c = 0;
#pragma unroll
    for (int k = 1; k < 2 * L - 1; ++k) t[k] = addc(t[k], t[k], c);
    t[2 * L - 1] = c;
  }
  // diagonal
  {
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
      u64 P = (u64)a[i] * a[i];
      t[2 * i] = addc(t[2 * i], (u32)P, c);
      t[2 * i + 1] = addc(t[2 * i + 1], (u32)(P >> 32), c);
    }
  }
}

// Montgomery reduction without the final conditional subtraction: returns (t + m p) / 2^(32L),
// which is < t / 2^(32L) + p.  Clobbers t.
template <class C>
BGLS_HD Fp<C> redc_raw(u32 (&t)[2 * C::L]) {
  constexpr int L = C::L;
  u32 top = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) {
    u32 m = t[i] * C::N0INV;
    u64 P[L];
#pragma unroll
    for (int j = 0; j < L; ++j) P[j] = (u64)m * C::P[j] + t[i + j];
    u32 c = 0;
#pragma unroll
    for (int j = 1; j < L; ++j) t[i + j] = addc((u32)P[j], (u32)(P[j - 1] >> 32), c);
    u64 s = (u64)t[i + L] + (u32)(P[L - 1] >> 32);
    s += c;
    s += top;
    t[i + L] = (u32)s;
    top = (u32)(s >> 32);
  }
  Fp<C> r;
#pragma unroll
  for (int j = 0; j < L; ++j) r.v[j] = t[L + j];
  return r;
}

// Montgomery reduction: t < p * 2^(32L)  ->  t / 2^(32L) mod p, fully reduced.  Clobbers t.
template <class C>
BGLS_HD Fp<C> redc(u32 (&t)[2 * C::L]) {
  return fp_reduce_once<C>(redc_raw<C>(t));
}

// Same for lazily accumulated inputs t < K p * 2^(32L) (result < (K+1) p before the K subtractions).
template <class C, int K>
BGLS_HD Fp<C> redc_k(u32 (&t)[2 * C::L]) {
  Fp<C> r = redc_raw<C>(t);
#pragma unroll
  for (int k = 0; k < K; ++k) r = fp_reduce_once<C>(r);
  return r;
}

template <class C>
BGLS_FN Fp<C> fp_mul(const Fp<C>& a, const Fp<C>& b) {
  u32 t[2 * C::L];
  mul_wide<C>(t, a.v, b.v);
  return redc<C>(t);
}

template <class C>
BGLS_FN Fp<C> fp_sqr(const Fp<C>& a) {
  u32 t[2 * C::L];
  sqr_wide<C>(t, a.v);
  return redc<C>(t);
}

// plain integer (< 2^(32L)) -> Montgomery form, fully reduced (valid for ANY L-limb input)
template <class C>
BGLS_HD Fp<C> fp_to_mont(const Fp<C>& a) {
  return fp_mul<C>(a, fp_load<C>(C::R2));
}

// Montgomery form -> canonical integer in [0, p)
template <class C>
BGLS_HD Fp<C> fp_from_mont(const Fp<C>& a) {
  u32 t[2 * C::L];
#pragma unroll
  for (int j = 0; j < C::L; ++j) {
    t[j] = a.v[j];
    t[C::L + j] = 0;
  }
  return redc<C>(t);
}

// a^e for a public, wave-uniform exponent e given as NL little-endian u32 limbs.
template <class C, int NL>
BGLS_FN Fp<C> fp_pow(const Fp<C>& a, const u32* e) {
  Fp<C> r = fp_one<C>();
  bool started = false;
  for (int i = NL * 32 - 1; i >= 0; --i) {
    u32 bit = (e[i >> 5] >> (i & 31)) & 1u;
    if (started) r = fp_sqr<C>(r);
    if (bit) {
      r = started ? fp_mul<C>(r, a) : a;
      started = true;
    }
  }
  return r;
}

// In-place variants for hot loops (no call, operands stay in VGPRs)
template <class C>
BGLS_HD Fp<C> fp_mul_inl(const Fp<C>& a, const Fp<C>& b) {
  u32 t[2 * C::L];
  mul_wide<C>(t, a.v, b.v);
  return redc<C>(t);
}
template <class C>
BGLS_HD Fp<C> fp_sqr_inl(const Fp<C>& a) {
  u32 t[2 * C::L];
  sqr_wide<C>(t, a.v);
  return redc<C>(t);
}

// a^e, fixed 4-bit windows, public wave-uniform exponent (NL limbs); the loop body is expanded
// in place once (4 squarings + 1 multiplication), the 16-entry table lives in private memory.
template <class C, int NL>
BGLS_FN Fp<C> fp_pow_w4(const Fp<C>& a, const u32* e) {
  Fp<C> tab[16];
  tab[0] = fp_one<C>();
  tab[1] = a;
#pragma unroll 1
  for (int k = 2; k < 16; ++k) tab[k] = fp_mul_inl<C>(tab[k - 1], a);
  int d = NL * 8 - 1;
  while (d > 0 && ((e[d >> 3] >> ((d & 7) * 4)) & 15u) == 0) --d;
  Fp<C> r = tab[(e[d >> 3] >> ((d & 7) * 4)) & 15u];
#pragma unroll 1
  for (--d; d >= 0; --d) {
    r = fp_sqr_inl<C>(r);
    r = fp_sqr_inl<C>(r);
    r = fp_sqr_inl<C>(r);
    r = fp_sqr_inl<C>(r);
    const u32 w = (e[d >> 3] >> ((d & 7) * 4)) & 15u;
    if (w) r = fp_mul_inl<C>(r, tab[w]);
  }
  return r;
}

// Modular inverse by the binary extended Euclidean algorithm (0 -> 0).  The inverse is unique, so
// this returns the same field element as the reference's big.Int ModInverse / a^(p-2)
// (curves/hash.go:109,139) at a fraction of the cost of an exponentiation.  Since round 4 fp_inv is the division-step form
// below (fp_inv_ds); this one stays for the host-side cross-check and the microbenchmark (tools/mb_inv.hip).
template <class C>
BGLS_FN Fp<C> fp_inv_euclid(const Fp<C>& a) {
  constexpr int L = C::L;
  if (fp_is_zero<C>(a)) return a;
  u32 u[L], v[L];
  Fp<C> x1 = fp_zero<C>(), x2 = fp_zero<C>();
  x1.v[0] = 1;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    u[j] = a.v[j];
    v[j] = C::P[j];
  }
  auto is_one = [](const u32(&w)[L]) {
    u32 o = w[0] ^ 1u;
#pragma unroll
    for (int j = 1; j < L; ++j) o |= w[j];
    return o == 0;
  };
  auto shr1 = [](u32(&w)[L], u32 top) {
#pragma unroll
    for (int j = 0; j < L - 1; ++j) w[j] = (w[j] >> 1) | (w[j + 1] << 31);
    w[L - 1] = (w[L - 1] >> 1) | (top << 31);
  };
  auto half_mod = [&](Fp<C>& x) {       // x/2 mod p
    u32 c = 0;
    const u32 mask = 0u - (x.v[0] & 1u);
#pragma unroll
    for (int j = 0; j < L; ++j) x.v[j] = addc(x.v[j], C::P[j] & mask, c);
    shr1(x.v, c);
  };
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1u)) {
      shr1(u, 0);
      half_mod(x1);
    }
    while (!(v[0] & 1u)) {
      shr1(v, 0);
      half_mod(x2);
    }
    u32 bw = 0;
    u32 d[L];
#pragma unroll
    for (int j = 0; j < L; ++j) d[j] = subb(u[j], v[j], bw);
    if (!bw) {                           // u >= v
#pragma unroll
      for (int j = 0; j < L; ++j) u[j] = d[j];
      x1 = fp_sub<C>(x1, x2);
    } else {
      bw = 0;
#pragma unroll
      for (int j = 0; j < L; ++j) v[j] = subb(v[j], u[j], bw);
      x2 = fp_sub<C>(x2, x1);
    }
  }
  Fp<C> r = is_one(u) ? x1 : x2;        // (a R)^-1 as a plain residue
  const Fp<C> r2 = fp_load<C>(C::R2);
  return fp_mul_inl<C>(fp_mul_inl<C>(r, r2), r2);   // -> a^-1 R
}

// Modular inverse by BATCHED DIVISION STEPS (Bernstein-Yang "safegcd" in the form libsecp256k1 made familiar: half-delta
// division steps, 30 per batch on the low words, one 2 x 2 transition matrix per batch applied to the full-width numbers).
// Round 4: fp_inv's binary Euclid is two nested data-dependent loops -- a lone lane needs ~110 us (alt-bn128) / ~250 us
// (BLS12-381) for it, a wave whose 64 lanes hold different values 300 / 610 us (tools/mb_inv.hip), and every latency-bound
// record pays one or two of them (the easy part of the final exponentiation, the to-affine step of a key sum).  Here the
// control flow does not depend on the data at all: a batch is 30 branch-free steps on two 32-bit words and four rows of
// multiply-adds over NW signed 30-bit limbs, and the only loop condition (g = 0 on every active lane) is wave-uniform.
// Invariants: d a = f, e a = g (mod p); f = p, g = a, d = 0, e = 1 at the start; at the end g = 0, f = +-1 and a^-1 = f d.
// Same field element as fp_inv (the inverse is unique); 0 -> 0.
template <class C>
BGLS_FN Fp<C> fp_inv_ds(const Fp<C>& a) {
  constexpr int L = C::L;
  constexpr int NW = (32 * L + 29) / 30 + ((32 * L) % 30 == 0 ? 1 : 0);      // 9 limbs for 256 bits, 13 for 384: a sign bit to spare
  constexpr i32 M30 = (i32)((1u << 30) - 1u);
  static_assert(30 * NW >= 32 * L + 2, "limbs hold (-2p, 2p)");
  auto limb = [](const u32* w, int i) -> i32 {                               // bits [30 i, 30 i + 30) of an L-word number
    const int bit = 30 * i, q = bit >> 5, sh = bit & 31;
    if (q >= L) return 0;
    u64 x = w[q];
    if (q + 1 < L) x |= (u64)w[q + 1] << 32;
    return (i32)((u32)(x >> sh) & (u32)M30);
  };
  i32 f[NW], g[NW], d[NW], e[NW], m[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    m[i] = limb(C::P, i);
    f[i] = m[i];
    g[i] = limb(a.v, i);
    d[i] = 0;
    e[i] = i == 0 ? 1 : 0;
  }
  u32 minv = C::P[0];                                                         // p^-1 mod 2^32 by Newton's iteration (p odd)
  minv *= 2u - C::P[0] * minv;
  minv *= 2u - C::P[0] * minv;
  minv *= 2u - C::P[0] * minv;
  minv *= 2u - C::P[0] * minv;
  minv *= 2u - C::P[0] * minv;
  i32 zeta = -1;
  for (int batch = 0; batch < 64; ++batch) {                                  // 590 steps suffice for 256 bits, ~870 for 384: 20 / 29 batches
    i32 u = 1, v = 0, q = 0, r = 1;
    {
      u32 fl = (u32)f[0] | ((u32)f[1] << 30), gl = (u32)g[0] | ((u32)g[1] << 30);
#pragma unroll
      for (int i = 0; i < 30; ++i) {
        i32 c1 = zeta >> 31;
        const i32 c2 = -(i32)(gl & 1u);
        const u32 x = (fl ^ (u32)c1) - (u32)c1;
        const i32 y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        gl += x & (u32)c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;
        zeta = (zeta ^ c1) - 1;
        fl += gl & (u32)c1;
        u += q & c1;
        v += r & c1;
        gl >>= 1;
        u = (i32)((u32)u << 1);
        v = (i32)((u32)v << 1);
      }
    }
    // 2^30 (f', g') = (u f + v g, q f + r g);  (d', e') = the same combination divided by 2^30 modulo p
    {
      const i32 sd = d[NW - 1] >> 31, se = e[NW - 1] >> 31;
      i32 md = (u & sd) + (v & se), me = (q & sd) + (r & se);
      i64 cd = (i64)u * d[0] + (i64)v * e[0], ce = (i64)q * d[0] + (i64)r * e[0];
      md -= (i32)((minv * (u32)cd + (u32)md) & (u32)M30);
      me -= (i32)((minv * (u32)ce + (u32)me) & (u32)M30);
      cd += (i64)m[0] * md;
      ce += (i64)m[0] * me;
      cd >>= 30;
      ce >>= 30;
#pragma unroll
      for (int i = 1; i < NW; ++i) {
        cd += (i64)u * d[i] + (i64)v * e[i] + (i64)m[i] * md;
        ce += (i64)q * d[i] + (i64)r * e[i] + (i64)m[i] * me;
        d[i - 1] = (i32)cd & M30;
        e[i - 1] = (i32)ce & M30;
        cd >>= 30;
        ce >>= 30;
      }
      d[NW - 1] = (i32)cd;
      e[NW - 1] = (i32)ce;
    }
    u32 gnz;
    {
      i64 cf = (i64)u * f[0] + (i64)v * g[0], cg = (i64)q * f[0] + (i64)r * g[0];
      cf >>= 30;
      cg >>= 30;
      gnz = 0;
#pragma unroll
      for (int i = 1; i < NW; ++i) {
        cf += (i64)u * f[i] + (i64)v * g[i];
        cg += (i64)q * f[i] + (i64)r * g[i];
        f[i - 1] = (i32)cf & M30;
        g[i - 1] = (i32)cg & M30;
        gnz |= (u32)g[i - 1];
        cf >>= 30;
        cg >>= 30;
      }
      f[NW - 1] = (i32)cf;
      g[NW - 1] = (i32)cg;
      gnz |= (u32)g[NW - 1];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(gnz != 0) == 0ull) break;                                    // wave-uniform: every active lane is done
#else
    if (gnz == 0) break;
#endif
  }
  // f = +-1 (gcd): a^-1 = f d, d in (-2p, p).  Bring sign(f) d + 2p into 32-bit words and subtract p while it is >= p.
  u32 fone = (u32)f[0] ^ 1u, fmone = (u32)(f[0] ^ M30);
#pragma unroll
  for (int i = 1; i < NW; ++i) { fone |= (u32)f[i]; fmone |= (u32)(f[i] ^ (i == NW - 1 ? -1 : M30)); }
  if (fone != 0 && fmone != 0) return fp_zero<C>();                           // a = 0 (f stays p): no inverse
  const i32 sf = fone == 0 ? 0 : -1;                                          // negate d when f = -1
  i64 c = 0;
  u32 w[L + 1];
  {
    // value = sum (sign d_i + 2 m_i) 2^(30 i), accumulated into 32-bit words
    u64 acc = 0;
    int have = 0, wi = 0;
    i64 carry = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      carry += (i64)((d[i] ^ sf) - sf) + 2 * (i64)m[i];
      const u32 lo = (u32)carry & (u32)M30;                                   // 30 bits of the non-negative total
      carry >>= 30;
      acc |= (u64)lo << have;
      have += 30;
      if (have >= 32) {
        if (wi <= L) w[wi] = (u32)acc;
        ++wi;
        acc >>= 32;
        have -= 32;
      }
    }
    acc |= (u64)(u32)carry << have;                                           // what is left above the last limb (small, non-negative)
    while (wi <= L) { w[wi++] = (u32)acc; acc >>= 32; }
    (void)c;
  }
  Fp<C> rr;
#pragma unroll
  for (int j = 0; j < L; ++j) rr.v[j] = w[j];
  u32 top = w[L];
#pragma unroll
  for (int k = 0; k < 4; ++k) {                                               // value < 4p: at most three subtractions
    u32 bw = 0;
    u32 t[L];
#pragma unroll
    for (int j = 0; j < L; ++j) t[j] = subb(rr.v[j], C::P[j], bw);
    const bool ge = top != 0 || bw == 0;
    if (ge) {
#pragma unroll
      for (int j = 0; j < L; ++j) rr.v[j] = t[j];
      top -= bw;
    }
  }
  const Fp<C> r2 = fp_load<C>(C::R2);
  return fp_mul_inl<C>(fp_mul_inl<C>(rr, r2), r2);                            // (a R)^-1 -> a^-1 R
}
// THE inverse of the library (0 -> 0)
template <class C>
BGLS_HD Fp<C> fp_inv(const Fp<C>& a) {
  return fp_inv_ds<C>(a);
}

// Legendre symbol (a/p) by the binary Jacobi algorithm: +1, -1, or 0 for a == 0.  Decides exactly what
// the reference decides with an exponentiation -- isQuadRes (curves/hash.go:254-265) and the
// "root^2 == y^2" acceptance test of try-and-increment (curves/hash.go:62-66) -- at a few percent of the
// cost (shifts and subtractions only).  Works on the Montgomery residue directly: (aR/p) = (a/p) because
// R = 2^(32L) is a perfect square.
// Round 6: the pair shrinks by about three bits per iteration, so most iterations shift, subtract and select limbs that are zero on every
// lane.  The loop runs in PHASES of LC = L, L - 2, .. 2 limbs: a phase ends when the top two limbs of both numbers are zero on every lane of the
// wave that is still at work (one ballot per iteration), and the next one touches two limbs less -- same iterations, same decisions, on average
// half the instructions (the iteration is 130 instructions at twelve limbs, 11 L + 16 in general).
template <int L, int LC>
BGLS_HD void fp_jacobi_phase(u32 (&a)[L], u32 (&n)[L], u32& t, bool& done) {
  for (int guard = 0; guard < 64 * L + 64; ++guard) {   // at most ~2 * 32L rounds in all: every round removes a bit
    if (!done) {
      const u32 low = a[0];
      const u32 sft = low ? (u32)__builtin_ctz(low) : 31u;
#pragma unroll
      for (int j = 0; j < LC - 1; ++j) a[j] = (u32)((((u64)a[j + 1] << 32) | a[j]) >> sft);
      a[LC - 1] >>= sft;
      const u32 n8 = n[0] & 7u;
      t ^= (sft & 1u) & (u32)(n8 == 3u || n8 == 5u);                 // (2/n) = -1 iff n = 3,5 mod 8
      const bool odd = (a[0] & 1u) != 0;
      u32 d[LC], e[LC];
      u32 bw = 0, bw2 = 0;
#pragma unroll
      for (int j = 0; j < LC; ++j) d[j] = subb(a[j], n[j], bw);       // a - n
#pragma unroll
      for (int j = 0; j < LC; ++j) e[j] = subb(n[j], a[j], bw2);      // n - a
      const bool lt = bw != 0;
      const bool swp = odd && lt;
      t ^= (u32)(swp && (a[0] & 3u) == 3u && (n[0] & 3u) == 3u);     // quadratic reciprocity
      u32 z = 0;
#pragma unroll
      for (int j = 0; j < LC; ++j) {
        const u32 na = odd ? (lt ? e[j] : d[j]) : a[j];
        n[j] = swp ? a[j] : n[j];
        a[j] = na;
        z |= na;
      }
      if (z == 0) done = true;
    }
    bool again = !done, shrink = false;
    if constexpr (LC > 2) shrink = !done && ((a[LC - 1] | a[LC - 2] | n[LC - 1] | n[LC - 2]) != 0);      // this lane still needs LC limbs
#if defined(__HIP_DEVICE_COMPILE__)
    again = __ballot(again) != 0;
    if constexpr (LC > 2) shrink = __ballot(shrink) != 0;
#endif
    if (!again) return;
    if constexpr (LC > 2) {
      if (!shrink) return;            // every lane at work fits LC - 2 limbs: the caller goes on there
    }
  }
}
template <int L, int LC>
BGLS_HD void fp_jacobi_phases(u32 (&a)[L], u32 (&n)[L], u32& t, bool& done) {
  fp_jacobi_phase<L, LC>(a, n, t, done);
  if constexpr (LC > 2) fp_jacobi_phases<L, LC - 2>(a, n, t, done);
}
template <class C>
BGLS_FN int fp_jacobi(const Fp<C>& x) {
  constexpr int L = C::L;
  static_assert(L % 2 == 0, "limb pairs");
  u32 a[L], n[L];
  u32 nz = 0;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    a[j] = x.v[j];
    n[j] = C::P[j];
    nz |= a[j];
  }
  u32 t = 0;                                  // sign: 1 means -1
  // One uniform iteration = strip up to 31 factors of two from a, then (if a became odd) order the pair,
  // apply reciprocity and subtract -- all with selects, so the lanes of a wave do not serialise on
  // data-dependent branches.  A lane whose input is zero sits the loop out.
  bool done = nz == 0;
  fp_jacobi_phases<L, L>(a, n, t, done);
  if (nz == 0) return 0;
  u32 one = n[0] ^ 1u;
#pragma unroll
  for (int j = 1; j < L; ++j) one |= n[j];
  if (one != 0) return 0;                     // gcd > 1 (impossible for prime p and a != 0)
  return t ? -1 : 1;
}

// candidate square root a^((p+1)/4) (calcQuadRes, curves/hash.go:178-190); caller checks r^2 == a
template <class C>
BGLS_HD Fp<C> fp_sqrt_candidate(const Fp<C>& a) {
  return fp_pow_w4<C, C::L>(a, C::EXP_SQRT);
}

// big-endian bytes (FP_BYTES) -> limbs (plain integer)
template <class C>
BGLS_HD Fp<C> fp_from_be(const uint8_t* b) {
  Fp<C> r;
#pragma unroll
  for (int j = 0; j < C::L; ++j) {
    const uint8_t* q = b + 4 * (C::L - 1 - j);
    r.v[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
  }
  return r;
}

template <class C>
BGLS_HD void fp_to_be(uint8_t* b, const Fp<C>& a) {
#pragma unroll
  for (int j = 0; j < C::L; ++j) {
    uint8_t* q = b + 4 * (C::L - 1 - j);
    q[0] = (uint8_t)(a.v[j] >> 24);
    q[1] = (uint8_t)(a.v[j] >> 16);
    q[2] = (uint8_t)(a.v[j] >> 8);
    q[3] = (uint8_t)(a.v[j]);
  }
}

}  // namespace bgls
