// G2 key sums (the main pass of AggregatePoints, curves/curve.go:73-121) on LANE PAIRS in the carry-free form: rx_jacpair.hpp.
//   k_sumpair_main      pair t adds keys t, t + T, t + 2T, ... (T = lane pairs in the launch) into its own Jacobian sum
//   k_sumpairseg_main   the same for nsets key sets in one launch (KoskVerifyBatchMultiSignature, bgls/blsKosk.go:126-133)
// The 32 sums of a block (one wave) are then added in the block, five levels through LDS (general Jacobian additions shared by
// the two lane pairs whose sums they join), so that a block leaves ONE partial: 3072 for 2^20 keys instead of 98 304, and the tree above them (k_sum_coop:
// one wave per addition) starts where it has few enough additions to be worth a wave each.  Partials leave in the library's
// 32-bit Montgomery Jacobian form (the even lane writes the real parts, the odd lane the imaginary parts).  Own translation unit: everything is
// expanded in place for this kernel's register budget (three waves per SIMD).
#include "dev_common.hpp"
#include "coop.hpp"
#include "rx_jacpair.hpp"
#include "launch.hpp"

namespace bgls {

// SRC: where the keys come from.  0 = wire bytes (parsed, canonical encoding and the curve equation checked per key);
// 1 = a key set's Montgomery affine points (no checks: validated at upload; two conversions into the carry-free form per key);
// 2 = a key set's SUM-READY records (k_g2_sumready below): the carry-free limbs of (x_re, y_re | x_im, y_im), 2 NL dwords per
// lane, fetched with 16-byte loads -- nothing but the mixed addition is left in the loop (11 Fp2 products per key instead of 16).
template <class C>
struct SumRec {
  static constexpr int LANE_DW = (2 * C::RX_NL + 3) & ~3;      // x limbs, y limbs, padding
};
template <class C, int SRC>
__device__ __forceinline__ bool sumpair_fetch(AffP<C>& q, const uint8_t* pts, size_t k, bool odd) {
  if constexpr (SRC == 2) {
    constexpr int N = C::RX_NL;
    constexpr int RS = SumRec<C>::LANE_DW;                      // a lane's record: whole 16-byte words (2 N limbs, padded: 20 / 20 / 28 dwords)
    const uint4* rec = reinterpret_cast<const uint4*>(pts) + (k * 2 * RS + (odd ? RS : 0)) / 4;
    u32 w[RS];
#pragma unroll
    for (int j = 0; j < RS / 4; ++j) {
      const uint4 v = rec[j];
      w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
    }
    q.inf = (w[0] >> 31) != 0;                                  // the flag rides on bit 31 of the record's first limb (limbs are 28 / 29 bits)
    w[0] &= 0x7FFFFFFFu;
#pragma unroll
    for (int i = 0; i < N; ++i) { q.x.v[i] = (i32)w[i]; q.y.v[i] = (i32)w[N + i]; }
    return true;
  } else if constexpr (SRC == 1) {
    q = affp_from_mont<C>(reinterpret_cast<const Aff<F2<C>>*>(pts)[k], odd);
    return true;
  } else {
    const bool ok = affp_from_bytes<C>(q, pts + k * 4 * C::FP_BYTES, odd);
    return affp_on_curve<C>(q, odd) && ok;
  }
}

template <class C>
__device__ __forceinline__ void sumpair_store(Jac<F2<C>>* out, const JacP<C>& acc, bool odd) {
  Fp<C>* o = reinterpret_cast<Fp<C>*>(out) + (odd ? 1 : 0);      // Jac = X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1
  if (acc.inf) {                                                  // jac_inf: (1, 1, 0)
    const Fp<C> one = odd ? fp_zero<C>() : fp_one<C>();
    o[0] = one;
    o[2] = one;
    o[4] = fp_zero<C>();
    return;
  }
  o[0] = sxp_to_mont<C>(acc.X);
  o[2] = sxp_to_mont<C>(acc.Y);
  o[4] = sxp_to_mont<C>(acc.Z);
}

// the block's 32 per-pair sums -> one (in pair 0), five levels through LDS (limb-major columns of the 64 lanes: conflict-free).
//
// A level adds the sums of pairs s .. 2s-1 ("B") to those of pairs 0 .. s-1 ("A").  Round 3 let A do the whole general addition
// (add-2007-bl, 16 Fp2 products) while B idled -- and a wave's instruction costs its SIMD the same with 16 or 32 pairs active, so
// the five levels cost as much as 7 of the ~11 keys a pair adds in its main loop.  Here A and B share the addition: both
// points are in registers where they are needed, the products split 8 + 7 with a critical path of NINE (one squaring, three
// products, three products, two), and three exchanges of two values each go through the columns:
//     both   ZZ = Z^2                                    | exchange Z, ZZ
//     both   U = X oZZ,  S = (Y oZ) oZZ                  | exchange U, S       H = U2 - U1,  r = 2 (S2 - S1)
//     A      I = (2H)^2,  J = H I,  V = U1 I
//     B      R2 = r^2,  (Z1 + Z2)^2,  Z3 = ((Z1 + Z2)^2 - Z1Z1 - Z2Z2) H        | B hands R2, Z3 to A
//     A      X3 = R2 - J - 2V,  Y3 = r (V - X3) - 2 S1 J
// one instruction stream for both roles (operand selects, no divergent branches).  The exceptional cases of the group law (a
// sum at infinity, equal x: the same key in both halves) are decided for the whole wave by ballot and take the plain
// one-pair addition for that level, so the result is the same point in every case (AggregatePoints, curves/curve.go:73-121).
template <class C>
__device__ __noinline__ JacP<C> sumpair_block_tree(JacP<C> acc, bool odd) {
  constexpr int N = C::RX_NL;
  __shared__ i32 cols[3 * N * 64];
  __shared__ int infs[64];
  const int lane = threadIdx.x & 63, pi = lane >> 1;
  auto put = [&](int slot, const auto& v) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < N; ++i) cols[(slot * N + i) * 64 + lane] = v.v[i];
  };
  auto get = [&](int slot, int src) __attribute__((always_inline)) {
    Sx<C, SX_F> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = cols[(slot * N + i) * 64 + src];
    return r;
  };
#pragma unroll 1
  for (int s = 16; s >= 1; s >>= 1) {
    const bool isA = pi < s, isB = pi >= s && pi < 2 * s;
    const int partner = isA ? lane + 2 * s : (isB ? lane - 2 * s : lane);
    bool plain = __ballot((isA || isB) && acc.inf) != 0;            // a sum at infinity somewhere: the plain addition handles it
    Sx<C, SX_T> U, S;
    Sx<C, SX_F> H, rr, oZ, oZZ;
    Sx<C, SX_T> ZZ;
    if (!plain) {
      ZZ = pair_sqr<C>(acc.Z, odd);
      put(0, acc.Z); put(1, ZZ);
      wave_sync();
      oZ = get(0, partner); oZZ = get(1, partner);
      wave_sync();
      U = pair_mul<C>(acc.X, oZZ, odd);
      S = pair_mul<C>(pair_mul<C>(acc.Y, oZ, odd), oZZ, odd);
      put(0, U); put(1, S);
      wave_sync();
      const Sx<C, SX_F> oU = get(0, partner), oS = get(1, partner);
      wave_sync();
      const auto du = sx_sub<C>(oU, U), ds = sx_sub<C>(oS, S);      // A: U2 - U1, S2 - S1;  B: the negatives
      const auto Hd = sx_select<C>(isA, du, sx_neg<C>(du));
      const auto Rd = sx_select<C>(isA, ds, sx_neg<C>(ds));
      plain = __ballot((isA || isB) && pair_both(sx_is_zero_mod_p<C>(Hd))) != 0;     // equal x somewhere: P = Q or P = -Q
      H = sx_normf<C>(Hd);
      rr = sx_normf<C>(sx_mulc<2, C>(Rd));
    }
    if (plain) {                                                    // round 3's level: B publishes its sum, A adds it
      put(0, acc.X); put(1, acc.Y); put(2, acc.Z);
      infs[lane] = acc.inf ? 1 : 0;
      wave_sync();
      if (isA) {
        JacP<C> o;
        o.X = get(0, partner); o.Y = get(1, partner); o.Z = get(2, partner);
        o.inf = infs[partner] != 0;
        acc = jacp_add<C>(acc, o, odd);
      }
      wave_sync();
      continue;
    }
    const Sx<C, SX_F> a1 = sx_select<C>(isA, sx_normf<C>(sx_mulc<2, C>(H)), rr);
    const Sx<C, SX_T> P1 = pair_sqr<C>(a1, odd);                                     // A: I = (2H)^2        B: R2 = r^2
    const Sx<C, SX_F> Zs = sx_normf<C>(sx_add<C>(acc.Z, oZ));
    const Sx<C, SX_F> P1f = sx_as<SX_F, C>(P1);
    const Sx<C, SX_T> P2 = pair_mul<C>(sx_select<C>(isA, H, Zs), sx_select<C>(isA, P1f, Zs), odd);      // A: J = H I   B: (Z1 + Z2)^2
    const Sx<C, SX_F> zz = sx_normf<C>(sx_sub<C>(sx_sub<C>(P2, ZZ), oZZ));
    const Sx<C, SX_T> P3 = pair_mul<C>(sx_select<C>(isA, sx_as<SX_F, C>(U), zz), sx_select<C>(isA, P1f, H), odd);   // A: V = U1 I  B: Z3 = zz H
    put(0, P1); put(1, P3);
    wave_sync();
    if (isA) {
      const Sx<C, SX_F> R2 = get(0, partner), Z3 = get(1, partner);
      JacP<C> r;
      r.X = sx_normf<C>(sx_sub<C>(sx_sub<C>(R2, P2), sx_mulc<2, C>(P3)));
      r.Y = pair_mulsub_f<C>(rr, sx_normf<C>(sx_sub<C>(P3, r.X)), sx_mulc<2, C>(S), P2, odd);
      r.Z = Z3;
      r.inf = false;
      acc = r;
    }
    wave_sync();
  }
  return acc;
}

template <class C, int SRC>
__global__ void __launch_bounds__(64, 3) k_sumpair_main(const uint8_t* pts, size_t n, Jac<F2<C>>* out, uint32_t* flags) {
  const size_t T = (size_t)gridDim.x * 32;
  const size_t t = (size_t)blockIdx.x * 32 + (threadIdx.x >> 1);
  const bool odd = threadIdx.x & 1;
  JacP<C> acc = jacp_inf<C>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = t; k < n; k += T) {
    AffP<C> q;
    bad = !sumpair_fetch<C, SRC>(q, pts, k, odd) || bad;
    acc = jacp_madd<C>(acc, q, odd);
  }
  if (bad && !odd) atomicOr(flags, FLAG_ENC);
  acc = sumpair_block_tree<C>(acc, odd);
  if (threadIdx.x < 2) sumpair_store<C>(out + blockIdx.x, acc, odd);
}

template <class C>
__global__ void __launch_bounds__(64, 3) k_sumpairseg_main(const uint8_t* pts, const uint64_t* off, unsigned P, Jac<F2<C>>* out, uint32_t* flags) {
  const unsigned per = P / 32;                 // blocks per set: P partials of one lane pair each
  const size_t b = blockIdx.x / per;
  const size_t t = (size_t)(blockIdx.x % per) * 32 + (threadIdx.x >> 1);
  const bool odd = threadIdx.x & 1;
  const size_t lo = off[b], hi = off[b + 1];
  JacP<C> acc = jacp_inf<C>();
  bool bad = false;
#pragma unroll 1
  for (size_t k = lo + t; k < hi; k += P) {
    AffP<C> q;
    bad = !sumpair_fetch<C, 0>(q, pts, k, odd) || bad;
    acc = jacp_madd<C>(acc, q, odd);
  }
  if (bad && !odd) atomicOr(flags, FLAG_ENC);
  acc = sumpair_block_tree<C>(acc, odd);
  if (threadIdx.x < 2) sumpair_store<C>(out + blockIdx.x, acc, odd);      // partials of set b: blocks b * per .. (b + 1) * per - 1
}

// a key set's sum-ready copy: per key 4 NL dwords = x_re y_re | x_im y_im in the carry-free form (tight limbs, value < 2 p),
// bit 31 of each half's first dword = point at infinity
template <class C>
__global__ void k_g2_sumready(const Aff<F2<C>>* in, size_t n, u32* out) {
  constexpr int N = C::RX_NL;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Aff<F2<C>> a = in[i];
  const Ux<C> xr = to_ux<C>(a.x.c0), xi = to_ux<C>(a.x.c1), yr = to_ux<C>(a.y.c0), yi = to_ux<C>(a.y.c1);
  constexpr int RS = SumRec<C>::LANE_DW;
  u32* o = out + i * 2 * RS;
  const u32 f = a.inf ? 0x80000000u : 0u;
#pragma unroll
  for (int k = 0; k < RS; ++k) o[k] = o[RS + k] = 0u;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    o[k] = a.inf ? 0u : xr.v[k];
    o[N + k] = a.inf ? 0u : yr.v[k];
    o[RS + k] = a.inf ? 0u : xi.v[k];
    o[RS + N + k] = a.inf ? 0u : yi.v[k];
  }
  o[0] |= f;
  o[RS] |= f;
}

namespace kl {

// The number form the key-sum kernels run on: alt-bn128 on nine limbs of 29 bits (BN254W, as k_miller_x60 since round 5: 81 instead of
// 100 multiplier instructions per limb product, one reduction more per addition), BLS12-381 on fourteen of 28.  Points and partial
// sums cross the launchers in the library's 32-bit Montgomery form, whose layout does not depend on the form class.
template <class C>
struct SumForm { typedef C type; };
#ifndef SUM_BN_W28
template <>
struct SumForm<BN254> { typedef BN254W type; };
#endif

// `pairs` = lane pairs in the launch (a multiple of 32); pairs / 32 Jacobian partial sums are written, one per block.
// src: 0 wire bytes, 1 Montgomery affine points of a key set, 2 its sum-ready records
template <class C>
void sumpair_main(hipStream_t st, int src, const uint8_t* pts, size_t n, unsigned pairs, void* out, uint32_t* flags) {
  typedef typename SumForm<C>::type X;
  if (src == 2) k_sumpair_main<X, 2><<<pairs / 32, 64, 0, st>>>(pts, n, (Jac<F2<X>>*)out, flags);
  else if (src == 1) k_sumpair_main<X, 1><<<pairs / 32, 64, 0, st>>>(pts, n, (Jac<F2<X>>*)out, flags);
  else k_sumpair_main<X, 0><<<pairs / 32, 64, 0, st>>>(pts, n, (Jac<F2<X>>*)out, flags);
}
template <class C>
void g2_sumready(hipStream_t st, const void* mont, size_t n, void* out) {
  typedef typename SumForm<C>::type X;
  k_g2_sumready<X><<<nblk(n, 128), 128, 0, st>>>((const Aff<F2<X>>*)mont, n, (u32*)out);
}
template <class C>
size_t g2_sumready_bytes() { return 2 * SumRec<typename SumForm<C>::type>::LANE_DW * 4; }
template <class C>
void sumpairseg_main(hipStream_t st, const uint8_t* pts, const uint64_t* off, size_t nsets, unsigned P, void* out, uint32_t* flags) {
  typedef typename SumForm<C>::type X;
  k_sumpairseg_main<X><<<(unsigned)(nsets * (P / 32)), 64, 0, st>>>(pts, off, P, (Jac<F2<X>>*)out, flags);
}
template void sumpair_main<BN254>(hipStream_t, int, const uint8_t*, size_t, unsigned, void*, uint32_t*);
template void sumpair_main<BLS381>(hipStream_t, int, const uint8_t*, size_t, unsigned, void*, uint32_t*);
template void g2_sumready<BN254>(hipStream_t, const void*, size_t, void*);
template void g2_sumready<BLS381>(hipStream_t, const void*, size_t, void*);
template size_t g2_sumready_bytes<BN254>();
template size_t g2_sumready_bytes<BLS381>();
template void sumpairseg_main<BN254>(hipStream_t, const uint8_t*, const uint64_t*, size_t, unsigned, void*, uint32_t*);
template void sumpairseg_main<BLS381>(hipStream_t, const uint8_t*, const uint64_t*, size_t, unsigned, void*, uint32_t*);

}  // namespace kl
}  // namespace bgls
