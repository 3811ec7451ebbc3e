// The tree above the main pass of a G2 key sum (AggregatePoints, curves/curve.go:73-121) as ONE launch.
//
// The main pass (k_sumpair_main) leaves a few thousand Jacobian partial sums; every addition above them is on the critical
// path of a multi-signature check (verifyMultiSignature, bgls/bgls.go:89-92).  Round 3 ran the levels as one launch each
// (k_sum_coop: one wave per addition, 12 launches of ~20 us for 3072 partials, then k_jac_to_bytes).  Here one wave starts
// at every pair of leaves and CLIMBS: having the sum of node i of a level it parks that sum in the node's slot, takes a
// ticket on the sibling pair, and either leaves (first to arrive: the sibling's wave will pick the sum up) or adds the
// sibling's parked sum and moves on to the parent (second to arrive).  No wave ever waits: the dependencies are carried by
// the tickets, the launch costs the 12 dependent additions of its longest path and nothing else, and the wave that completes
// the root converts it to affine and writes the wire bytes (the reference's Point).  Additions are jac_coop.hpp's (one wave
// per addition, five levels of independent Fp2 products); the order in which two sums meet changes the Jacobian
// representative, never the point.
#include "dev_common.hpp"
#include <stdlib.h>
#include "jac_coop.hpp"
#include "points_inl.hpp"
#include "finalx.hpp"
#include "rx_pair.hpp"
#include "launch_tail.hpp"

using namespace bgls;

// The hand-over between waves below (relaxed agent-scope atomics, s_waitcnt vmcnt(0) in front of the ticket, no release / acquire fence) is correct
// under the memory behaviour of THIS architecture -- stores counted in vmcnt, sc1 write-through to the shared level -- and has only been validated
// there.  Building the library for anything else must fail loudly rather than verify against a stale key sum.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_sumtree.hip: the fence-free ticket hand-over is written for gfx950 (MI355X) only"
#endif

// Hand-over of a parked sum between two waves on different CUs / XCDs.  The per-XCD L2s are not coherent with each other and
// an agent-scope release fence writes a whole L2's dirty lines back (MI355X_MICROARCH.md, inter-workgroup visibility: ~3.5 us
// per __threadfence(), several times that next to the main pass's freshly written partials -- the first version of this kernel
// fenced twice per level and was SLOWER than twelve launches).  So the record travels as 8-byte relaxed agent-scope atomics:
// write-through (sc1) stores by 6 L / 2 lanes, drained (s_waitcnt vmcnt(0)) before the ticket is taken, sc1 loads behind
// the returned ticket on the other side.  No fence anywhere.
// ------------------------------------------------------------------------------------------------ the tree on the carry-free limbs (round 4b; the 32-bit form k_sum_tree it replaced was removed in round 5)
// The additions on the carry-free limbs, one Fp2 product per LANE PAIR (rx_pair.hpp), in FOUR rounds of independent products
// instead of jac_coop.hpp's five levels of 32-bit products (~20 us an addition on a lone wave: most of the ~25 us a level costs).
// The formulas are add-2007-bl written out so that nothing is more than four products deep:
//     Z1Z1 = Z1^2, Z2Z2 = Z2^2, ZZ = (Z1 + Z2)^2, t1 = Y1 Z2, t2 = Y2 Z1                                    (round 1)
//     U1 = X1 Z2Z2, U2 = X2 Z1Z1, S1 = t1 Z2Z2, S2 = t2 Z1Z1;    H = U2 - U1, r = 2 (S2 - S1)                (round 2)
//     I = (2H)^2, R2 = r^2, Z3 = (ZZ - Z1Z1 - Z2Z2) H, rU = r U1, rH = r H, sH = S1 H                        (round 3)
//     J = H I, V = U1 I, Ya = I (3 rU + rH - 2 sH), Yb = r R2                                                (round 4)
//     X3 = R2 - J - 2V,   Y3 = r (V - X3) - 2 S1 J = Ya - Yb
// A lane pair forms its two factors as small integer combinations of the wave's LDS entries (lin3, as in k_millerlatx.hip).  The
// running sum stays in this form in LDS from level to level and travels between waves in it (records of 3 ES dwords); the leaves
// arrive and the root leaves in the library's 32-bit Montgomery form.  Exceptional inputs are exact as in coop_jac_add: infinity
// is a copy, equal x (H = 0) takes jac_coop.hpp's doubling through a conversion (or returns infinity).
template <class C>
struct TreeX {
  typedef FX<C> E;
  static constexpr int ES = E::ES, HS = E::HS, N = C::RX_NL;
  enum { AX = 0, AY, AZ, BX, BY, BZ, Z1Z1, Z2Z2, ZZ, T1, T2, U1, U2, S1, S2, II, R2, RU, RH, SH, JJ, VV, YA, YB, NENT };
  static constexpr int REC_DW = 3 * ES;                       // a parked sum: entries AX, AY, AZ as they lie in LDS
  static constexpr int COOP_DW = CoopF2<C>::WAVE_DW > 6 * C::L ? CoopF2<C>::WAVE_DW : 6 * C::L;   // doubling fall-back / one 32-bit record
  static constexpr int LDS_DW = COOP_DW + NENT * ES;
  static_assert(REC_DW % 2 == 0 && REC_DW / 2 <= 64, "a record is at most 64 eight-byte words");
};

#ifdef TREEX_DBG
// development only: shader-clock stamps of the wave that carries the sum upwards, [level][event] (tools/exp/treex_steps.py)
__device__ unsigned long long g_treex_t[20][8];
#define TREEX_T(lv, k) do { if ((threadIdx.x & 63) == 0 && (lv) < 20) g_treex_t[lv][k] = clock64(); } while (0)
#else
#define TREEX_T(lv, k) do { } while (0)
#endif

template <class C>
__global__ void __launch_bounds__(64) k_sum_tree_x(const Jac<F2<C>>* in, size_t cnt, u32* store, uint32_t* tickets, uint8_t* d_bytes,
                                                   Jac<F2<C>>* d_jac) {
  typedef F2<C> F;
  typedef TreeX<C> T;
  typedef FX<C> E;
  extern __shared__ u32 lds[];
  constexpr int ES = T::ES, HS = T::HS, N = T::N, L = C::L;
  const int lane = threadIdx.x, q = lane >> 1;
  const bool odd = lane & 1;
  const int ebase = T::COOP_DW;
  auto LD = [&](int e) { return fx_ld<C>(ebase + e * ES + (odd ? HS : 0)); };
  auto ST = [&](int e, const Sx<C, SX_T>& v, bool active) {
    if (active) fx_st<C>(ebase + e * ES + (odd ? HS : 0), v);
  };
  auto Fn = [&](const auto& v) { return sx_normf<C>(v); };
  auto lin3 = [&](int e0, int c0, int e1, int c1, int e2, int c2) {
    const Sx<C, SX_T> v0 = LD(e0), v1 = LD(e1), v2 = LD(e2);
    Sx<C, 96> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = c0 * v0.v[i] + c1 * v1.v[i] + c2 * v2.v[i];
    return r;
  };
  // 32-bit record (Jac<F2>: X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1, L words each) at LDS dword `rec`  <->  entries e0 .. e0+2
  auto from_rec = [&](int rec, int e0) {
    if (lane < 6) {
      Fp<C> v;
#pragma unroll
      for (int k = 0; k < L; ++k) v.v[k] = lds[rec + lane * L + k];
      fx_st<C>(ebase + (e0 + (lane >> 1)) * ES + ((lane & 1) ? HS : 0), sx_from_mont<C>(v));
    }
    wave_sync();
  };
  auto to_rec = [&](int rec, int e0) {
    if (lane < 6) {
      const Fp<C> v = sx_to_mont<C>(fx_ld<C>(ebase + (e0 + (lane >> 1)) * ES + ((lane & 1) ? HS : 0)));
#pragma unroll
      for (int k = 0; k < L; ++k) lds[rec + lane * L + k] = v.v[k];
    }
    wave_sync();
  };
  auto load_leaf = [&](const Jac<F>* src, int e0) {
    const u32* g = reinterpret_cast<const u32*>(src);
    if (lane < 3 * L) reinterpret_cast<uint2*>(lds)[lane] = reinterpret_cast<const uint2*>(g)[lane];      // 6 L words
    wave_sync();
    from_rec(0, e0);
  };
  // both halves of entry e are zero mod p (entries that are products' outputs or their differences)
  auto is_zero = [&](const auto& own) {
    const bool z = sx_is_zero_mod_p<C>(own);
    return z && pair_swap1(z ? 1 : 0) != 0;
  };
  // A <- A + B
  auto add = [&]() {
    const bool ainf = is_zero(LD(T::AZ)), binf = is_zero(LD(T::BZ));     // the same on every lane
    if (binf) return;
    if (ainf) {
      const Sx<C, SX_T> v = LD(T::BX + (q % 3));
      wave_sync();
      ST(T::AX + q, v, q < 3);
      wave_sync();
      return;
    }
    {   // round 1
      const int ia = q == 0 ? T::AZ : q == 1 ? T::BZ : q == 2 ? T::AZ : q == 3 ? T::AY : T::BY;
      const int ib = q == 2 ? T::BZ : q == 3 ? T::BZ : q == 4 ? T::AZ : ia;
      const Sx<C, SX_F> a = Fn(lin3(ia, 1, ib, q == 2 ? 1 : 0, ia, 0));
      const Sx<C, SX_F> b = q < 3 ? a : sx_as<SX_F, C>(LD(ib));
      ST(T::Z1Z1 + q, pair_mul<C>(a, b, odd), q < 5);         // Z1Z1, Z2Z2, ZZ, T1, T2 are consecutive
      wave_sync();
    }
    {   // round 2
      const int ia = q == 0 ? T::AX : q == 1 ? T::BX : q == 2 ? T::T1 : T::T2;
      const int ib = (q == 0 || q == 2) ? T::Z2Z2 : T::Z1Z1;
      ST(T::U1 + q, pair_mul<C>(sx_as<SX_F, C>(LD(ia)), sx_as<SX_F, C>(LD(ib)), odd), q < 4);    // U1, U2, S1, S2 are consecutive
      wave_sync();
    }
    const bool hz = is_zero(sx_sub<C>(LD(T::U2), LD(T::U1)));
    if (hz) {                                                 // same x: P = Q (doubling on the 32-bit wave form) or P = -Q
      const bool rz = is_zero(sx_sub<C>(LD(T::S2), LD(T::S1)));
      Jac<F> r = jac_inf<F>();
      if (rz) {
        to_rec(0, T::AX);
        const Jac<F> p = *reinterpret_cast<const Jac<F>*>(lds);
        wave_sync();
        const CoopF2<C> k(0);
        r = coop_jac_dbl<C>(k, p);
      }
      wave_sync();
      if (lane == 0) *reinterpret_cast<Jac<F>*>(lds) = r;
      wave_sync();
      from_rec(0, T::AX);
      return;
    }
    {   // round 3: I = (2H)^2, R2 = r^2, Z3 = (ZZ - Z1Z1 - Z2Z2) H, rU = r U1, rH = r H, sH = S1 H
      const bool ar = q == 1 || q == 3 || q == 4;             // a = r = 2 (S2 - S1)
      const Sx<C, SX_F> a = Fn(lin3(q == 0 ? T::U2 : ar ? T::S2 : q == 2 ? T::ZZ : T::S1, (q == 0 || ar) ? 2 : 1,
                                    q == 0 ? T::U1 : ar ? T::S1 : T::Z1Z1, (q == 0 || ar) ? -2 : q == 2 ? -1 : 0, T::Z2Z2, q == 2 ? -1 : 0));
      const Sx<C, SX_F> bb = Fn(lin3(q == 3 ? T::U1 : T::U2, 1, T::U1, q == 3 ? 0 : -1, T::U1, 0));      // U1 | H
      const Sx<C, SX_F> b = sx_select<C>(q < 2, a, bb);
      const Sx<C, SX_T> r = pair_mul<C>(a, b, odd);
      // nothing below reads Z1, Z2 again: Z3 goes straight to its place
      ST(q == 0 ? T::II : q == 1 ? T::R2 : q == 2 ? T::AZ : q == 3 ? T::RU : q == 4 ? T::RH : T::SH, r, q < 6);
      wave_sync();
    }
    {   // round 4: J = H I, V = U1 I, Ya = I (3 rU + rH - 2 sH), Yb = r R2
      const Sx<C, SX_F> a = Fn(lin3(q == 0 ? T::U2 : q == 1 ? T::U1 : q == 2 ? T::II : T::S2, q == 3 ? 2 : 1, q == 3 ? T::S1 : T::U1,
                                    q == 0 ? -1 : q == 3 ? -2 : 0, T::U1, 0));
      const Sx<C, SX_F> b = Fn(lin3(q == 2 ? T::RU : q == 3 ? T::R2 : T::II, q == 2 ? 3 : 1, T::RH, q == 2 ? 1 : 0, T::SH, q == 2 ? -2 : 0));
      ST(T::JJ + q, pair_mul<C>(a, b, odd), q < 4);           // JJ, VV, YA, YB are consecutive
      wave_sync();
    }
    {   // X3 = R2 - J - 2V, Y3 = Ya - Yb
      const Sx<C, SX_T> v = sx_norm<C>(lin3(q == 0 ? T::R2 : T::YA, 1, q == 0 ? T::JJ : T::YB, -1, T::VV, q == 0 ? -2 : 0));
      ST(q == 0 ? T::AX : T::AY, v, q < 2);
      wave_sync();
    }
  };
  // a parked sum between waves: 8-byte relaxed agent-scope words, written through and drained before the ticket (see tree_park)
  auto park = [&](u32* slot) {
    if (lane < T::REC_DW / 2) {
      const unsigned long long w = *reinterpret_cast<const unsigned long long*>(lds + ebase + T::AX * ES + 2 * lane);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot) + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
  };
  auto fetch = [&](const u32* slot) {
    if (lane < T::REC_DW / 2) {
      const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(slot) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *reinterpret_cast<unsigned long long*>(lds + ebase + T::BX * ES + 2 * lane) = w;
    }
    wave_sync();
  };

  size_t i = blockIdx.x;                        // node index at the current level
  int lv = 0;
  (void)lv;
  TREEX_T(0, 0);
  load_leaf(in + 2 * i, T::AX);
  if (2 * i + 1 < cnt) {
    load_leaf(in + 2 * i + 1, T::BX);
    TREEX_T(0, 1);
    add();
  }
  TREEX_T(0, 2);
  size_t n = (cnt + 1) / 2, off = 0;            // nodes at this level, offset of the level's slots
  while (n > 1) {
    const size_t sib = i ^ 1;
    ++lv;
    if (sib < n) {
      TREEX_T(lv, 0);
      park(store + (off + i) * T::REC_DW);
      TREEX_T(lv, 1);
      unsigned t = 0;
      if (lane == 0) t = atomicAdd(&tickets[off + (i & ~(size_t)1)], 1u);
      t = __shfl(t, 0);
      if (t == 0) return;                       // first of the pair: the sibling's wave carries both sums on
      TREEX_T(lv, 2);
      fetch(store + (off + sib) * T::REC_DW);
      TREEX_T(lv, 3);
      add();
      TREEX_T(lv, 4);
      if (lane == 0) tickets[off + (i & ~(size_t)1)] = 0;        // left clean for the next launch
    }
    off += n;
    i >>= 1;
    n = (n + 1) / 2;
  }
  // the root: the Jacobian record in the library's form if asked for, and the affine wire bytes x = X / Z^2, y = Y / Z^3 -- on
  // the lane pairs as well (one lane walking jac_to_aff in the 32-bit form took 60 us: an Fp2 inversion and five Fp2 products
  // of dependent carry chains): |Z|^2 and the products are rounds of the pair arithmetic, ONE value goes through fp_inv
  TREEX_T(19, 0);
  const bool zinf = is_zero(LD(T::AZ));
  if (d_jac || zinf) {
    to_rec(0, T::AX);
    if (lane == 0) {
      const Jac<F> acc = *reinterpret_cast<const Jac<F>*>(lds);
      if (d_jac) *d_jac = acc;
      if (d_bytes && zinf) aff_to_bytes<F>(d_bytes, jac_to_aff<F>(acc));
    }
  }
  TREEX_T(19, 1);
  if (d_bytes && !zinf) {
    const Sx<C, SX_T> own = LD(T::AZ);
    const Sx<C, SX_T> sq = pair_muls<C>(own, own);                       // z0^2 | z1^2
    Sx<C, 2 * SX_T> nn;
#pragma unroll
    for (int i = 0; i < N; ++i) nn.v[i] = sq.v[i] + pair_swap1(sq.v[i]);
    const Sx<C, SX_T> ni = sx_from_mont<C>(fp_inv<C>(sx_to_mont<C>(nn)));   // 1 / |Z|^2, the same on every lane
    ST(T::T1, pair_muls<C>(sx_select<C>(odd, sx_neg<C>(own), own), ni), q == 0);      // 1 / Z = conj(Z) / |Z|^2
    wave_sync();
    {
      const Sx<C, SX_F> zi = sx_as<SX_F, C>(LD(T::T1));
      ST(T::T2, pair_mul<C>(zi, zi, odd), q == 0);                        // Z^-2
      wave_sync();
    }
    ST(q == 0 ? T::U1 : T::U2, pair_mul<C>(sx_as<SX_F, C>(LD(q == 0 ? T::AX : T::T2)), sx_as<SX_F, C>(LD(q == 0 ? T::T2 : T::T1)), odd), q < 2);   // x | Z^-3
    wave_sync();
    ST(T::S1, pair_mul<C>(sx_as<SX_F, C>(LD(T::AY)), sx_as<SX_F, C>(LD(T::U2)), odd), q == 0);     // y
    wave_sync();
    if (lane < 4) {                                                       // x.c1 | x.c0 | y.c1 | y.c0, big-endian (g2_to_bytes)
      const Sx<C, SX_T> v = fx_ld<C>(ebase + (lane < 2 ? T::U1 : T::S1) * ES + ((lane & 1) ? 0 : HS));
      fp_to_be<C>(d_bytes + lane * C::FP_BYTES, fp_from_mont<C>(sx_to_mont<C>(v)));
    }
  }
  TREEX_T(19, 2);
}

namespace bgls {
namespace kl {

template <class C>
size_t sum_tree_store_bytes(size_t cnt) { return (cnt + 64) * (size_t)TreeX<C>::REC_DW * 4; }
template size_t sum_tree_store_bytes<BN254>(size_t);
template size_t sum_tree_store_bytes<BLS381>(size_t);

template <class C>
void sum_tree(hipStream_t st, const void* in, size_t cnt, void* store, uint32_t* tickets, uint8_t* d_bytes, void* d_jac) {
  k_sum_tree_x<C><<<(unsigned)((cnt + 1) / 2), 64, TreeX<C>::LDS_DW * 4, st>>>((const Jac<F2<C>>*)in, cnt, (u32*)store, tickets, d_bytes, (Jac<F2<C>>*)d_jac);
}
template void sum_tree<BN254>(hipStream_t, const void*, size_t, void*, uint32_t*, uint8_t*, void*);
template void sum_tree<BLS381>(hipStream_t, const void*, size_t, void*, uint32_t*, uint8_t*, void*);

}  // namespace kl
}  // namespace bgls
#ifdef TREEX_DBG
extern "C" int bgls_dbg_treex_dump(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_treex_t), sizeof(g_treex_t)); }
#endif
