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
#include "jac_coop.hpp"
#include "points_inl.hpp"
#include "launch_tail.hpp"

using namespace bgls;

// Hand-over of a parked sum between two waves on different CUs / XCDs.  The per-XCD L2s are not coherent with each other and
// an agent-scope release fence writes a whole L2's dirty lines back (MI355X_MICROARCH.md, inter-workgroup visibility: ~3.5 us
// per __threadfence(), several times that next to the main pass's freshly written partials -- the first version of this kernel
// fenced twice per level and was SLOWER than twelve launches).  So the record travels as 8-byte relaxed agent-scope atomics:
// write-through (sc1) stores by 6 L / 2 lanes, drained (s_waitcnt vmcnt(0)) before the ticket is taken, sc1 loads behind
// the returned ticket on the other side.  No fence anywhere.
template <class C>
__device__ __forceinline__ void tree_park(Jac<F2<C>>* slot, const Jac<F2<C>>& v, int lds_base) {
  extern __shared__ u32 lds[];
  constexpr int NW = 6 * C::L / 2;              // 8-byte words of a record
  static_assert(sizeof(Jac<F2<C>>) == NW * 8 && 2 * NW <= CoopF2<C>::WAVE_DW, "record fits the wave's LDS scratch");
  const int lane = threadIdx.x & 63;
  if (lane == 0) *reinterpret_cast<Jac<F2<C>>*>(lds + lds_base) = v;     // every lane holds the same record
  wave_sync();
  if (lane < NW) {
    const unsigned long long w = *reinterpret_cast<const unsigned long long*>(lds + lds_base + 2 * lane);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot) + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();
}
template <class C>
__device__ __forceinline__ Jac<F2<C>> tree_fetch(const Jac<F2<C>>* slot, int lds_base) {
  extern __shared__ u32 lds[];
  constexpr int NW = 6 * C::L / 2;
  const int lane = threadIdx.x & 63;
  if (lane < NW) {
    const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(slot) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<unsigned long long*>(lds + lds_base + 2 * lane) = w;
  }
  wave_sync();
  const Jac<F2<C>> r = *reinterpret_cast<const Jac<F2<C>>*>(lds + lds_base);
  wave_sync();
  return r;
}

template <class C>
__global__ void __launch_bounds__(64) k_sum_tree(const Jac<F2<C>>* in, size_t cnt, Jac<F2<C>>* store, uint32_t* tickets, uint8_t* d_bytes,
                                                 Jac<F2<C>>* d_jac) {
  typedef F2<C> F;
  const int lane = threadIdx.x;
  const CoopF2<C> k(0);
  size_t i = blockIdx.x;                        // node index at the current level
  Jac<F> acc = in[2 * i];
  if (2 * i + 1 < cnt) acc = coop_jac_add<C>(k, acc, in[2 * i + 1]);
  size_t n = (cnt + 1) / 2, off = 0;            // nodes at this level, offset of the level's slots
  while (n > 1) {
    const size_t sib = i ^ 1;
    if (sib < n) {
      tree_park<C>(store + off + i, acc, 0);    // visible (written through, drained) before the ticket is taken
      unsigned t = 0;
      if (lane == 0) t = atomicAdd(&tickets[off + (i & ~(size_t)1)], 1u);
      t = __shfl(t, 0);
      if (t == 0) return;                       // first of the pair: the sibling's wave carries both sums on
      const Jac<F> other = tree_fetch<C>(store + off + sib, 0);
      acc = coop_jac_add<C>(k, acc, other);
      if (lane == 0) tickets[off + (i & ~(size_t)1)] = 0;        // left clean for the next launch
    }
    off += n;
    i >>= 1;
    n = (n + 1) / 2;
  }
  if (lane == 0) {
    if (d_jac) *d_jac = acc;
    if (d_bytes) aff_to_bytes<F>(d_bytes, jac_to_aff<F>(acc));
  }
}

namespace bgls {
namespace kl {

template <class C>
void sum_tree(hipStream_t st, const void* in, size_t cnt, void* store, uint32_t* tickets, uint8_t* d_bytes, void* d_jac) {
  k_sum_tree<C><<<(unsigned)((cnt + 1) / 2), 64, CoopF2<C>::WAVE_DW * 4, st>>>((const Jac<F2<C>>*)in, cnt, (Jac<F2<C>>*)store, tickets, d_bytes,
                                                                                (Jac<F2<C>>*)d_jac);
}
template void sum_tree<BN254>(hipStream_t, const void*, size_t, void*, uint32_t*, uint8_t*, void*);
template void sum_tree<BLS381>(hipStream_t, const void*, size_t, void*, uint32_t*, uint8_t*, void*);

}  // namespace kl
}  // namespace bgls
