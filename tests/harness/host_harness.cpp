// TEST-ONLY: compiles the device arithmetic headers for the host so that every routine can be
// diffed against the oracle in the CPU test tier (no GPU here).  Never linked into the product.
#include <type_traits>
#include <string.h>
#include <atomic>
#include <thread>
#define BGLS_RX_CHECK 1
#include "../../bgls_amd/csrc/pairing.hpp"
#include "../../bgls_amd/csrc/h2c.hpp"
#include "../../bgls_amd/csrc/wire.hpp"
#include "../../bgls_amd/csrc/r28.hpp"
#include "../../bgls_amd/csrc/rx_pair.hpp"
#include "../../bgls_amd/csrc/rx_pow.hpp"
#include "../../bgls_amd/csrc/h2c_x.hpp"

// The file compiles as ONE translation unit (no HT_PART: the sanitizer build includes it whole) or as four parts compiled in parallel and linked
// together (-DHT_PART=0..3; tests/conftest.py, __graft_entry__.py): the single unit takes five minutes of an -O1 compile, the parts under two.
#ifdef HT_PART
#define HT_HAS(p) (HT_PART == (p))
#else
#define HT_HAS(p) 1
#endif

#if HT_HAS(0)
namespace bgls { int g_rx_overflow = 0; }
#endif
#if HT_HAS(0)

// ---- host emulation of a lane pair (rx_pair.hpp): two threads in lock-step, values exchanged through a rendezvous
static std::atomic<int> g_pair_slot[2];
std::atomic<int> g_pair_cnt{0};
static std::atomic<int> g_pair_gen{0};
thread_local int tl_pair_lane = 0;
static void pair_barrier() {
  const int gen = g_pair_gen.load();
  if (g_pair_cnt.fetch_add(1) == 1) {
    g_pair_cnt.store(0);
    g_pair_gen.fetch_add(1);
  } else {
    while (g_pair_gen.load(std::memory_order_acquire) == gen) { }
  }
}
int rx_host_pair_swap(int v) {
  g_pair_slot[tl_pair_lane].store(v);
  pair_barrier();
  const int r = g_pair_slot[1 - tl_pair_lane].load();
  pair_barrier();
  return r;
}

#else
extern thread_local int tl_pair_lane;
extern std::atomic<int> g_pair_cnt;
int rx_host_pair_swap(int v);
#endif
using namespace bgls;

#if HT_HAS(0)

template <class C>
static int fp_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fp<C> x = fp_to_mont<C>(fp_from_be<C>(a));
  Fp<C> y = fp_to_mont<C>(fp_from_be<C>(b));
  Fp<C> r;
  switch (op) {
    case 0: r = fp_mul<C>(x, y); break;
    case 1: r = fp_sqr<C>(x); break;
    case 2: r = fp_add<C>(x, y); break;
    case 3: r = fp_sub<C>(x, y); break;
    case 4: r = fp_neg<C>(x); break;
    case 5: r = fp_inv<C>(x); break;
    case 6: r = fp_sqrt_candidate<C>(x); break;
    case 7: return fp_jacobi<C>(x) + 10;
    case 8: return fp_jacobi<C>(fp_from_mont<C>(x)) + 10;
    case 9: r = fp_inv_euclid<C>(x); break;          // the binary Euclid fp_inv was until round 4 (cross-check of the division-step form)
    default: return -1;
  }
  fp_to_be<C>(out, fp_from_mont<C>(r));
  return 0;
}

template <class C>
static int f2_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {  // re||im
  constexpr int N = C::FP_BYTES;
  Fp2<C> x = {fp_to_mont<C>(fp_from_be<C>(a)), fp_to_mont<C>(fp_from_be<C>(a + N))};
  Fp2<C> y = {fp_to_mont<C>(fp_from_be<C>(b)), fp_to_mont<C>(fp_from_be<C>(b + N))};
  Fp2<C> r;
  switch (op) {
    case 0: r = f2_mul<C>(x, y); break;
    case 1: r = f2_sqr<C>(x); break;
    case 2: r = f2_mulxi<C>(x); break;
    case 3: r = f2_inv<C>(x); break;
    default: return -1;
  }
  fp_to_be<C>(out, fp_from_mont<C>(r.c0));
  fp_to_be<C>(out + N, fp_from_mont<C>(r.c1));
  return 0;
}

template <class C>
static int f12_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  Fp12<C> x, y, r;
  if (!gt_from_bytes<C>(x, a)) return -2;
  if (b && !gt_from_bytes<C>(y, b)) return -2;
  switch (op) {
    case 0: r = f12_mul<C>(x, y); break;
    case 1: r = f12_sqr<C>(x); break;
    case 2: r = f12_inv<C>(x); break;
    case 3: r = f12_frob<C>(x, 1); break;
    case 4: r = f12_frob<C>(x, 2); break;
    case 5: r = f12_frob<C>(x, 3); break;
    case 6: r = f12_cyclo_sqr<C>(x); break;
    case 7: r = f12_conj<C>(x); break;
    case 8: r = final_exp<C>(x); break;
    default: return -1;
  }
  gt_to_bytes<C>(out, r);
  return 0;
}

template <class C>
static int miller(const uint8_t* g1, const uint8_t* g2, uint8_t* out) {
  Aff<F1<C>> P;
  Aff<F2<C>> Q;
  if (!g1_from_bytes<C>(P, g1) || !g2_from_bytes<C>(Q, g2)) return -2;
  gt_to_bytes<C>(out, miller_loop<C>(P, Q));
  return 0;
}

template <class C>
static int group_op(int op, const uint8_t* a, const uint8_t* b, const uint8_t* k_be32, uint8_t* out) {
  u32 k[8];
  if (k_be32)
    for (int j = 0; j < 8; ++j) {
      const uint8_t* q = k_be32 + 4 * (7 - j);
      k[j] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | q[3];
    }
  if (op < 10) {
    typedef F1<C> F;
    Aff<F> P, Q;
    if (!g1_from_bytes<C>(P, a)) return -2;
    Jac<F> r;
    if (op == 0) {
      if (!g1_from_bytes<C>(Q, b)) return -2;
      r = jac_add_aff<F>(jac_from_aff<F>(P), Q);
    } else if (op == 1) {
      r = jac_mul<F>(P, k, 256);
    } else if (op == 2) {
      if (!g1_from_bytes<C>(Q, b)) return -2;
      r = jac_add<F>(jac_dbl<F>(jac_from_aff<F>(P)), jac_dbl<F>(jac_from_aff<F>(Q)));  // 2P + 2Q
    } else if (op == 3) {
      return aff_on_curve<F>(P) ? 1 : 0;
    } else if (op == 4) {      // the scale kernels' form: signed radix-16 windows over the scalar's actual bit length
      int top = -1;
      for (int j = 7; j >= 0 && top < 0; --j)
        if (k[j]) top = j * 32 + 31 - __builtin_clz(k[j]);
      r = jac_mul_w4<F>(P, k, top + 1);
    } else
      return -1;
    g1_to_bytes<C>(out, jac_to_aff<F>(r));
  } else {
    typedef F2<C> F;
    Aff<F> P, Q;
    if (!g2_from_bytes<C>(P, a)) return -2;
    Jac<F> r;
    if (op == 10) {
      if (!g2_from_bytes<C>(Q, b)) return -2;
      r = jac_add_aff<F>(jac_from_aff<F>(P), Q);
    } else if (op == 11) {
      r = jac_mul<F>(P, k, 256);
    } else if (op == 12) {
      if (!g2_from_bytes<C>(Q, b)) return -2;
      r = jac_add<F>(jac_dbl<F>(jac_from_aff<F>(P)), jac_dbl<F>(jac_from_aff<F>(Q)));
    } else if (op == 13) {
      return aff_on_curve<F>(P) ? 1 : 0;
    } else if (op == 14) {
      int top = -1;
      for (int j = 7; j >= 0 && top < 0; --j)
        if (k[j]) top = j * 32 + 31 - __builtin_clz(k[j]);
      r = jac_mul_w4<F>(P, k, top + 1);
    } else
      return -1;
    g2_to_bytes<C>(out, jac_to_aff<F>(r));
  }
  return 0;
}

extern "C" {
int ht_fp_op(int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  return curve == 0 ? fp_op<BN254>(op, a, b, out) : fp_op<BLS381>(op, a, b, out);
}
int ht_f2_op(int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  return curve == 0 ? f2_op<BN254>(op, a, b, out) : f2_op<BLS381>(op, a, b, out);
}
int ht_f12_op(int curve, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  return curve == 0 ? f12_op<BN254>(op, a, b, out) : f12_op<BLS381>(op, a, b, out);
}
int ht_miller(int curve, const uint8_t* g1, const uint8_t* g2, uint8_t* out) {
  return curve == 0 ? miller<BN254>(g1, g2, out) : miller<BLS381>(g1, g2, out);
}
int ht_group_op(int curve, int op, const uint8_t* a, const uint8_t* b, const uint8_t* k, uint8_t* out) {
  return curve == 0 ? group_op<BN254>(op, a, b, k, out) : group_op<BLS381>(op, a, b, k, out);
}
int ht_hash_to_g1(int curve, const uint8_t* msg, size_t len, uint8_t* out) {
  if (curve == 0) {
    Aff<F1<BN254>> p;
    if (!bn_hash_to_g1(msg, len, p)) return -3;
    g1_to_bytes<BN254>(out, p);
  } else {
    g1_to_bytes<BLS381>(out, bls_hash_to_g1(msg, len));
  }
  return 0;
}
int ht_keccak256(const uint8_t* msg, size_t len, uint8_t prefix, uint8_t* out) {
  ByteSrc s; s.msg = msg; s.len = len; s.pre[0] = prefix; s.npre = 1; s.nsuf = 0;
  u32 d[8]; keccak256_legacy(s, d);
  for (int i = 0; i < 8; ++i) { out[4*i] = d[i] >> 24; out[4*i+1] = d[i] >> 16; out[4*i+2] = d[i] >> 8; out[4*i+3] = d[i]; }
  return 0;
}
int ht_blake2b(const uint8_t* msg, size_t len, int k, uint8_t* out) {
  ByteSrc s; s.msg = msg; s.len = len; s.npre = 0; s.pre[0] = 0; s.suf[0]='G'; s.suf[1]='1'; s.suf[2]='_'; s.suf[3]='0'+k; s.nsuf = 4;
  u32 d[16]; blake2b512(s, d);
  for (int i = 0; i < 16; ++i) { out[4*i] = d[i] >> 24; out[4*i+1] = d[i] >> 16; out[4*i+2] = d[i] >> 8; out[4*i+3] = d[i]; }
  return 0;
}
// compressed alt-bn128 forms (wire.hpp): op 0 = compress G1 (64 -> 32), 1 = compress G2 (128 -> 64),
// 2 = decompress G1 (32 -> 64), 3 = decompress G2 (64 -> 128).  Returns 1 ok / 0 "nil,false" / -1 bad input.
int ht_wire(int op, const uint8_t* in, uint8_t* out) {
  typedef BN254 C;
  if (op == 0) {
    Aff<F1<C>> p;
    if (!g1_from_bytes<C>(p, in) || !aff_on_curve<F1<C>>(p)) return -1;
    g1_compress<C>(out, p);
    return 1;
  }
  if (op == 1) {
    Aff<F2<C>> p;
    if (!g2_from_bytes<C>(p, in) || !aff_on_curve<F2<C>>(p)) return -1;
    g2_compress<C>(out, p);
    return 1;
  }
  if (op == 2) {
    Aff<F1<C>> p;
    if (!g1_decompress<C>(p, in)) return 0;
    g1_to_bytes<C>(out, p);
    return 1;
  }
  if (op == 3) {
    Aff<F2<C>> p;
    if (!g2_decompress<C>(p, in)) return 0;
    g2_to_bytes<C>(out, p);
    return 1;
  }
  // BLS12-381, ebfull/pairing layout (wire.hpp second half): 4 / 5 compress G1 / G2, 6 / 7 decode 48 / 96 bytes (before Check())
  typedef BLS381 B;
  if (op == 4) {
    Aff<F1<B>> p;
    if (!g1_from_bytes<B>(p, in) || !aff_on_curve<F1<B>>(p)) return -1;
    g1_compress_zc<B>(out, p);
    return 1;
  }
  if (op == 5) {
    Aff<F2<B>> p;
    if (!g2_from_bytes<B>(p, in) || !aff_on_curve<F2<B>>(p)) return -1;
    g2_compress_zc<B>(out, p);
    return 1;
  }
  if (op == 6) {
    Aff<F1<B>> p;
    if (!g1_decompress_zc<B>(p, in)) return 0;
    g1_to_bytes<B>(out, p);
    return 1;
  }
  if (op == 7) {
    Aff<F2<B>> p;
    if (!g2_decompress_zc<B>(p, in)) return 0;
    g2_to_bytes<B>(out, p);
    return 1;
  }
  return -1;
}
// one BLAKE2Xb expansion node (hashes.hpp blake2xb_node) from a 64-byte root
int ht_blake2xb_node(const uint8_t* root, uint32_t i, uint32_t xof_len, uint32_t take, uint8_t* out) {
  u64 r[8], o[8];
  for (int k = 0; k < 8; ++k) { r[k] = 0; for (int b = 7; b >= 0; --b) r[k] = (r[k] << 8) | root[8 * k + b]; }
  blake2xb_node(r, i, xof_len, take, o);
  for (uint32_t b = 0; b < take; ++b) out[b] = (uint8_t)(o[b >> 3] >> (8 * (b & 7)));
  return 0;
}
// r28.hpp (28-bit-limb consumer arithmetic, alt-bn128): inputs / outputs are Fp2 as 64-byte big-endian (re || im).
// op 0: from_r28(to_r28(a))   op 1: from_r28(r28_f2_mul(to_r28(a), to_r28(b)))   op 2: xi * a (through a product by one)
// op 3: a five-term dot product sum_t a_t * b_t with a_t = a^(t+1)-ish derived values, both ways; returns 1 when equal
int ht_r28(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  typedef BN254 C;
  auto ld = [](const uint8_t* p) { return Fp2<C>{fp_to_mont<C>(fp_from_be<C>(p)), fp_to_mont<C>(fp_from_be<C>(p + 32))}; };
  auto st = [](uint8_t* p, const Fp2<C>& v) { fp_to_be<C>(p, fp_from_mont<C>(v.c0)); fp_to_be<C>(p + 32, fp_from_mont<C>(v.c1)); };
  Fp2<C> x = ld(a), y = ld(b);
  if (op == 0) { st(out, from_r28<C>(to_r28<C>(x))); return 0; }
  if (op == 1) { st(out, from_r28<C>(r28_f2_mul<C>(to_r28<C>(x), to_r28<C>(y)))); return 0; }
  if (op == 2) {   // the xi multiple is a lazy value (up to ~73 p): bring it down with a product by one before converting back
    F28x2 one = {r28_load<C>(C::R28_ONE), F28{}};
    for (int i = 0; i < 10; ++i) one.c1.v[i] = 0;
    st(out, from_r28<C>(r28_f2_mul<C>(r28_mulxi<C>(to_r28<C>(x)), one)));
    return 0;
  }
  if (op == 3) {
    Fp2<C> as[5], bs[5];
    as[0] = x; bs[0] = y;
    for (int t = 1; t < 5; ++t) { as[t] = f2_add<C>(f2_sqr<C>(as[t - 1]), y); bs[t] = f2_mulxi<C>(f2_add<C>(bs[t - 1], x)); }
    Fp2<C> want = f2_zero<C>();
    for (int t = 0; t < 5; ++t) want = f2_add<C>(want, f2_mul<C>(as[t], bs[t]));
    u64 cr[20], ci[20];
    for (int k = 0; k < 20; ++k) cr[k] = ci[k] = 0;
    for (int t = 0; t < 5; ++t) {
      F28x2 ea = to_r28<C>(as[t]), eb = to_r28<C>(bs[t]);
      r28_acc(cr, ea.c0, eb.c0); r28_acc(cr, ea.c1, r28_fatneg<C>(eb.c1));
      r28_acc(ci, ea.c0, eb.c1); r28_acc(ci, ea.c1, eb.c0);
    }
    F28x2 got = {r28_redc<C>(cr), r28_redc<C>(ci)};
    st(out, from_r28<C>(got));
    uint8_t w[64]; st(w, want);
    return memcmp(w, out, 64) == 0 ? 1 : 0;
  }
  if (op == 4) {   // three-term dot product in the Karatsuba form, with a lazy xi multiple as one of the right operands
    Fp2<C> as[3], bs[3];
    as[0] = x; bs[0] = y;
    for (int t = 1; t < 3; ++t) { as[t] = f2_add<C>(f2_sqr<C>(as[t - 1]), y); bs[t] = f2_add<C>(f2_mul<C>(bs[t - 1], x), y); }
    Fp2<C> want = f2_zero<C>();
    for (int t = 0; t < 3; ++t) want = f2_add<C>(want, f2_mul<C>(as[t], t == 1 ? f2_mulxi<C>(bs[t]) : bs[t]));
    u64 v0[20], v1[20], ss[20];
    for (int k = 0; k < 20; ++k) v0[k] = v1[k] = ss[k] = 0;
    for (int t = 0; t < 3; ++t) {
      F28x2 eb = to_r28<C>(bs[t]);
      if (t == 1) eb = r28_mulxi<C>(eb);                  // as published by the consumer: up to ~73 p, tight limbs
      r28_kara_term(v0, v1, ss, to_r28<C>(as[t]), eb);
    }
    st(out, from_r28<C>(r28_kara_finish<C>(v0, v1, ss)));
    uint8_t w[64]; st(w, want);
    return memcmp(w, out, 64) == 0 ? 1 : 0;
  }
  return -1;
}
}
#endif  // part 0

#if HT_HAS(1)

// ---- rx.hpp, consumer side, on RAW limbs (so that the unit tests can feed worst-case limb patterns): A and B hold three
// Fp2 operands each as [t][half][NL] u32; out = [half][NL].  Returns the overflow flag of the checked column arithmetic.
//   op 0: ux_dot_k2p<3>   op 1: ux_sqr_dot with kinds k0..k3 packed in `arg` (2 bits each; operands t = 0..3 from A / B
//   with A, B holding FOUR operands)   op 2: ux_mulxi(A[0])   op 3: to_ux / from_ux round trip is ht_rx_conv below
template <class C>
static int rx_raw(int op, int arg, const u32* A, const u32* Bv, u32* out) {
  constexpr int N = C::RX_NL;
  g_rx_overflow = 0;
  auto ld = [&](const u32* base, int t, int h) { Ux<C> r; for (int i = 0; i < N; ++i) r.v[i] = base[(t * 2 + h) * N + i]; return r; };
  Ux2<C> r;
  if (op == 0) r = ux_dot_k2p<C, 3>([&](int t, int h) { return ld(A, t, h); }, [&](int t, int h) { return ld(Bv, t, h); });
  else if (op == 1) {
    if constexpr (rx_lazy<C>) r = ux_sqr_dot<C>([&](int t) { return (arg >> (2 * t)) & 3; }, [&](int t, int h) { return ld(A, t, h); }, [&](int t, int h) { return ld(Bv, t, h); });
    else return -1;
  }
  else if (op == 2) { Ux2<C> a = {ld(A, 0, 0), ld(A, 0, 1)}; r = ux_mulxi<C>(a); }
  else if (op == 3) {   // the 29-bit form's squaring: pile A = operands 0, 1, pile B = operands 2, 3 (A, B hold FOUR operands); arg bit 0: odd row
    // (A undoubled and doubled after its reduction, B = 2 x operand 2, operand 3 unused), else even row (operands 0 and 2 doubled inside)
    const bool twice = arg & 1;
    auto lm = [&](const u32* base, int t, int h, bool used) { Ux<C> v = ld(base, t, h); if (!used) for (int i = 0; i < N; ++i) v.v[i] = 0; return v; };
    if constexpr (!rx_lazy<C>)
      r = ux_sqr_dot3<C>([&](int t, int side, int h) { return side ? ld(Bv, t, h) : lm(A, t, h, true); },
                         [&](int t, int side, int h) { return side ? ld(Bv, 2 + t, h) : lm(A, 2 + t, h, !(twice && t == 1)); },
                         [&](int t) { return t == 0 && !twice; }, [&](int t) { return t == 0; }, twice);
    else return -1;
  }
  else if (op == 4) {   // quasi-reduction, levels arg >> 4 .. arg & 15, of A[0] (raw limbs, not necessarily tight)
    Ux2<C> a = {ld(A, 0, 0), ld(A, 0, 1)};
    if constexpr (!rx_lazy<C>) {
      if (arg == 0x42) r = ux_quasi<C, 4, 2>(a);
      else if (arg == 0x21) r = ux_quasi<C, 2, 1>(a);
      else return -1;
    } else return -1;
  }
  else return -1;
  for (int i = 0; i < N; ++i) { out[i] = r.c0.v[i]; out[N + i] = r.c1.v[i]; }
  return g_rx_overflow;
}
extern "C" int ht_rx_raw(int curve, int op, int arg, const u32* A, const u32* Bv, u32* out) {
  return curve == 0 ? rx_raw<BN254>(op, arg, A, Bv, out) : (curve == 2 ? rx_raw<BN254W>(op, arg, A, Bv, out) : rx_raw<BLS381>(op, arg, A, Bv, out));
}
// a (FP_BYTES big-endian, canonical) -> limbs of to_ux(a R);  and back: from_ux(limbs) -> canonical bytes
template <class C>
static int rx_conv(int dir, uint8_t* bytes, u32* limbs) {
  constexpr int N = C::RX_NL;
  g_rx_overflow = 0;
  if (dir == 0) {
    const Ux<C> u = to_ux<C>(fp_to_mont<C>(fp_from_be<C>(bytes)));
    for (int i = 0; i < N; ++i) limbs[i] = u.v[i];
  } else {
    Ux<C> u;
    for (int i = 0; i < N; ++i) u.v[i] = limbs[i];
    fp_to_be<C>(bytes, fp_from_mont<C>(from_ux<C>(u)));
  }
  return g_rx_overflow;
}
extern "C" int ht_rx_conv(int curve, int dir, uint8_t* bytes, u32* limbs) {
  return curve == 0 ? rx_conv<BN254>(dir, bytes, limbs) : (curve == 2 ? rx_conv<BN254W>(dir, bytes, limbs) : rx_conv<BLS381>(dir, bytes, limbs));
}

// ---- rx_pow.hpp: sqrt exponent powers on the carry-free limbs.  op 0: sx_sqr of raw limbs (in/out: NL limbs, signed
// top limb) -- the unit test feeds worst-case limbs; op 1: a^((p+1)/4) of canonical bytes, sliding windows W = 3 and 4 against
// fp_sqrt_candidate (returns 0 when both agree, writes the result); op 2: a^((p-3)/4) against fp_pow_w4.
template <class C>
static int rx_pow(int op, uint8_t* bytes, i32* limbs) {
  constexpr int N = C::RX_NL;
  g_rx_overflow = 0;
  if (op == 0) {
    Sx<C, SX_T> a;
    for (int i = 0; i < N; ++i) a.v[i] = limbs[i];
    const Sx<C, SX_T> r = sx_sqr<C>(a);
    for (int i = 0; i < N; ++i) limbs[i] = r.v[i];
    return g_rx_overflow ? -3 : 0;
  }
  const Fp<C> a = fp_to_mont<C>(fp_from_be<C>(bytes));
  i32 tab[8 * N];
  auto ld = [&](int e, int i) { return tab[e * N + i]; };
  auto st = [&](int e, int i, i32 v) { tab[e * N + i] = v; };
  u32 e[C::L];
  for (int k = 0; k < C::L; ++k) e[k] = C::EXP_SQRT[k];
  if (op == 2) e[0] -= 1u;
  auto word = [&](int k) { return e[k]; };
  const Sx<C, SX_T> x = ux_to_sx<C>(to_ux<C>(a));
  const Sx<C, SX_T> r3 = sx_pow_sw<C, 3, 32 * C::L>(x, word, ld, st);
  const Sx<C, SX_T> r4 = sx_pow_sw<C, 4, 32 * C::L>(x, word, ld, st);
  auto back = [&](const Sx<C, SX_T>& v) {
    Ux<C> u;
    for (int i = 0; i < N; ++i) u.v[i] = (u32)v.v[i];
    return from_ux<C>(u);
  };
  const Sx<C, SX_T> r5 = op == 2 ? sx_pow_sqrt<C, 3, true>(x, ld, st) : sx_pow_sqrt<C, 3, false>(x, ld, st);   // the compile-time schedule
  for (int i = 0; i < N; ++i) if (r5.v[i] != r3.v[i]) return -5;
  const Sx<C, SX_T> r6 = op == 2 ? sx_pow_sqrt<C, 3, true, true>(x, ld, st) : sx_pow_sqrt<C, 3, false, true>(x, ld, st);   // entry 0 in registers, three slots
  for (int i = 0; i < N; ++i) if (r6.v[i] != r3.v[i]) return -6;
  const Fp<C> want = op == 1 ? fp_sqrt_candidate<C>(a) : fp_pow_w4<C, C::L>(a, e);
  fp_to_be<C>(bytes, fp_from_mont<C>(back(r4)));
  if (g_rx_overflow) return -3;
  for (int i = 0; i < N; ++i) if (r3.v[i] < 0 || r4.v[i] < 0 || (i + 1 < N && ((u32)r3.v[i] > C::RX_MASK || (u32)r4.v[i] > C::RX_MASK))) return -2;
  return (fp_eq<C>(back(r3), want) ? 0 : 1) + (fp_eq<C>(back(r4), want) ? 0 : 2);
}
// h2c_x.hpp: one Shallue-van de Woestijne work item on the carry-free limbs (what k_bls_sw_jacobi runs per lane) against h2c.hpp's 32-bit form of
// curves/hash.go:97-167.  out: the affine point (96 bytes) of the carry-free path.  Returns the kind (0..3) when both agree, -1 on a different
// kind, -2 on a different point, -3 on a column overflow.
static int bls_sw_x_digest(const u32 (&d)[16], uint8_t* out) {
  typedef BLS381 C;
  g_rx_overflow = 0;
  static thread_local i32 tab[4 * C::RX_NL];
  auto ld = [&](int e, int i) { return tab[e * C::RX_NL + i]; };
  auto st = [&](int e, int i, i32 v) { tab[e * C::RX_NL + i] = v; };
  Jac<F1<C>> pt;
  const u32 kind = bls_sw_jac_x<3, true>(d, pt, ld, st);
  if (g_rx_overflow) return -3;
  // the 32-bit form from the same digest, as bls_h2c_t: (lo + hi 2^384) mod q
  Fp<C> lo, hi = fp_zero<C>();
  for (int j = 0; j < 12; ++j) lo.v[j] = d[15 - j];
  for (int j = 0; j < 4; ++j) hi.v[j] = d[3 - j];
  const Fp<C> tm = fp_add<C>(fp_mul<C>(lo, fp_load<C>(C::R2)), fp_mul<C>(hi, fp_load<C>(C::R3)));
  const Fp<C> t = fp_from_mont<C>(tm);
  u32 want = H2C_SW;
  if (fp_is_zero<C>(t)) want = H2C_INF;
  else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT1))) want = H2C_PLUS_G1;
  else if (fp_eq<C>(t, fp_load<C>(C::FT_ROOT2))) want = H2C_MINUS_G1;
  if (kind != want) return -1;
  if (kind != H2C_SW) return (int)kind;
  const Aff<F1<C>> ref = bls_sw_encode(tm, fp_plain_parity<C>(t));
  const Aff<F1<C>> got = jac_to_aff<F1<C>>(pt);
  g1_to_bytes<C>(out, got);
  if (got.inf != ref.inf || !fp_eq<C>(got.x, ref.x) || !fp_eq<C>(got.y, ref.y)) return -2;
  return (int)kind;
}
extern "C" int ht_bls_sw_x(const uint8_t* msg, size_t len, int k, uint8_t* out) {
  u32 d[16];
  bls_h2c_digest(msg, len, k, d);
  return bls_sw_x_digest(d, out);
}
// the same from a given 64-byte digest (big-endian, as BLAKE2b writes it): the degenerate t = 0, +-sqrt(-5) and unreduced values no message reaches
extern "C" int ht_bls_sw_x_digest(const uint8_t* digest, uint8_t* out) {
  u32 d[16];
  for (int i = 0; i < 16; ++i) d[i] = ((u32)digest[4 * i] << 24) | ((u32)digest[4 * i + 1] << 16) | ((u32)digest[4 * i + 2] << 8) | digest[4 * i + 3];
  return bls_sw_x_digest(d, out);
}
extern "C" int ht_rx_pow(int curve, int op, uint8_t* bytes, i32* limbs) {
  return curve == 0 ? rx_pow<BN254>(op, bytes, limbs) : (curve == 2 ? rx_pow<BN254W>(op, bytes, limbs) : rx_pow<BLS381>(op, bytes, limbs));
}
#endif  // part 1

#if HT_HAS(2)

// ---- rx_pair.hpp: the whole sequence of point steps of one Miller loop on an emulated lane pair, every line (as handed
// to the consumer: three tight Fp2 entries, scaled by the hash point) and the final point compared with pairing.hpp's
// dbl_step_t / add_step_t.  Returns 0 when everything matches, else 1 + the index of the first differing step; -3 on overflow.
template <class C>
struct RxMillerRun {
  static constexpr int N = C::RX_NL;
  static constexpr int MAXS = 96;
  Ux<C> got[MAXS][3][2];       // [step][entry][half]
  Sx<C, SX_T> fin[3][2];
  int nsteps[2];
  Aff<F1<C>> P;
  Aff<F2<C>> Q;
  struct Env {
    Sx<C, SX_T> nyp, xp, xq_, yq_;
    Sx<C, SX_T> nyP() const { return nyp; }
    Sx<C, SX_T> xP() const { return xp; }
    Sx<C, SX_T> xq() const { return xq_; }
    Sx<C, SX_T> yq() const { return yq_; }
  };
  void lane(int l) {
    tl_pair_lane = l;
    const bool odd = l == 1;
    auto half = [&](const Fp2<C>& v) { return ux_to_sx<C>(to_ux<C>(odd ? v.c1 : v.c0)); };
    Env env;
    env.nyp = ux_to_sx<C>(to_ux<C>(fp_neg<C>(P.y)));
    env.xp = ux_to_sx<C>(to_ux<C>(P.x));
    const Sx<C, SX_T> xq = half(Q.x), yq = half(Q.y);
    PointX<C> T;
    T.X = xq;
    T.Y = yq;
    T.Z = sx_select<C>(odd, ux_to_sx<C>(ux_zero<C>()), sx_const<C>(C::RX_ONE));
    int s = 0;
    auto emit = [&](int which, const auto& v) {
      const int entry = which == 1 ? 1 : ((which == 0) == C::TWIST_D ? 0 : 2);
      if constexpr (rx_lazy<C>) got[s][entry][l] = sx_to_ux<C>(v);            // as miller_x.hpp's emit
      else if constexpr (std::is_same<std::decay_t<decltype(v)>, Sx<C, SX_T>>::value) got[s][entry][l] = sx_to_ux_p<C>(v);
      else got[s][entry][l] = sx_to_ux_k<1, C>(v);
    };
    for (int i = 1; i < C::LOOP_LEN; ++i) {
      dbl_step_x<C>(T, env, odd, emit);
      ++s;
      const int d = C::LOOP_NAF[i];
      if (d != 0) {
        env.xq_ = xq;
        env.yq_ = d > 0 ? yq : sx_neg<C>(yq);
        add_step_x<C>(T, env, odd, emit);
        ++s;
      }
    }
    if constexpr (C::CURVE_ID == 0) {
      const Fp2<C> x1 = f2_mul<C>(f2_conj<C>(Q.x), gamma_const<C>(1, 2)), y1 = f2_mul<C>(f2_conj<C>(Q.y), gamma_const<C>(1, 3));
      const Fp2<C> x2 = f2_mul<C>(Q.x, gamma_const<C>(2, 2)), y2 = f2_neg<C>(f2_mul<C>(Q.y, gamma_const<C>(2, 3)));
      env.xq_ = half(x1); env.yq_ = half(y1);
      add_step_x<C>(T, env, odd, emit);
      ++s;
      env.xq_ = half(x2); env.yq_ = half(y2);
      add_step_x<C>(T, env, odd, emit);
      ++s;
    }
    nsteps[l] = s;
    fin[0][l] = T.X; fin[1][l] = T.Y; fin[2][l] = T.Z;
  }
};
template <class C>
static int rx_miller_check(const uint8_t* g1, const uint8_t* g2) {
  static RxMillerRun<C> run;
  if (!g1_from_bytes<C>(run.P, g1) || !g2_from_bytes<C>(run.Q, g2)) return -2;
  g_rx_overflow = 0;
  g_pair_cnt.store(0);
  std::thread t1([&] { run.lane(1); });
  run.lane(0);
  t1.join();
  if (g_rx_overflow) return -3;
  // reference walk
  G2Proj<C> R = {run.Q.x, run.Q.y, f2_one<C>()};
  int s = 0;
  auto same = [&](const Fp2<C>& want, const Ux<C>& h0, const Ux<C>& h1) {
    // handed-over values are lazy (up to ~21 p): a product by one brings them below 2 p, which from_ux accepts
    const Ux<C> one = ux_load<C>(C::RX_ONE), zero = ux_zero<C>();
    const Ux2<C> red = ux_dot_k2p<C, 1>([&](int, int h) { return h ? h1 : h0; }, [&](int, int h) { return h ? zero : one; });
    const Fp2<C> g = {from_ux<C>(red.c0), from_ux<C>(red.c1)};
    return f2_eq<C>(g, want);
  };
  auto check_line = [&](const LineCoeffs<C>& l) {
    const Fp2<C> a = f2_muls<C>(l.c0, run.P.y), b = f2_muls<C>(l.c1, run.P.x);
    const Fp2<C> e0 = C::TWIST_D ? a : l.c2, e2 = C::TWIST_D ? l.c2 : a;
    const bool ok = same(e0, run.got[s][0][0], run.got[s][0][1]) && same(b, run.got[s][1][0], run.got[s][1][1]) && same(e2, run.got[s][2][0], run.got[s][2][1]);
    // handed-over entries must be tight and non-negative
    bool tight = true;
    for (int e = 0; e < 3; ++e) for (int h = 0; h < 2; ++h) for (int i = 0; i + 1 < C::RX_NL; ++i) tight = tight && run.got[s][e][h].v[i] < (1u << C::RX_W);
    ++s;
    return ok && tight;
  };
  for (int i = 1; i < C::LOOP_LEN; ++i) {
    if (!check_line(dbl_step_t<C, true>(R))) return s;
    const int d = C::LOOP_NAF[i];
    if (d != 0 && !check_line(add_step_t<C, true>(R, run.Q.x, d > 0 ? run.Q.y : f2_neg<C>(run.Q.y)))) return s;
  }
  if constexpr (C::CURVE_ID == 0) {
    const Fp2<C> x1 = f2_mul<C>(f2_conj<C>(run.Q.x), gamma_const<C>(1, 2)), y1 = f2_mul<C>(f2_conj<C>(run.Q.y), gamma_const<C>(1, 3));
    const Fp2<C> x2 = f2_mul<C>(run.Q.x, gamma_const<C>(2, 2)), y2 = f2_neg<C>(f2_mul<C>(run.Q.y, gamma_const<C>(2, 3)));
    if (!check_line(add_step_t<C, true>(R, x1, y1))) return s;
    if (!check_line(add_step_t<C, true>(R, x2, y2))) return s;
  }
  if (s != run.nsteps[0] || s != run.nsteps[1]) return 1000 + s;
  // the running point itself (signed tight halves -> non-negative -> library form)
  const Fp2<C> want[3] = {R.X, R.Y, R.Z};
  for (int k = 0; k < 3; ++k)
    if (!same(want[k], sx_to_ux<C>(run.fin[k][0]), sx_to_ux<C>(run.fin[k][1]))) return 2000 + k;
  return 0;
}
extern "C" int ht_rx_miller(int curve, const uint8_t* g1, const uint8_t* g2) {
  return curve == 0 ? rx_miller_check<BN254>(g1, g2) : (curve == 2 ? rx_miller_check<BN254W>(g1, g2) : rx_miller_check<BLS381>(g1, g2));
}
#endif  // part 2

#if HT_HAS(3)

// ---- rx_jac.hpp: a G2 key sum on the carry-free arithmetic (jacx_madd over the wire-format points, in order) against the
// library's own Jacobian mixed additions; exceptional cases included by the caller's choice of points.  out = the sum's
// wire bytes; returns 0, -2 on a non-canonical / off-curve point, -3 on a column overflow.
#include "../../bgls_amd/csrc/rx_jac.hpp"
template <class C>
static int rx_sum(const uint8_t* pts, int n, uint8_t* out, uint8_t* out_ref) {
  typedef F2<C> F;
  g_rx_overflow = 0;
  JacX<C> acc = jacx_inf<C>();
  Jac<F> ref = jac_inf<F>();
  for (int i = 0; i < n; ++i) {
    AffX<C> q;
    if (!affx_from_bytes<C>(q, pts + (size_t)i * 4 * C::FP_BYTES) || !affx_on_curve<C>(q)) return -2;
    acc = jacx_madd<C>(acc, q);
    Aff<F> a;
    if (!g2_from_bytes<C>(a, pts + (size_t)i * 4 * C::FP_BYTES) || !aff_on_curve<F>(a)) return -2;
    ref = jac_add_aff<F>(ref, a);
    // also through the resident (Montgomery) form of a key set
    const AffX<C> q2 = affx_from_mont<C>(a);
    for (int k = 0; k < C::RX_NL; ++k)
      if (q2.x.c0.v[k] != q.x.c0.v[k] || q2.y.c1.v[k] != q.y.c1.v[k]) return -4;
  }
  g2_to_bytes<C>(out, jac_to_aff<F>(jacx_to_mont<C>(acc)));
  g2_to_bytes<C>(out_ref, jac_to_aff<F>(ref));
  return g_rx_overflow ? -3 : 0;
}
extern "C" int ht_rx_sum(int curve, const uint8_t* pts, int n, uint8_t* out, uint8_t* out_ref) {
  return curve == 0 ? rx_sum<BN254>(pts, n, out, out_ref) : rx_sum<BLS381>(pts, n, out, out_ref);
}

// ---- rx_jacpair.hpp: the same key sum on an emulated LANE PAIR (two lock-stepped threads; jacp_madd over the wire-format
// points, in order, and once more through the resident Montgomery form).  out = the sum's wire bytes as the two lanes write
// them (sxp_to_mont halves of the Jacobian partial).  Returns 0, -2 bad point, -3 column overflow, -5 the two forms differ.
#include "../../bgls_amd/csrc/rx_jacpair.hpp"
template <class C>
struct RxSumPair {
  const uint8_t* pts;
  int n;
  Fp<C> part[3][6];            // [form][X.c0 X.c1 Y.c0 Y.c1 Z.c0 Z.c1]; form 2: two interleaved partial sums joined by jacp_add
  bool inf[3];
  int bad = 0;
  void lane(int l) {
    tl_pair_lane = l;
    const bool odd = l == 1;
    for (int form = 0; form < 3; ++form) {
      JacP<C> acc = jacp_inf<C>(), acc2 = jacp_inf<C>();
      for (int i = 0; i < n; ++i) {
        AffP<C> q;
        const uint8_t* b = pts + (size_t)i * 4 * C::FP_BYTES;
        if (form != 1) {
          const bool ok = affp_from_bytes<C>(q, b, odd);
          if (!(affp_on_curve<C>(q, odd) && ok)) bad = 1;
        } else {
          Aff<F2<C>> a;
          g2_from_bytes<C>(a, b);
          q = affp_from_mont<C>(a, odd);
        }
        if (form == 2 && (i & 1)) acc2 = jacp_madd<C>(acc2, q, odd);
        else acc = jacp_madd<C>(acc, q, odd);
      }
      if (form == 2) acc = jacp_add<C>(acc, acc2, odd);
      inf[form] = acc.inf;
      if (!acc.inf) {
        part[form][0 + l] = sxp_to_mont<C>(acc.X);
        part[form][2 + l] = sxp_to_mont<C>(acc.Y);
        part[form][4 + l] = sxp_to_mont<C>(acc.Z);
      }
    }
  }
};
template <class C>
static int rx_sumpair(const uint8_t* pts, int n, uint8_t* out) {
  typedef F2<C> F;
  g_rx_overflow = 0;
  g_pair_cnt.store(0);
  RxSumPair<C> run;
  run.pts = pts;
  run.n = n;
  std::thread t1([&] { run.lane(1); });
  run.lane(0);
  t1.join();
  if (run.bad) return -2;
  if (g_rx_overflow) return -3;
  Aff<F> got[3];
  for (int form = 0; form < 3; ++form) {
    Jac<F> j = jac_inf<F>();
    if (!run.inf[form]) j = {{run.part[form][0], run.part[form][1]}, {run.part[form][2], run.part[form][3]}, {run.part[form][4], run.part[form][5]}};
    got[form] = jac_to_aff<F>(j);
  }
  uint8_t other[4 * 48], third[4 * 48];
  g2_to_bytes<C>(out, got[0]);
  g2_to_bytes<C>(other, got[1]);
  g2_to_bytes<C>(third, got[2]);
  if (memcmp(out, third, 4 * C::FP_BYTES)) return -6;
  return memcmp(out, other, 4 * C::FP_BYTES) ? -5 : 0;
}
extern "C" int ht_rx_sumpair(int curve, const uint8_t* pts, int n, uint8_t* out) {
  return curve == 0 ? rx_sumpair<BN254>(pts, n, out) : (curve == 2 ? rx_sumpair<BN254W>(pts, n, out) : rx_sumpair<BLS381>(pts, n, out));
}

#endif  // part 3