// Host side of the aggregate-verify engine: contexts, workspaces, the stage pipeline and the C ABI declared in
// include/bgls_hip.h.  Pipeline of bgls.VerifyAggregateSignature (bgls/bgls.go:94-119) on the device:
//   dup_check     exact duplicate-message scan          (containsDuplicateMessage, bgls.go:139-150)
//   hash_to_g1    H(m_i) for every message              (concurrentHash, bgls.go:107-111,134-137)
//   g1_parse      -sigma                                (aggsig.Mul(-1), bgls.go:112)
//   miller        Miller values of every (H(m_i), pk_i) and of (-sigma, g2), multiplied into per-group partial products
//                                                       (concurrentPair + GT Add tree, curves/curve.go:132-169,217-223)
//   reduce        product of the partial products
//   final36       ONE final exponentiation, compare with 1 (Equals(GetGTIdentity), bgls.go:115-118)
// Kernels live in the k_*.hip units and are reached through launch.hpp.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include <atomic>
#include <string>
#include <thread>
#include <memory>
#include <optional>
#include <unordered_map>
#include <dlfcn.h>
// RCCL is reached through dlopen only (see struct Rccl): the few types and constants it needs are declared here, so the
// library also builds where the RCCL development headers are not installed.  Values follow nccl.h's ABI.
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;

#include <new>
#include <stdexcept>
#include <system_error>
#include "dev_common.hpp"
#include "coop.hpp"
#include "coop_r28.hpp"
#include "finalexp.hpp"
#include "miller_kernels.hpp"
#include "launch.hpp"
#include "launch_tail.hpp"
#include "../../include/bgls_hip.h"

using namespace bgls;

// ======================================================================= host side
namespace {

thread_local std::string g_err;

int fail(int code, const char* what, hipError_t e = hipSuccess) noexcept {
  char buf[256];
  if (e != hipSuccess)
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else
    snprintf(buf, sizeof buf, "%s", what);
  try { g_err = buf; } catch (...) {}      // the message is best effort, the code is the contract
  return code;
}

// The exception barrier of the C ABI ("no exceptions, no abort()": include/bgls_hip.h; the reference never panics and returns
// (nil, false): curves/curve.go:15-22).  Every extern "C" body below is a function-try-block ending in BGLS_ABI_GUARD: host
// containers sized by the caller (std::vector), std::string, std::thread and std::shared_ptr can throw, and an exception must
// not unwind into a cgo / ctypes frame.
int abi_catch() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    return fail(BGLS_ERR_NOMEM, "out of host memory");
  } catch (const std::length_error&) {
    return fail(BGLS_ERR_NOMEM, "host container size limit");
  } catch (const std::system_error& e) {
    return fail(BGLS_ERR_HIP, e.what());
  } catch (const std::exception& e) {
    return fail(BGLS_ERR_ARG, e.what());
  } catch (...) {
    return fail(BGLS_ERR_ARG, "unknown C++ exception");
  }
}
#define BGLS_ABI_GUARD catch (...) { return abi_catch(); }

// Error-path guard of a fork onto a context's side stream: kernels launched there read and write the context's workspaces, so
// an early return between the fork and the join must not leave them running into the next call on the same context.  Armed at
// the fork, disarmed once the main stream waits on the join event; an error return in between drains the side stream.
struct SideJoin {
  hipStream_t side;
  bool armed = false;
  ~SideJoin() {
    if (armed) (void)hipStreamSynchronize(side);
  }
};

#define HIPCHK(expr)                                         \
  do {                                                       \
    hipError_t e_ = (expr);                                  \
    if (e_ != hipSuccess) return fail(BGLS_ERR_HIP, #expr, e_); \
  } while (0)

// Largest batch one call accepts: the duplicate table (2n rounded up to a power of two, u32 slots), the hashing work
// lists (u32 indices) and the grid computations all assume n < 2^30.
constexpr size_t MAX_BATCH = (size_t)1 << 30;
// batches up to this many pairings take the latency form of the Miller loop (one block per pairing, k_miller_lat)
constexpr size_t LAT_MAX = 128;

// workspace slots
enum { WS_G1S = 0, WS_F_A, WS_F_B, WS_FLAGS, WS_TABLE, WS_IN_A, WS_IN_B, WS_IN_C, WS_IN_D, WS_OUT, WS_JAC_A, WS_JAC_B, WS_PART, WS_TMP, WS_TMP2, WS_H2C_LIST, WS_H2C_CNT, WS_H2C_PTS, WS_H2C_KIND, WS_HAE_ROOT, WS_HAE_T, WS_HAE_KEYS, WS_HAE_APK, WS_HAE_SIGN, WS_FLAGS2, WS_GEN_TMP, WS_LINES, WS_SUMJ, WS_QP, WS_MSM_AFF, WS_MSM_CNT, WS_MSM_START, WS_MSM_LIST, WS_SEG_OFF, WS_SEG_KEYS, WS_EPI, WS_TREE_S, WS_TREE_T, WS_NUM };

struct Ctx {
  std::mutex mu;
  int device = 0;
  bool ready = false;
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;                // a lone verification's independent stages run beside each other (fork / join with the two events)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::vector<std::pair<void*, size_t>> ws;  // cached device workspaces by slot
  uint32_t* h_res = nullptr;                 // pinned host words {verdict, final-stage flags, caller flags} of the verification in flight
  bool res_pending = false;
  hipStream_t res_stream = nullptr;
  // optional per-stage timing with HIP events on the launch stream (bench.py roofline leg)
  bool prof = false;
  struct Pending { hipEvent_t a, b; int stage; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;
  double stage_ms[8] = {0};
  unsigned long long stage_cnt[8] = {0};
  hipEvent_t ev() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void collect() {
    for (auto& p : pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { stage_ms[p.stage] += ms; stage_cnt[p.stage] += 1; }
      ev_pool.push_back(p.a);
      ev_pool.push_back(p.b);
    }
    pending.clear();
  }

  // makes this context's device current for the calling thread and creates the stream on first use
  int enter() {
    if (!ready) {
      int cnt = 0;
      hipError_t e = hipGetDeviceCount(&cnt);
      if (e != hipSuccess || cnt <= 0) return fail(BGLS_ERR_NO_DEVICE, "no HIP device available", e);
      if (device < 0 || device >= cnt) return fail(BGLS_ERR_NO_DEVICE, "device index out of range");
    }
    HIPCHK(hipSetDevice(device));
    if (ready) return 0;
    HIPCHK(hipStreamCreate(&stream));
    HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void**)&h_res, 64));
    ws.assign(WS_NUM + 8, {nullptr, 0});
    ready = true;
    return 0;
  }
  int get(int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (ws[slot].second < bytes) {
      if (ws[slot].first) HIPCHK(hipFree(ws[slot].first));
      ws[slot] = {nullptr, 0};
      size_t cap = bytes + bytes / 4;
      HIPCHK(hipMalloc(&ws[slot].first, cap));
      ws[slot].second = cap;
    }
    *out = ws[slot].first;
    return 0;
  }
};

// Contexts: each owns a stream, its workspaces and its stage timers.  A host thread works on context (device, index) =
// (bgls_init's device or bgls_select_device's, bgls_select_context's index, default 0): several contexts of one device
// let one thread keep several verifications in flight, so the serial, latency-bound stages of one (hashing rounds,
// reduction tail, final exponentiation) overlap the Miller launch of another; contexts of different devices are what
// the multi-GPU entry points drive from their worker threads.
constexpr int NCTX = 16;
constexpr int MAX_DEVICES = 16;
std::atomic<int> g_default_device{0};
thread_local int g_sel = 0;
thread_local int g_dev = -1;           // -1: the process default
Ctx* ctx_pool() {
  static Ctx c[MAX_DEVICES][NCTX];
  static std::once_flag once;
  std::call_once(once, [] {
    for (int d = 0; d < MAX_DEVICES; ++d)
      for (int k = 0; k < NCTX; ++k) c[d][k].device = d;
  });
  return &c[0][0];
}
int cur_device() { return g_dev >= 0 ? g_dev : g_default_device.load(); }
Ctx& ctx_of(int device, int index) { return ctx_pool()[(size_t)device * NCTX + index]; }
Ctx& ctx() { return ctx_of(cur_device(), g_sel); }

// Per-device tables built once: the fixed-argument line coefficients of the generator g2 (k_gen_lines), one per curve,
// and the window multiples d 2^(8j) g of both generators for batch key generation (k_fb_build).
// Built under the device's lock on a private stream and published only after the build has completed, so every stream
// of every context of that device may read them without further ordering.
struct DeviceTables {
  std::mutex mu;
  void* gen_lines[2] = {nullptr, nullptr};
  void* fixed_base[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [curve][group - 1]: window multiples of the generator (k_fb_build)
};
DeviceTables& tables_of(int device) {
  static DeviceTables t[MAX_DEVICES];
  return t[device];
}

// Throughput mode (bgls_set_throughput_mode / BGLS_THROUGHPUT=1): alt-bn128 batches use k_miller_s60, whose launches are
// meant to overlap with their neighbours (several verifications in flight).
std::atomic<int> g_throughput{-1};
bool throughput_mode() {
  int v = g_throughput.load();
  if (v < 0) {
    const char* e = getenv("BGLS_THROUGHPUT");
    v = (e && e[0] == '1') ? 1 : 0;
    g_throughput.store(v);
  }
  return v == 1;
}

// Miller shape (bgls_set_miller_shape): 0 = fused producer/consumer blocks (k_miller_ab64 / k_miller_s60), 1..3 = decoupled
// k_lines + k_fold through a line table in HBM (1: 32-bit limbs, 2: 28-bit limbs, 3: 28-bit limbs with Karatsuba dot
// products; alt-bn128 only for 2 and 3), ng = pairings folded per group and squaring.
// 0 also means "automatic": k_miller_x60 (carry-free 28-bit limbs, both curves) wherever it wins, see Engine::miller.
// 4 = k_miller_x60 always (the second argument is then its role / priority mode), 5 = the 32-bit fused kernels always.
// BGLS_MILLER_SHAPE / BGLS_X60_ROT preset them from the environment.
std::atomic<int> g_shape{-1}, g_ng{6}, g_x60_rot{-1};      // x60 mode -1: automatic (see Engine::miller)
int miller_shape() {
  // environment presets are read exactly once (thread-safe static initialiser), before any setter's value can be overwritten
  static const bool env_once = [] {
    const char* e = getenv("BGLS_MILLER_SHAPE");
    int v = e ? atoi(e) : 0;
    if (v < 0 || v > 5) v = 0;
    const char* r = getenv("BGLS_X60_ROT");
    int expect = -1;
    if (r && atoi(r) >= 0 && atoi(r) <= 31 && (atoi(r) & 3) != 3) g_x60_rot.compare_exchange_strong(expect, atoi(r));
    expect = -1;
    g_shape.compare_exchange_strong(expect, v);
    return true;
  }();
  (void)env_once;
  const int v = g_shape.load();
  return v < 0 ? 0 : v;
}

// BLS12-381 G1 scalar multiplications at the seam on the carry-free limbs (k_g1x.hip; measured at 2^18 points: Sign 55 -> 43 ms,
// ScalePoints 36 -> 29 ms, identical bytes; alt-bn128 gains nothing -- ten 28-bit limbs against eight 32-bit ones -- and keeps the
// 32-bit kernels); BGLS_G1X=0 keeps them on BLS12-381 as well (A/B runs)
bool g1x() {
  static const bool on = [] { const char* e = getenv("BGLS_G1X"); return !(e && e[0] == '0'); }();
  return on;
}

// G2 key sums on the carry-free 28-bit-limb form.  Mode 2 (default): one key sum partial per LANE PAIR (k_sumpair.hip,
// rx_jacpair.hpp: half of every Fp2 value per lane, three waves per SIMD).  Mode 1: one partial per lane (k_sumx.hip; measured
// at 2^20 keys, main pass + tree: BLS12-381 1.20 vs 1.30 ms for the 32-bit form, alt-bn128 0.66 vs 0.60 ms -- one lane cannot
// hold the three piles of a Karatsuba Fp2 product next to a Jacobian point).  Mode 0: the 32-bit form (k_points.hip k_sum_main).
// BGLS_SUMX=0 / 1 / 2 forces a mode for both curves (A/B measurements).
template <class C>
int sum_mode() {
  static const int v = [] { const char* e = getenv("BGLS_SUMX"); return e ? atoi(e) : -1; }();
  return v < 0 || v > 2 ? 2 : v;
}

#ifdef BGLS_DEV
// development builds only: BGLS_MILLER_DBG=1/2 times the producer / consumer half of the fused Miller kernels (WRONG results)
int miller_dbg() {
  static const int v = [] { const char* e = getenv("BGLS_MILLER_DBG"); return e ? atoi(e) : 0; }();
  return v;
}
#else
constexpr int miller_dbg() { return 0; }
#endif

// ST_SUM is opened ONCE per key sum (main pass + tree + conversion); ST_SUM_MAIN brackets the main-pass kernel alone, inside it
enum { ST_DUP = 0, ST_H2C, ST_MILLER, ST_REDUCE, ST_FINAL, ST_SUM, ST_SUM_MAIN, ST_NUM };
const char* const STAGE_NAMES[ST_NUM] = {"dup_check", "h2c", "miller", "reduce", "final_exp", "sum_points", "sum_main"};

struct Scope {  // brackets the launches of one stage with events when profiling is on
  Ctx& c; hipStream_t st; int stage; hipEvent_t a = nullptr;
  Scope(Ctx& c_, hipStream_t st_, int stage_) : c(c_), st(st_), stage(stage_) {
    if (c.prof) { a = c.ev(); (void)hipEventRecord(a, st); }
  }
  ~Scope() {
    if (c.prof && a) { hipEvent_t b = c.ev(); (void)hipEventRecord(b, st); c.pending.push_back({a, b, stage}); }
  }
};

template <class C>
struct Engine {
  typedef F1<C> G1F;
  typedef F2<C> G2F;
  static constexpr size_t FB = C::FP_BYTES, G1B = 2 * FB, G2B = 4 * FB, GTB = 12 * FB;

  static int dup_scan(Ctx& c, hipStream_t st, MsgView mv, size_t n, uint32_t* d_flags) {
    if (n < 2) return 0;
    if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
    size_t cap = 1;
    while (cap < 2 * n) cap <<= 1;
    void* tab;
    int rc;
    if ((rc = c.get(WS_TABLE, cap * 4, &tab))) return rc;
    Scope sc(c, st, ST_DUP);
    HIPCHK(hipMemsetAsync(tab, 0, cap * 4, st));
    kl::dup_check(st, mv, n, (uint32_t*)tab, (uint32_t)(cap - 1), d_flags);
    HIPCHK(hipGetLastError());
    return 0;
  }

  // d_flags: device u32 (already zeroed by caller).  Writes the product of the n (+1) Miller values to d_partial (GT
  // bytes, no final exponentiation).
  // d_w16 != nullptr: pair i is (w_i H(m_i), pk_i) with 16-byte big-endian weights (hashed aggregation exponents).
  static int miller_product(Ctx& c, hipStream_t st, const uint8_t* d_sig, const uint8_t* d_keys, MsgView mv, size_t n,
                            int check_dups, uint8_t* d_partial, uint32_t* d_flags, const uint8_t* d_w16 = nullptr) {
    if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
    const bool raw = C::CURVE_ID == 1 && n > 0;           // BLS12-381: uncleared hash points, cofactor applied once in GT (any batch size:
                                                          // clearing it per message is a 126-bit scalar multiplication on a lone lane, 4.8 ms)
    void* g1s;
    int rc;
    if ((rc = c.get(WS_G1S, (n + 2) * sizeof(Aff<G1F>), &g1s))) return rc;
    // The signature pair (-sigma, g2) depends on nothing but sigma.  A verification with the machine to itself (one stream, not
    // throughput mode) walks it on the context's side stream WHILE THE MESSAGES ARE HASHED, and joins before the Miller launch:
    // the epilogue behind the reduce stage is then only rest^h and one product (0.4 ms off the serial tail on alt-bn128).  The join
    // is early on purpose: a block still running beside a launch that fills the chip exactly would displace one of its blocks
    // into a second round (measured: Miller stage 4.9 -> 6.0 ms at 2^16).
    static const bool sig_early_on = [] { const char* e = getenv("BGLS_EPIX"); const char* f = getenv("BGLS_SIG_EARLY"); return !(e && e[0] == '0') && !(f && f[0] == '0'); }();
    const bool sig_early = d_sig != nullptr && sig_early_on && !throughput_mode() && n > LAT_MAX && (miller_shape() == 0 || miller_shape() == 4);
    Fp2<C>* sig_half = nullptr;
    SideJoin sj{c.side};
    if (d_sig && sig_early) {
      const LineCoeffs<C>* gl = nullptr;
      void* tmp;
      if ((rc = gen_lines(c, &gl))) return rc;
      if ((rc = c.get(WS_EPI, 12 * sizeof(Fp2<C>), &tmp))) return rc;
      kl::g1_parse<C>(st, d_sig, 1, 1, (Aff<G1F>*)g1s + n, d_flags);
      HIPCHK(hipEventRecord(c.ev_fork, st));
      HIPCHK(hipStreamWaitEvent(c.side, c.ev_fork, 0));
      sj.armed = true;
      kl::cofactor_epiloguex_part<C>(c.side, 1, nullptr, (const Aff<G1F>*)g1s + n, gl, (Fp2<C>*)tmp, nullptr);
      HIPCHK(hipEventRecord(c.ev_join, c.side));
      sig_half = (Fp2<C>*)tmp;
    }
    if (check_dups && (rc = dup_scan(c, st, mv, n, d_flags))) return rc;
    if (n) {
      Scope sc(c, st, ST_H2C);
      if ((rc = hash_to_g1(c, st, mv, n, (Aff<G1F>*)g1s, d_flags, raw))) return rc;
      if (d_w16) kl::scale_g1_inplace<C>(st, (Aff<G1F>*)g1s, d_w16, n);
    }
    if (d_sig && !sig_early) kl::g1_parse<C>(st, d_sig, 1, 1, (Aff<G1F>*)g1s + n, d_flags);
    if (sig_half) HIPCHK(hipStreamWaitEvent(st, c.ev_join, 0));
    sj.armed = false;
    return miller(c, st, (const Aff<G1F>*)g1s, d_keys, n, d_sig ? (const Aff<G1F>*)g1s + n : nullptr, d_partial, d_flags, raw, sig_half);
  }

  // The same product against a PREPARED key range (prepared.hpp): no point steps, the hash points only scale the resident
  // line ratios.  BLS12-381 hash points stay uncleared as on the unprepared path: the cofactor is applied once in GT.
  static int miller_product_prepared(Ctx& c, hipStream_t st, const uint8_t* d_sig, const uint32_t* d_prep, const uint8_t* d_kinf, size_t n_pad,
                                     MsgView mv, size_t n, uint8_t* d_partial, uint32_t* d_flags) {
    if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
    void *g1s, *ptab, *pa, *pb;
    int rc;
    if ((rc = c.get(WS_G1S, (n + 2) * sizeof(Aff<G1F>), &g1s))) return rc;
    const bool raw = C::CURVE_ID == 1 && n > 0;
    if (n) {
      Scope sc(c, st, ST_H2C);
      if ((rc = hash_to_g1(c, st, mv, n, (Aff<G1F>*)g1s, d_flags, raw))) return rc;
    }
    const Aff<G1F>* sig = nullptr;
    const LineCoeffs<C>* gl = nullptr;
    if (d_sig) {
      kl::g1_parse<C>(st, d_sig, 1, 1, (Aff<G1F>*)g1s + n, d_flags);
      sig = (const Aff<G1F>*)g1s + n;
      if ((rc = gen_lines(c, &gl))) return rc;
    }
    if (n == 0) return miller(c, st, (const Aff<G1F>*)g1s, nullptr, 0, sig, d_partial, d_flags, false);
    // pairings per squaring: as many as still leave about two waves per SIMD
    const int ng = n_pad >= ((size_t)1 << 19) ? 24 : n_pad >= ((size_t)1 << 18) ? 12 : 6;
    const kl::PrepSizes ps = kl::prep_sizes<C>();
    const size_t groups = n_pad / ng;
    if ((rc = c.get(WS_LINES, n_pad * ps.point_bytes, &ptab))) return rc;
    if ((rc = c.get(WS_F_A, (groups + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
    if ((rc = c.get(WS_F_B, (groups / REDUCE_R + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
    {
      Scope sc(c, st, ST_MILLER);
      kl::prep_points<C>(st, (const Aff<G1F>*)g1s, d_kinf, n, n_pad, (uint32_t*)ptab);
      kl::fold_prep<C>(st, d_prep, (const uint32_t*)ptab, n_pad, ng, (Fp2<C>*)pa);
      HIPCHK(hipGetLastError());
    }
    Fp2<C>* red = nullptr;
    if ((rc = reduce(c, st, (Fp2<C>*)pa, (Fp2<C>*)pb, groups, &red))) return rc;
    return emit_partial(c, st, red, raw || sig != nullptr, sig, gl, d_partial);
  }

  // H(m_i) as affine Montgomery points.  raw (BLS12-381 only): points before cofactor clearing, for the cofactor-in-GT
  // verification path.
  static int hash_to_g1(Ctx& c, hipStream_t st, MsgView mv, size_t n, Aff<G1F>* out, uint32_t* d_flags, bool raw = false) {
    if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
    int rc;
    void *lists = nullptr, *cnts = nullptr;
    if constexpr (C::CURVE_ID == 0) {
      if (n >= 256) {
        if ((rc = c.get(WS_H2C_LIST, 2 * n * 4, &lists))) return rc;
        if ((rc = c.get(WS_H2C_CNT, 64, &cnts))) return rc;
        HIPCHK(hipMemsetAsync(cnts, 0, 64, st));
      }
      kl::h2c_bn(st, mv, n, (uint32_t*)lists, (uint32_t*)cnts, out, d_flags, throughput_mode());
    } else {
      void *pts, *kinds;
      if ((rc = c.get(WS_H2C_PTS, 2 * n * sizeof(Jac<G1F>), &pts))) return rc;
      if ((rc = c.get(WS_H2C_KIND, 2 * n * 4, &kinds))) return rc;
      kl::h2c_bls(st, mv, n, (Jac<G1F>*)pts, (uint32_t*)kinds, out, raw);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }

  // fixed-argument lines of g2 for this device (DeviceTables)
  static int gen_lines(Ctx& c, const LineCoeffs<C>** out) {
    DeviceTables& t = tables_of(c.device);
    std::lock_guard<std::mutex> lk(t.mu);
    if (!t.gen_lines[C::CURVE_ID]) {
      void *tab = nullptr, *cnt = nullptr;
      hipStream_t bs = nullptr;
      HIPCHK(hipMalloc(&tab, 160 * sizeof(LineCoeffs<C>)));
      hipError_t e = hipMalloc(&cnt, 16);
      if (e == hipSuccess) e = hipStreamCreate(&bs);
      if (e == hipSuccess) {
        kl::gen_lines<C>(bs, (LineCoeffs<C>*)tab, (int*)cnt);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(bs);
      }
      if (bs) (void)hipStreamDestroy(bs);
      if (cnt) (void)hipFree(cnt);
      if (e != hipSuccess) {
        (void)hipFree(tab);
        return fail(BGLS_ERR_HIP, "building the generator line table", e);
      }
      t.gen_lines[C::CURVE_ID] = tab;
    }
    *out = (const LineCoeffs<C>*)t.gen_lines[C::CURVE_ID];
    return 0;
  }

  // product of the per-group partial products (w-basis Fp12 arrays, 6 Fp2 each) down to one.  Fan-in REDUCE_R per pass: a pass
  // costs a launch boundary plus R - 1 dependent products (~11 us each on six lanes); three per pass is the shortest chain at
  // the sizes that matter (65 partials of an n = 64 verification: 4 passes x 2 products instead of 4 x 3, 1.73 -> 1.69 ms;
  // 10 240 of a 2^16 batch: 0.24 -> 0.22 ms).  The second buffer (b) must hold count / REDUCE_R + 1 products.
  static constexpr int REDUCE_R = 3;
  static constexpr size_t REDUCE_FX_MAX = 12288;
  static int reduce(Ctx& c, hipStream_t st, Fp2<C>* a, Fp2<C>* b, size_t cnt, Fp2<C>** out) {
    Scope sc(c, st, ST_REDUCE);
    const int R = REDUCE_R;
    // passes with at most REDUCE_FX_MAX products left run one product per BLOCK on the carry-free 36-lane form (k_reduce_fx:
    // ~2-3 us per dependent product instead of 11 / 22 us); BGLS_REDUCEX=0 keeps k_reduce_coop everywhere (A/B runs)
    static const bool rfx = [] { const char* e = getenv("BGLS_REDUCEX"); return !(e && e[0] == '0'); }();
    static const int rfx_r = [] { const char* e = getenv("BGLS_REDUCEX_R"); const int v = e ? atoi(e) : 0; return v >= 3 && v <= 12 ? v : 10; }();      // >= REDUCE_R: the second buffer is sized for that
    while (cnt > 1) {
      const bool fx = rfx && cnt <= REDUCE_FX_MAX;
      const int Rp = fx ? rfx_r : R;                          // operands converted side by side into LDS slots (at most 12), then a bare chain of ~1.5 us products: fewer, longer passes
      const size_t nout = (cnt + Rp - 1) / Rp;
      if (fx) kl::reduce_fx<C>(st, a, cnt, Rp, b);
      else kl::reduce_coop<C>(st, a, cnt, Rp, b);
      Fp2<C>* t = a;
      a = b;
      b = t;
      cnt = nout;
    }
    HIPCHK(hipGetLastError());
    *out = a;
    return 0;
  }

  // Miller product of npairs (g1s[i], g2s[i]) pairs and, when sig != nullptr, of (*sig, g2) on the generator's
  // pre-computed lines; GT bytes (no final exponentiation) to d_partial.  sig must be an element of the g1s array
  // (verification stores -sigma behind the hash points).  cofactor: the g1s are uncleared BLS12-381 hash points, the
  // product of their Miller values is raised to the G1 cofactor before the signature pair is folded in.
  // sig_half != nullptr: the signature pair's Miller value is already there (12 Fp2 of scratch, the first six filled by
  // cofactor_epiloguex_part 1): the x60 shape's epilogue is then rest^h and the product only.
  static int miller(Ctx& c, hipStream_t st, const Aff<G1F>* g1s, const uint8_t* g2s, size_t npairs, const Aff<G1F>* sig,
                    uint8_t* d_partial, uint32_t* d_flags, bool cofactor = false, Fp2<C>* sig_half = nullptr) {
    if (npairs == 0 && !sig) {
      HIPCHK(hipMemsetAsync(d_partial, 0, GTB, st));
      HIPCHK(hipMemsetAsync(d_partial + GTB - 1, 1, 1, st));
      return 0;
    }
    int rc;
    // the epilogue kernels of the throughput shapes always raise a BLS12-381 product to the G1 cofactor: cleared hash points
    // together with a signature pair would come out as prod^h * e(-sigma, g2) there.  No caller does that (every BLS12-381
    // verification pairs uncleared points); refuse it rather than return a wrong GT value.
    if (C::CURVE_ID == 1 && sig && !cofactor && npairs > LAT_MAX)
      return fail(BGLS_ERR_ARG, "BLS12-381: a signature pair with cleared hash points is only served by the latency shape (<= 128 pairings)");
    const LineCoeffs<C>* gl = nullptr;
    if (sig && (rc = gen_lines(c, &gl))) return rc;
    void *pa, *pb;
    Fp2<C>* red = nullptr;
    bool epilogue = cofactor;
    if (npairs <= LAT_MAX && miller_shape() == 0) {
      // a handful of pairings: one block per pairing (k_miller_latx), the signature pair as one more block -- or, with uncleared
      // BLS12-381 hash points, in the epilogue that raises the other pairs' product to the cofactor
      const bool sig_block = sig && !cofactor;
      const size_t blocks = npairs + (sig_block ? 1 : 0);
      if ((rc = c.get(WS_F_A, (blocks + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
      if ((rc = c.get(WS_F_B, (blocks / REDUCE_R + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
      {
        Scope sc(c, st, ST_MILLER);
        // k_miller_latx: the same two-wave block on the carry-free limbs (BGLS_LATX=0 keeps k_miller_lat: A/B runs)
        static const bool latx = [] { const char* e = getenv("BGLS_LATX"); return !(e && e[0] == '0'); }();
        if (latx) kl::miller_latx<C>(st, g1s, g2s, npairs, sig_block ? (long long)(sig - g1s) : -1LL, gl, (Fp2<C>*)pa, d_flags);
        else kl::miller_lat<C>(st, g1s, g2s, npairs, sig_block ? (long long)(sig - g1s) : -1LL, gl, (Fp2<C>*)pa, d_flags);
        HIPCHK(hipGetLastError());
      }
      if ((rc = reduce(c, st, (Fp2<C>*)pa, (Fp2<C>*)pb, blocks, &red))) return rc;
      return emit_partial(c, st, red, cofactor, cofactor ? sig : nullptr, gl, d_partial);
    }
    // k_miller_x60 (both roles on carry-free 28-bit limbs; the signature pair goes to the epilogue kernel) is the default above
    // the latency shape.  1024 blocks are resident at a time.  Its 60-pairing block is the faster one per pairing; the
    // 64-pairing block (a seventh line in four of a block's ten groups: the consumer's step is a fold longer) needs fewer
    // blocks, and a verification with the machine to itself takes it where that saves a nearly empty last round of blocks --
    // 61 441 .. 65 536 pairings (exactly 2^16: BASELINE configs 2 / 3) are ONE round instead of two.  (Round 3 fell back to
    // the 32-bit k_miller_ab64 there.)  In throughput mode the neighbours' blocks fill the last round and the 60-form stays.
    const bool x60_auto = miller_shape() == 0 && npairs > LAT_MAX;
    if ((miller_shape() == 4 || x60_auto) && npairs >= 1) {
      constexpr size_t RES = 1024;                      // resident blocks: 256 CUs x 4
      const size_t nb60 = (npairs + 59) / 60, nb64 = (npairs + 63) / 64;
      const size_t r60 = (nb60 + RES - 1) / RES, r64 = (nb64 + RES - 1) / RES;
      // A/B runs and tests: BGLS_X_NP=60 / 64, or bgls_set_miller_shape(4, mode) with mode bit 16 = the 64-form (clear = the 60-form)
      static const int env_np = [] { const char* e = getenv("BGLS_X_NP"); return e ? atoi(e) : 0; }();
      const int force_np = miller_shape() == 4 && g_x60_rot.load() >= 0 ? ((g_x60_rot.load() & 16) ? 64 : 60) : env_np;
      const bool np64 = force_np == 64 || (force_np != 60 && !throughput_mode() && r64 < r60 && r64 <= 2);
      // role / priority mode: consumers placed by SIMD; the producers get issue priority only when the whole batch is one round of
      // resident blocks with the machine to itself (there the slowest block is the launch: 5.5 instead of 7.2 ms for 61 440
      // BLS12-381 pairings), in steady state it costs 3-6 % (measured at 2^20, four verifications in flight)
      const size_t nb = np64 ? nb64 : nb60, NPB = np64 ? 64 : 60, groups = nb * 10;
      const int xmode = g_x60_rot.load() >= 0 ? (g_x60_rot.load() & 15) : ((nb <= RES && !throughput_mode()) ? 8 : 0);
      constexpr size_t XB = 32768;                      // blocks per launch (the lanes' parked operands take up to 57 / 49 KB per block)
      void* park;
      const size_t pblocks = nb < XB ? nb : XB;
      if ((rc = c.get(WS_QP, np64 ? kl::miller_x_park_bytes<C, 64>(pblocks) : kl::miller_x_park_bytes<C, 60>(pblocks), &park))) return rc;
      if ((rc = c.get(WS_F_A, (groups + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
      if ((rc = c.get(WS_F_B, (groups / REDUCE_R + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
      {
        Scope sc(c, st, ST_MILLER);
        for (size_t blk0 = 0; blk0 < nb; blk0 += XB) {
          const size_t nblocks = nb - blk0 < XB ? nb - blk0 : XB;
          const size_t p0 = blk0 * NPB;
          if (np64) kl::miller_x<C, 64>(st, (unsigned)nblocks, g1s + p0, g2s + p0 * G2B, npairs - p0, (Fp2<C>*)pa + blk0 * 60, d_flags, (uint32_t*)park, xmode);
          else kl::miller_x<C, 60>(st, (unsigned)nblocks, g1s + p0, g2s + p0 * G2B, npairs - p0, (Fp2<C>*)pa + blk0 * 60, d_flags, (uint32_t*)park, xmode);
        }
        HIPCHK(hipGetLastError());
      }
      if ((rc = reduce(c, st, (Fp2<C>*)pa, (Fp2<C>*)pb, groups, &red))) return rc;
      if (sig_half) {
        kl::cofactor_epiloguex_part<C>(st, 2, red, sig, gl, sig_half, d_partial);
        HIPCHK(hipGetLastError());
        return 0;
      }
      return emit_partial(c, st, red, cofactor || sig != nullptr, sig, gl, d_partial);
    }
    if (miller_shape() > 0 && miller_shape() < 4 && npairs >= 1) {
      // decoupled: line table in HBM, then folds; batches above 2^16 pairings go chunk by chunk through one table
      int variant = miller_shape() - 1;
      if (C::CURVE_ID != 0 && variant > 0) variant = 0;
      const int ng = g_ng.load();
      const size_t chunk = (size_t)1 << 18;
      const size_t per_wave = (size_t)ng * 10;
      const size_t max_pad = ((npairs < chunk ? npairs : chunk) + per_wave - 1) / per_wave * per_wave;
      const size_t groups_total = ((npairs + chunk - 1) / chunk) * (max_pad / ng);
      void* tab;
      if ((rc = c.get(WS_LINES, kl::lines_bytes<C>(variant, max_pad), &tab))) return rc;
      if ((rc = c.get(WS_F_A, (groups_total + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
      if ((rc = c.get(WS_F_B, (groups_total / REDUCE_R + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
      size_t gdone = 0;
      {
        Scope sc(c, st, ST_MILLER);
        for (size_t p0 = 0; p0 < npairs; p0 += chunk) {
          const size_t np = npairs - p0 < chunk ? npairs - p0 : chunk;
          const size_t n_pad = (np + per_wave - 1) / per_wave * per_wave;
          kl::miller_lines<C>(st, variant, g1s + p0, g2s + p0 * G2B, np, n_pad, (uint32_t*)tab, d_flags);
          kl::miller_fold<C>(st, variant, (const uint32_t*)tab, n_pad, ng, (Fp2<C>*)pa + gdone * 6);
          gdone += n_pad / ng;
        }
        HIPCHK(hipGetLastError());
      }
      if ((rc = reduce(c, st, (Fp2<C>*)pa, (Fp2<C>*)pb, gdone, &red))) return rc;
      return emit_partial(c, st, red, cofactor || sig != nullptr, sig, gl, d_partial);
    }
    if constexpr (C::CURVE_ID == 0) {
      if (throughput_mode() && npairs >= 1) {
        // 60 pairings per block, 28-bit-limb consumer; the signature pair goes to the epilogue kernel
        const size_t nb60 = (npairs + 59) / 60, groups = nb60 * 10;
        void* qp;
        if ((rc = c.get(WS_QP, kl::miller_qp_bytes<C>(nb60 < 8192 ? nb60 : 8192), &qp))) return rc;
        if ((rc = c.get(WS_F_A, (groups + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
        if ((rc = c.get(WS_F_B, (groups / REDUCE_R + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
        {
          Scope sc(c, st, ST_MILLER);
          for (size_t blk0 = 0; blk0 < nb60; blk0 += 8192) {
            const size_t nblocks = nb60 - blk0 < 8192 ? nb60 - blk0 : 8192;
            const size_t p0 = blk0 * 60;
            kl::miller_s60<C>(st, (unsigned)nblocks, g1s + p0, g2s + p0 * G2B, npairs - p0, (Fp2<C>*)pa + blk0 * 60, d_flags, miller_dbg(), (uint32_t*)qp);
          }
          HIPCHK(hipGetLastError());
        }
        if ((rc = reduce(c, st, (Fp2<C>*)pa, (Fp2<C>*)pb, groups, &red))) return rc;
        epilogue = sig != nullptr;
        return emit_partial(c, st, red, epilogue, sig, gl, d_partial);
      }
    }
    // 64 pairings per block at 256 registers: one 2^16 batch is exactly 1024 resident blocks; larger batches run as
    // launches of up to 16384 blocks (one launch drains once: 5 % faster alone than sixteen launches of 1024).  Block 0 of the first launch also scales the generator lines for the
    // signature pair (unless the epilogue kernel does: cofactor path).
    const size_t nb64 = npairs ? (npairs + 63) / 64 : 1, groups = nb64 * 10;
    void* qp;
    constexpr size_t AB = 16384;                        // blocks per launch: 2^20 pairings; the parked operands of a launch take 200 / 300 MB
    if ((rc = c.get(WS_QP, kl::miller_qp_bytes<C>(nb64 < AB ? nb64 : AB), &qp))) return rc;
    if ((rc = c.get(WS_F_A, (groups + 1) * 6 * sizeof(Fp2<C>), &pa))) return rc;
    if ((rc = c.get(WS_F_B, (groups / REDUCE_R + 2) * 6 * sizeof(Fp2<C>), &pb))) return rc;
    {
      Scope sc(c, st, ST_MILLER);
      for (size_t blk0 = 0; blk0 < nb64; blk0 += AB) {
        const size_t nblocks = nb64 - blk0 < AB ? nb64 - blk0 : AB;
        const size_t p0 = blk0 * 64;
        const size_t np = npairs - p0 < nblocks * 64 ? npairs - p0 : nblocks * 64;
        const long long sig_at = (blk0 == 0 && sig && !cofactor) ? (long long)(sig - (g1s + p0)) : -1LL;
        kl::miller_ab64<C>(st, (unsigned)nblocks, g1s + p0, g2s + p0 * G2B, np, sig_at, gl, (Fp2<C>*)pa + blk0 * 10 * 6, d_flags, miller_dbg(), (uint32_t*)qp);
      }
      HIPCHK(hipGetLastError());
    }
    if ((rc = reduce(c, st, (Fp2<C>*)pa, (Fp2<C>*)pb, groups, &red))) return rc;
    return emit_partial(c, st, red, cofactor, cofactor ? sig : nullptr, gl, d_partial);
  }

  // serialise the reduced product; with `epilogue`: raise it to the G1 cofactor (BLS12-381 raw hash points; a no-op
  // exponent on alt-bn128) and fold the signature pair in on the 36-lane arithmetic
  static int emit_partial(Ctx& c, hipStream_t st, const Fp2<C>* w, bool epilogue, const Aff<G1F>* sig, const LineCoeffs<C>* gl,
                          uint8_t* d_partial) {
    if (!epilogue) {
      kl::w_to_bytes<C>(st, w, d_partial);
    } else {
      // k_epilogue_ax / _bx: the two chains on the carry-free limbs, side by side as two blocks (BGLS_EPIX=0 keeps k_cofactor_epilogue)
      static const bool epix = [] { const char* e = getenv("BGLS_EPIX"); return !(e && e[0] == '0'); }();
      void* tmp = nullptr;
      int rc;
      if (epix && (rc = c.get(WS_EPI, 12 * sizeof(Fp2<C>), &tmp))) return rc;
      if (epix) kl::cofactor_epiloguex<C>(st, w, sig, gl, (Fp2<C>*)tmp, d_partial, nullptr);
      else kl::cofactor_epilogue<C>(st, w, sig, gl, d_partial);
    }
    HIPCHK(hipGetLastError());
    return 0;
  }

  // enqueue product-of-partials + final exponentiation + compare; the verdict lands in the context's pinned words
  static int finalize_submit(Ctx& c, hipStream_t st, const uint8_t* d_partials, size_t count, int do_final_exp, const uint32_t* d_flags_in,
                             uint8_t* h_gt_out) {
    void *tmp, *fl;
    int rc;
    if (c.res_pending) return fail(BGLS_ERR_ARG, "a verification is already in flight on this context (collect it first)");
    if ((rc = c.get(WS_TMP, GTB + 16, &tmp))) return rc;
    if ((rc = c.get(WS_OUT, 16, &fl))) return rc;
    uint8_t* d_gt = (uint8_t*)tmp;
    uint32_t* d_verdict = (uint32_t*)(d_gt + GTB);        // three words: {verdict, final-stage flags, caller flags} on the finalx path
    uint32_t* d_fl2 = (uint32_t*)fl;
    // finalx.hpp: the same 36-lane split on the carry-free limbs (BGLS_FINALX=0 keeps the 32-bit form of finalexp.hpp: A/B runs)
    static const bool finalx = [] { const char* e = getenv("BGLS_FINALX"); return !(e && e[0] == '0'); }();
    if (!finalx) HIPCHK(hipMemsetAsync(d_fl2, 0, 4, st));
    {
      Scope sc(c, st, ST_FINAL);
      if (finalx) kl::finalx_res<C>(st, d_partials, count, do_final_exp, d_gt, d_verdict, d_flags_in);
      else kl::final36<C>(st, d_partials, count, do_final_exp, d_gt, d_verdict, d_fl2);
    }
    HIPCHK(hipGetLastError());
    c.h_res[0] = c.h_res[1] = c.h_res[2] = 0;
    if (finalx) {
      HIPCHK(hipMemcpyAsync(&c.h_res[0], d_verdict, 12, hipMemcpyDeviceToHost, st));      // one copy: the kernel gathered the three words
    } else {
      HIPCHK(hipMemcpyAsync(&c.h_res[0], d_verdict, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&c.h_res[1], d_fl2, 4, hipMemcpyDeviceToHost, st));
      if (d_flags_in) HIPCHK(hipMemcpyAsync(&c.h_res[2], d_flags_in, 4, hipMemcpyDeviceToHost, st));
    }
    if (h_gt_out) HIPCHK(hipMemcpyAsync(h_gt_out, d_gt, GTB, hipMemcpyDeviceToHost, st));
    c.res_pending = true;
    c.res_stream = st;
    return 0;
  }
  // wait for the verification in flight; returns 1/0 or <0
  static int finalize_collect(Ctx& c) {
    if (!c.res_pending) return fail(BGLS_ERR_ARG, "no verification in flight on this context");
    c.res_pending = false;
    HIPCHK(hipStreamSynchronize(c.res_stream));
    c.collect();
    uint32_t f = c.h_res[1] | c.h_res[2];
    if (f & FLAG_ENC) return fail(BGLS_ERR_ENCODING, "non-canonical coordinate or point not on curve");
    if (f & FLAG_SUBGROUP) return fail(BGLS_ERR_ENCODING, "G2 point outside the order-r subgroup");
    if (f & FLAG_DEGENERATE) return fail(BGLS_ERR_ENCODING, "degenerate point step (small-order key)");
    if (f & FLAG_HASH) return fail(BGLS_ERR_HASH, "try-and-increment exhausted");
    if (f & FLAG_DUP) return 0;
    return c.h_res[0] ? 1 : 0;
  }
  // returns 1/0 or <0; optionally copies the GT bytes out
  static int finalize(Ctx& c, hipStream_t st, const uint8_t* d_partials, size_t count, int do_final_exp, const uint32_t* d_flags_in,
                      uint8_t* h_gt_out) {
    int rc;
    if ((rc = finalize_submit(c, st, d_partials, count, do_final_exp, d_flags_in, h_gt_out))) return rc;
    return finalize_collect(c);
  }

  // AggregatePoints (curves/curve.go:73-121): affine bytes of the sum of n points to d_out.
  // src: 0 = wire bytes; 1 = the Montgomery affine points of a key set; 2 = its sum-ready records (lane-pair kernel only)
  static int sum_points(Ctx& c, hipStream_t st, int group, const uint8_t* d_pts, size_t n, uint8_t* d_out, uint32_t* d_flags,
                        int src = 0) {
    const size_t PTB = group == BGLS_G1 ? G1B : G2B;
    if (n == 0) {
      HIPCHK(hipMemsetAsync(d_out, 0, PTB, st));
      return 0;
    }
    void* jac;
    int rc;
    if ((rc = c.get(WS_SUMJ, 4 * kl::jac_bytes<C>(group), &jac))) return rc;
    Scope sc(c, st, ST_SUM);                                   // one scope per key sum: the stage count equals the number of sums
    bool bytes_done = false;
    if ((rc = sum_points_jac(c, st, group, d_pts, n, jac, d_flags, src, false, d_out, &bytes_done))) return rc;
    if (!bytes_done) kl::jac_to_bytes<C>(st, group, jac, 1, d_out);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // nsets sums in one pass: set b = points d_off[b] .. d_off[b+1] (d_off: nsets + 1 offsets on the device, max_set = the
  // largest set); wire bytes of the nsets sums to d_out.  One main launch for all sets (P partials per set), then the
  // usual tree levels over the flat array of nsets * P partials -- P is a power of two, so no level pairs two sets.
  static int sum_sets(Ctx& c, hipStream_t st, int group, const uint8_t* d_pts, const uint64_t* d_off, size_t nsets, size_t max_set,
                      uint8_t* d_out, uint32_t* d_flags) {
    if (nsets == 0) return 0;
    // P partials (lane pairs / lanes) per set, a power of two.  The lane-pair kernel adds the 32 sums of a block itself and
    // leaves ONE partial per block, so a set of a few keys needs one block (P = 32) and no tree; the other kernels write one
    // partial per lane of at least one wave (P = 64).  Workspaces are sized by what the selected kernel writes.
    const bool pairs = group == BGLS_G2 && sum_mode<C>() == 2;
    size_t P = pairs ? 32 : 64;
    while (P < 8192 && P * 4 <= max_set && nsets * P * 2 <= (size_t)131072) P *= 2;     // >= 4 points per partial, <= 2048 waves in all
    const size_t written = pairs ? nsets * (P / 32) : nsets * P;                         // partial sums the main pass leaves
    const size_t blocks = pairs ? nsets * (P / 32) : nsets * (P / 64);
    const size_t JB = kl::jac_bytes<C>(group);
    // bounds of one call: the grid of the main pass and the partials' workspace (the caller can cut a larger job into calls)
    if (blocks > ((size_t)1 << 30) || (written + 1) * JB > ((size_t)8 << 30))
      return fail(BGLS_ERR_ARG, "too many key sets for one call (cut the batch: at most 2^30 blocks / 8 GiB of partial sums)");
    void *ja, *jb;
    int rc;
    Scope sc(c, st, ST_SUM);
    if ((rc = c.get(WS_JAC_A, (written + 1) * JB, &ja))) return rc;
    if ((rc = c.get(WS_JAC_B, (written / 2 + 2) * JB, &jb))) return rc;
    if (group == BGLS_G2 && sum_mode<C>() == 2) kl::sumpairseg_main<C>(st, d_pts, d_off, nsets, (unsigned)P, ja, d_flags);
    else if (group == BGLS_G2 && sum_mode<C>() == 1) kl::sumxseg_main<C>(st, d_pts, d_off, nsets, (unsigned)P, ja, d_flags);
    else kl::sumseg_main<C>(st, group, d_pts, d_off, nsets, (unsigned)P, ja, d_flags);
    void *a = ja, *b = jb;
    size_t p = pairs ? P / 32 : P, cnt = written;
    while (p > 1) {
      if (group == BGLS_G2 && cnt <= 8192) {
        kl::sum_coop<C>(st, a, cnt, b);
        p /= 2; cnt /= 2;
      } else if (cnt > 4096 || p < 64) {
        kl::sum_pair<C>(st, group, a, cnt, b);
        p /= 2; cnt /= 2;
      } else {
        kl::sum_wave<C>(st, group, a, cnt, b);
        p /= 64; cnt /= 64;
      }
      void* t = a;
      a = b;
      b = t;
    }
    kl::jac_to_bytes<C>(st, group, a, nsets, d_out);
    HIPCHK(hipGetLastError());
    return 0;
  }
  // the same sum left in Jacobian form at d_jac (multi-device key sums exchange projective partials);
  // n == 0 gives the point at infinity (all-zero record: Z = 0)
  // d_bytes != nullptr: the caller wants the affine wire bytes as well; where the one-launch tree (k_sumtree.hip) serves the
  // sum it writes them itself and *bytes_done is set (no separate conversion launch)
  static int sum_points_jac(Ctx& c, hipStream_t st, int group, const uint8_t* d_pts, size_t n, void* d_jac, uint32_t* d_flags,
                            int src = 0, bool own_scope = true, uint8_t* d_bytes = nullptr, bool* bytes_done = nullptr) {
    if (n == 0) {
      HIPCHK(hipMemsetAsync(d_jac, 0, kl::jac_bytes<C>(group), st));
      return 0;
    }
    // main pass: at most two waves per SIMD (2048 waves), at least ~4 points per lane; then the per-thread partials
    // are folded 64 at a time
    size_t waves = (n + 255) / 256;
    if (waves > 2048) waves = 2048;
    const bool pairs = group == BGLS_G2 && sum_mode<C>() == 2;
    if (pairs) {                      // one running sum per lane pair, three waves per SIMD resident (3072), >= ~4 keys per pair;
      waves = (n + 127) / 128;        // a block (one wave) adds its 32 sums itself and leaves ONE partial
      static const size_t cap = [] { const char* e = getenv("BGLS_SUM_WAVES"); const long v = e ? atol(e) : 0; return v >= 64 && v <= 8192 ? (size_t)v : (size_t)3072; }();
      if (waves > cap) waves = cap;
    }
    const size_t partials = pairs ? waves : waves * 64;
    void *ja, *jb;
    int rc;
    std::optional<Scope> sc;
    if (own_scope) sc.emplace(c, st, ST_SUM);
    const size_t JB = kl::jac_bytes<C>(group);
    if ((rc = c.get(WS_JAC_A, (partials + 1) * JB, &ja))) return rc;
    if ((rc = c.get(WS_JAC_B, (partials / 2 + 2) * JB, &jb))) return rc;
    std::optional<Scope> scm;
    scm.emplace(c, st, ST_SUM_MAIN);
    if (src == 2 && !pairs) return fail(BGLS_ERR_ARG, "sum-ready records need the lane-pair key-sum kernel");
    if (pairs) kl::sumpair_main<C>(st, src, d_pts, n, (unsigned)(waves * 32), ja, d_flags);                                     // lane pairs, carry-free limbs (rx_jacpair.hpp)
    else if (group == BGLS_G2 && sum_mode<C>() == 1) kl::sumx_main<C>(st, src == 1, d_pts, n, (unsigned)waves, ja, d_flags); // one lane, carry-free limbs (rx_jac.hpp)
    else kl::sum_main<C>(st, group, src == 1, d_pts, n, (unsigned)waves, ja, d_flags);
    scm.reset();
    void *a = ja, *b = jb;
    size_t cnt = partials;
    static const bool one_launch_tree = [] { const char* e = getenv("BGLS_SUMTREE"); return !(e && e[0] == '0'); }();
    while (cnt > 1) {
      if (group == BGLS_G2 && cnt <= 8192 && one_launch_tree) {
        // the whole tree in ONE launch: a wave per pair of leaves climbs by tickets (k_sumtree.hip); the root's wave writes the
        // Jacobian record and, if asked, the affine bytes
        void *store, *tick;
        if ((rc = c.get(WS_TREE_S, std::max((cnt + 64) * JB, kl::sum_tree_store_bytes<C>(cnt)), &store))) return rc;
        if ((rc = c.get(WS_TREE_T, (size_t)(8192 + 64) * 4, &tick))) return rc;
        // The tickets of THIS tree are zeroed in front of every launch (cnt + 64 words: the levels' offsets sum below cnt + 64).  The
        // kernel also leaves them at zero, but a launch that was aborted -- or a second sum submitted on the same context through a
        // user stream while the first was in flight -- would leave ones behind, every later tree would then take its waves for
        // "first arrivals", the root would never be written and a STALE key sum would be verified against.
        HIPCHK(hipMemsetAsync(tick, 0, (cnt + 64) * 4, st));
        kl::sum_tree<C>(st, a, cnt, store, (uint32_t*)tick, d_bytes, d_jac);
        HIPCHK(hipGetLastError());
        if (bytes_done) *bytes_done = d_bytes != nullptr;
        return 0;
      }
      // halving launches while there is parallelism to speak of, then 64 -> 1 per wave with lane shuffles
      if (group == BGLS_G2 && cnt <= 8192) {
        kl::sum_coop<C>(st, a, cnt, b);                    // few additions left: one wave per addition, ~12 us a level
        cnt = (cnt + 1) / 2;
      } else if (cnt > 4096) {
        kl::sum_pair<C>(st, group, a, cnt, b);
        cnt = (cnt + 1) / 2;
      } else {
        kl::sum_wave<C>(st, group, a, cnt, b);
        cnt = (cnt + 63) / 64;
      }
      void* t = a;
      a = b;
      b = t;
    }
    HIPCHK(hipMemcpyAsync(d_jac, a, JB, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    return 0;
  }
};

int flags_to_rc(uint32_t f) {
  if (f & FLAG_ENC) return fail(BGLS_ERR_ENCODING, "non-canonical coordinate or point not on curve");
  if (f & FLAG_SUBGROUP) return fail(BGLS_ERR_ENCODING, "point outside the order-r subgroup");
  if (f & FLAG_DEGENERATE) return fail(BGLS_ERR_ENCODING, "degenerate point step (small-order key)");
  if (f & FLAG_HASH) return fail(BGLS_ERR_HASH, "try-and-increment exhausted");
  return 0;
}

#define DISPATCH(curve, CALL)                                    \
  do {                                                           \
    if ((curve) == BGLS_CURVE_ALTBN128) {                        \
      typedef BN254 CV;                                          \
      return CALL;                                               \
    } else if ((curve) == BGLS_CURVE_BLS12_381) {                \
      typedef BLS381 CV;                                         \
      return CALL;                                               \
    }                                                            \
    return fail(BGLS_ERR_ARG, "unknown curve id");               \
  } while (0)

template <class C>
int verify_aggregate_t(const uint8_t* sig, const uint8_t* keys, const uint8_t* blob, const uint64_t* off, size_t n, int allow_dups) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = n ? off[n] : 0;
  void *d_sig, *d_keys, *d_blob, *d_off, *d_flags, *d_part;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, mv, n, !allow_dups, (uint8_t*)d_part,
                              (uint32_t*)d_flags)))
    return rc;
  return E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int verify_multi_dev_t(Ctx& c, hipStream_t st, const uint8_t* d_sig, const uint8_t* d_keys, size_t n, const uint8_t* d_msg,
                       size_t msg_len, bool submit_only = false, int key_src = 0) {
  typedef Engine<C> E;
  int rc;
  void *d_flags, *d_g2s, *d_g1s, *d_part;
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_TMP2, 2 * E::G2B, &d_g2s))) return rc;
  if ((rc = c.get(WS_G1S, 4 * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  // pairs (H(msg), apk) and (-sig, g2) -- the reference's e(sig, g2) = e(H(msg), apk) (bgls/bgls.go:59-70) as a product that
  // must be 1: one message through the batch hashing path, then the two-pairing product on the cooperative Miller kernel
  // with the (-sig, g2) pair on the pre-computed generator lines.
  // H(m) and -sig do not depend on the key sum: a verification with the machine to itself hashes on the context's side stream
  // while the keys are added (0.3 ms off its latency); with several in flight (throughput mode) the neighbours fill the machine
  // and one stream per verification is what the hardware queues are budgeted for.
  MsgView mv = {d_msg, nullptr, msg_len, msg_len};
  Aff<F1<C>>* g1s = (Aff<F1<C>>*)d_g1s;
  constexpr bool raw = C::CURVE_ID == 1;        // BLS12-381: H(m) before cofactor clearing, the cofactor applied in GT
  const bool fork = !throughput_mode() && n >= 4096;
  hipStream_t hs = fork ? c.side : st;
  SideJoin sj{c.side};
  if (fork) {
    HIPCHK(hipEventRecord(c.ev_fork, st));
    HIPCHK(hipStreamWaitEvent(c.side, c.ev_fork, 0));
    sj.armed = true;
  } else {
    // apk = sum(keys)  (AggregatePoints)
    if ((rc = E::sum_points(c, st, BGLS_G2, d_keys, n, (uint8_t*)d_g2s, (uint32_t*)d_flags, key_src))) return rc;
  }
  if ((rc = E::hash_to_g1(c, hs, mv, 1, g1s, (uint32_t*)d_flags, raw))) return rc;          // H(m)
  kl::g1_parse<C>(hs, d_sig, 1, 1, g1s + 1, (uint32_t*)d_flags);                            // -sig
  if (fork) {
    HIPCHK(hipEventRecord(c.ev_join, c.side));
    if ((rc = E::sum_points(c, st, BGLS_G2, d_keys, n, (uint8_t*)d_g2s, (uint32_t*)d_flags, key_src))) return rc;
    HIPCHK(hipStreamWaitEvent(st, c.ev_join, 0));
    sj.armed = false;
  }
  if ((rc = E::miller(c, st, g1s, (const uint8_t*)d_g2s, 1, g1s + 1, (uint8_t*)d_part, (uint32_t*)d_flags, raw))) return rc;
  if (submit_only) return E::finalize_submit(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
  return E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
}

// KoskVerifyBatchMultiSignature's body (bgls/blsKosk.go:126-133): aggsig = sum(sigs), key_b = sum(set b), then ONE aggregate
// verification over the nsets pairs (key_b, msg_b) -- one Miller launch, one final exponentiation for all the sets.
template <class C>
int verify_multi_batch_dev_t(Ctx& c, hipStream_t st, const uint8_t* d_sigs, const uint8_t* d_keys, const uint64_t* d_key_off, size_t nsets,
                             size_t max_set, MsgView mv, int allow_dups, bool submit_only) {
  typedef Engine<C> E;
  int rc;
  void *d_flags, *d_akeys, *d_sig, *d_part;
  if (nsets >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_SEG_KEYS, (nsets + 1) * E::G2B, &d_akeys))) return rc;
  if ((rc = c.get(WS_TMP2, 2 * E::G2B, &d_sig))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if ((rc = E::sum_sets(c, st, BGLS_G2, d_keys, d_key_off, nsets, max_set, (uint8_t*)d_akeys, (uint32_t*)d_flags))) return rc;   // AggregateKeys x nsets
  if ((rc = E::sum_points(c, st, BGLS_G1, d_sigs, nsets, (uint8_t*)d_sig, (uint32_t*)d_flags))) return rc;                          // AggregateSignatures
  if ((rc = E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_akeys, mv, nsets, !allow_dups, (uint8_t*)d_part, (uint32_t*)d_flags)))
    return rc;
  if (submit_only) return E::finalize_submit(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
  return E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int verify_multi_batch_t(const uint8_t* sigs, const uint8_t* keys, const uint64_t* key_off, size_t nsets, const uint8_t* blob, const uint64_t* off,
                         int allow_dups) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  size_t max_set = 0;
  for (size_t i = 0; i < nsets; ++i) {
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
    if (key_off[i + 1] < key_off[i]) return fail(BGLS_ERR_ARG, "key_off not monotone");
    if (key_off[i + 1] - key_off[i] > max_set) max_set = key_off[i + 1] - key_off[i];
  }
  const size_t nkeys = nsets ? key_off[nsets] - key_off[0] : 0, k0 = nsets ? key_off[0] : 0;
  if (nkeys >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  const size_t blob_len = nsets ? off[nsets] : 0;
  void *d_sigs, *d_keys, *d_blob, *d_off, *d_koff;
  if ((rc = c.get(WS_IN_A, (nsets + 1) * E::G1B, &d_sigs))) return rc;
  if ((rc = c.get(WS_IN_B, (nkeys + 1) * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (nsets + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_SEG_OFF, (nsets + 1) * 8, &d_koff))) return rc;
  std::vector<uint64_t> rel(nsets + 1);
  for (size_t i = 0; i <= nsets; ++i) rel[i] = nsets ? key_off[i] - k0 : 0;
  if (nsets) HIPCHK(hipMemcpyAsync(d_sigs, sigs, nsets * E::G1B, hipMemcpyHostToDevice, st));
  if (nkeys) HIPCHK(hipMemcpyAsync(d_keys, keys + k0 * E::G2B, nkeys * E::G2B, hipMemcpyHostToDevice, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (nsets + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_koff, rel.data(), (nsets + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));                       // rel goes out of scope
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  return verify_multi_batch_dev_t<C>(c, st, (const uint8_t*)d_sigs, (const uint8_t*)d_keys, (const uint64_t*)d_koff, nsets, max_set, mv, allow_dups, false);
}

// AggregatePoints over nsets sets in one pass (host buffers): out = nsets points
template <class C>
int aggregate_sets_t(int group, const uint8_t* pts, const uint64_t* set_off, size_t nsets, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  size_t max_set = 0;
  for (size_t i = 0; i < nsets; ++i) {
    if (set_off[i + 1] < set_off[i]) return fail(BGLS_ERR_ARG, "set_off not monotone");
    if (set_off[i + 1] - set_off[i] > max_set) max_set = set_off[i + 1] - set_off[i];
  }
  if (nsets == 0) return 0;
  const size_t k0 = set_off[0], n = set_off[nsets] - k0;
  if (n >= MAX_BATCH || nsets >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  void *d_pts, *d_off, *d_out, *d_flags;
  if ((rc = c.get(WS_IN_B, (n + 1) * PB, &d_pts))) return rc;
  if ((rc = c.get(WS_SEG_OFF, (nsets + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_SEG_KEYS, (nsets + 1) * PB, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  std::vector<uint64_t> rel(nsets + 1);
  for (size_t i = 0; i <= nsets; ++i) rel[i] = set_off[i] - k0;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (n) HIPCHK(hipMemcpyAsync(d_pts, pts + k0 * PB, n * PB, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, rel.data(), (nsets + 1) * 8, hipMemcpyHostToDevice, st));
  if ((rc = E::sum_sets(c, st, group, (const uint8_t*)d_pts, (const uint64_t*)d_off, nsets, max_set, (uint8_t*)d_out, (uint32_t*)d_flags))) return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, nsets * PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  c.collect();
  return flags_to_rc(f);
}

template <class C>
int verify_multi_t(const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  void *d_sig, *d_keys, *d_msg;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, msg_len, &d_msg))) return rc;
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (msg_len) HIPCHK(hipMemcpyAsync(d_msg, msg, msg_len, hipMemcpyHostToDevice, st));
  return verify_multi_dev_t<C>(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, n, (const uint8_t*)d_msg, msg_len);
}

template <class C>
int pairing_product_t(const uint8_t* g1s, const uint8_t* g2s, size_t n, uint8_t* gt_out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  void *d_g1b, *d_g2b, *d_g1s, *d_flags, *d_part;
  if ((rc = c.get(WS_IN_A, n * E::G1B, &d_g1b))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_g2b))) return rc;
  if ((rc = c.get(WS_G1S, (n + 1) * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (n == 0) {
    memset(gt_out, 0, E::GTB);
    gt_out[E::GTB - 1] = 1;
    return 0;
  }
  HIPCHK(hipMemcpyAsync(d_g1b, g1s, n * E::G1B, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_g2b, g2s, n * E::G2B, hipMemcpyHostToDevice, st));
  kl::g1_parse<C>(st, (const uint8_t*)d_g1b, n, 0, (Aff<F1<C>>*)d_g1s, (uint32_t*)d_flags);
  if ((rc = E::miller(c, st, (const Aff<F1<C>>*)d_g1s, (const uint8_t*)d_g2b, n, nullptr, (uint8_t*)d_part, (uint32_t*)d_flags))) return rc;
  rc = E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, gt_out);
  return rc < 0 ? rc : 0;
}

template <class C>
int hash_to_g1_t(const uint8_t* blob, const uint64_t* off, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = off[n];
  void *d_blob, *d_off, *d_g1s, *d_out, *d_flags;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_G1S, n * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_IN_A, n * E::G1B, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::hash_to_g1(c, st, mv, n, (Aff<F1<C>>*)d_g1s, (uint32_t*)d_flags))) return rc;
  kl::g1_to_bytes<C>(st, (const Aff<F1<C>>*)d_g1s, n, (uint8_t*)d_out);
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * E::G1B, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int aggregate_points_t(int group, const uint8_t* pts, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_in, *d_out, *d_flags;
  if ((rc = c.get(WS_IN_B, n * PB, &d_in))) return rc;
  if ((rc = c.get(WS_OUT, PB, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (n) HIPCHK(hipMemcpyAsync(d_in, pts, n * PB, hipMemcpyHostToDevice, st));
  rc = E::sum_points(c, st, group, (const uint8_t*)d_in, n, (uint8_t*)d_out, (uint32_t*)d_flags);
  if (rc) return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int scale_points_t(int group, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_in, *d_sc, *d_sg, *d_out, *d_flags;
  if ((rc = c.get(WS_IN_B, n * PB, &d_in))) return rc;
  if ((rc = c.get(WS_IN_C, n * 32, &d_sc))) return rc;
  if ((rc = c.get(WS_IN_D, n, &d_sg))) return rc;
  if ((rc = c.get(WS_IN_A, n * PB, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_in, pts, n * PB, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, st));
  if (signs) HIPCHK(hipMemcpyAsync(d_sg, signs, n, hipMemcpyHostToDevice, st));
  const uint8_t* sg = signs ? (const uint8_t*)d_sg : nullptr;
  if (group == BGLS_G1 && C::CURVE_ID == 1 && g1x()) kl::scale_g1x<C>(st, (const uint8_t*)d_in, (const uint8_t*)d_sc, sg, n, (uint8_t*)d_out, (uint32_t*)d_flags, 32);
  else kl::scale<C>(st, group, (const uint8_t*)d_in, (const uint8_t*)d_sc, sg, n, (uint8_t*)d_out, (uint32_t*)d_flags, 32);
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int point_check_t(int group, const uint8_t* a) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_in, *d_flags;
  if ((rc = c.get(WS_IN_B, PB, &d_in))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_in, a, PB, hipMemcpyHostToDevice, st));
  kl::check<C>(st, group, (const uint8_t*)d_in, 1, (uint32_t*)d_flags, nullptr);
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return f ? 0 : 1;
}

template <class C>
int check_points_t(int group, const uint8_t* pts, size_t n, uint8_t* ok_out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_in, *d_ok, *d_flags;
  if ((rc = c.get(WS_IN_B, n * PB, &d_in))) return rc;
  if ((rc = c.get(WS_IN_D, n, &d_ok))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_in, pts, n * PB, hipMemcpyHostToDevice, st));
  kl::check<C>(st, group, (const uint8_t*)d_in, n, (uint32_t*)d_flags, (uint8_t*)d_ok);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(ok_out, d_ok, n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

template <class C>
int generator_t(int group, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void* d_out;
  if ((rc = c.get(WS_OUT, PB, &d_out))) return rc;
  kl::generator<C>(st, group, (uint8_t*)d_out);
  HIPCHK(hipMemcpyAsync(out, d_out, PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

template <class C>
int gt_mul_t(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  void* d_in;
  if ((rc = c.get(WS_IN_A, 2 * E::GTB, &d_in))) return rc;
  HIPCHK(hipMemcpyAsync(d_in, a, E::GTB, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync((uint8_t*)d_in + E::GTB, b, E::GTB, hipMemcpyHostToDevice, st));
  rc = E::finalize(c, st, (const uint8_t*)d_in, 2, 0, nullptr, out);
  return rc < 0 ? rc : 0;
}

template <class C>
int gt_pow_t(const uint8_t* gt, const uint8_t* k_be32, int negate, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  void *d_in, *d_flags;
  if ((rc = c.get(WS_IN_A, 2 * E::GTB + 32, &d_in))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  uint8_t* d = (uint8_t*)d_in;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d, gt, E::GTB, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d + E::GTB, k_be32, 32, hipMemcpyHostToDevice, st));
  kl::gt_pow<C>(st, d, d + E::GTB, negate, d + E::GTB + 32, (uint32_t*)d_flags);
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d + E::GTB + 32, E::GTB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int miller_product_dev_t(const void* d_sig, const void* d_keys, const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n,
                         int check_dups, void* d_partial, void* d_flags, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  MsgView mv = {(const uint8_t*)d_msgs, nullptr, msg_len, msg_stride};
  return E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, mv, n, check_dups, (uint8_t*)d_partial,
                           (uint32_t*)d_flags);
}

// containsDuplicateMessage (bgls/bgls.go:139-150) over device-resident fixed-stride messages: exact byte comparison
int duplicate_scan_dev(const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n, void* d_flags, void* stream) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  MsgView mv = {(const uint8_t*)d_msgs, nullptr, msg_len, msg_stride};
  return Engine<BN254>::dup_scan(c, st, mv, n, (uint32_t*)d_flags);      // curve-independent
}

template <class C>
int final_verify_dev_t(const void* d_partials, size_t count, const void* d_flags, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  return E::finalize(c, st, (const uint8_t*)d_partials, count, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int final_verify_submit_dev_t(const void* d_partials, size_t count, const void* d_flags, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  return E::finalize_submit(c, st, (const uint8_t*)d_partials, count, 1, (const uint32_t*)d_flags, nullptr);
}

template <class C>
int aggregate_points_dev_t(int group, const void* d_pts, size_t n, void* d_out, void* stream) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  void* d_flags;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  rc = E::sum_points(c, st, group, (const uint8_t*)d_pts, n, (uint8_t*)d_out, (uint32_t*)d_flags);
  if (rc) return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

template <class C>
int verify_multi_dev_entry_t(const void* d_sig, const void* d_keys, size_t n, const void* d_msg, size_t msg_len, void* stream,
                             bool submit_only = false) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  return verify_multi_dev_t<C>(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, n, (const uint8_t*)d_msg, msg_len, submit_only);
}

template <class C>
int verify_multi_batch_sub_t(const void* d_sigs, const void* d_keys, const void* d_key_off, size_t nsets, size_t max_set, const void* d_msgs,
                             size_t msg_len, size_t msg_stride, int allow_dups, void* stream, bool submit_only) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  MsgView mv = {(const uint8_t*)d_msgs, nullptr, msg_len, msg_stride};
  return verify_multi_batch_dev_t<C>(c, st, (const uint8_t*)d_sigs, (const uint8_t*)d_keys, (const uint64_t*)d_key_off, nsets, max_set, mv, allow_dups,
                                     submit_only);
}

// ---- hashed aggregation exponents / weighted sums: host flows -------------------------------------------------
// Root digest of BLAKE2Xb (hashes.hpp has the device-side tables; these are the host's own copies).  One sequential
// compression chain over all key bytes -- by construction not parallel -- computed while the keys travel to the device.
namespace host_blake2 {
const uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                        0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
const uint8_t SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
inline uint64_t ror(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
inline void compress(uint64_t h[8], const uint8_t* block, uint64_t t, bool last) {
  uint64_t m[16], v[16];
  memcpy(m, block, 128);                                     // little-endian host
  for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
  v[12] ^= t;
  if (last) v[14] = ~v[14];
#define BGLS_G(a, b, c, d, x, y)                                                    \
  v[a] += v[b] + (x); v[d] = ror(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 24); \
  v[a] += v[b] + (y); v[d] = ror(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 63);
  for (int r = 0; r < 12; ++r) {
    const uint8_t* s = SIGMA[r];
    BGLS_G(0, 4, 8, 12, m[s[0]], m[s[1]]) BGLS_G(1, 5, 9, 13, m[s[2]], m[s[3]])
    BGLS_G(2, 6, 10, 14, m[s[4]], m[s[5]]) BGLS_G(3, 7, 11, 15, m[s[6]], m[s[7]])
    BGLS_G(0, 5, 10, 15, m[s[8]], m[s[9]]) BGLS_G(1, 6, 11, 12, m[s[10]], m[s[11]])
    BGLS_G(2, 7, 8, 13, m[s[12]], m[s[13]]) BGLS_G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
#undef BGLS_G
  for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}
// h <- BLAKE2Xb root of data[0..len) for an XOF of xof_len bytes (x/crypto/blake2b/blake2x.go Reset + Write + finalize)
void xb_root(const uint8_t* data, size_t len, uint32_t xof_len, uint64_t h[8]) {
  for (int i = 0; i < 8; ++i) h[i] = IV[i];
  h[0] ^= 0x01010040ull;
  h[1] ^= (uint64_t)xof_len << 32;
  size_t off = 0;
  while (len - off > 128) {
    compress(h, data + off, (uint64_t)off + 128, false);
    off += 128;
  }
  uint8_t lastb[128];
  memset(lastb, 0, 128);
  if (len > off) memcpy(lastb, data + off, len - off);
  compress(h, lastb, (uint64_t)len, true);
}
}  // namespace host_blake2

// d_t (WS_HAE_T) <- the n 16-byte exponents of hashPubKeysToExponents (blsHAE.go:80-93) for the keys' wire bytes
template <class C>
int hae_exponents_dev(Ctx& c, hipStream_t st, const uint8_t* h_keys, size_t n, void** d_t) {
  typedef Engine<C> E;
  if (n >= (1ull << 28)) return fail(BGLS_ERR_ARG, "XOF length 16 n must fit a uint32 (blsHAE.go:81)");
  int rc;
  void* d_root;
  if ((rc = c.get(WS_HAE_ROOT, 64, &d_root))) return rc;
  if ((rc = c.get(WS_HAE_T, n * 16, d_t))) return rc;
  if (n == 0) return 0;
  uint64_t root[8];
  const uint32_t xof_len = (uint32_t)(16 * n);
  host_blake2::xb_root(h_keys, n * E::G2B, xof_len, root);
  HIPCHK(hipMemcpyAsync(d_root, root, 64, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));                          // root[] is a stack buffer
  kl::blake2x_expand(st, (const uint64_t*)d_root, xof_len, (uint8_t*)*d_t);
  HIPCHK(hipGetLastError());
  return 0;
}

template <class C>
int hae_exponents_t(const uint8_t* keys, size_t n, uint8_t* t_out) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  void* d_t;
  if ((rc = hae_exponents_dev<C>(c, c.stream, keys, n, &d_t))) return rc;
  if (n) HIPCHK(hipMemcpyAsync(t_out, d_t, n * 16, hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipStreamSynchronize(c.stream));
  return 0;
}

// Weighted sums below this many points keep one double-and-add per point (bgls_set_msm_min; tests pin both paths)
std::atomic<size_t> g_msm_min{32};

// d_out (affine bytes) <- sum_i w_i P_i over device-resident points and 16-byte weights, one scalar multiplication per
// point (k_wsum_first) followed by the addition tree
template <class C>
int weighted_sum_naive(Ctx& c, hipStream_t st, int group, const uint8_t* d_pts, const uint8_t* d_w16, const uint8_t* d_signs, size_t n,
                       uint8_t* d_out, uint32_t* d_flags) {
  void *ja, *jb;
  int rc;
  const size_t JB = kl::jac_bytes<C>(group);
  if ((rc = c.get(WS_JAC_A, (n + 1) * JB, &ja))) return rc;
  if ((rc = c.get(WS_JAC_B, (n / 2 + 2) * JB, &jb))) return rc;
  kl::wsum_first<C>(st, group, d_pts, d_w16, d_signs, n, ja, d_flags);
  void *a = ja, *b = jb;
  size_t cnt = n;
  while (cnt > 1) {
    size_t r16 = (cnt + 131071) / 131072;                 // same fan-in rule as Engine::sum_points
    const int R = (int)(r16 < 2 ? 2 : r16 > 16 ? 16 : r16);
    size_t nout = (cnt + R - 1) / R;
    kl::sum_next<C>(st, group, a, cnt, R, b);
    void* t = a;
    a = b;
    b = t;
    cnt = nout;
  }
  kl::jac_to_bytes<C>(st, group, a, 1, d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// The same sum by the bucket method (k_msm.hip): n W mixed additions instead of n (128 doublings + 64 additions).
// Bucket populations are only balanced for weights that look random (hashed exponents do); when the largest bucket is
// far above the mean -- small multiplicities, repeated weights -- the per-point form is the faster one and is used.
template <class C>
int weighted_sum_dev(Ctx& c, hipStream_t st, int group, const uint8_t* d_pts, const uint8_t* d_w16, const uint8_t* d_signs, size_t n,
                     uint8_t* d_out, uint32_t* d_flags) {
  const size_t PTB = group == BGLS_G1 ? Engine<C>::G1B : Engine<C>::G2B;
  if (n == 0) {
    HIPCHK(hipMemsetAsync(d_out, 0, PTB, st));
    return 0;
  }
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  Scope sc(c, st, ST_SUM);
  const kl::MsmPlan p = kl::msm_plan(n);
  if (n < g_msm_min.load() || (uint64_t)n * (uint64_t)p.W >= (1ull << 32))            // list positions are 32-bit
    return weighted_sum_naive<C>(c, st, group, d_pts, d_w16, d_signs, n, d_out, d_flags);
  const size_t JB = kl::jac_bytes<C>(group);
  void *aff, *cnt, *start, *list, *buckets, *tail;
  int rc;
  if ((rc = c.get(WS_MSM_AFF, n * kl::msm_aff_bytes<C>(group), &aff))) return rc;
  if ((rc = c.get(WS_MSM_CNT, ((size_t)p.NB + 2) * 4, &cnt))) return rc;
  if ((rc = c.get(WS_MSM_START, ((size_t)p.NB + 1) * 4, &start))) return rc;
  if ((rc = c.get(WS_MSM_LIST, n * (size_t)p.W * 4, &list))) return rc;
  if ((rc = c.get(WS_JAC_A, (size_t)p.NB * p.S * JB, &buckets))) return rc;
  const size_t half = p.S > 1 ? (size_t)p.NB * p.S / 2 : 0;                  // second buffer of the partials' pairwise folds
  if ((rc = c.get(WS_JAC_B, (half + kl::msm_tail_points(p)) * JB, &tail))) return rc;
  uint32_t* d_meta = (uint32_t*)cnt + p.NB;
  HIPCHK(hipMemsetAsync(cnt, 0, ((size_t)p.NB + 2) * 4, st));
  kl::msm_parse<C>(st, group, d_pts, d_w16, d_signs, n, p, aff, (uint32_t*)cnt, d_flags);
  kl::msm_scan(st, (uint32_t*)cnt, p.NB, (uint32_t*)start, d_meta);
  HIPCHK(hipGetLastError());
  uint32_t meta[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(meta, d_meta, 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const size_t mean = n >> p.c;
  if (meta[0] > (mean * 8 > 64 ? mean * 8 : 64)) return weighted_sum_naive<C>(c, st, group, d_pts, d_w16, d_signs, n, d_out, d_flags);
  void* res = nullptr;
  kl::msm_scatter<C>(st, group, aff, d_w16, n, p, (uint32_t*)cnt, (uint32_t*)list);
  kl::msm_buckets<C>(st, group, aff, (const uint32_t*)list, (const uint32_t*)start, p, buckets);
  void *a = buckets, *b = tail;
  for (size_t cntp = (size_t)p.NB * p.S; cntp > p.NB; cntp /= 2) {            // S partials per bucket -> one, halving
    kl::sum_pair<C>(st, group, a, cntp, b);
    std::swap(a, b);
  }
  kl::msm_tail<C>(st, group, a, p, (uint8_t*)tail + half * JB, &res);
  kl::jac_to_bytes<C>(st, group, res, 1, d_out);
  HIPCHK(hipGetLastError());
  return 0;
}

// verify_multi with apk = sum w_i pk_i: VerifyMultiSignatureWithHAE (blsHAE.go:56-58; weights hashed from the keys) when
// mult == nullptr, the core of KoskVerifyMultiSignatureWithMultiplicity (blsKosk.go:137-150) otherwise.
template <class C>
int verify_multi_weighted_t(const uint8_t* sig, const uint8_t* keys, const int64_t* mult, size_t n, const uint8_t* msg, size_t msg_len) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  void *d_sig, *d_keys, *d_msg, *d_apk, *d_fl2, *d_t = nullptr, *d_sg = nullptr;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, msg_len, &d_msg))) return rc;
  if ((rc = c.get(WS_HAE_APK, E::G2B, &d_apk))) return rc;
  if ((rc = c.get(WS_FLAGS2, 16, &d_fl2))) return rc;
  HIPCHK(hipMemsetAsync(d_fl2, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (msg_len) HIPCHK(hipMemcpyAsync(d_msg, msg, msg_len, hipMemcpyHostToDevice, st));
  if (!mult) {
    if ((rc = hae_exponents_dev<C>(c, st, keys, n, &d_t))) return rc;
  } else {
    std::vector<uint8_t> w(n * 16, 0), sg(n, 0);
    for (size_t i = 0; i < n; ++i) {
      const int64_t m = mult[i];
      uint64_t mag = m < 0 ? (uint64_t)0 - (uint64_t)m : (uint64_t)m;
      sg[i] = m < 0 ? 1 : 0;
      for (int b = 0; b < 8; ++b) w[i * 16 + 15 - b] = (uint8_t)(mag >> (8 * b));
    }
    if ((rc = c.get(WS_HAE_T, n * 16, &d_t))) return rc;
    if ((rc = c.get(WS_HAE_SIGN, n, &d_sg))) return rc;
    if (n) {
      HIPCHK(hipMemcpyAsync(d_t, w.data(), n * 16, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_sg, sg.data(), n, hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));                      // w, sg are locals
    }
  }
  if ((rc = weighted_sum_dev<C>(c, st, BGLS_G2, (const uint8_t*)d_keys, (const uint8_t*)d_t, (const uint8_t*)d_sg, n,
                                                   (uint8_t*)d_apk, (uint32_t*)d_fl2)))
    return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(&f, d_fl2, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if ((rc = flags_to_rc(f))) return rc;
  return verify_multi_dev_t<C>(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_apk, 1, (const uint8_t*)d_msg, msg_len);
}

// getAggregatePubKey over device-resident points and weights (blsHAE.go:74-77): d_out <- sum_i w_i P_i as affine bytes
template <class C>
int weighted_sum_dev_t(int group, const void* d_pts, const void* d_w16, size_t n, void* d_out, void* stream) {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  void* d_flags;
  if ((rc = c.get(WS_FLAGS2, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if ((rc = weighted_sum_dev<C>(c, st, group, (const uint8_t*)d_pts, (const uint8_t*)d_w16, nullptr, n, (uint8_t*)d_out, (uint32_t*)d_flags)))
    return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  c.collect();
  return flags_to_rc(f);
}

// VerifyAggregateSignatureWithHAE (blsHAE.go:49-53): keys scaled by their exponents, then verifyAggSig with duplicates allowed
template <class C>
int verify_aggregate_hae_t(const uint8_t* sig, const uint8_t* keys, const uint8_t* blob, const uint64_t* off, size_t n) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = n ? off[n] : 0;
  void *d_sig, *d_keys, *d_blob, *d_off, *d_flags, *d_part, *d_t;
  if ((rc = c.get(WS_IN_A, E::G1B, &d_sig))) return rc;
  if ((rc = c.get(WS_IN_B, n * E::G2B, &d_keys))) return rc;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  if ((rc = c.get(WS_PART, E::GTB, &d_part))) return rc;
  HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
  if (n) HIPCHK(hipMemcpyAsync(d_keys, keys, n * E::G2B, hipMemcpyHostToDevice, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if ((rc = hae_exponents_dev<C>(c, st, keys, n, &d_t))) return rc;
  // e(H(m_i), t_i pk_i) = e(t_i H(m_i), pk_i): the exponent goes to the G1 side (a third of the G2 work, same GT value)
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_keys, mv, n, 0, (uint8_t*)d_part, (uint32_t*)d_flags,
                              (const uint8_t*)d_t)))
    return rc;
  return E::finalize(c, st, (const uint8_t*)d_part, 1, 1, (const uint32_t*)d_flags, nullptr);
}

// AggregateSignaturesWithHAE (blsHAE.go:39-46): sum_i t_i sigma_i
template <class C>
int aggregate_signatures_hae_t(const uint8_t* sigs, const uint8_t* keys, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  void *d_sigs, *d_out, *d_flags, *d_t;
  if ((rc = c.get(WS_IN_B, n * E::G1B, &d_sigs))) return rc;
  if ((rc = c.get(WS_OUT, E::G1B, &d_out))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (n) HIPCHK(hipMemcpyAsync(d_sigs, sigs, n * E::G1B, hipMemcpyHostToDevice, st));
  if ((rc = hae_exponents_dev<C>(c, st, keys, n, &d_t))) return rc;
  if ((rc = weighted_sum_dev<C>(c, st, BGLS_G1, (const uint8_t*)d_sigs, (const uint8_t*)d_t, nullptr, n, (uint8_t*)d_out,
                                                   (uint32_t*)d_flags)))
    return rc;
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, E::G1B, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}

// Marshal / Unmarshal* compressed branch over a batch.  alt-bn128: the reference's own 32 / 64-byte forms
// (curves/altbn128.go:81-89,203-221,296-376).  BLS12-381: 48 / 96 bytes in the ebfull/pairing layout the reference names as
// its target (curves/bls12_381.go:54-62,115-123,242-264; wire.hpp) -- unpinned against the un-vendored dis2/bls12.
int wire_points(int curve, int group, bool compress, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
  if (curve != BGLS_CURVE_ALTBN128 && curve != BGLS_CURVE_BLS12_381) return fail(BGLS_ERR_ARG, "unknown curve id");
  const bool bls = curve == BGLS_CURVE_BLS12_381;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  const size_t CB = (bls ? 48 : 32) * (group == BGLS_G1 ? 1 : 2), UB = 2 * CB;
  const size_t in_b = compress ? UB : CB, out_b = compress ? CB : UB;
  void *d_in, *d_out, *d_ok, *d_flags;
  if ((rc = c.get(WS_IN_B, n * in_b, &d_in))) return rc;
  if ((rc = c.get(WS_IN_A, n * out_b, &d_out))) return rc;
  if ((rc = c.get(WS_IN_D, n, &d_ok))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  HIPCHK(hipMemcpyAsync(d_in, in, n * in_b, hipMemcpyHostToDevice, st));
  {
    Scope sc(c, st, ST_SUM);
    if (compress) {
      if (bls) kl::compress_bls(st, group, (const uint8_t*)d_in, n, (uint8_t*)d_out, (uint32_t*)d_flags);
      else kl::compress_bn(st, group, (const uint8_t*)d_in, n, (uint8_t*)d_out, (uint32_t*)d_flags);
    } else {
      if (bls) kl::decompress_bls(st, group, (const uint8_t*)d_in, n, (uint8_t*)d_out, (uint8_t*)d_ok);
      else kl::decompress_bn(st, group, (const uint8_t*)d_in, n, (uint8_t*)d_out, (uint8_t*)d_ok);
    }
  }
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * out_b, hipMemcpyDeviceToHost, st));
  if (!compress) HIPCHK(hipMemcpyAsync(ok, d_ok, n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  c.collect();
  return flags_to_rc(f);
}

// window multiples of a generator for this device (DeviceTables), built on first use
template <class C>
int fixed_base_table(Ctx& c, int group, const void** out) {
  DeviceTables& t = tables_of(c.device);
  std::lock_guard<std::mutex> lk(t.mu);
  void*& slot = t.fixed_base[C::CURVE_ID][group - 1];
  if (!slot) {
    void* tab = nullptr;
    hipStream_t bs = nullptr;
    HIPCHK(hipMalloc(&tab, kl::fb_table_bytes<C>(group)));
    hipError_t e = hipStreamCreate(&bs);
    if (e == hipSuccess) {
      kl::fb_build<C>(bs, group, tab);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(bs);
    }
    if (bs) (void)hipStreamDestroy(bs);
    if (e != hipSuccess) {
      (void)hipFree(tab);
      return fail(BGLS_ERR_HIP, "building the fixed-base table", e);
    }
    slot = tab;
  }
  *out = slot;
  return 0;
}

// LoadPublicKey over a batch (bgls/bgls.go:40-43): out[i] = sk_i * g2 (group = BGLS_G2) or sk_i * g1, one mixed addition
// per scalar byte from the resident table of window multiples
template <class C>
int scale_generator_t(int group, const uint8_t* sks, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  const size_t PB = group == BGLS_G1 ? E::G1B : E::G2B;
  void *d_sc, *d_out;
  const void* tab;
  if ((rc = fixed_base_table<C>(c, group, &tab))) return rc;
  if ((rc = c.get(WS_IN_C, n * 32, &d_sc))) return rc;
  if ((rc = c.get(WS_IN_A, n * PB, &d_out))) return rc;
  HIPCHK(hipMemcpyAsync(d_sc, sks, n * 32, hipMemcpyHostToDevice, st));
  kl::fb_scale<C>(st, group, tab, (const uint8_t*)d_sc, n, (uint8_t*)d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d_out, n * PB, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return 0;
}

// Sign over a batch (bgls/bgls.go:46-56): out[i] = sk_i * HashToG1(msg_i); the hash points never leave the device
template <class C>
int sign_batch_t(const uint8_t* sks, const uint8_t* blob, const uint64_t* off, size_t n, uint8_t* out) {
  typedef Engine<C> E;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  if (n == 0) return 0;
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  const size_t blob_len = off[n];
  void *d_blob, *d_off, *d_g1s, *d_out, *d_flags, *d_sc;
  if ((rc = c.get(WS_IN_C, blob_len, &d_blob))) return rc;
  if ((rc = c.get(WS_IN_D, (n + 1) * 8, &d_off))) return rc;
  if ((rc = c.get(WS_G1S, n * sizeof(Aff<F1<C>>), &d_g1s))) return rc;
  if ((rc = c.get(WS_IN_A, n * E::G1B, &d_out))) return rc;
  if ((rc = c.get(WS_IN_B, n * 32, &d_sc))) return rc;
  if ((rc = c.get(WS_FLAGS, 16, &d_flags))) return rc;
  HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
  if (blob_len) HIPCHK(hipMemcpyAsync(d_blob, blob, blob_len, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_off, off, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_sc, sks, n * 32, hipMemcpyHostToDevice, st));
  MsgView mv = {(const uint8_t*)d_blob, (const uint64_t*)d_off, 0, 0};
  if ((rc = E::hash_to_g1(c, st, mv, n, (Aff<F1<C>>*)d_g1s, (uint32_t*)d_flags))) return rc;
  if (C::CURVE_ID == 1 && g1x()) kl::scale_aff_g1x<C>(st, (const Aff<F1<C>>*)d_g1s, (const uint8_t*)d_sc, n, (uint8_t*)d_out);
  else kl::scale_aff<C>(st, BGLS_G1, (const Aff<F1<C>>*)d_g1s, (const uint8_t*)d_sc, n, (uint8_t*)d_out);
  HIPCHK(hipGetLastError());
  uint32_t f = 0;
  HIPCHK(hipMemcpyAsync(out, d_out, n * E::G1B, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return flags_to_rc(f);
}


// ======================================================================= key sets and multi-device verification
// RCCL is loaded at run time (dlopen) so that the library has no link-time dependency on it: the exchange of the
// per-device partials falls back to peer copies whenever RCCL is missing, a device id repeats, or a call fails.
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false;
  Rccl() {
    if (const char* e = getenv("BGLS_NO_RCCL")) { if (e[0] == '1') return; }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
    if (!so) return;
    CommInitAll = (decltype(CommInitAll))dlsym(so, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(so, "ncclAllGather");
    GroupStart = (decltype(GroupStart))dlsym(so, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(so, "ncclGroupEnd");
    ok = CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
  }
};
Rccl& rccl() {
  static Rccl r;
  return r;
}

struct KeyShard {
  int device = 0;
  size_t lo = 0, hi = 0;
  void* d_wire = nullptr;    // (hi - lo) wire-format keys
  void* d_mont = nullptr;    // the same keys as Aff<F2<C>> (Montgomery form): prepared lines, weighted sums, the one-lane key sums
  void* d_sumr = nullptr;    // the same keys as sum-ready records (k_g2_sumready): what the lane-pair key sum reads
  void* d_rec = nullptr;     // this shard's exchange record: GT partial + status word (send buffer)
  void* d_all = nullptr;     // every shard's record (receive buffer)
  // prepared sets (BGLS_KEYS_PREPARE, prepared.hpp): normalised line ratios of every key and step, [step][n_pad] rows
  void* d_prep = nullptr;
  void* d_kinf = nullptr;    // n_pad bytes: 1 = key at infinity / padding
  size_t n_pad = 0;          // hi - lo rounded up to whole fold groups for every NG in use
};
struct KeySet {
  int curve = 0;
  size_t n = 0;
  std::vector<KeyShard> shards;
  bool prepared = false;
  std::vector<ncclComm_t> comms;     // one per shard when the RCCL exchange is usable, else empty
  std::mutex mu;                     // one verification at a time per key set (the shards' buffers are part of it)
  int base_ctx = 0;                  // context of shard 0 in the verification in flight (the caller's bgls_select_context)
  ~KeySet() {
    for (auto cm : comms) if (cm) (void)rccl().CommDestroy(cm);
    for (auto& sh : shards) {
      if (hipSetDevice(sh.device) != hipSuccess) continue;
      for (void* q : {sh.d_wire, sh.d_mont, sh.d_sumr, sh.d_rec, sh.d_all, sh.d_prep, sh.d_kinf}) if (q) (void)hipFree(q);
    }
  }
};
std::mutex g_keys_mu;
std::unordered_map<uint64_t, std::shared_ptr<KeySet>> g_keys;
uint64_t g_keys_next = 1;
thread_local int g_last_exchange = 0;

std::shared_ptr<KeySet> keyset(bgls_keys_t h) {
  std::lock_guard<std::mutex> lk(g_keys_mu);
  auto it = g_keys.find(h);
  return it == g_keys.end() ? nullptr : it->second;
}

constexpr size_t REC_PAD = 16;     // status word + padding behind the GT bytes of an exchange record
constexpr size_t PREP_PAD = 240;   // prepared sets are padded to whole waves of 10 groups for NG = 6, 12 and 24

template <class C>
int keys_upload_t(const uint8_t* keys, size_t n, const int* devices, int n_devices, unsigned flags, bgls_keys_t* out) {
  typedef Engine<C> E;
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "key set too large (n must be below 2^30)");
  auto ks = std::make_shared<KeySet>();
  ks->curve = C::CURVE_ID;
  ks->n = n;
  ks->prepared = (flags & BGLS_KEYS_PREPARE) != 0;
  const size_t REC = E::GTB + REC_PAD;
  bool distinct = true;
  for (int s = 0; s < n_devices; ++s) {
    KeyShard sh;
    sh.device = devices ? devices[s] : s;
    if (sh.device < 0 || sh.device >= MAX_DEVICES) return fail(BGLS_ERR_NO_DEVICE, "device index out of range");
    for (int t = 0; t < s; ++t) distinct = distinct && ks->shards[t].device != sh.device;
    sh.lo = n * (size_t)s / n_devices;
    sh.hi = n * (size_t)(s + 1) / n_devices;
    ks->shards.push_back(sh);
  }
  const int prev_dev = g_dev, prev_sel = g_sel;
  int rc = 0;
  for (int s = 0; s < n_devices && rc == 0; ++s) {
    KeyShard& sh = ks->shards[s];
    const size_t cnt = sh.hi - sh.lo;
    g_dev = sh.device;
    g_sel = 0;
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    rc = [&]() -> int {
      int r;
      if ((r = c.enter())) return r;
      HIPCHK(hipMalloc(&sh.d_wire, cnt ? cnt * E::G2B : 16));
      HIPCHK(hipMalloc(&sh.d_mont, cnt ? cnt * kl::g2_parsed_bytes<C>() : 16));
      HIPCHK(hipMalloc(&sh.d_sumr, cnt ? cnt * kl::g2_sumready_bytes<C>() : 16));
      HIPCHK(hipMalloc(&sh.d_rec, REC));
      HIPCHK(hipMalloc(&sh.d_all, REC * n_devices));
      void* d_flags;
      if ((r = c.get(WS_FLAGS, 16, &d_flags))) return r;
      HIPCHK(hipMemsetAsync(d_flags, 0, 4, c.stream));
      if (cnt) {
        HIPCHK(hipMemcpyAsync(sh.d_wire, keys + sh.lo * E::G2B, cnt * E::G2B, hipMemcpyHostToDevice, c.stream));
        kl::g2_parse<C>(c.stream, (const uint8_t*)sh.d_wire, cnt, (flags & BGLS_KEYS_CHECK) ? 1 : 0, sh.d_mont, (uint32_t*)d_flags);
        kl::g2_sumready<C>(c.stream, sh.d_mont, cnt, sh.d_sumr);
        HIPCHK(hipGetLastError());
      }
      uint32_t f = 0;
      HIPCHK(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, c.stream));
      HIPCHK(hipStreamSynchronize(c.stream));
      if (f & FLAG_ENC) return fail(BGLS_ERR_ENCODING, "key set: non-canonical coordinate or key not on the twist");
      if (f & FLAG_SUBGROUP) return fail(BGLS_ERR_ENCODING, "key set: key outside the order-r subgroup");
      if (flags & BGLS_KEYS_PREPARE) {
        // line ratios of every key (k_prepare), in chunks so that the scratch stays a few GB
        const kl::PrepSizes ps = kl::prep_sizes<C>();
        sh.n_pad = (cnt + PREP_PAD - 1) / PREP_PAD * PREP_PAD;
        if (sh.n_pad == 0) sh.n_pad = PREP_PAD;
        HIPCHK(hipMalloc(&sh.d_prep, sh.n_pad * ps.line_bytes_per_key));
        HIPCHK(hipMalloc(&sh.d_kinf, sh.n_pad));
        const size_t chunk = (size_t)1 << 17;
        void* tmp = nullptr;
        HIPCHK(hipMalloc(&tmp, (sh.n_pad < chunk ? sh.n_pad : chunk) * ps.tmp_bytes_per_key));
        HIPCHK(hipMemsetAsync(d_flags, 0, 4, c.stream));
        for (size_t i0 = 0; i0 < sh.n_pad; i0 += chunk) {
          const size_t count = sh.n_pad - i0 < chunk ? sh.n_pad - i0 : chunk;
          kl::prepare_keys<C>(c.stream, sh.d_mont, cnt, sh.n_pad, i0, count, (uint32_t*)sh.d_prep, (uint8_t*)sh.d_kinf, (uint32_t*)tmp, (uint32_t*)d_flags);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, c.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(BGLS_ERR_HIP, "preparing the key set", e);
        if (f & FLAG_DEGENERATE) return fail(BGLS_ERR_ENCODING, "key set: a key has a degenerate Miller step and cannot be prepared (upload it without BGLS_KEYS_PREPARE)");
      }
      return 0;
    }();
  }
  g_dev = prev_dev;
  g_sel = prev_sel;
  if (rc) return rc;                  // ~KeySet frees what was allocated
  if (n_devices > 1 && distinct && rccl().ok) {
    std::vector<int> devs;
    for (auto& sh : ks->shards) devs.push_back(sh.device);
    ks->comms.assign(n_devices, nullptr);
    if (rccl().CommInitAll(ks->comms.data(), n_devices, devs.data()) != ncclSuccess) ks->comms.clear();
  }
  std::lock_guard<std::mutex> lk(g_keys_mu);
  const uint64_t h = g_keys_next++;
  g_keys[h] = ks;
  *out = h;
  return 0;
}

// Gather every shard's exchange record on shard 0 (sh[0].d_all, shard order).  Each shard's record is complete on its
// context-`s` stream.  RCCL: one all-gather enqueued on every device's stream; else peer / device copies.
template <class C>
int exchange_records(KeySet& ks) {
  typedef Engine<C> E;
  const size_t REC = E::GTB + REC_PAD;
  const int S = (int)ks.shards.size();
  g_last_exchange = S > 1 ? 1 : 0;
  if (!ks.comms.empty()) {
    bool good = rccl().GroupStart() == ncclSuccess;
    for (int s = 0; s < S && good; ++s) {
      KeyShard& sh = ks.shards[s];
      good = hipSetDevice(sh.device) == hipSuccess &&
             rccl().AllGather(sh.d_rec, sh.d_all, REC, ncclUint8, ks.comms[s], ctx_of(sh.device, (ks.base_ctx + s) % NCTX).stream) == ncclSuccess;
    }
    good = (rccl().GroupEnd() == ncclSuccess) && good;
    if (good) {
      g_last_exchange = 2;
      return 0;
    }
  }
  KeyShard& root = ks.shards[0];
  for (int s = 0; s < S; ++s) {
    KeyShard& sh = ks.shards[s];
    hipStream_t st = ctx_of(sh.device, (ks.base_ctx + s) % NCTX).stream;
    HIPCHK(hipSetDevice(sh.device));
    uint8_t* dst = (uint8_t*)root.d_all + (size_t)s * REC;
    if (sh.device == root.device) HIPCHK(hipMemcpyAsync(dst, sh.d_rec, REC, hipMemcpyDeviceToDevice, st));
    else HIPCHK(hipMemcpyPeerAsync(dst, root.device, sh.d_rec, sh.device, REC, st));
    if (s) HIPCHK(hipStreamSynchronize(st));         // shard 0 continues on its own stream
  }
  return 0;
}

// runs fn(shard index) on one host thread per shard, each bound to its shard's device and to context `s`
template <class Fn>
int for_each_shard(KeySet& ks, Fn&& fn) {
  const int S = (int)ks.shards.size();
  std::vector<int> rcs(S, 0);
  std::vector<std::string> errs(S);
  const int base = g_sel;                 // the context the caller selected (bgls_select_context): shard s takes (base + s) mod NCTX,
  ks.base_ctx = base;                     // so a one-shard key set runs on the caller's context and threads on distinct contexts do not serialise
  auto body = [&](int s) noexcept {        // runs on a thread of its own: nothing may leave it
    g_dev = ks.shards[s].device;
    g_sel = (base + s) % NCTX;
    try {
      rcs[s] = fn(s);
    } catch (...) {
      rcs[s] = abi_catch();
    }
    try {
      if (rcs[s] < 0) errs[s] = g_err;
    } catch (...) {}
  };
  if (S == 1) {
    const int pd = g_dev, ps = g_sel;
    body(0);
    g_dev = pd;
    g_sel = ps;
  } else {
    std::vector<std::thread> th;
    th.reserve(S);
    int started = 0;
    try {
      for (; started < S; ++started) th.emplace_back(body, started);
    } catch (...) {                        // a thread could not be created: the shards without one fail, the others are joined below
      for (int s = started; s < S; ++s) rcs[s] = fail(BGLS_ERR_HIP, "could not start a shard's host thread");
    }
    for (auto& t : th) t.join();
  }
  for (int s = 0; s < S; ++s)
    if (rcs[s] < 0) { g_err = errs[s]; return rcs[s]; }
  return 0;
}

int merged_flags_rc(uint32_t f) {
  if (f & FLAG_ENC) return fail(BGLS_ERR_ENCODING, "non-canonical coordinate or point not on curve");
  if (f & FLAG_SUBGROUP) return fail(BGLS_ERR_ENCODING, "point outside the order-r subgroup");
  if (f & FLAG_DEGENERATE) return fail(BGLS_ERR_ENCODING, "degenerate point step (small-order key)");
  if (f & FLAG_HASH) return fail(BGLS_ERR_HASH, "try-and-increment exhausted");
  return 0;
}

template <class C>
int verify_aggregate_h_t(KeySet& ks, const uint8_t* sig, const uint8_t* blob, const uint64_t* off, size_t n, int allow_dups, uint8_t* gt_out) {
  typedef Engine<C> E;
  if (n != ks.n) return fail(BGLS_ERR_ARG, "message count differs from the key set's size");
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return fail(BGLS_ERR_ARG, "msg_off not monotone");
  std::lock_guard<std::mutex> lk_set(ks.mu);
  const size_t REC = E::GTB + REC_PAD;
  const int S = (int)ks.shards.size();
  int rc = for_each_shard(ks, [&](int s) -> int {
    KeyShard& sh = ks.shards[s];
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    int r;
    if ((r = c.enter())) return r;
    hipStream_t st = c.stream;
    const size_t cnt = sh.hi - sh.lo;
    // shard 0 holds ALL messages: the duplicate rule is a property of the whole list (two equal messages may sit in
    // different shards); the other shards upload their own range only
    const size_t mlo = s == 0 ? 0 : sh.lo, mhi = s == 0 ? n : sh.hi;
    const size_t bytes = off[mhi] - off[mlo];
    void *d_sig, *d_blob, *d_off, *d_flags;
    if ((r = c.get(WS_IN_A, E::G1B, &d_sig))) return r;
    if ((r = c.get(WS_IN_C, bytes, &d_blob))) return r;
    if ((r = c.get(WS_IN_D, (mhi - mlo + 1) * 8, &d_off))) return r;
    if ((r = c.get(WS_FLAGS, 16, &d_flags))) return r;
    HIPCHK(hipMemsetAsync(d_flags, 0, 4, st));
    if (s == 0) HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
    if (bytes) HIPCHK(hipMemcpyAsync(d_blob, blob + off[mlo], bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_off, off + mlo, (mhi - mlo + 1) * 8, hipMemcpyHostToDevice, st));
    // offsets keep their global values: the view's base is shifted instead (never dereferenced below d_blob)
    MsgView all = {(const uint8_t*)d_blob - off[mlo], (const uint64_t*)d_off, 0, 0};
    if (s == 0 && !allow_dups && (r = E::dup_scan(c, st, all, n, (uint32_t*)d_flags))) return r;
    MsgView mine = all;
    if (s == 0) mine.off = (const uint64_t*)d_off + sh.lo;      // sh.lo == 0; kept for clarity
    if (ks.prepared) {
      if ((r = E::miller_product_prepared(c, st, s == 0 ? (const uint8_t*)d_sig : nullptr, (const uint32_t*)sh.d_prep, (const uint8_t*)sh.d_kinf, sh.n_pad,
                                          mine, cnt, (uint8_t*)sh.d_rec, (uint32_t*)d_flags)))
        return r;
    } else if ((r = E::miller_product(c, st, s == 0 ? (const uint8_t*)d_sig : nullptr, (const uint8_t*)sh.d_wire, mine, cnt, 0,
                                      (uint8_t*)sh.d_rec, (uint32_t*)d_flags)))
      return r;
    HIPCHK(hipMemcpyAsync((uint8_t*)sh.d_rec + E::GTB, d_flags, 4, hipMemcpyDeviceToDevice, st));
    if (s) HIPCHK(hipStreamSynchronize(st));       // record complete before the exchange reads it (shard 0: stream order)
    return 0;
  });
  if (rc) return rc;
  if ((rc = exchange_records<C>(ks))) return rc;
  // shard 0: product of the S partials, one final exponentiation, compare with 1; status words OR-ed on the host
  KeyShard& root = ks.shards[0];
  const int pd = g_dev, ps = g_sel;
  g_dev = root.device;
  g_sel = ks.base_ctx % NCTX;             // shard 0's context
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  rc = [&]() -> int {
    int r;
    if ((r = c.enter())) return r;
    void* d_parts;
    if ((r = c.get(WS_HAE_KEYS, (size_t)S * E::GTB, &d_parts))) return r;
    std::vector<uint8_t> recs((size_t)S * REC);
    HIPCHK(hipMemcpyAsync(recs.data(), root.d_all, recs.size(), hipMemcpyDeviceToHost, c.stream));
    for (int s = 0; s < S; ++s)
      HIPCHK(hipMemcpyAsync((uint8_t*)d_parts + (size_t)s * E::GTB, (uint8_t*)root.d_all + (size_t)s * REC, E::GTB, hipMemcpyDeviceToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    uint32_t f = 0;
    for (int s = 0; s < S; ++s) {
      uint32_t w;
      memcpy(&w, recs.data() + (size_t)s * REC + E::GTB, 4);
      f |= w;
    }
    if ((r = merged_flags_rc(f))) return r;
    r = E::finalize(c, c.stream, (const uint8_t*)d_parts, (size_t)S, 1, nullptr, gt_out);
    if (r < 0) return r;
    return (f & FLAG_DUP) ? 0 : r;
  }();
  g_dev = pd;
  g_sel = ps;
  return rc;
}

template <class C>
int verify_multi_h_t(KeySet& ks, const uint8_t* sig, const uint8_t* msg, size_t msg_len) {
  typedef Engine<C> E;
  std::lock_guard<std::mutex> lk_set(ks.mu);
  const int S = (int)ks.shards.size();
  const size_t JB = kl::jac_bytes<C>(BGLS_G2);
  const size_t REC = E::GTB + REC_PAD;
  static_assert(E::GTB >= 3 * 2 * C::L * 4, "a projective G2 partial fits an exchange record");
  // per-device partial key sums (projective): the exchange record carries the Jacobian point instead of a GT partial
  int rc = for_each_shard(ks, [&](int s) -> int {
    KeyShard& sh = ks.shards[s];
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    int r;
    if ((r = c.enter())) return r;
    void* d_flags;
    if ((r = c.get(WS_FLAGS, 16, &d_flags))) return r;
    HIPCHK(hipMemsetAsync(d_flags, 0, 4, c.stream));
    HIPCHK(hipMemsetAsync(sh.d_rec, 0, REC, c.stream));
    const bool ready = sum_mode<C>() == 2;       // the lane-pair kernel reads the sum-ready records, the others the Montgomery points
    if ((r = E::sum_points_jac(c, c.stream, BGLS_G2, (const uint8_t*)(ready ? sh.d_sumr : sh.d_mont), sh.hi - sh.lo, sh.d_rec, (uint32_t*)d_flags, ready ? 2 : 1))) return r;
    if (s) HIPCHK(hipStreamSynchronize(c.stream));
    return 0;
  });
  if (rc) return rc;
  if ((rc = exchange_records<C>(ks))) return rc;
  KeyShard& root = ks.shards[0];
  const int pd = g_dev, ps = g_sel;
  g_dev = root.device;
  g_sel = ks.base_ctx % NCTX;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  rc = [&]() -> int {
    int r;
    if ((r = c.enter())) return r;
    hipStream_t st = c.stream;
    void *d_jacs, *d_tmp, *d_apk, *d_sig, *d_msg;
    if ((r = c.get(WS_JAC_A, (size_t)(S + 64) * JB, &d_jacs))) return r;
    if ((r = c.get(WS_JAC_B, 4 * JB, &d_tmp))) return r;
    if ((r = c.get(WS_HAE_APK, E::G2B, &d_apk))) return r;
    if ((r = c.get(WS_IN_A, E::G1B, &d_sig))) return r;
    if ((r = c.get(WS_IN_C, msg_len, &d_msg))) return r;
    for (int s = 0; s < S; ++s)
      HIPCHK(hipMemcpyAsync((uint8_t*)d_jacs + (size_t)s * JB, (uint8_t*)root.d_all + (size_t)s * REC, JB, hipMemcpyDeviceToDevice, st));
    void* cur = d_jacs;
    if (S > 1) {
      kl::sum_wave<C>(st, BGLS_G2, d_jacs, (size_t)S, d_tmp);       // S <= 16 partials: one wave
      cur = d_tmp;
    }
    kl::jac_to_bytes<C>(st, BGLS_G2, cur, 1, (uint8_t*)d_apk);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(d_sig, sig, E::G1B, hipMemcpyHostToDevice, st));
    if (msg_len) HIPCHK(hipMemcpyAsync(d_msg, msg, msg_len, hipMemcpyHostToDevice, st));
    return verify_multi_dev_t<C>(c, st, (const uint8_t*)d_sig, (const uint8_t*)d_apk, 1, (const uint8_t*)d_msg, msg_len);
  }();
  g_dev = pd;
  g_sel = ps;
  return rc;
}

bool group_ok(int g) { return g == BGLS_G1 || g == BGLS_G2; }

}  // namespace

// ======================================================================= C ABI
extern "C" {

int bgls_abi_version(void) { return 2; }

const char* bgls_last_error(void) { return g_err.c_str(); }

int bgls_init(int device) try {
  if (device < 0 || device >= MAX_DEVICES) return fail(BGLS_ERR_NO_DEVICE, "device index out of range");
  g_default_device.store(device);
  g_dev = -1;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  return c.enter();
} BGLS_ABI_GUARD

size_t bgls_fp_size(int curve) { return curve == BGLS_CURVE_ALTBN128 ? 32 : curve == BGLS_CURVE_BLS12_381 ? 48 : 0; }
size_t bgls_g1_size(int curve) { return 2 * bgls_fp_size(curve); }
size_t bgls_g2_size(int curve) { return 4 * bgls_fp_size(curve); }
size_t bgls_gt_size(int curve) { return 12 * bgls_fp_size(curve); }

int bgls_verify_aggregate(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob, const uint64_t* msg_off,
                          size_t n, int allow_duplicates) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!sig || !msg_off || (n && !keys)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_aggregate_t<CV>(sig, keys, msg_blob, msg_off, n, allow_duplicates));
} BGLS_ABI_GUARD

int bgls_verify_multi(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!sig || (n && !keys) || (msg_len && !msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_t<CV>(sig, keys, n, msg, msg_len));
} BGLS_ABI_GUARD

int bgls_verify_multi_batch(int curve, const uint8_t* sigs, const uint8_t* keys, const uint64_t* key_off, size_t n_sets, const uint8_t* msg_blob,
                            const uint64_t* msg_off, int allow_duplicates) try {
  if (n_sets >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!key_off || !msg_off || (n_sets && (!sigs || !msg_blob)) || (n_sets && key_off[n_sets] > key_off[0] && !keys)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_batch_t<CV>(sigs, keys, key_off, n_sets, msg_blob, msg_off, allow_duplicates));
} BGLS_ABI_GUARD

int bgls_aggregate_sets(int curve, int group, const uint8_t* pts, const uint64_t* set_off, size_t n_sets, uint8_t* out) try {
  if (!group_ok(group) || !set_off || (n_sets && !out) || (n_sets && set_off[n_sets] > set_off[0] && !pts)) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, aggregate_sets_t<CV>(group, pts, set_off, n_sets, out));
} BGLS_ABI_GUARD

int bgls_verify_multi_batch_dev(int curve, const void* d_sigs, const void* d_keys, const void* d_key_off, size_t n_sets, size_t max_set,
                                const void* d_msgs, size_t msg_len, size_t msg_stride, int allow_duplicates, void* stream) try {
  if (n_sets >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (n_sets && (!d_sigs || !d_keys || !d_key_off || !d_msgs)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_batch_sub_t<CV>(d_sigs, d_keys, d_key_off, n_sets, max_set, d_msgs, msg_len, msg_stride, allow_duplicates, stream, false));
} BGLS_ABI_GUARD
int bgls_verify_multi_batch_submit_dev(int curve, const void* d_sigs, const void* d_keys, const void* d_key_off, size_t n_sets, size_t max_set,
                                       const void* d_msgs, size_t msg_len, size_t msg_stride, int allow_duplicates, void* stream) try {
  if (n_sets >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (n_sets && (!d_sigs || !d_keys || !d_key_off || !d_msgs)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_batch_sub_t<CV>(d_sigs, d_keys, d_key_off, n_sets, max_set, d_msgs, msg_len, msg_stride, allow_duplicates, stream, true));
} BGLS_ABI_GUARD

int bgls_pairing_product(int curve, const uint8_t* g1s, const uint8_t* g2s, size_t n, uint8_t* gt_out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!gt_out || (n && (!g1s || !g2s))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, pairing_product_t<CV>(g1s, g2s, n, gt_out));
} BGLS_ABI_GUARD

int bgls_hash_to_g1(int curve, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n, uint8_t* g1_out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (n && (!msg_off || !g1_out)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, hash_to_g1_t<CV>(msg_blob, msg_off, n, g1_out));
} BGLS_ABI_GUARD

int bgls_aggregate_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!group_ok(group) || !out || (n && !pts)) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, aggregate_points_t<CV>(group, pts, n, out));
} BGLS_ABI_GUARD

int bgls_scale_points(int curve, int group, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n,
                      uint8_t* out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!group_ok(group) || (n && (!pts || !scalars || !out))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, scale_points_t<CV>(group, pts, scalars, signs, n, out));
} BGLS_ABI_GUARD

int bgls_point_add(int curve, int group, const uint8_t* a, const uint8_t* b, uint8_t* out) try {
  if (!group_ok(group) || !a || !b || !out) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  size_t pb = group == BGLS_G1 ? bgls_g1_size(curve) : bgls_g2_size(curve);
  if (!pb) return fail(BGLS_ERR_ARG, "unknown curve id");
  std::vector<uint8_t> two(2 * pb);
  memcpy(two.data(), a, pb);
  memcpy(two.data() + pb, b, pb);
  return bgls_aggregate_points(curve, group, two.data(), 2, out);
} BGLS_ABI_GUARD

int bgls_point_check(int curve, int group, const uint8_t* a) try {
  if (!group_ok(group) || !a) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, point_check_t<CV>(group, a));
} BGLS_ABI_GUARD

int bgls_check_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* ok_out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!group_ok(group) || (n && (!pts || !ok_out))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, check_points_t<CV>(group, pts, n, ok_out));
} BGLS_ABI_GUARD

int bgls_select_device(int device) try {
  if (device < -1 || device >= MAX_DEVICES) return fail(BGLS_ERR_NO_DEVICE, "device index out of range");
  g_dev = device;
  return 0;
} BGLS_ABI_GUARD

int bgls_keys_upload(int curve, const uint8_t* keys, size_t n, const int* devices, int n_devices, unsigned flags, bgls_keys_t* handle_out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!handle_out || (n && !keys)) return fail(BGLS_ERR_ARG, "NULL argument");
  if (n_devices < 1 || n_devices > NCTX) return fail(BGLS_ERR_ARG, "n_devices out of range (1..16)");
  int dflt = cur_device();
  if (!devices && n_devices == 1) devices = &dflt;
  DISPATCH(curve, keys_upload_t<CV>(keys, n, devices, n_devices, flags, handle_out));
} BGLS_ABI_GUARD

int bgls_keys_free(bgls_keys_t handle) try {
  std::shared_ptr<KeySet> ks;
  {
    std::lock_guard<std::mutex> lk(g_keys_mu);
    auto it = g_keys.find(handle);
    if (it == g_keys.end()) return fail(BGLS_ERR_ARG, "unknown key-set handle");
    ks = it->second;
    g_keys.erase(it);
  }
  std::lock_guard<std::mutex> lk(ks->mu);      // wait for a verification in progress
  return 0;
} BGLS_ABI_GUARD

int bgls_keys_info(bgls_keys_t handle, int* curve, size_t* n, int* n_devices) try {
  auto ks = keyset(handle);
  if (!ks) return fail(BGLS_ERR_ARG, "unknown key-set handle");
  if (curve) *curve = ks->curve;
  if (n) *n = ks->n;
  if (n_devices) *n_devices = (int)ks->shards.size();
  return 0;
} BGLS_ABI_GUARD

int bgls_verify_aggregate_h(bgls_keys_t handle, const uint8_t* sig, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n,
                            int allow_duplicates) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  auto ks = keyset(handle);
  if (!ks) return fail(BGLS_ERR_ARG, "unknown key-set handle");
  if (!sig || !msg_off) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(ks->curve, verify_aggregate_h_t<CV>(*ks, sig, msg_blob, msg_off, n, allow_duplicates, nullptr));
} BGLS_ABI_GUARD

int bgls_verify_aggregate_h_gt(bgls_keys_t handle, const uint8_t* sig, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n,
                               int allow_duplicates, uint8_t* gt_out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  auto ks = keyset(handle);
  if (!ks) return fail(BGLS_ERR_ARG, "unknown key-set handle");
  if (!sig || !msg_off || !gt_out) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(ks->curve, verify_aggregate_h_t<CV>(*ks, sig, msg_blob, msg_off, n, allow_duplicates, gt_out));
} BGLS_ABI_GUARD

int bgls_rccl_available(void) try { return rccl().ok ? 1 : 0; } BGLS_ABI_GUARD

// device-resident messages against a one-device key set, on the calling thread's context (several verifications in flight
// on several contexts: the handle's resident arrays are read-only)
int bgls_miller_product_keys_dev(bgls_keys_t handle, const void* d_sig, const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n,
                                 int check_duplicates, void* d_partial_out, void* d_flags, void* stream) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  auto ks = keyset(handle);
  if (!ks) return fail(BGLS_ERR_ARG, "unknown key-set handle");
  if (ks->shards.size() != 1) return fail(BGLS_ERR_ARG, "device entry point: the key set must live on one device");
  if (n != ks->n) return fail(BGLS_ERR_ARG, "message count differs from the key set's size");
  if (!d_partial_out || !d_flags || (n && !d_msgs && msg_len)) return fail(BGLS_ERR_ARG, "NULL argument");
  const int pd = g_dev;
  g_dev = ks->shards[0].device;
  Ctx& c = ctx();
  int rc;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    rc = [&]() -> int {
      int r;
      if ((r = c.enter())) return r;
      hipStream_t st = stream ? (hipStream_t)stream : c.stream;
      const KeyShard& sh = ks->shards[0];
      MsgView mv = {(const uint8_t*)d_msgs, nullptr, msg_len, msg_stride};
      if (ks->curve == BGLS_CURVE_ALTBN128) {
        typedef Engine<BN254> E;
        if (check_duplicates && (r = E::dup_scan(c, st, mv, n, (uint32_t*)d_flags))) return r;
        if (ks->prepared) return E::miller_product_prepared(c, st, (const uint8_t*)d_sig, (const uint32_t*)sh.d_prep, (const uint8_t*)sh.d_kinf, sh.n_pad, mv, n, (uint8_t*)d_partial_out, (uint32_t*)d_flags);
        return E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)sh.d_wire, mv, n, 0, (uint8_t*)d_partial_out, (uint32_t*)d_flags);
      }
      typedef Engine<BLS381> E;
      if (check_duplicates && (r = E::dup_scan(c, st, mv, n, (uint32_t*)d_flags))) return r;
      if (ks->prepared) return E::miller_product_prepared(c, st, (const uint8_t*)d_sig, (const uint32_t*)sh.d_prep, (const uint8_t*)sh.d_kinf, sh.n_pad, mv, n, (uint8_t*)d_partial_out, (uint32_t*)d_flags);
      return E::miller_product(c, st, (const uint8_t*)d_sig, (const uint8_t*)sh.d_wire, mv, n, 0, (uint8_t*)d_partial_out, (uint32_t*)d_flags);
    }();
  }
  g_dev = pd;
  return rc;
} BGLS_ABI_GUARD

// verifyMultiSignature against a one-device key set with signature and message already on the device, on the calling thread's
// context (several checks in flight on several contexts: the handle's resident arrays are read-only)
static int verify_multi_keys_dev(bgls_keys_t handle, const void* d_sig, const void* d_msg, size_t msg_len, void* stream, bool submit_only) {
  auto ks = keyset(handle);
  if (!ks) return fail(BGLS_ERR_ARG, "unknown key-set handle");
  if (ks->shards.size() != 1) return fail(BGLS_ERR_ARG, "device entry point: the key set must live on one device");
  if (!d_sig || (msg_len && !d_msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  const int pd = g_dev;
  g_dev = ks->shards[0].device;
  Ctx& c = ctx();
  int rc;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    rc = [&]() -> int {
      int r;
      if ((r = c.enter())) return r;
      hipStream_t st = stream ? (hipStream_t)stream : c.stream;
      const KeyShard& sh = ks->shards[0];
      if (ks->curve == BGLS_CURVE_ALTBN128) {
        const bool ready = sum_mode<BN254>() == 2;
        return verify_multi_dev_t<BN254>(c, st, (const uint8_t*)d_sig, (const uint8_t*)(ready ? sh.d_sumr : sh.d_mont), ks->n, (const uint8_t*)d_msg, msg_len, submit_only, ready ? 2 : 1);
      }
      const bool ready = sum_mode<BLS381>() == 2;
      return verify_multi_dev_t<BLS381>(c, st, (const uint8_t*)d_sig, (const uint8_t*)(ready ? sh.d_sumr : sh.d_mont), ks->n, (const uint8_t*)d_msg, msg_len, submit_only, ready ? 2 : 1);
    }();
  }
  g_dev = pd;
  return rc;
}
int bgls_verify_multi_keys_dev(bgls_keys_t handle, const void* d_sig, const void* d_msg, size_t msg_len, void* stream) try {
  return verify_multi_keys_dev(handle, d_sig, d_msg, msg_len, stream, false);
} BGLS_ABI_GUARD
int bgls_verify_multi_keys_submit_dev(bgls_keys_t handle, const void* d_sig, const void* d_msg, size_t msg_len, void* stream) try {
  return verify_multi_keys_dev(handle, d_sig, d_msg, msg_len, stream, true);
} BGLS_ABI_GUARD

int bgls_verify_multi_h(bgls_keys_t handle, const uint8_t* sig, const uint8_t* msg, size_t msg_len) try {
  auto ks = keyset(handle);
  if (!ks) return fail(BGLS_ERR_ARG, "unknown key-set handle");
  if (!sig || (msg_len && !msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(ks->curve, verify_multi_h_t<CV>(*ks, sig, msg, msg_len));
} BGLS_ABI_GUARD

int bgls_verify_aggregate_multi(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob, const uint64_t* msg_off,
                                size_t n, int allow_duplicates, const int* devices, int n_devices) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  bgls_keys_t h;
  // host keys arrive unvalidated here (no Point construction in between): the upload checks the order-r subgroup as the
  // reference's constructors do, so a small-order twist point is an encoding error, not an unspecified verdict
  int rc = bgls_keys_upload(curve, keys, n, devices, n_devices, BGLS_KEYS_CHECK, &h);
  if (rc) return rc;
  rc = bgls_verify_aggregate_h(h, sig, msg_blob, msg_off, n, allow_duplicates);
  const int ex = g_last_exchange;
  (void)bgls_keys_free(h);
  g_last_exchange = ex;
  return rc;
} BGLS_ABI_GUARD

int bgls_verify_multi_multi(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len,
                            const int* devices, int n_devices) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  bgls_keys_t h;
  int rc = bgls_keys_upload(curve, keys, n, devices, n_devices, BGLS_KEYS_CHECK, &h);
  if (rc) return rc;
  rc = bgls_verify_multi_h(h, sig, msg, msg_len);
  const int ex = g_last_exchange;
  (void)bgls_keys_free(h);
  g_last_exchange = ex;
  return rc;
} BGLS_ABI_GUARD

int bgls_last_exchange(void) { return g_last_exchange; }

int bgls_generator(int curve, int group, uint8_t* out) try {
  if (!group_ok(group) || !out) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, generator_t<CV>(group, out));
} BGLS_ABI_GUARD

int bgls_pair(int curve, const uint8_t* g1, const uint8_t* g2, uint8_t* gt_out) try {
  return bgls_pairing_product(curve, g1, g2, 1, gt_out);
} BGLS_ABI_GUARD

int bgls_gt_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* out) try {
  if (!a || !b || !out) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, gt_mul_t<CV>(a, b, out));
} BGLS_ABI_GUARD

int bgls_gt_pow(int curve, const uint8_t* gt, const uint8_t* k_be32, int negative, uint8_t* out) try {
  if (!gt || !k_be32 || !out) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, gt_pow_t<CV>(gt, k_be32, negative ? 1 : 0, out));
} BGLS_ABI_GUARD

int bgls_gt_identity(int curve, uint8_t* out) try {
  size_t n = bgls_gt_size(curve);
  if (!n || !out) return fail(BGLS_ERR_ARG, "unknown curve id or NULL argument");
  memset(out, 0, n);
  out[n - 1] = 1;
  return 0;
} BGLS_ABI_GUARD

int bgls_profile_enable(int on) try {
  Ctx* all = ctx_pool();
  for (int k = 0; k < MAX_DEVICES * NCTX; ++k) {
    Ctx& c = all[k];
    std::lock_guard<std::mutex> lk(c.mu);
    c.prof = on != 0;
    for (int i = 0; i < 8; ++i) { c.stage_ms[i] = 0; c.stage_cnt[i] = 0; }
  }
  return 0;
} BGLS_ABI_GUARD

int bgls_profile_get(const char* stage, double* total_ms, unsigned long long* launches) try {
  if (!stage || !total_ms || !launches) return fail(BGLS_ERR_ARG, "NULL argument");
  for (int i = 0; i < ST_NUM; ++i)
    if (!strcmp(stage, STAGE_NAMES[i])) {
      *total_ms = 0;
      *launches = 0;
      Ctx* all = ctx_pool();
      for (int k = 0; k < MAX_DEVICES * NCTX; ++k) {        // summed over the contexts of every device
        std::lock_guard<std::mutex> lk(all[k].mu);
        *total_ms += all[k].stage_ms[i];
        *launches += all[k].stage_cnt[i];
      }
      return 0;
    }
  return fail(BGLS_ERR_ARG, "unknown stage name");
} BGLS_ABI_GUARD

// Self-test of the exception barrier (no device needed): raises the named C++ exception inside a guarded body -- kind 0
// std::bad_alloc, 1 std::length_error, 2 std::system_error, 3 std::runtime_error, 4 a non-standard object, 5 a vector whose
// size no allocator can serve, 6 an exception on a shard's host thread -- and returns what the guard made of it.
int bgls_selftest_exception_barrier(int kind) try {
  switch (kind) {
    case 0: throw std::bad_alloc();
    case 1: throw std::length_error("selftest");
    case 2: throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "selftest");
    case 3: throw std::runtime_error("selftest");
    case 4: throw 42;
    case 5: {
      std::vector<uint64_t> v((size_t)-1 / 16);
      return (int)v.size();
    }
    case 6: {
      KeySet ks;
      ks.shards.resize(2);
      ks.shards[0].device = ks.shards[1].device = 0;
      return for_each_shard(ks, [&](int s) -> int {
        if (s == 1) throw std::bad_alloc();
        return 0;
      });
    }
    default: return 0;
  }
} BGLS_ABI_GUARD

int bgls_probe_mad_peak(double* mac_per_s) try {
  if (!mac_per_s) return fail(BGLS_ERR_ARG, "NULL argument");
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = c.stream;
  void* sink;
  if ((rc = c.get(WS_OUT, 16, &sink))) return rc;
  const int iters = 4096, blocks = 256 * 8, threads = 256;
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  double best = 0;
  for (int rep = 0; rep < 4; ++rep) {
    HIPCHK(hipEventRecord(a, st));
    kl::mad_probe(st, blocks, threads, 12345u + rep, iters, (uint64_t*)sink);
    HIPCHK(hipEventRecord(b, st));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    double macs = (double)blocks * threads * iters * 16.0;
    double rate = macs / (ms * 1e-3);
    if (rep > 0 && rate > best) best = rate;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *mac_per_s = best;
  return 0;
} BGLS_ABI_GUARD

int bgls_miller_product_dev(int curve, const void* d_sig, const void* d_keys, const void* d_msgs, size_t msg_len,
                            size_t msg_stride, size_t n, int check_duplicates, void* d_partial_out, void* d_flags,
                            void* stream) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!d_partial_out || !d_flags || (n && (!d_keys || (!d_msgs && msg_len)))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, miller_product_dev_t<CV>(d_sig, d_keys, d_msgs, msg_len, msg_stride, n, check_duplicates, d_partial_out,
                                           d_flags, stream));
} BGLS_ABI_GUARD

int bgls_set_throughput_mode(int on) try {
  g_throughput.store(on ? 1 : 0);
  return 0;
} BGLS_ABI_GUARD

int bgls_set_miller_shape(int shape, int pairings_per_group) try {
  if (shape < 0 || shape > 5 || pairings_per_group < 0 || pairings_per_group > 4096) return fail(BGLS_ERR_ARG, "bad Miller shape");
  if (shape >= 4) {                       // 4: k_miller_x60 always (second argument: mode word, see bgls_hip.h); 5: the 32-bit fused kernels always
    if (shape == 4 && (pairings_per_group > 31 || (pairings_per_group & 3) == 3)) return fail(BGLS_ERR_ARG, "bad k_miller_x60 mode word");
    if (shape == 4) g_x60_rot.store(pairings_per_group);
    g_shape.store(shape);
    return 0;
  }
  if (pairings_per_group < 1) return fail(BGLS_ERR_ARG, "bad Miller shape");
  if (shape == 0) g_x60_rot.store(-1);    // back to the automatic role / priority mode
  g_shape.store(shape);
  g_ng.store(pairings_per_group);
  return 0;
} BGLS_ABI_GUARD

int bgls_set_msm_min(size_t n) try {
  g_msm_min.store(n);
  return 0;
} BGLS_ABI_GUARD

int bgls_weighted_sum_dev(int curve, int group, const void* d_pts, const void* d_w16, size_t n, void* d_out, void* stream) try {
  if (group != BGLS_G1 && group != BGLS_G2) return fail(BGLS_ERR_ARG, "bad group");
  if (!d_out || (n && (!d_pts || !d_w16))) return fail(BGLS_ERR_ARG, "NULL argument");
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  DISPATCH(curve, weighted_sum_dev_t<CV>(group, d_pts, d_w16, n, d_out, stream));
} BGLS_ABI_GUARD

int bgls_select_context(int index) try {
  if (index < 0 || index >= NCTX) return fail(BGLS_ERR_ARG, "context index out of range");
  g_sel = index;
  return 0;
} BGLS_ABI_GUARD

int bgls_final_verify_submit_dev(int curve, const void* d_partials, size_t count, const void* d_flags, void* stream) try {
  if (!d_partials || !count) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, final_verify_submit_dev_t<CV>(d_partials, count, d_flags, stream));
} BGLS_ABI_GUARD

int bgls_final_verify_collect(int curve) try {
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  DISPATCH(curve, Engine<CV>::finalize_collect(c));
} BGLS_ABI_GUARD

int bgls_duplicate_scan_dev(const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n, void* d_flags, void* stream) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!d_flags || (n && !d_msgs && msg_len)) return fail(BGLS_ERR_ARG, "NULL argument");
  if (n >= (1ull << 30)) return fail(BGLS_ERR_ARG, "too many messages for one scan");
  return duplicate_scan_dev(d_msgs, msg_len, msg_stride, n, d_flags, stream);
} BGLS_ABI_GUARD

int bgls_message_digests_dev(const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n, void* d_out16, void* stream) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (n && (!d_out16 || (!d_msgs && msg_len))) return fail(BGLS_ERR_ARG, "NULL argument");
  if (((uintptr_t)d_out16 & 15) != 0) return fail(BGLS_ERR_ARG, "digest buffer must be 16-byte aligned");
  if (n == 0) return 0;
  Ctx& c = ctx();
  std::lock_guard<std::mutex> lk(c.mu);
  int rc;
  if ((rc = c.enter())) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : c.stream;
  MsgView mv = {(const uint8_t*)d_msgs, nullptr, msg_len, msg_stride};
  kl::msg_digest(st, mv, n, (uint8_t*)d_out16);
  HIPCHK(hipGetLastError());
  return 0;
} BGLS_ABI_GUARD

int bgls_final_verify_dev(int curve, const void* d_partials, size_t count, const void* d_flags, void* stream) try {
  if (!d_partials || !count) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, final_verify_dev_t<CV>(d_partials, count, d_flags, stream));
} BGLS_ABI_GUARD

int bgls_aggregate_points_dev(int curve, int group, const void* d_pts, size_t n, void* d_out, void* stream) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!group_ok(group) || !d_out || (n && !d_pts)) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, aggregate_points_dev_t<CV>(group, d_pts, n, d_out, stream));
} BGLS_ABI_GUARD

int bgls_verify_multi_dev(int curve, const void* d_sig, const void* d_keys, size_t n, const void* d_msg, size_t msg_len,
                          void* stream) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!d_sig || (n && !d_keys) || (msg_len && !d_msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_dev_entry_t<CV>(d_sig, d_keys, n, d_msg, msg_len, stream));
} BGLS_ABI_GUARD

int bgls_verify_multi_submit_dev(int curve, const void* d_sig, const void* d_keys, size_t n, const void* d_msg, size_t msg_len,
                                 void* stream) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!d_sig || (n && !d_keys) || (msg_len && !d_msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_dev_entry_t<CV>(d_sig, d_keys, n, d_msg, msg_len, stream, true));
} BGLS_ABI_GUARD

/* ---- hashed aggregation exponents (bgls/blsHAE.go) and multiplicities (bgls/blsKosk.go:137-150) ---- */
int bgls_hae_exponents(int curve, const uint8_t* keys, size_t n, uint8_t* t_out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (n && (!keys || !t_out)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, hae_exponents_t<CV>(keys, n, t_out));
} BGLS_ABI_GUARD

int bgls_aggregate_signatures_hae(int curve, const uint8_t* sigs, const uint8_t* keys, size_t n, uint8_t* out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!out || (n && (!sigs || !keys))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, aggregate_signatures_hae_t<CV>(sigs, keys, n, out));
} BGLS_ABI_GUARD

int bgls_verify_multi_hae(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!sig || (n && !keys) || (msg_len && !msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_multi_weighted_t<CV>(sig, keys, nullptr, n, msg, msg_len));
} BGLS_ABI_GUARD

int bgls_verify_aggregate_hae(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob, const uint64_t* msg_off,
                              size_t n) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!sig || !msg_off || (n && !keys)) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, verify_aggregate_hae_t<CV>(sig, keys, msg_blob, msg_off, n));
} BGLS_ABI_GUARD

int bgls_verify_multi_multiplicity(int curve, const uint8_t* sig, const uint8_t* keys, const int64_t* multiplicity, size_t n,
                                   const uint8_t* msg, size_t msg_len) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!sig || (n && !keys) || (msg_len && !msg)) return fail(BGLS_ERR_ARG, "NULL argument");
  if (!multiplicity) DISPATCH(curve, verify_multi_t<CV>(sig, keys, n, msg, msg_len));
  DISPATCH(curve, verify_multi_weighted_t<CV>(sig, keys, multiplicity, n, msg, msg_len));
} BGLS_ABI_GUARD

/* ---- compressed wire formats (alt-bn128; curves/altbn128.go:81-89,203-221,296-376) ---- */
int bgls_compress_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!group_ok(group) || (n && (!pts || !out))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  return wire_points(curve, group, true, pts, n, out, nullptr);
} BGLS_ABI_GUARD

int bgls_decompress_points(int curve, int group, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!group_ok(group) || (n && (!in || !out || !ok))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  return wire_points(curve, group, false, in, n, out, ok);
} BGLS_ABI_GUARD

/* ---- batch key generation / signing (bgls/bgls.go:40-56) ---- */
int bgls_scale_generator(int curve, int group, const uint8_t* scalars, size_t n, uint8_t* out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!group_ok(group) || (n && (!scalars || !out))) return fail(BGLS_ERR_ARG, "bad group or NULL argument");
  DISPATCH(curve, scale_generator_t<CV>(group, scalars, n, out));
} BGLS_ABI_GUARD

int bgls_sign_batch(int curve, const uint8_t* sks, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n, uint8_t* sigs_out) try {
  if (n >= MAX_BATCH) return fail(BGLS_ERR_ARG, "batch too large (n must be below 2^30)");
  if (!msg_off || (n && (!sks || !sigs_out))) return fail(BGLS_ERR_ARG, "NULL argument");
  DISPATCH(curve, sign_batch_t<CV>(sks, msg_blob, msg_off, n, sigs_out));
} BGLS_ABI_GUARD

}  // extern "C"


