// Host-callable launchers of every kernel in the library.  Each k_*.hip translation unit defines the launchers of the
// kernels it holds (a <<<>>> launch must sit in the unit that defines the kernel; the units are compiled in parallel and
// linked without relocatable device code); engine.hip only sees these declarations.
#pragma once
#include "dev_common.hpp"

namespace bgls {
namespace kl {

// ---- k_hash.hip
void digest_pack(hipStream_t st, const uint8_t* dig, size_t n, uint8_t* out, size_t cap, uint32_t n_buckets, uint32_t* counts, uint32_t* flags);
constexpr uint32_t DUP_MAX_PROBE = 1023;      // bucketed duplicate scan: probes per record before it reports "undecided"
void dup_check(hipStream_t st, MsgView mv, size_t n, uint32_t* table, uint32_t mask, uint32_t* flags, uint32_t bucket = 0, uint32_t n_buckets = 1, uint64_t seed = 0,
               bool unbounded = false);
void msg_digest(hipStream_t st, MsgView mv, size_t n, uint8_t* out);
void h2c_bn(hipStream_t st, MsgView mv, size_t n, uint32_t* lists, uint32_t* counters, Aff<F1<BN254>>* out, uint32_t* flags, bool lean);
void h2c_bls(hipStream_t st, MsgView mv, size_t n, Jac<F1<BLS381>>* pts, uint32_t* kinds, Aff<F1<BLS381>>* out, bool raw);   // pts: 2n Jacobian work items
void blake2x_expand(hipStream_t st, const uint64_t* root, uint32_t xof_len, uint8_t* out);

// ---- k_points.hip   (group = BGLS_G1 / BGLS_G2; Jacobian workspaces are passed as void*)
template <class C> void g1_to_bytes(hipStream_t st, const Aff<F1<C>>* in, size_t n, uint8_t* out);
template <class C> void g1_parse(hipStream_t st, const uint8_t* in, size_t n, int negate, Aff<F1<C>>* out, uint32_t* flags);
template <class C> void sum_main(hipStream_t st, int group, bool parsed, const uint8_t* pts, size_t n, unsigned waves, void* out, uint32_t* flags);
template <class C> void sumseg_main(hipStream_t st, int group, const uint8_t* pts, const uint64_t* off, size_t nsets, unsigned P, void* out, uint32_t* flags);
// ---- k_sumpair.hip   (G2 key sums on lane pairs, carry-free limbs: rx_jacpair.hpp)
template <class C> void sumpair_main(hipStream_t st, int src, const uint8_t* pts, size_t n, unsigned partials, void* out, uint32_t* flags);
template <class C> void sumpairseg_main(hipStream_t st, const uint8_t* pts, const uint64_t* off, size_t nsets, unsigned P, void* out, uint32_t* flags);
template <class C> void sum_wave(hipStream_t st, int group, const void* in, size_t n, void* out);
template <class C> void sum_pair(hipStream_t st, int group, const void* in, size_t n, void* out);
template <class C> void sum_coop(hipStream_t st, const void* in, size_t n, void* out);        // G2, one wave per addition (jac_coop.hpp)
template <class C> void sum_next(hipStream_t st, int group, const void* in, size_t n, int R, void* out);
template <class C> void jac_to_bytes(hipStream_t st, int group, const void* in, size_t n, uint8_t* out);
template <class C> void wsum_first(hipStream_t st, int group, const uint8_t* pts, const uint8_t* w16, const uint8_t* signs, size_t n, void* out, uint32_t* flags);
template <class C> void scale(hipStream_t st, int group, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs, size_t n, uint8_t* out, uint32_t* flags, int sbytes);
template <class C> void scale_aff(hipStream_t st, int group, const Aff<F1<C>>* g1_pts, const uint8_t* scalars, size_t n, uint8_t* out);
template <class C> void check(hipStream_t st, int group, const uint8_t* pts, size_t n, uint32_t* flags, uint8_t* ok);
template <class C> void g2_parse(hipStream_t st, const uint8_t* in, size_t n, int check_subgroup, void* out, uint32_t* flags);
template <class C> size_t g2_parsed_bytes();
template <class C> void g2_sumready(hipStream_t st, const void* mont, size_t n, void* out);      // k_sumpair.hip: a key set's sum-ready records
template <class C> size_t g2_sumready_bytes();
template <class C> void generator(hipStream_t st, int group, uint8_t* out);
void compress_bn(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint32_t* flags);
void decompress_bn(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok);
// ---- k_wire.hip   (BLS12-381 compressed forms, ebfull/pairing layout)
void compress_bls(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint32_t* flags);
void decompress_bls(hipStream_t st, int group, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok);
void mad_probe(hipStream_t st, unsigned blocks, unsigned threads, uint32_t seed, int iters, uint64_t* sink);
template <class C> size_t jac_bytes(int group) { return (size_t)3 * (group == 1 ? 1 : 2) * C::L * 4; }

// ---- k_msm.hip   (bucket-method weighted sums, fixed-base generator multiples, in-place G1 scaling)
struct MsmPlan { int c, W, K, S; uint32_t NB, NQ, NCH; };    // window bits, windows, digits per chunk, threads per bucket, buckets, chunks per window, chunks
MsmPlan msm_plan(size_t n);
template <class C> size_t msm_aff_bytes(int group);
template <class C>
void msm_parse(hipStream_t st, int group, const uint8_t* pts, const uint8_t* w16, const uint8_t* signs, size_t n, const MsmPlan& p, void* aff,
               uint32_t* cnt, uint32_t* flags);
void msm_scan(hipStream_t st, uint32_t* cnt, uint32_t NB, uint32_t* start, uint32_t* meta);
template <class C>
void msm_scatter(hipStream_t st, int group, const void* aff, const uint8_t* w16, size_t n, const MsmPlan& p, uint32_t* cursor, uint32_t* list);
template <class C>
void msm_buckets(hipStream_t st, int group, const void* aff, const uint32_t* list, const uint32_t* start, const MsmPlan& p, void* parts);
// buckets (one Jacobian point each) -> the weighted sum; scratch: msm_tail_points(p) Jacobian points, the result is *result
size_t msm_tail_points(const MsmPlan& p);
template <class C> void msm_tail(hipStream_t st, int group, const void* buckets, const MsmPlan& p, void* scratch, void** result);
template <class C> size_t fb_table_bytes(int group);
template <class C> void fb_build(hipStream_t st, int group, void* table);
template <class C> void fb_scale(hipStream_t st, int group, const void* table, const uint8_t* scalars, size_t n, uint8_t* out);
template <class C> void scale_g1_inplace(hipStream_t st, Aff<F1<C>>* pts, const uint8_t* w16, size_t n);

// ---- k_miller_bn.hip / k_miller_bls.hip   (dbg != 0 only in -DBGLS_DEV builds: timing variants, wrong results)
template <class C>
void miller_ab64(hipStream_t st, unsigned nblocks, const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at,
                 const LineCoeffs<C>* gen_lines, Fp2<C>* out, uint32_t* flags, int dbg, uint32_t* qp);
template <class C> constexpr size_t miller_qp_bytes(size_t nblocks) { return nblocks * 64 * 6 * C::L * 4; }    // parked Q, P of every producer lane

// ---- k_millerx_{bn,bls}.hip (NP = 60), k_millerx64_{bn,bls}.hip (NP = 64)
template <class C, int NP>
void miller_x(hipStream_t st, unsigned nblocks, const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, Fp2<C>* out, uint32_t* flags, uint32_t* park, int rot_mode);
template <class C, int NP> size_t miller_x_park_bytes(size_t nblocks);


// prepared key sets (prepared.hpp): bytes per key of the line table, per pairing of the point table, per key of k_prepare's scratch
struct PrepSizes { size_t line_bytes_per_key, point_bytes, tmp_bytes_per_key; };
template <class C> PrepSizes prep_sizes();
template <class C>
void prepare_keys(hipStream_t st, const void* keys_mont, size_t n, size_t n_pad, size_t i0, size_t count, uint32_t* table, uint8_t* kinf,
                  uint32_t* tmp, uint32_t* flags);
template <class C> void prep_points(hipStream_t st, const Aff<F1<C>>* g1s, const uint8_t* kinf, size_t n, size_t n_pad, uint32_t* ptab);
template <class C> void fold_prep(hipStream_t st, const uint32_t* table, const uint32_t* ptab, size_t n_pad, int ng, Fp2<C>* out);

// ---- k_tail_bn.hip / k_tail_bls.hip
template <class C> void gen_lines(hipStream_t st, LineCoeffs<C>* table, int* nsteps);
template <class C> void reduce_coop(hipStream_t st, const Fp2<C>* in, size_t count, int R, Fp2<C>* out);
template <class C> void w_to_bytes(hipStream_t st, const Fp2<C>* in, uint8_t* out);
template <class C> void final36(hipStream_t st, const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out, uint32_t* verdict, uint32_t* flags);
template <class C> void finalx(hipStream_t st, const uint8_t* partials, size_t count, int do_final_exp, uint8_t* gt_out, uint32_t* verdict, uint32_t* flags);
template <class C>
void miller_lat(hipStream_t st, const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at, const LineCoeffs<C>* gen_lines, Fp2<C>* out,
                uint32_t* flags);
template <class C>
void miller_latx(hipStream_t st, const Aff<F1<C>>* g1s, const uint8_t* g2s, size_t n, long long sig_at, const LineCoeffs<C>* gen_lines, Fp2<C>* out,
                uint32_t* flags);
template <class C> void gt_pow(hipStream_t st, const uint8_t* gt_in, const uint8_t* k_be32, int negate, uint8_t* gt_out, uint32_t* flags);
template <class C> void cofactor_epilogue(hipStream_t st, const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines, uint8_t* out);
template <class C> void cofactor_epiloguex(hipStream_t st, const Fp2<C>* rest, const Aff<F1<C>>* sig, const LineCoeffs<C>* gen_lines, Fp2<C>* tmp, uint8_t* out, uint32_t* flags);

}  // namespace kl
}  // namespace bgls
