/*
 * bgls_hip.h -- C ABI of the MI355X aggregate-signature verification engine.
 *
 * Drop-in boundary for the hot path of Project-Arda/bgls: everything the Go package `curves`
 * reaches through its CurveSystem / Point / PointT interfaces (curves/curve.go:12-70) and the
 * goroutine helpers built on them (curves/curve.go:73-223), restated as BATCH entry points so
 * one cgo call replaces n goroutines.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   curve   : BGLS_CURVE_ALTBN128 (curves.Altbn128, curves/altbn128.go:32) or
 *             BGLS_CURVE_BLS12_381 (curves.Bls12, curves/bls12_381.go:31)
 *   return  : verify-style calls return 1 (valid) / 0 (invalid); every call returns < 0 on a
 *             usage or runtime error.  The Go shim maps anything != 1 to `false` / `nil,false`,
 *             which is the reference's only error convention (curves/curve.go:15-22,46-48;
 *             bgls/bgls.go:95-97,115-118).  No exceptions, no abort().
 *   formats : uncompressed big-endian wire formats of the reference --
 *             G1 = x||y (curves/altbn128.go:42-57; curves/testcases/bls12G1Hash.dat),
 *             G2 = x_im||x_re||y_im||y_re (curves/altbn128.go:157-179, altbn128_test.go:26-38;
 *                  curves/bls12_381.go:147-158,209-226),
 *             infinity = all-zero bytes (curves/altbn128.go:431-439),
 *             GT = 12 field elements (384 B / 576 B; curves/altbn128.go:378-387), layout in
 *                  DESIGN.md (byte-parity with the upstream Go libraries is unpinned).
 *             Sizes: bgls_g1_size / bgls_g2_size / bgls_gt_size.
 *   memory  : the caller owns every buffer; inputs are const and never modified (the reference's
 *             in-place mutation quirks, curves/altbn128.go:306-309,344-349 and
 *             curves/bls12_381.go:70,131, are NOT reproduced).
 *   threads : every entry point may be called concurrently; calls on one context serialise.
 *   device  : all arithmetic runs in HIP kernels on the selected GPU; there is no CPU fallback.
 *             If no gfx950 device is usable every compute call returns BGLS_ERR_NO_DEVICE.
 */
#ifndef BGLS_HIP_H
#define BGLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGLS_CURVE_ALTBN128 0
#define BGLS_CURVE_BLS12_381 1

#define BGLS_G1 1
#define BGLS_G2 2

#define BGLS_ERR_ARG (-1)       /* bad curve/group id, NULL pointer, inconsistent offsets */
#define BGLS_ERR_ENCODING (-2)  /* a coordinate >= q, a point not on its curve, or (where checked) a point outside the order-r subgroup */
#define BGLS_ERR_HASH (-3)      /* try-and-increment exhausted 256 counters (probability 2^-256) */
#define BGLS_ERR_NO_DEVICE (-4) /* no usable HIP device */
#define BGLS_ERR_HIP (-5)       /* a HIP runtime call failed; see bgls_last_error() */
#define BGLS_ERR_NOMEM (-6)     /* a host allocation failed inside the library (std::bad_alloc); nothing was verified */

/* ---- runtime ---------------------------------------------------------------------------- */
/* Select the default HIP device of the calling process (default 0) and bring it up.  May be called again with another
 * device; contexts, workspaces and key sets of devices used earlier stay valid. */
int bgls_init(int device);
/* Device used by the calling thread's subsequent calls (-1 = the process default set by bgls_init). */
int bgls_select_device(int device);
/* Human-readable text for the last error on this thread ("" if none). */
const char* bgls_last_error(void);
/* ABI version; bumped on any signature change. */
int bgls_abi_version(void);

size_t bgls_fp_size(int curve); /* 32 / 48 */
size_t bgls_g1_size(int curve); /* 64 / 96 */
size_t bgls_g2_size(int curve); /* 128 / 192 */
size_t bgls_gt_size(int curve); /* 384 / 576 */

/* ---- the hot path, host buffers ------------------------------------------------------------ */
/* The Verify* entry points take keys as the reference's Verify* functions take Points: already constructed, i.e.
 * validated by MakeG2Point / UnmarshalG2 (here: bgls_point_check, bgls_check_points, bgls_decompress_points or
 * bgls_keys_upload with BGLS_KEYS_CHECK).  They re-check canonical encoding and curve membership of every key (an
 * encoding error is BGLS_ERR_ENCODING, never an accept) but not G2 subgroup membership, exactly as bgls.Verify*
 * does not re-validate its Point arguments. */

/* bgls.VerifyAggregateSignature (bgls/bgls.go:82-84) -> verifyAggSig (bgls/bgls.go:94-119).
 * keys: n G2 points; messages: msg_blob[msg_off[i] .. msg_off[i+1]), i < n (msg_off has n+1
 * entries).  allow_duplicates = 0 reproduces the duplicate-message rejection
 * (containsDuplicateMessage, bgls/bgls.go:139-150); 1 is the Kosk / distinct-message callers'
 * mode (bgls/blsKosk.go:100-106, bgls/blsDistinctMessage.go:45-57).
 * Computes e(-sig, g2) * prod_i e(HashToG1(m_i), pk_i) == 1. */
int bgls_verify_aggregate(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob,
                          const uint64_t* msg_off, size_t n, int allow_duplicates);

/* verifyMultiSignature (bgls/bgls.go:89-92): apk = sum(keys) (AggregatePoints,
 * curves/curve.go:73-121) then VerifySingleSignature (bgls/bgls.go:59-70) on msg.
 * KoskVerifyMultiSignature (bgls/blsKosk.go:117-120) is this call with 0x01 prepended to msg. */
int bgls_verify_multi(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg,
                      size_t msg_len);

/* KoskVerifyBatchMultiSignature's body (bgls/blsKosk.go:126-133): aggsig = AggregateSignatures(sigs) (n_sets G1 points),
 * key_b = AggregateKeys(set b) with set b = keys[key_off[b] .. key_off[b+1]) (counts of points, n_sets + 1 offsets), then
 * verifyAggSig(aggsig, key_0 .. key_{n_sets-1}, msgs, allow_duplicates) -- ONE launch for all key sums, one Miller launch and
 * one final exponentiation for all the sets.  The Go function prepends 0x01 to every message and passes
 * allow_duplicates = true (KoskVerifyAggregateSignature, bgls/blsKosk.go:100-106); the host mirrors do the same.
 * Keys are checked for canonical encoding and curve membership, NOT for subgroup membership: like every Verify* entry point
 * this call takes Points the caller has constructed (validated) already -- bgls_check_points / bgls_keys_upload(BGLS_KEYS_CHECK)
 * are the constructors' checks; a degenerate point step of a small-order key is reported as BGLS_ERR_ENCODING.
 * One call handles what fits one launch: at most 2^30 blocks of the key-sum pass and 8 GiB of partial sums (a set of up to
 * 128 keys takes one block and one 192 / 288-byte partial: 2^25 such sets); larger jobs return BGLS_ERR_ARG and are cut by
 * the caller. */
int bgls_verify_multi_batch(int curve, const uint8_t* sigs, const uint8_t* keys, const uint64_t* key_off, size_t n_sets,
                            const uint8_t* msg_blob, const uint64_t* msg_off, int allow_duplicates);
/* AggregatePoints (curves/curve.go:73-121) over n_sets sets in one pass: out[b] = sum of pts[set_off[b] .. set_off[b+1])
 * (an empty set gives the point at infinity).  What the n_sets AggregateKeys calls of blsKosk.go:128-131 cost. */
int bgls_aggregate_sets(int curve, int group, const uint8_t* pts, const uint64_t* set_off, size_t n_sets, uint8_t* out);

/* CurveSystem.PairingProduct (curves/altbn128.go:143-145, curves/bls12_381.go:238-240 ->
 * concurrentPairingProduct, curves/curve.go:125-170): gt_out = prod_i e(g1s[i], g2s[i]). */
int bgls_pairing_product(int curve, const uint8_t* g1s, const uint8_t* g2s, size_t n, uint8_t* gt_out);

/* CurveSystem.HashToG1 over a batch (curves/altbn128.go:509-513, curves/bls12_381.go:349-351;
 * the per-message goroutines of bgls/bgls.go:107-111,134-137). g1_out: n G1 points. */
int bgls_hash_to_g1(int curve, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n, uint8_t* g1_out);

/* curves.AggregatePoints (curves/curve.go:73-121): out = sum of n points of `group`.
 * n == 0 yields infinity (the reference spins forever there, curves/curve.go:94-108). */
int bgls_aggregate_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* out);

/* curves.ScalePoints (curves/curve.go:190-214) -> Point.Mul (curves/altbn128.go:107-121,235-249;
 * curves/bls12_381.go:65-76,126-137): out[i] = k_i * pts[i].  scalars: n x 32-byte big-endian
 * magnitudes; signs: NULL (all non-negative) or n bytes, 0 = +, 1 = negative (negate-then-mul),
 * 2 = nil factor (copy the point). */
int bgls_scale_points(int curve, int group, const uint8_t* pts, const uint8_t* scalars, const uint8_t* signs,
                      size_t n, uint8_t* out);

/* ---- hashed aggregation exponents (bgls/blsHAE.go) and multiplicities (bgls/blsKosk.go:137-150) ---- */
/* hashPubKeysToExponents (bgls/blsHAE.go:80-93): t_out = n 16-byte big-endian exponents read from BLAKE2Xb
 * (golang.org/x/crypto/blake2b NewXOF(16 n, nil)) over MarshalUncompressed(pk_0) || ... || pk_{n-1}, i.e. over the
 * n x bgls_g2_size bytes of `keys` (curves/altbn128.go:223-225; BLS12-381's upstream layout is unpinned).
 * The root digest is one sequential compression chain over all key bytes (host side of the boundary); the XOF
 * expansion runs on the device.  n < 2^28 (the XOF length is a uint32). */
int bgls_hae_exponents(int curve, const uint8_t* keys, size_t n, uint8_t* t_out);
/* AggregateSignaturesWithHAE (bgls/blsHAE.go:39-46): out = sum_i t_i * sigs[i] (G1); the caller checks the lengths. */
int bgls_aggregate_signatures_hae(int curve, const uint8_t* sigs, const uint8_t* keys, size_t n, uint8_t* out);
/* VerifyMultiSignatureWithHAE (bgls/blsHAE.go:56-58) -> getAggregatePubKey (:74-77): apk = sum_i t_i * keys[i], then
 * VerifySingleSignature (bgls/bgls.go:59-70). */
int bgls_verify_multi_hae(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg,
                          size_t msg_len);
/* VerifyAggregateSignatureWithHAE (bgls/blsHAE.go:49-53): keys scaled by their exponents (ScalePoints), then
 * verifyAggSig with duplicate messages allowed. */
int bgls_verify_aggregate_hae(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob,
                              const uint64_t* msg_off, size_t n);
/* getAggregatePubKey over device-resident inputs (bgls/blsHAE.go:74-77 = AggregatePoints(ScalePoints(points, w)),
 * curves/curve.go:73-121,190-214): d_out (affine bytes of the group) = sum_i w_i P_i, weights = n 16-byte big-endian
 * magnitudes.  Computed by the bucket method (k_msm.hip: a counting sort of the (point, window) pairs by digit, one
 * thread per bucket, running sums) when n >= the threshold below and the digits are balanced, else by one
 * double-and-add per point; both give the same bytes. */
int bgls_weighted_sum_dev(int curve, int group, const void* d_pts, const void* d_w16, size_t n, void* d_out, void* stream);
/* Smallest n for which weighted sums take the bucket method (default 32; tests pin both paths with 0 / SIZE_MAX). */
int bgls_set_msm_min(size_t n);
/* verifyMultiSignature over ScalePoints(keys, multiplicity) -- the body of KoskVerifyMultiSignatureWithMultiplicity
 * (bgls/blsKosk.go:137-150; that function prepends 0x01 to msg like KoskVerifyMultiSignature, the host mirror does
 * the same).  multiplicity: n int64 factors, negative = negate-then-multiply (curves/curve.go:190-214); NULL = plain
 * bgls_verify_multi. */
int bgls_verify_multi_multiplicity(int curve, const uint8_t* sig, const uint8_t* keys, const int64_t* multiplicity,
                                   size_t n, const uint8_t* msg, size_t msg_len);

/* ---- batch key generation and signing ---------------------------------------------------- */
/* LoadPublicKey over a batch (bgls/bgls.go:40-43: curve.GetG2().Mul(sk)): out[i] = scalars[i] * generator of `group`
 * (BGLS_G2 for public keys).  scalars: n x 32-byte big-endian. */
int bgls_scale_generator(int curve, int group, const uint8_t* scalars, size_t n, uint8_t* out);
/* Sign over a batch (bgls/bgls.go:46-56: HashToG1(msg).Mul(sk)); KoskSign is this call with 0x01 prepended to every
 * message (bgls/blsKosk.go:73-77).  sigs_out: n G1 points. */
int bgls_sign_batch(int curve, const uint8_t* sks, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n,
                    uint8_t* sigs_out);

/* ---- compressed wire formats ------------------------------------------------------------- */
/* Point.Marshal over a batch: out = n compressed points.  Inputs are validated like every other point (BGLS_ERR_ENCODING).
 *   alt-bn128 (curves/altbn128.go:81-89 G1, :203-221 G2; the reference's OWN format): G1 = x (32-byte big-endian) with the top
 *     bit of byte 0 set iff 2y > q; G2 = x_im || x_re with the top bits set iff 2 y_im > q / 2 y_re > q; infinity = zeros.
 *   BLS12-381 (curves/bls12_381.go:54-62 G1, :115-123 G2): 48 / 96 bytes in the ebfull/pairing ("ZCash") layout, the layout the
 *     reference names as its target ("TODO Make this match ebfull/pairing marshalling"): G1 = x big-endian, G2 = x.c1 || x.c0;
 *     byte 0 bit 7 = compressed, bit 6 = infinity (every other bit zero), bit 5 = y is the lexicographically larger of
 *     {y, -y} (G2: compare c1, then c0).  PARITY UNPINNED against the un-vendored github.com/dis2/bls12 the reference links
 *     (no vector exists in the reference); pinned by the format's public known-answer values (the generators' encodings,
 *     tests/test_wire.py) and by round trips. */
int bgls_compress_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* out);
/* UnmarshalG1 / UnmarshalG2, compressed branches.  out = n uncompressed points (zeros where rejected), ok[i] = 1 / 0 = the
 * reference's (Point, bool).
 *   alt-bn128 (curves/altbn128.go:296-327, :329-376): square roots by calcQuadRes / calcComplexQuadRes (curves/hash.go:178-223),
 *     the component-wise sign rule, then the MakeG*Point validation (G2: subgroup membership included).
 *   BLS12-381 (curves/bls12_381.go:242-264, inputs of 48 / 96 bytes): the flag rules above (a clear compression flag, an
 *     infinity flag with any other bit set, x >= p are refused), y = sqrt(x^3 + b) selected by the sort flag, then Check():
 *     membership in the order-r subgroup, G1 and G2 alike. */
int bgls_decompress_points(int curve, int group, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok);

/* ---- per-point operations backing the Go Point / PointT methods --------------------------- */
/* Point.Add (curves/altbn128.go:59-66,181-188; curves/bls12_381.go:33-41,94-102) */
int bgls_point_add(int curve, int group, const uint8_t* a, const uint8_t* b, uint8_t* out);
/* MakeG1Point/MakeG2Point/Unmarshal* validation (curves/altbn128.go:42-57,157-179;
 * curves/bls12_381.go:196-264 pt.Check()): 1 if the coordinates are canonical, the point is on the curve AND in the
 * order-r subgroup (G2 on both curves; G1 on BLS12-381, whose E(Fp) has cofactor (x-1)^2/3 -- alt-bn128's G1 is the whole
 * curve), 0 otherwise. */
int bgls_point_check(int curve, int group, const uint8_t* a);
/* The same validation over a batch (what constructing n Points costs in the reference: MakeG1Point / MakeG2Point with
 * check, UnmarshalG1 / UnmarshalG2; curves/altbn128.go:149-179,296-376, curves/bls12_381.go:196-264): ok_out[i] = 1 iff
 * point i has canonical coordinates, lies on its curve and in the order-r subgroup: G2 on both curves (endomorphism
 * criterion, exact), G1 on BLS12-381 ([r]P = infinity; alt-bn128's G1 is the whole curve).  The pairing value itself does
 * not see a G1 point's cofactor-order component (e(P + T, Q) = e(P, Q)), which is exactly why such points must be refused
 * at construction: sigma + T would be a second valid encoding of a signature, and scalars reduced mod r act wrongly on it.
 * Returns 0 or < 0. */
int bgls_check_points(int curve, int group, const uint8_t* pts, size_t n, uint8_t* ok_out);
/* GetG1 / GetG2 (curves/altbn128.go:423-429, curves/bls12_381.go:275-281) */
int bgls_generator(int curve, int group, uint8_t* out);
/* CurveSystem.Pair (curves/altbn128.go:130-141, curves/bls12_381.go:228-236) */
int bgls_pair(int curve, const uint8_t* g1, const uint8_t* g2, uint8_t* gt_out);
/* PointT.Add = Fp12 multiplication (curves/altbn128.go:264-271, curves/bls12_381.go:160-168) */
int bgls_gt_mul(int curve, const uint8_t* a, const uint8_t* b, uint8_t* out);
/* PointT.Mul (curves/altbn128.go:273-281, curves/bls12_381.go:170-173): gt^k, k a 32-byte big-endian magnitude,
 * negative != 0 for k < 0 (the inverse of a GT element is its conjugate: GT elements are unitary). */
int bgls_gt_pow(int curve, const uint8_t* gt, const uint8_t* k_be32, int negative, uint8_t* out);
/* GetGTIdentity (curves/altbn128.go:441-443,478; curves/bls12_381.go:295-297,341) */
int bgls_gt_identity(int curve, uint8_t* out);

/* ---- device-resident variants (inputs already in HBM; used by bench.py and multi-GPU) ------ */
/* All pointers prefixed d_ are device pointers on the current device; `stream` is a hipStream_t
 * (NULL = the context's own stream).  Calls are asynchronous unless they return a verdict. */

/* Partial Miller product of one shard: d_partial_out (bgls_gt_size bytes, GT wire format of the
 * UN-exponentiated Fp12 value) = [miller(-sig, g2) if d_sig != NULL] * prod_i miller(H(m_i), pk_i).
 * Messages are fixed-stride: message i is d_msgs[i*msg_stride .. i*msg_stride + msg_len).
 * check_duplicates != 0 runs the exact duplicate-message scan on the device; *d_flags (one
 * uint32 in HBM, bit 0 = duplicate found, bit 1 = bad encoding, bit 2 = hash failure) is OR-ed. */
int bgls_miller_product_dev(int curve, const void* d_sig, const void* d_keys, const void* d_msgs,
                            size_t msg_len, size_t msg_stride, size_t n, int check_duplicates,
                            void* d_partial_out, void* d_flags, void* stream);

/* containsDuplicateMessage (bgls/bgls.go:139-150) on its own: sets bit 0 of *d_flags when two of the n device-resident
 * fixed-stride messages are byte-identical.  The multi-GPU path runs it over the all-gathered messages of every shard
 * (a duplicate may straddle two shards); the per-shard scan inside bgls_miller_product_dev covers one GPU. */
int bgls_duplicate_scan_dev(const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n, void* d_flags,
                            void* stream);
/* The same scan restricted to ONE bucket of the records: only those whose first byte is `bucket` mod n_buckets (bucket < n_buckets
 * <= 256) enter the table.  Equal records share a bucket, so rank r of an N-rank verification scanning bucket r of the all-gathered
 * 16-byte digests finds, between the ranks, exactly what one scan of everything finds -- with 1/N of the inserts each, so the scan
 * scales with the number of GPUs (the OR of the ranks' words travels with their status words).  The table is sized for twice the
 * bucket's fair share; if it fills up (records that are not digests, or chosen by an adversary) bit 0 is set: callers treat a hit
 * as "undecided" and settle it with the exact scan over the messages, so this can cost time but never miss a duplicate. */
int bgls_duplicate_scan_bucket_dev(const void* d_recs, size_t rec_len, size_t rec_stride, size_t n, unsigned bucket,
                                   unsigned n_buckets, void* d_flags, void* stream);
/* The digest exchange as an all-to-all by bucket (round 6; bgls_amd/sharding.py enqueue_digest_probe(exchange="all_to_all")).  An
 * all-gather hands all N x n digests to every rank, which then discards (N - 1) / N of them; here rank r receives only the digests it
 * owns (first byte = r mod N).  bgls_digest_pack_dev sorts a rank's n digests (16 bytes each, as bgls_message_digests_dev writes
 * them) into n_buckets slots of `cap` records each in d_out (n_buckets x cap x 16 bytes): slot b is what goes to rank b.  Unused
 * records are padding that belongs to another bucket, so all chunks have one size and no counts are exchanged.  A slot that would
 * overflow sets bit 0 of *d_flags ("undecided": settled by the exact scan, like any digest hit; cap = 1.25 x n / n_buckets + 1024 never
 * overflows on real digests).  bgls_duplicate_scan_packed_dev is the receiving half: the exact scan of the n_slots = n_buckets x cap
 * records a rank holds after the exchange (its own bucket's digests; the padding is skipped), table sized for all of them.
 * Replaces nothing in the reference (one process there): containsDuplicateMessage, bgls/bgls.go:139-150, across GPUs. */
int bgls_digest_pack_dev(const void* d_digests16, size_t n, unsigned n_buckets, size_t cap, void* d_out, void* d_flags, void* stream);
int bgls_duplicate_scan_packed_dev(const void* d_recs16, size_t n_slots, unsigned bucket, unsigned n_buckets, void* d_flags, void* stream);
/* 16-byte digests (the first 16 bytes of BLAKE2b-512) of n device-resident fixed-stride messages, to d_out16 (n x 16 bytes,
 * 16-byte aligned).  What the ranks of a multi-GPU verification exchange for the global containsDuplicateMessage rule
 * (bgls/bgls.go:139-150) instead of the messages themselves: no two equal digests => no two equal messages; a pair of equal
 * digests is settled by the exact scan over the messages (bgls_amd/sharding.py global_duplicate_scan). */
int bgls_message_digests_dev(const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n, void* d_out16,
                             void* stream);

/* Multiply `count` partial products (count x bgls_gt_size bytes, e.g. the all-gathered shards),
 * apply the single shared final exponentiation and compare with 1.  Returns 1 / 0 / < 0.
 * d_flags (may be NULL) is read: any bit set forces 0 (duplicate) or the matching error. */
int bgls_final_verify_dev(int curve, const void* d_partials, size_t count, const void* d_flags,
                          void* stream);

/* The same, split in two so that a second verification can be enqueued while this one's serial tail (reduction, final
 * exponentiation) is still running: submit returns at once, collect waits and returns 1 / 0 / < 0.  One verification in
 * flight per context. */
int bgls_final_verify_submit_dev(int curve, const void* d_partials, size_t count, const void* d_flags, void* stream);
int bgls_final_verify_collect(int curve);
/* Throughput mode: tells the engine that several verifications are in flight, so that Miller launches keep the block form
 * that is fastest per pairing (60 pairings per block) even where a launch's last round of resident blocks is nearly empty --
 * the neighbours fill it -- and a multi-signature's key sum runs at one wave per SIMD so that the other checks' latency-bound tails find
 * wave slots.  Results are identical in every mode.  Off by default; BGLS_THROUGHPUT=1 turns it on from the environment.  on = 2: one stream
 * per verification (no fork onto the context's side stream) but the launch shapes of a verification that has the machine to itself -- for
 * timing ONE verification stage by stage. */
int bgls_set_throughput_mode(int on);
/* Environment switches, each read ONCE per process (four in all).  None changes a result: they select between kernels that
 * compute the same bytes and exist for A/B measurements and for the legacy-path tests (tests/test_gpu_legacy_paths.py,
 * tests/test_gpu_x60.py).
 *   BGLS_THROUGHPUT=1     = bgls_set_throughput_mode(1)
 *   BGLS_MILLER_SHAPE=4|5 = bgls_set_miller_shape(shape, 0) below
 *   BGLS_LEGACY=<mask>    the one 32-bit-limb fallback kept per stage: 1 final exponentiation (k_final36), 2 latency Miller loop
 *                         (k_miller_lat), 4 epilogue (k_cofactor_epilogue), 8 G2 key sums (k_sum_main), 16 key-sum tree as one launch
 *                         per level, 32 BLS12-381 G1 scalar multiplications on 32-bit limbs, 64 every reduce pass on k_reduce_coop
 *   BGLS_NO_RCCL=1        host exchange between the devices of a key set instead of RCCL */
/* Shape of the Miller stage.  0 (default): automatic -- up to 128 pairings the latency kernel; above, k_miller_x60 (carry-free
 * limbs: nine of 29 bits on alt-bn128, fourteen of 28 on BLS12-381; lane-pair point steps) with 60 pairings per block, or with 64
 * per block where that saves a nearly empty last round of the 1024 resident blocks outside throughput mode (61 441..65 536
 * pairings: exactly 2^16 is one round).  4: k_miller_x60 for every batch; `mode` is then a development mode word, validated: bits
 * 0-1 role placement (0 by SIMD id, 1 wave 2 consumes, 2 rotate by block), bit 2 consumer priority, bit 3 producer priority, bit 4
 * the 64-pairing block form (clear: the 60-pairing form); values above 31 are rejected.  5: the 32-bit fused kernel
 * (k_miller_ab64) for every batch.  `mode` is ignored for shapes 0 and 5; any other shape is BGLS_ERR_ARG (shapes 1-3, the Miller loop
 * as two kernels joined through a line table in HBM, were removed in round 5: never faster than the fused kernels).
 * Results (partial products, GT bytes, verdicts) are identical for every shape. */
int bgls_set_miller_shape(int shape, int mode);
/* Contexts 0..15: each owns a HIP stream, its device workspaces and stage timers; the calling thread works on the one it
 * selected (default 0).  Two contexts let one host thread keep two verifications in flight (bench.py). */
int bgls_select_context(int index);

/* Device-resident key sum for the multisig path: d_out = projective partial sum of n G2 keys,
 * serialised as affine G2 bytes (bgls_g2_size).  Shards combine with bgls_aggregate_points. */
int bgls_aggregate_points_dev(int curve, int group, const void* d_pts, size_t n, void* d_out, void* stream);

/* bgls_verify_multi_batch with everything on the device: d_key_off = n_sets + 1 uint64 offsets (from 0), max_set = the
 * largest set's size, messages at a fixed stride.  _submit_dev enqueues only (collect with bgls_final_verify_collect). */
int bgls_verify_multi_batch_dev(int curve, const void* d_sigs, const void* d_keys, const void* d_key_off, size_t n_sets, size_t max_set,
                                const void* d_msgs, size_t msg_len, size_t msg_stride, int allow_duplicates, void* stream);
int bgls_verify_multi_batch_submit_dev(int curve, const void* d_sigs, const void* d_keys, const void* d_key_off, size_t n_sets, size_t max_set,
                                       const void* d_msgs, size_t msg_len, size_t msg_stride, int allow_duplicates, void* stream);
/* verify_multi with keys already on the device. */
int bgls_verify_multi_dev(int curve, const void* d_sig, const void* d_keys, size_t n, const void* d_msg,
                          size_t msg_len, void* stream);
/* The same without waiting: the verdict is collected with bgls_final_verify_collect on the same context. */
int bgls_verify_multi_submit_dev(int curve, const void* d_sig, const void* d_keys, size_t n, const void* d_msg,
                                 size_t msg_len, void* stream);

/* ---- measurement hooks (bench.py; not part of the reference's interface) ------------------ */
/* ---- device-resident key sets (opaque handles so keys uploaded once are verified many times) -------------
 * A key set is n G2 public keys, parsed and validated once, resident in HBM as Montgomery-form affine points next to
 * their wire bytes, cut into contiguous ranges over n_devices GPUs (devices[] lists them; NULL = 0 .. n_devices-1; an id
 * may repeat, which is how the multi-device code is exercised on a one-GPU box).  This is what the Go shim's
 * altbn128Point2 / bls12Point2 slices become (curves/altbn128.go:19-29, curves/bls12_381.go:18-28): the shim uploads a
 * []Point once, keeps the handle (runtime.SetFinalizer -> bgls_keys_free) and passes it to every Verify*.
 * flags: BGLS_KEYS_CHECK also requires every key to lie in the order-r subgroup (the reference's construction-time
 * check); any invalid key fails the upload with BGLS_ERR_ENCODING.
 * BGLS_KEYS_PREPARE additionally walks the G2 point steps of the Miller loop once per key and keeps every step's line
 * function in HBM (13-17 KB per key: a 2^20-key set is 14-18 GB of the 288), so that verifications against the set only
 * scale and fold those lines (prepared.hpp): the fixed-argument precomputation pairing libraries offer as "prepared G2".
 * Verdicts and the final GT element are identical to the unprepared path. */
typedef uint64_t bgls_keys_t;
#define BGLS_KEYS_CHECK 1u
#define BGLS_KEYS_PREPARE 2u
int bgls_keys_upload(int curve, const uint8_t* keys, size_t n, const int* devices, int n_devices, unsigned flags, bgls_keys_t* handle_out);
/* Lifetime: a key set must not be freed while a call that uses it is in progress or has device work outstanding (the *_dev
 * entry points return before their kernels finish: synchronise the stream first); calls on one key set from several threads
 * are otherwise safe, each running on the calling thread's context. */
int bgls_keys_free(bgls_keys_t handle);
int bgls_keys_info(bgls_keys_t handle, int* curve, size_t* n, int* n_devices);
/* verifyAggSig (bgls/bgls.go:94-119) against a resident key set: message i belongs to key i; n must equal the set's size.
 * Every device of the set hashes its message range and multiplies its Miller values (one host thread per device), the
 * 384 / 576-byte partial products and status words meet on the first device (ncclAllGather over the devices' streams
 * when RCCL is usable and the devices are distinct, peer copies otherwise), which runs the single final exponentiation.
 * Same GT element, hence the same verdict, for any number of devices. */
int bgls_verify_aggregate_h(bgls_keys_t handle, const uint8_t* sig, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n,
                            int allow_duplicates);
/* The same verification, also returning the GT element e(-sig, g2) * prod_i e(H(m_i), pk_i) (the PairingProduct value of
 * bgls/bgls.go:113-114; the identity iff the verdict is 1): canonical bytes, identical for any number of devices. */
int bgls_verify_aggregate_h_gt(bgls_keys_t handle, const uint8_t* sig, const uint8_t* msg_blob, const uint64_t* msg_off, size_t n,
                               int allow_duplicates, uint8_t* gt_out);
/* verifyMultiSignature (bgls/bgls.go:89-92) against a ONE-device key set, signature and message already on the device, on the
 * calling thread's context and the given stream: the key sum reads the set's resident sum-ready records -- the keys are the
 * reference's already-constructed Points (parsed and validated at upload), so nothing but the n - 1 additions of
 * AggregatePoints (curves/curve.go:73-121) is left per key.  _submit_ enqueues and returns; the verdict is collected with
 * bgls_final_verify_collect(curve) on the same context, as for bgls_verify_multi_submit_dev. */
int bgls_verify_multi_keys_dev(bgls_keys_t handle, const void* d_sig, const void* d_msg, size_t msg_len, void* stream);
int bgls_verify_multi_keys_submit_dev(bgls_keys_t handle, const void* d_sig, const void* d_msg, size_t msg_len, void* stream);
/* bgls_miller_product_dev against a ONE-device key set (prepared or not) with device-resident fixed-stride messages, on
 * the calling thread's context and the given stream; finish with bgls_final_verify_(submit_)dev. */
int bgls_miller_product_keys_dev(bgls_keys_t handle, const void* d_sig, const void* d_msgs, size_t msg_len, size_t msg_stride, size_t n,
                                 int check_duplicates, void* d_partial_out, void* d_flags, void* stream);
/* verifyMultiSignature (bgls/bgls.go:89-92) against a resident key set: per-device partial key sums (projective G2
 * points) are gathered on the first device, added, and the two-pairing check runs there. */
int bgls_verify_multi_h(bgls_keys_t handle, const uint8_t* sig, const uint8_t* msg, size_t msg_len);
/* Contexts: a key-set verification runs shard s on context (selected + s) mod 16 of the shard's device, `selected` being the
 * calling thread's bgls_select_context (default 0); the final product / exponentiation runs on shard 0's context. */
/* The same two calls with host keys, for callers without a resident set: upload (with BGLS_KEYS_CHECK: the keys have not been
 * through a Point constructor, so subgroup membership is checked here; a key outside G2 gives BGLS_ERR_ENCODING), verify, free. */
int bgls_verify_aggregate_multi(int curve, const uint8_t* sig, const uint8_t* keys, const uint8_t* msg_blob, const uint64_t* msg_off,
                                size_t n, int allow_duplicates, const int* devices, int n_devices);
int bgls_verify_multi_multi(int curve, const uint8_t* sig, const uint8_t* keys, size_t n, const uint8_t* msg, size_t msg_len,
                            const int* devices, int n_devices);
/* Which exchange the last multi-device call on this thread used: 0 none (one device), 1 peer / device copies, 2 RCCL. */
int bgls_last_exchange(void);
/* 1 if librccl was found and its entry points resolved (dlopen at first use), else 0. */
int bgls_rccl_available(void);

/* Per-stage device time, measured with HIP events on the stream the kernels are launched on.
 * Stages: "dup_check", "h2c", "miller", "reduce", "final_exp", "sum_points" (one scope per key sum: main pass, tree and
 * conversion), "sum_main" (the main-pass kernel of a key sum alone, nested in "sum_points"). */
int bgls_profile_enable(int on); /* also resets the counters */
int bgls_profile_get(const char* stage, double* total_ms, unsigned long long* launches);
/* Measured peak of dependent-free v_mad_u64_u32 chains on this GPU, in 32x32->64 MAC/s:
 * the roofline denominator for the integer-multiply-bound kernels. */
int bgls_probe_mad_peak(double* mac_per_s);
/* Self-test of the ABI's exception barrier, usable without a device: raises a C++ exception of the given kind inside the
 * library (0 bad_alloc, 1 length_error, 2 system_error, 3 runtime_error, 4 a non-standard object, 5 an unservable std::vector,
 * 6 bad_alloc on a shard's host thread) and returns the error code the barrier maps it to (BGLS_ERR_NOMEM / _HIP / _ARG).
 * No entry point lets an exception unwind into the caller (the reference never panics: curves/curve.go:15-22). */
int bgls_selftest_exception_barrier(int kind);

#ifdef __cplusplus
}
#endif
#endif /* BGLS_HIP_H */
