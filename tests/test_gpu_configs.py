"""GPU tier: BASELINE.json configs 4 and 5 and the headline record (alt-bn128 aggregate verify, 2^20 signers) at their full sizes.

config 4: alt-bn128 VerifyMultiSignature, 2^20 signers on one message (bgls/blsKosk_test.go:35-64 at scale)
config 5: BLS12-381 aggregate verify, 2^20 signers cut into 8 shards (the 8-GPU decomposition, run here on one GPU
          through the same per-shard entry point bench.py --gpus 8 uses; bgls/bgls_test.go:40-77 at scale)

The oracle cannot replay 2^20 pairings in test time, so parity at these sizes rests on what it CAN check exactly --
the aggregated key bytes (2^20 G2 additions, ~1 s of CPU), the GT product of the shard partials, the final
exponentiation of that product -- plus size-independent properties: a valid instance verifies, one signer less or one
flipped bit rejects, a duplicate message rejects even when the two copies sit in different shards, and the partial
product bytes do not depend on how the batch was cut."""
import ctypes
import random

import pytest

from oracle import coracle

pytestmark = pytest.mark.gpu

ORDER = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
         1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}
N20 = 1 << 20


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def out(n):
    return (ctypes.c_uint8 * max(1, n))()


def gen_keys(lib, cid, fp, sks):
    n = len(sks)
    g2 = out(4 * fp)
    assert lib.bgls_generator(cid, 2, g2) == 0
    keys = out(n * 4 * fp)
    assert lib.bgls_scale_generator(cid, 2, B(b"".join(s.to_bytes(32, "big") for s in sks)), n, keys) == 0
    return keys


def test_config4_multisig_2pow20_altbn128(gpu_lib):
    import torch
    lib, cid, fp, n = gpu_lib, 0, 32, N20
    rnd = random.Random(0xB6150000 + 4)
    sks = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
    keys = gen_keys(lib, cid, fp, sks)
    msg = b"\x01" + rnd.randbytes(64)
    off = (ctypes.c_uint64 * 2)(0, len(msg))
    h = out(2 * fp)
    assert lib.bgls_hash_to_g1(cid, B(msg), off, 1, h) == 0
    assert bytes(h) == coracle.hash_to_g1(cid, msg)
    sig = coracle.scale_point(cid, 1, bytes(h), sum(sks) % ORDER[cid])
    dev = torch.device("cuda:0")
    t_keys = torch.frombuffer(bytearray(bytes(keys)), dtype=torch.uint8).to(dev)
    t_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).to(dev)
    t_msg = torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev)
    # AggregatePoints over all 2^20 keys: same bytes as the oracle's 2^20 - 1 additions
    apk = torch.zeros(4 * fp, dtype=torch.uint8, device=dev)
    assert lib.bgls_aggregate_points_dev(cid, 2, t_keys.data_ptr(), n, apk.data_ptr(), None) == 0
    apk_b = bytes(apk.cpu().numpy())
    assert apk_b == coracle.aggregate_points(cid, 2, bytes(keys), n)
    # ... and the same point as (sum sk) g2, computed independently
    g2 = out(4 * fp); lib.bgls_generator(cid, 2, g2)
    assert apk_b == coracle.scale_point(cid, 2, bytes(g2), sum(sks) % ORDER[cid])
    # verdicts: all signers / one signer missing / wrong message / host-buffer entry point
    assert lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_msg.data_ptr(), len(msg), None) == 1
    assert lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n - 1, t_msg.data_ptr(), len(msg), None) == 0
    bad = bytearray(msg); bad[17] ^= 2
    t_bad = torch.frombuffer(bad, dtype=torch.uint8).to(dev)
    assert lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_bad.data_ptr(), len(msg), None) == 0
    assert lib.bgls_verify_multi(cid, B(sig), keys, n, B(msg), len(msg)) == 1
    # the two-pairing tail agrees with the oracle on the aggregated key
    assert coracle.verify_multi(cid, sig, apk_b, 1, msg) == 1
    # the key sum's other launch shapes (round 6): throughput mode adds the 2^20 keys with one wave per SIMD (1024 waves), mode 2 is the lone
    # shape on one stream -- same verdicts
    try:
        for mode in (1, 2):
            assert lib.bgls_set_throughput_mode(mode) == 0
            assert lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_msg.data_ptr(), len(msg), None) == 1, mode
            assert lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n - 1, t_msg.data_ptr(), len(msg), None) == 0, mode
            assert lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_bad.data_ptr(), len(msg), None) == 0, mode
    finally:
        lib.bgls_set_throughput_mode(0)


@pytest.mark.parametrize("cid,fp", [(0, 32), (1, 48)])
def test_key_sum_tree_shapes_match_the_oracle(gpu_lib, cid, fp):
    """AggregatePoints (curves/curve.go:73-121) on G2 at sizes that give the one-launch tree (k_sumtree.hip) every shape: one
    partial, two, odd counts whose single nodes pass through levels unpaired, counts around the 128-key blocks of the main pass,
    the 3072-partial cap -- same bytes as the oracle's sequential additions and as (sum sk) g2; with the point at infinity and a
    repeated key (P + P in the tree: the doubling case of the group law) among the inputs; and the same sums with the tree as one
    launch per level (BGLS_LEGACY bit 16 is read once per process, so that form runs in the legacy-path tests)."""
    lib = gpu_lib
    rnd = random.Random(77 + cid)
    nmax = 70000
    sks = [rnd.randrange(1, ORDER[cid]) for _ in range(nmax)]
    keys = bytes(gen_keys(lib, cid, fp, sks))
    g2 = out(4 * fp); lib.bgls_generator(cid, 2, g2)
    PT = 4 * fp
    for n in (1, 2, 3, 127, 128, 129, 255, 257, 300, 385, 600, 641, 1000, 4097, 65535, 70000):
        got = out(PT)
        assert lib.bgls_aggregate_points(cid, 2, B(keys[:n * PT]), n, got) == 0
        assert bytes(got) == coracle.scale_point(cid, 2, bytes(g2), sum(sks[:n]) % ORDER[cid]), n
        if n <= 1000:
            assert bytes(got) == coracle.aggregate_points(cid, 2, keys[:n * PT], n), n
    # infinity among the keys, the same key twice in one pair of leaves and across blocks, a key and its negation
    n = 700
    pts = bytearray(keys[:n * PT])
    pts[5 * PT:6 * PT] = bytes(PT)                                   # infinity
    pts[129 * PT:130 * PT] = pts[128 * PT:129 * PT]                  # neighbours equal
    pts[400 * PT:401 * PT] = pts[10 * PT:11 * PT]                    # equal keys in different blocks
    neg = coracle.scale_point(cid, 2, bytes(pts[20 * PT:21 * PT]), ORDER[cid] - 1)
    pts[500 * PT:501 * PT] = neg                                     # P and -P
    got = out(PT)
    assert lib.bgls_aggregate_points(cid, 2, B(bytes(pts)), n, got) == 0
    assert bytes(got) == coracle.aggregate_points(cid, 2, bytes(pts), n)
    # a sum that IS the point at infinity (P + (-P), alone and spread over two blocks): the root's infinity path
    for pair in (bytes(pts[20 * PT:21 * PT]) + neg, bytes(pts[20 * PT:21 * PT]) + keys[:127 * PT] + neg + coracle.scale_point(cid, 2, bytes(g2), (ORDER[cid] - sum(sks[:127])) % ORDER[cid])):
        cnt = len(pair) // PT
        got = out(PT)
        assert lib.bgls_aggregate_points(cid, 2, B(pair), cnt, got) == 0
        assert bytes(got) == coracle.aggregate_points(cid, 2, pair, cnt), cnt
    # two blocks whose sums are equal: the tree's first addition is a doubling
    two = keys[:128 * PT] * 2
    got = out(PT)
    assert lib.bgls_aggregate_points(cid, 2, B(two), 256, got) == 0
    assert bytes(got) == coracle.scale_point(cid, 2, bytes(g2), 2 * sum(sks[:128]) % ORDER[cid])


def test_config5_bls12_2pow20_in_8_shards(gpu_lib):
    aggregate_2pow20(gpu_lib, 1, 48, 0xB6150000 + 5)


def test_headline_altbn128_aggregate_2pow20(gpu_lib):
    """THE headline workload of BASELINE.json's metric (bench.py's default record): alt-bn128 VerifyAggregateSignature over 2^20
    signers with distinct messages (bgls/bgls_test.go:186-202 at scale), seed 0xB6150000 + 2: valid -> 1, one flipped bit -> 0, a
    straddling duplicate -> 0 by the duplicate rule alone, partial-product bytes identical for 1 and 8 shards, and the oracle's
    final exponentiation of that product is the identity."""
    aggregate_2pow20(gpu_lib, 0, 32, 0xB6150000 + 2)


def aggregate_2pow20(gpu_lib, cid, fp, seed):
    import torch
    lib, n, shards = gpu_lib, N20, 8
    gtb = 12 * fp
    rnd = random.Random(seed)
    sks = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
    kb = b"".join(s.to_bytes(32, "big") for s in sks)
    keys = gen_keys(lib, cid, fp, sks)
    msgs = rnd.randbytes(64 * n)
    off = (ctypes.c_uint64 * (n + 1))(*range(0, 64 * (n + 1), 64))
    sigs = out(n * 2 * fp)
    assert lib.bgls_sign_batch(cid, B(kb), B(msgs), off, n, sigs) == 0
    for i in (0, 77777, n - 1):                                  # spot-check signatures against the oracle
        hi = coracle.hash_to_g1(cid, msgs[64 * i:64 * i + 64])
        assert bytes(sigs[2 * fp * i:2 * fp * (i + 1)]) == coracle.scale_point(cid, 1, hi, sks[i])
    agg = out(2 * fp)
    assert lib.bgls_aggregate_points(cid, 1, sigs, n, agg) == 0
    dev = torch.device("cuda:0")
    t_keys = torch.frombuffer(bytearray(bytes(keys)), dtype=torch.uint8).to(dev)
    t_sig = torch.frombuffer(bytearray(bytes(agg)), dtype=torch.uint8).to(dev)
    t_msgs = torch.frombuffer(bytearray(msgs), dtype=torch.uint8).to(dev)

    def run(t_m, nshards, t_s=t_sig, scan=True):
        """per-shard partial Miller products (rank 0 carries the signature pair) + the global duplicate scan, then the
        combine + final exponentiation: what bench.py --gpus N does with one shard per rank"""
        parts = torch.zeros(nshards * gtb, dtype=torch.uint8, device=dev)
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        if scan:
            assert lib.bgls_duplicate_scan_dev(t_m.data_ptr(), 64, 64, n, flags.data_ptr(), None) == 0
        for s in range(nshards):
            lo, hi = n * s // nshards, n * (s + 1) // nshards
            assert lib.bgls_miller_product_dev(cid, t_s.data_ptr() if s == 0 else None, t_keys.data_ptr() + lo * 4 * fp,
                                               t_m.data_ptr() + lo * 64, 64, 64, hi - lo, 0, parts.data_ptr() + s * gtb,
                                               flags.data_ptr(), None) == 0
        verdict = lib.bgls_final_verify_dev(cid, parts.data_ptr(), nshards, flags.data_ptr(), None)
        torch.cuda.synchronize()
        raw = bytes(parts.cpu().numpy())
        acc = raw[:gtb]
        for s in range(1, nshards):
            acc = coracle.gt_mul(cid, acc, raw[s * gtb:(s + 1) * gtb])
        return verdict, acc

    v1, p1 = run(t_msgs, 1)
    v8, p8 = run(t_msgs, shards)
    assert v1 == 1 and v8 == 1
    assert p1 == p8                                               # canonical partial product, however the batch is cut
    assert coracle.final_exp(cid, p8) == bytes(gtb - 1) + b"\x01"    # the oracle's final exponentiation lands on 1
    # one flipped message bit in shard 5
    bad = t_msgs.clone(); bad[64 * (5 * n // 8 + 1234) + 9] ^= 0x08
    torch.cuda.synchronize()
    assert run(bad, shards)[0] == 0
    # A duplicate message whose two copies straddle shards 3 | 4 (per-shard scans would miss it).  The instance is
    # re-signed so that the pairing equation HOLDS: only the duplicate rule (bgls/bgls.go:139-150) can reject it.
    a, b = 4 * n // 8 - 5, 4 * n // 8 + 11
    dup = t_msgs.clone()
    dup[64 * b:64 * b + 64] = dup[64 * a:64 * a + 64]
    h_a = coracle.hash_to_g1(cid, msgs[64 * a:64 * a + 64])
    old_b = bytes(sigs[2 * fp * b:2 * fp * (b + 1)])
    three = bytes(agg) + coracle.scale_point(cid, 1, old_b, ORDER[cid] - 1) + coracle.scale_point(cid, 1, h_a, sks[b])
    agg_dup = out(2 * fp)
    assert lib.bgls_aggregate_points(cid, 1, B(three), 3, agg_dup) == 0
    t_sig_dup = torch.frombuffer(bytearray(bytes(agg_dup)), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()
    assert run(dup, shards, t_sig_dup, scan=False)[0] == 1        # allowDuplicates = true would accept it
    assert run(dup, shards, t_sig_dup, scan=True)[0] == 0         # VerifyAggregateSignature does not
    # host-buffer door, whole batch in one call
    assert lib.bgls_verify_aggregate(cid, agg, keys, B(msgs), off, n, 0) == 1
    # the same batch against a PREPARED resident key set (round 6: the fold on the Miller kernel's carry-free limbs, 38 / 52 pairings per
    # group and squaring at this size -- one round of resident waves; the small sets of test_gpu_keys.py all run at 6): same verdicts, and the
    # partial product differs from the unprepared one only by factors the final exponentiation removes -- the oracle's says so
    h = ctypes.c_uint64()
    devs = (ctypes.c_int * 1)(0)
    assert lib.bgls_keys_upload(cid, keys, n, devs, 1, 2, ctypes.byref(h)) == 0          # BGLS_KEYS_PREPARE
    part = torch.zeros(gtb, dtype=torch.uint8, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    assert lib.bgls_miller_product_keys_dev(h, t_sig.data_ptr(), t_msgs.data_ptr(), 64, 64, n, 1, part.data_ptr(), flags.data_ptr(), None) == 0
    assert lib.bgls_final_verify_dev(cid, part.data_ptr(), 1, flags.data_ptr(), None) == 1
    torch.cuda.synchronize()
    assert coracle.final_exp(cid, bytes(part.cpu().numpy())) == bytes(gtb - 1) + b"\x01"
    flags.zero_()
    assert lib.bgls_miller_product_keys_dev(h, t_sig.data_ptr(), bad.data_ptr(), 64, 64, n, 1, part.data_ptr(), flags.data_ptr(), None) == 0
    assert lib.bgls_final_verify_dev(cid, part.data_ptr(), 1, flags.data_ptr(), None) == 0
    assert lib.bgls_keys_free(h) == 0


def test_bls12_hash_normalisation_batches_agree(gpu_lib):
    """BLS12-381 batches of 2^18 messages and more normalise the hash points four at a time with one shared inversion
    (k_bls_combine_raw_batched); smaller ones one at a time.  A ragged 2^18 + 3 batch in one call and the same batch cut at
    2^17 (two calls on the per-message kernel) give the same partial Miller product."""
    import torch
    lib, cid, fp = gpu_lib, 1, 48
    n = (1 << 18) + 3
    gtb = 12 * fp
    rnd = random.Random(0xB6150000 + 55)
    keys = gen_keys(lib, cid, fp, [rnd.randrange(1, ORDER[cid]) for _ in range(n)])
    dev = torch.device("cuda:0")
    t_keys = torch.frombuffer(bytearray(bytes(keys)), dtype=torch.uint8).to(dev)
    t_msgs = torch.frombuffer(bytearray(rnd.randbytes(64 * n)), dtype=torch.uint8).to(dev)

    def part(lo, hi):
        p = torch.zeros(gtb, dtype=torch.uint8, device=dev)
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        assert lib.bgls_miller_product_dev(cid, None, t_keys.data_ptr() + lo * 4 * fp, t_msgs.data_ptr() + lo * 64, 64, 64, hi - lo, 0,
                                           p.data_ptr(), flags.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert int(flags.cpu()[0]) == 0
        return bytes(p.cpu().numpy())

    whole = part(0, n)
    cut = coracle.gt_mul(cid, part(0, 1 << 17), part(1 << 17, n))
    assert whole == cut
