"""GPU tier: KoskVerifyBatchMultiSignature as ONE call (bgls_verify_multi_batch, bgls/blsKosk.go:126-133) and the segmented
key sums behind it (bgls_aggregate_sets): ragged set sizes around the sum kernels' tile sizes, against the C oracle's
AggregatePoints bytes and the oracle's verdicts; a wrong signer, swapped messages and an off-curve key are refused."""
import ctypes
import random

import pytest

from oracle import coracle

pytestmark = pytest.mark.gpu


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def out(n):
    return (ctypes.c_uint8 * max(1, n))()


def offs(counts):
    o = (ctypes.c_uint64 * (len(counts) + 1))()
    for i, c in enumerate(counts):
        o[i + 1] = o[i] + c
    return o


def instance(lib, cid, n_fp, sizes, seed):
    """len(sizes) multi-signatures: set b has sizes[b] signers on message b (keys and signatures made by the engine)"""
    rnd = random.Random(seed)
    n = sum(sizes)
    sks = [rnd.randrange(1, 1 << 250) for _ in range(n)]
    kb = b"".join(s.to_bytes(32, "big") for s in sks)
    keys = out(n * 4 * n_fp)
    assert lib.bgls_scale_generator(cid, 2, B(kb), n, keys) == 0
    msgs = [b"\x01" + rnd.randbytes(rnd.choice((8, 32, 64))) for _ in sizes]
    # every signer signs its set's message: sign_batch over (sk_i, msg of i's set)
    per = [m for m, c in zip(msgs, sizes) for _ in range(c)]
    sigs = out(n * 2 * n_fp)
    assert lib.bgls_sign_batch(cid, B(kb), B(b"".join(per)), offs([len(m) for m in per]), n, sigs) == 0
    aggs = out(len(sizes) * 2 * n_fp)
    assert lib.bgls_aggregate_sets(cid, 1, sigs, offs(sizes), len(sizes), aggs) == 0
    return bytes(keys), bytes(aggs), msgs, bytes(sigs)


def test_aggregate_sets_equals_oracle_sums(gpu_lib, curve):
    cid, n_fp = curve["id"], curve["fp"]
    sizes = [1, 2, 63, 64, 65, 0, 300, 7, 1025, 128]
    keys, aggs, msgs, sigs = instance(gpu_lib, cid, n_fp, sizes, 11 + cid)
    got = out(len(sizes) * 4 * n_fp)
    assert gpu_lib.bgls_aggregate_sets(cid, 2, B(keys), offs(sizes), len(sizes), got) == 0
    at = 0
    for b, c in enumerate(sizes):
        want = coracle.aggregate_points(cid, 2, keys[at * 4 * n_fp:(at + c) * 4 * n_fp], c) if c else bytes(4 * n_fp)
        assert bytes(got)[b * 4 * n_fp:(b + 1) * 4 * n_fp] == want, (b, c)
        wsig = coracle.aggregate_points(cid, 1, sigs[at * 2 * n_fp:(at + c) * 2 * n_fp], c) if c else bytes(2 * n_fp)
        assert aggs[b * 2 * n_fp:(b + 1) * 2 * n_fp] == wsig, (b, c)
        at += c


def test_batch_multi_signature_verdicts(gpu_lib, curve):
    cid, n_fp = curve["id"], curve["fp"]
    for sizes, seed in (([3], 1), ([1, 1], 2), ([5, 70, 2, 129, 64, 33], 3), ([40] * 70, 4)):
        keys, aggs, msgs, _ = instance(gpu_lib, cid, n_fp, sizes, seed)
        blob, moff, koff = b"".join(msgs), offs([len(m) for m in msgs]), offs(sizes)
        nb = len(sizes)
        assert gpu_lib.bgls_verify_multi_batch(cid, B(aggs), B(keys), koff, nb, B(blob), moff, 1) == 1, sizes
        # the oracle, stepwise as the reference writes it: aggregate signature, one key sum per set, aggregate verification
        aggsig = coracle.aggregate_points(cid, 1, aggs, nb)
        at, apks = 0, b""
        for c in sizes:
            apks += coracle.aggregate_points(cid, 2, keys[at * 4 * n_fp:(at + c) * 4 * n_fp], c)
            at += c
        if nb <= 8:
            assert coracle.verify_aggregate(cid, aggsig, apks, msgs, True, threads=8) == 1
        # one signer dropped from one set, two messages swapped
        if sum(sizes) > nb:
            b = max(range(nb), key=lambda i: sizes[i])
            at = sum(sizes[:b])
            kk = keys[:at * 4 * n_fp] + keys[(at + 1) * 4 * n_fp:]
            sz = list(sizes); sz[b] -= 1
            assert gpu_lib.bgls_verify_multi_batch(cid, B(aggs), B(kk), offs(sz), nb, B(blob), moff, 1) == 0
        if nb > 1 and msgs[0] != msgs[1]:
            sw = [msgs[1], msgs[0]] + msgs[2:]
            assert gpu_lib.bgls_verify_multi_batch(cid, B(aggs), B(keys), koff, nb, B(b"".join(sw)), offs([len(m) for m in sw]), 1) == 0
    # duplicates among the messages: refused without allow_duplicates, as verifyAggSig does (bgls/bgls.go:139-150)
    keys, aggs, msgs, _ = instance(gpu_lib, cid, n_fp, [2, 3], 9)
    same = [msgs[0], msgs[0]]
    assert gpu_lib.bgls_verify_multi_batch(cid, B(aggs), B(keys), offs([2, 3]), 2, B(b"".join(same)), offs([len(m) for m in same]), 0) == 0
    # an off-curve key anywhere is an encoding error, not a verdict
    bad = bytearray(keys); bad[4 * n_fp * 3 + 5] ^= 0x40
    assert gpu_lib.bgls_verify_multi_batch(cid, B(aggs), B(bytes(bad)), offs([2, 3]), 2, B(b"".join(msgs)), offs([len(m) for m in msgs]), 1) < 0


def test_many_small_sets_take_one_block_each(gpu_lib, curve):
    """Thousands of key sets of 1..3 keys (round-3 advisor finding: every set took 64 partial sums and two blocks whatever its
    size, so 2^16 sets needed 1.2 GB of workspace and 2^20 sets 19 GB).  The lane-pair kernel now gives a small set ONE block and
    ONE partial (no tree above it); the sums still equal the oracle's, set by set."""
    cid, n_fp = curve["id"], curve["fp"]
    rnd = random.Random(2024 + cid)
    base = 40
    sks = [rnd.randrange(1, 1 << 250) for _ in range(base)]
    kb = b"".join(s.to_bytes(32, "big") for s in sks)
    pool = out(base * 4 * n_fp)
    assert gpu_lib.bgls_scale_generator(cid, 2, B(kb), base, pool) == 0
    pool = [bytes(pool)[i * 4 * n_fp:(i + 1) * 4 * n_fp] for i in range(base)]
    nsets = 6000
    sizes = [rnd.choice((1, 1, 2, 3, 0)) for _ in range(nsets)]
    picks = [[rnd.randrange(base) for _ in range(c)] for c in sizes]
    keys = b"".join(pool[i] for p in picks for i in p)
    got = out(nsets * 4 * n_fp)
    assert gpu_lib.bgls_aggregate_sets(cid, 2, B(keys), offs(sizes), nsets, got) == 0
    got = bytes(got)
    memo = {}
    for b in list(range(0, nsets, 97)) + [nsets - 1]:
        key = tuple(sorted(picks[b]))
        if key not in memo:
            memo[key] = coracle.aggregate_points(cid, 2, b"".join(pool[i] for i in key), len(key)) if key else bytes(4 * n_fp)
        assert got[b * 4 * n_fp:(b + 1) * 4 * n_fp] == memo[key], (b, picks[b])
