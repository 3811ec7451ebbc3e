"""Development soak test (GPU): random batch sizes / message lengths / corruptions on both curves; every verdict is
checked against the expectation (valid -> 1, any single corruption -> 0) and, for small n, against the C oracle.
  python tools/gpu_soak.py [seconds]"""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bgls_amd import _lib
from oracle import coracle
L = _lib.load(); assert L.bgls_init(0) == 0
B = lambda b: (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = random.Random(int(time.time()))
t0 = time.time(); runs = 0; oracle_checks = 0; multis = 0; prepared_checks = 0


def offsets(msgs):
    off = (ctypes.c_uint64 * (len(msgs) + 1))(); acc = 0
    for i, m in enumerate(msgs):
        off[i] = acc; acc += len(m)
    off[len(msgs)] = acc
    return off


while time.time() - t0 < budget:
    cid = rnd.randrange(2); fp = 32 if cid == 0 else 48
    n = rnd.choice([1, 2, 3, 5, 6, 7, 59, 60, 61, 63, 64, 65, 127, 128, 129, 255, 256, 257, rnd.randrange(1, 700), rnd.randrange(1, 4000)])
    mlen = rnd.choice([8, 9, 31, 32, 33, 64, 100, 200])
    msgs = [rnd.randbytes(mlen) for _ in range(n)]
    if len(set(msgs)) != n:
        continue
    sks = [rnd.randrange(1, 1 << 250) for _ in range(n)]
    kb = b"".join(s.to_bytes(32, "big") for s in sks)
    keys = (ctypes.c_uint8 * (n * 4 * fp))(); assert L.bgls_scale_generator(cid, 2, B(kb), n, keys) == 0
    sigs = (ctypes.c_uint8 * (n * 2 * fp))(); assert L.bgls_sign_batch(cid, B(kb), B(b"".join(msgs)), offsets(msgs), n, sigs) == 0
    agg = (ctypes.c_uint8 * (2 * fp))(); assert L.bgls_aggregate_points(cid, 1, sigs, n, agg) == 0
    blob = b"".join(msgs); off = offsets(msgs)
    ok = L.bgls_verify_aggregate(cid, agg, keys, B(blob), off, n, 0)
    assert ok == 1, ("valid rejected", cid, n, mlen)
    kind = rnd.randrange(4)
    if kind == 0:
        bad = bytearray(blob); bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
        r = L.bgls_verify_aggregate(cid, agg, keys, B(bytes(bad)), off, n, 1)
    elif kind == 1 and n > 1:
        i, k = rnd.sample(range(n), 2); kk = bytearray(bytes(keys)); s = 4 * fp
        kk[i * s:(i + 1) * s], kk[k * s:(k + 1) * s] = bytes(keys)[k * s:(k + 1) * s], bytes(keys)[i * s:(i + 1) * s]
        r = L.bgls_verify_aggregate(cid, agg, B(bytes(kk)), B(blob), off, n, 0)
    elif kind == 2:
        i = rnd.randrange(n)
        r = L.bgls_verify_aggregate(cid, B(bytes(sigs)[i * 2 * fp:(i + 1) * 2 * fp]), keys, B(blob), off, n, 0) if n > 1 else 0
    else:
        dup = list(msgs); dup[-1] = dup[0]
        r = L.bgls_verify_aggregate(cid, agg, keys, B(b"".join(dup)), offsets(dup), n, 0) if n > 1 else 0
    assert r == 0, ("corruption accepted", cid, n, mlen, kind)
    # ---- the same instance against a resident key set, plain and PREPARED (round 6: the fold on the Miller kernel's carry-free limbs)
    if runs % 4 == 1:
        for flags in (1, 3):                  # BGLS_KEYS_CHECK, + BGLS_KEYS_PREPARE
            h = ctypes.c_uint64()
            assert L.bgls_keys_upload(cid, keys, n, (ctypes.c_int * 1)(0), 1, flags, ctypes.byref(h)) == 0
            assert L.bgls_verify_aggregate_h(h, agg, B(blob), off, n, 0) == 1, ("valid rejected by the key set", cid, n, flags)
            bad2 = bytearray(blob); bad2[rnd.randrange(len(bad2))] ^= 1 << rnd.randrange(8)
            assert L.bgls_verify_aggregate_h(h, agg, B(bytes(bad2)), off, n, 1) == 0, ("corruption accepted by the key set", cid, n, flags)
            assert L.bgls_keys_free(h) == 0
        prepared_checks += 1
    if n <= 40:
        assert coracle.verify_aggregate(cid, bytes(agg), bytes(keys), msgs, threads=8) == 1
        oracle_checks += 1
    runs += 1
    # ---- multi-signatures on one message (verifyMultiSignature, bgls/bgls.go:89-92) and the batched form (blsKosk.go:126-133):
    # key sets of random sizes around the key-sum kernels' tile boundaries, keys repeated inside a set (the doubling branch)
    if runs % 3 == 0:
        nm = rnd.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, rnd.randrange(1, 3000)])
        pool = min(nm, rnd.choice([1, 2, 7, 64, nm]))
        psk = [rnd.randrange(1, 1 << 250) for _ in range(pool)]
        pkb = b"".join(x.to_bytes(32, "big") for x in psk)
        pkeys = (ctypes.c_uint8 * (pool * 4 * fp))(); assert L.bgls_scale_generator(cid, 2, B(pkb), pool, pkeys) == 0
        idx = [rnd.randrange(pool) for _ in range(nm)]
        mkeys = b"".join(bytes(pkeys)[i * 4 * fp:(i + 1) * 4 * fp] for i in idx)
        m = rnd.randbytes(rnd.choice([1, 32, 77]))
        sk_sum = sum(psk[i] for i in idx)
        hm = (ctypes.c_uint8 * (2 * fp))(); assert L.bgls_hash_to_g1(cid, B(m), (ctypes.c_uint64 * 2)(0, len(m)), 1, hm) == 0
        order = [21888242871839275222246405745257275088548364400416034343698204186575808495617,
                 52435875175126190479447740508185965837690552500527637822603658699938581184513][cid]
        msig = coracle.scale_point(cid, 1, bytes(hm), sk_sum % order)
        assert L.bgls_verify_multi(cid, B(msig), B(mkeys), nm, B(m), len(m)) == 1, ("valid multisig rejected", cid, nm, pool)
        m2 = bytearray(m); m2[0] ^= 2
        assert L.bgls_verify_multi(cid, B(msig), B(mkeys), nm, B(bytes(m2)), len(m)) == 0, ("multisig on another message accepted", cid, nm)
        if nm > 1 and pool > 1:
            other = bytearray(mkeys); j = rnd.randrange(nm)
            repl = bytes(pkeys)[((idx[j] + 1) % pool) * 4 * fp:((idx[j] + 1) % pool + 1) * 4 * fp]
            other[j * 4 * fp:(j + 1) * 4 * fp] = repl
            assert L.bgls_verify_multi(cid, B(msig), B(bytes(other)), nm, B(m), len(m)) == 0, ("multisig with a swapped key accepted", cid, nm)
        # batched: two to five such sets, each on its own message
        nsets = rnd.randrange(2, 6)
        sets, sigs_b, msgs_b = [], [], []
        for _ in range(nsets):
            ns = rnd.choice([1, 2, 33, 64, rnd.randrange(1, 400)])
            ix = [rnd.randrange(pool) for _ in range(ns)]
            mb = rnd.randbytes(24)
            hb = (ctypes.c_uint8 * (2 * fp))(); assert L.bgls_hash_to_g1(cid, B(mb), (ctypes.c_uint64 * 2)(0, 24), 1, hb) == 0
            sigs_b.append(coracle.scale_point(cid, 1, bytes(hb), sum(psk[i] for i in ix) % order))
            sets.append(b"".join(bytes(pkeys)[i * 4 * fp:(i + 1) * 4 * fp] for i in ix)); msgs_b.append(mb)
        koff = (ctypes.c_uint64 * (nsets + 1))(); acc = 0
        for i, st in enumerate(sets):
            koff[i] = acc; acc += len(st) // (4 * fp)
        koff[nsets] = acc
        okb = L.bgls_verify_multi_batch(cid, B(b"".join(sigs_b)), B(b"".join(sets)), koff, nsets, B(b"".join(msgs_b)), offsets(msgs_b), 1)
        assert okb == 1, ("valid batch of multisigs rejected", cid, nsets)
        sigs_b[0], sigs_b[1] = sigs_b[1], sigs_b[0]          # the aggregate of the signatures is unchanged: still valid
        assert L.bgls_verify_multi_batch(cid, B(b"".join(sigs_b)), B(b"".join(sets)), koff, nsets, B(b"".join(msgs_b)), offsets(msgs_b), 1) == 1
        msgs_b[0] = rnd.randbytes(24)
        assert L.bgls_verify_multi_batch(cid, B(b"".join(sigs_b)), B(b"".join(sets)), koff, nsets, B(b"".join(msgs_b)), offsets(msgs_b), 1) == 0
        multis += 1
print("multi-signature instances:", multis, " key-set instances (plain + prepared):", prepared_checks)
print("soak ok: %d instances (%d also checked by the oracle) in %.0f s" % (runs, oracle_checks, time.time() - t0))
