"""GPU tier: the kernels that round 3 replaced stay selectable for A/B measurements (BGLS_FINALX=0: 36-lane final exponentiation
on 32-bit limbs, BGLS_LATX=0: k_miller_lat, BGLS_SUMX=0 / 1: key sums on 32-bit limbs / on one lane, BGLS_EPIX=0: k_cofactor_epilogue, BGLS_SUMTREE=0: k_sum_coop per level instead of k_sum_tree, BGLS_SUMTREEX=0: the tree's additions on 32-bit limbs (k_sum_tree) instead of k_sum_tree_x, BGLS_G1X=0: BLS12-381 Sign / ScalePoints / HashToG1 cofactor clearing on 32-bit limbs, BGLS_REDUCEX=0: every reduce pass on k_reduce_coop).  The switches are read once
per process, so each combination runs in a child process: PairingProduct of a handful of pairings (the latency path: k_miller_lat(x)
+ reduce + final exponentiation) must give the C oracle's GT bytes, and a 300-key aggregate of public keys the oracle's point."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, json, random, sys
sys.path.insert(0, %r)
from bgls_amd import _lib
from oracle import coracle
L = _lib.load()
assert L.bgls_init(0) == 0
B = lambda b: (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b))
gold = %r
res = {}
for cname, cid, fp in (("altbn128", 0, 32), ("bls12", 1, 48)):
    v = json.load(open(gold + "/vectors_" + cname + ".json"))
    g1 = bytes.fromhex(v["pairings"][3]["g1"]); g2 = bytes.fromhex(v["pairings"][3]["g2"])
    rnd = random.Random(5 + cid)
    n = 7
    g1s = [coracle.scale_point(cid, 1, g1, rnd.randrange(1, 1 << 250)) for _ in range(n)]
    g2s = [coracle.scale_point(cid, 2, g2, rnd.randrange(1, 1 << 250)) for _ in range(n)]
    a, b = b"".join(g1s), b"".join(g2s)
    o = (ctypes.c_uint8 * (12 * fp))()
    assert L.bgls_pairing_product(cid, B(a), B(b), n, o) == 0
    res[cname + "_gt"] = bytes(o) == coracle.pairing_product(cid, a, b, n, threads=4)
    m = 700                                                    # six blocks of the lane-pair main pass: a tree of six partials
    keys = [coracle.scale_point(cid, 2, g2, rnd.randrange(1, 1 << 250)) for _ in range(16)]
    pts = b"".join(keys[i %% 16] for i in range(m))            # repeated keys: the doubling branch of the mixed addition
    s = (ctypes.c_uint8 * (4 * fp))()
    assert L.bgls_aggregate_points(cid, 2, B(pts), m, s) == 0
    res[cname + "_sum"] = bytes(s) == coracle.aggregate_points(cid, 2, pts, m)
    # a small aggregate verification through the C ABI: hashing, latency-form Miller loop, epilogue (BLS12-381: uncleared
    # cofactor), final exponentiation; a flipped message bit must reject
    ns = 5
    sks = [rnd.randrange(1, 1 << 250) for _ in range(ns)]
    kb = b"".join(x.to_bytes(32, "big") for x in sks)
    msgs = [rnd.randbytes(40) for _ in range(ns)]
    off = (ctypes.c_uint64 * (ns + 1))(*[40 * i for i in range(ns + 1)])
    keys2 = (ctypes.c_uint8 * (ns * 4 * fp))(); assert L.bgls_scale_generator(cid, 2, B(kb), ns, keys2) == 0
    sigs = (ctypes.c_uint8 * (ns * 2 * fp))(); assert L.bgls_sign_batch(cid, B(kb), B(b"".join(msgs)), off, ns, sigs) == 0
    agg = (ctypes.c_uint8 * (2 * fp))(); assert L.bgls_aggregate_points(cid, 1, sigs, ns, agg) == 0
    ok = L.bgls_verify_aggregate(cid, agg, keys2, B(b"".join(msgs)), off, ns, 0)
    bad = bytearray(b"".join(msgs)); bad[7] ^= 4
    no = L.bgls_verify_aggregate(cid, agg, keys2, B(bytes(bad)), off, ns, 0)
    res[cname + "_verify"] = ok == 1 and no == 0 and coracle.verify_aggregate(cid, bytes(agg), bytes(keys2), msgs, False, 2, 0) == 1
    # the public HashToG1 (BLS12-381: cofactor cleared per message) and Sign = H(m).Mul(sk) against the oracle
    hs = (ctypes.c_uint8 * (ns * 2 * fp))(); assert L.bgls_hash_to_g1(cid, B(b"".join(msgs)), off, ns, hs) == 0
    res[cname + "_hash"] = all(bytes(hs)[2 * fp * i:2 * fp * (i + 1)] == coracle.hash_to_g1(cid, msgs[i]) for i in range(ns))
    res[cname + "_sign"] = all(bytes(sigs)[2 * fp * i:2 * fp * (i + 1)] == coracle.scale_point(cid, 1, coracle.hash_to_g1(cid, msgs[i]), sks[i]) for i in range(ns))
print("RESULT " + json.dumps(res))
"""


@pytest.mark.parametrize("env", [{"BGLS_FINALX": "0", "BGLS_LATX": "0", "BGLS_SUMX": "0", "BGLS_EPIX": "0"}, {"BGLS_SUMX": "1"}, {"BGLS_SUMTREE": "0"}, {"BGLS_SUMTREEX": "0"}, {"BGLS_G1X": "0"}, {"BGLS_REDUCEX": "0"}, {"BGLS_LATX2": "0"}, {}],
                         ids=["32-bit tails and key sum", "one-lane key sum", "key-sum tree as one launch per level", "key-sum tree with 32-bit additions", "32-bit G1 scalar multiplications", "six-lane reduce passes", "one-wave accumulator in the latency Miller kernel", "defaults"])
def test_replaced_kernels_still_match_the_oracle(env):
    golden = os.path.join(ROOT, "tests", "golden")
    code = CHILD % (ROOT, golden)
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res and all(res.values()), res
