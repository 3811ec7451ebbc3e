"""GPU tier: ONE 32-bit-limb fallback per stage stays selectable for A/B measurements, all behind the bit mask BGLS_LEGACY (read
once per process, so each combination runs in a child process): 1 k_final36 (final exponentiation), 2 k_miller_lat (latency Miller
loop), 4 k_cofactor_epilogue, 8 k_sum_main (G2 key sums), 16 the key-sum tree as one launch per level, 32 BLS12-381 G1 scalar
multiplications on 32-bit limbs, 64 every reduce pass on k_reduce_coop; BGLS_MILLER_SHAPE=5 puts the Miller stage on the 32-bit fused
kernel k_miller_ab64.  PairingProduct of a handful of pairings (latency Miller loop + reduce + final exponentiation) must give the C
oracle's GT bytes, a 700-key aggregate of public keys the oracle's point, a small verification the oracle's verdict."""
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
    # a Miller stage above the latency shape (k_miller_x60 by default, k_miller_ab64 under BGLS_MILLER_SHAPE=5): bilinearity pins the value,
    # prod e(a_i g1, b_i g2) = e((sum a_i b_i) g1, g2)
    nb = 700
    ka = [rnd.randrange(1, 1 << 250) for _ in range(nb)]; kbs = [rnd.randrange(1, 1 << 250) for _ in range(nb)]
    G1 = (ctypes.c_uint8 * (2 * fp))(); G2 = (ctypes.c_uint8 * (4 * fp))()
    assert L.bgls_generator(cid, 1, G1) == 0 and L.bgls_generator(cid, 2, G2) == 0
    p1 = (ctypes.c_uint8 * (nb * 2 * fp))(); p2 = (ctypes.c_uint8 * (nb * 4 * fp))()
    assert L.bgls_scale_points(cid, 1, B(bytes(G1) * nb), B(b"".join(x.to_bytes(32, "big") for x in ka)), None, nb, p1) == 0
    assert L.bgls_scale_points(cid, 2, B(bytes(G2) * nb), B(b"".join(x.to_bytes(32, "big") for x in kbs)), None, nb, p2) == 0
    big = (ctypes.c_uint8 * (12 * fp))(); assert L.bgls_pairing_product(cid, p1, p2, nb, big) == 0
    order = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617, 1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}[cid]
    ssum = sum(x * y for x, y in zip(ka, kbs)) %% order
    one = (ctypes.c_uint8 * (2 * fp))(); assert L.bgls_scale_points(cid, 1, G1, B(ssum.to_bytes(32, "big")), None, 1, one) == 0
    e1 = (ctypes.c_uint8 * (12 * fp))(); assert L.bgls_pairing_product(cid, one, G2, 1, e1) == 0
    res[cname + "_bilinear_700"] = bytes(big) == bytes(e1)
print("RESULT " + json.dumps(res))
"""


@pytest.mark.parametrize("env", [{"BGLS_LEGACY": "15"}, {"BGLS_LEGACY": "16"}, {"BGLS_LEGACY": "32"}, {"BGLS_LEGACY": "64"}, {"BGLS_LEGACY": "127", "BGLS_MILLER_SHAPE": "5"}, {}],
                         ids=["32-bit tails and key sum", "key-sum tree as one launch per level", "32-bit G1 scalar multiplications", "six-lane reduce passes",
                              "every fallback at once", "defaults"])
def test_replaced_kernels_still_match_the_oracle(env):
    """Round 6: the kernels that only BGLS_LEGACY bits 1 / 2 / 4 / 8 and BGLS_MILLER_SHAPE=5 reach live in libbgls_hip_legacy.so (`make LEGACY=1`,
    built by __graft_entry__.build()); the shipped library ignores those bits.  Every combination below runs on the legacy build; the last one
    ("defaults") also runs on the shipped library, where -- second test below -- asking for a fallback it does not carry changes nothing."""
    golden = os.path.join(ROOT, "tests", "golden")
    code = CHILD % (ROOT, golden)
    e = dict(os.environ)
    e.update(env)
    e["BGLS_LIB_VARIANT"] = "legacy"
    legacy_lib = os.path.join(ROOT, "bgls_amd", "libbgls_hip_legacy.so")
    assert os.path.exists(legacy_lib), "build it: make -C bgls_amd/csrc LEGACY=1 (or __graft_entry__.build())"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res and all(res.values()), res


def test_the_shipped_library_ignores_the_fallbacks_it_does_not_carry():
    """libbgls_hip.so has no k_final36 / k_miller_lat / k_cofactor_epilogue / k_sum_main<G2> / k_miller_ab64: BGLS_LEGACY=15 and
    BGLS_MILLER_SHAPE=5 must leave it on its default kernels (same oracle bytes), bgls_set_miller_shape(5) must refuse, and the symbols
    must be absent from the library."""
    golden = os.path.join(ROOT, "tests", "golden")
    code = CHILD % (ROOT, golden) + """
assert L.bgls_set_miller_shape(5, 0) < 0 and b"LEGACY" in L.bgls_last_error()
assert L.bgls_set_miller_shape(0, 0) == 0
"""
    e = dict(os.environ)
    e.update({"BGLS_LEGACY": "15", "BGLS_MILLER_SHAPE": "5"})
    e.pop("BGLS_LIB_VARIANT", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res and all(res.values()), res
    blob = open(os.path.join(ROOT, "bgls_amd", "libbgls_hip.so"), "rb").read()
    legacy_blob = open(os.path.join(ROOT, "bgls_amd", "libbgls_hip_legacy.so"), "rb").read()
    for mangled in (b"9k_final36I", b"12k_miller_latI", b"19k_cofactor_epilogueI", b"13k_miller_ab64I"):      # kernel symbols (Itanium mangling), not error texts
        assert mangled not in blob and mangled in legacy_blob, mangled
