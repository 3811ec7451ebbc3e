"""CPU tier: the oracle's BLAKE2Xb and hashed-aggregation-exponent (HAE) / multiplicity flows
(bgls/blsHAE.go, bgls/blsKosk.go:137-150).

No reference test stores an exponent or a BLAKE2Xb output (bgls/blsHAE_test.go uses fresh random keys), so the
anchor is the BLAKE2 authors' own C code inside CPython: the restated compression function + parameter block must
equal hashlib.blake2b on every parameter combination hashlib accepts; BLAKE2Xb is that function with the published
BLAKE2X parameter blocks (root: XOF length in bytes 12..15; nodes: fanout = depth = 0, leaf = inner = 64)."""
import hashlib
import json
import os
import random

import pytest

from oracle import coracle
from oracle.pyref import hashes, scheme
from oracle.pyref.pairing import Pairing

HERE = os.path.dirname(os.path.abspath(__file__))
CURVES = [(0, "altbn128"), (1, "bls12")]


def load(name):
    return json.load(open(os.path.join(HERE, "golden", "hae_%s.json" % name)))


def test_param_blake2b_equals_hashlib():
    rnd = random.Random(11)
    for _ in range(150):
        data = rnd.randbytes(rnd.choice([0, 1, 63, 64, 127, 128, 129, 255, 256, 300, 1000]))
        ds, fo, dp = rnd.randrange(1, 65), rnd.randrange(0, 256), rnd.randrange(1, 256)
        ls, no, xl = rnd.randrange(0, 2 ** 32), rnd.randrange(0, 2 ** 32), rnd.randrange(0, 2 ** 32)
        nd, isz = rnd.randrange(0, 256), rnd.randrange(0, 65)
        want = hashlib.blake2b(data, digest_size=ds, fanout=fo, depth=dp, leaf_size=ls, node_offset=no | (xl << 32), node_depth=nd,
                               inner_size=isz).digest()
        assert hashes.blake2b_param(data, ds, fo, dp, ls, no, xl, nd, isz) == want


def test_blake2xb_structure_and_c_twin():
    rnd = random.Random(12)
    for ln in (0, 1, 64, 128, 129, 1000, 4096):
        for ol in (1, 16, 63, 64, 65, 128, 160, 1000):
            d = rnd.randbytes(ln)
            out = hashes.blake2xb(d, ol)
            assert len(out) == ol and coracle.blake2xb(d, ol) == out
    # the root is plain BLAKE2b-512 apart from the XOF length field; prefixes of different lengths are unrelated
    assert hashes.blake2xb(b"abc", 64) != hashes.blake2xb(b"abc", 65)[:64]
    # a 16 n byte request is what blsHAE.go:81 makes: n = 4 fits exactly one expansion node
    r = hashlib.blake2b(b"k", digest_size=64, node_offset=64 << 32).digest()
    assert hashes.blake2xb(b"k", 64) == hashes.blake2b_param(r, 64, 0, 0, 64, 0, 64, 0, 64)


@pytest.mark.parametrize("cid,name", CURVES, ids=[c[1] for c in CURVES])
def test_hae_golden_against_c_oracle(cid, name):
    v = load(name)
    for row in v["xof"]:
        assert coracle.blake2xb(bytes.fromhex(row["in"]), row["out_len"]).hex() == row["out"]
    keys = b"".join(map(bytes.fromhex, v["exponents"]["keys"]))
    assert ["%032x" % t for t in coracle.hae_exponents(cid, keys, len(v["exponents"]["keys"]))] == v["exponents"]["t"]
    a = v["aggregate_signatures"]
    assert coracle.aggregate_signatures_hae(cid, b"".join(map(bytes.fromhex, a["sigs"])), b"".join(map(bytes.fromhex, a["keys"])),
                                            len(a["sigs"])).hex() == a["out"]
    for case in v["multi_cases"]:
        got = coracle.verify_multi_hae(cid, bytes.fromhex(case["sig"]), b"".join(map(bytes.fromhex, case["keys"])), len(case["keys"]),
                                       bytes.fromhex(case["msg"]))
        assert bool(got == 1) == case["expect"], case["name"]
    for case in v["aggregate_cases"]:
        if len(case["keys"]) != len(case["msgs"]):
            got = 0                                       # bgls/bgls.go:95-97
        else:
            got = coracle.verify_aggregate_hae(cid, bytes.fromhex(case["sig"]), b"".join(map(bytes.fromhex, case["keys"])),
                                               [bytes.fromhex(m) for m in case["msgs"]])
        assert bool(got == 1) == case["expect"], case["name"]
    for case in v["multiplicity_cases"]:
        keys = b"".join(map(bytes.fromhex, case["keys"]))
        if case["mult"] is None:
            got = coracle.verify_multi(cid, bytes.fromhex(case["sig"]), keys, len(case["keys"]), b"\x01" + bytes.fromhex(case["msg"]))
        elif len(case["mult"]) != len(case["keys"]):
            got = 0                                       # blsKosk.go:141-143
        else:
            got = coracle.verify_multi_multiplicity(cid, bytes.fromhex(case["sig"]), keys, len(case["keys"]), case["mult"],
                                                    bytes.fromhex(case["msg"]))
        assert bool(got == 1) == case["expect"], case["name"]


def test_hae_exponents_pyref_matches_fixture():
    from oracle.pyref.params import BN254
    v = load("altbn128")
    G = Pairing(BN254).G
    keys = [G.g2_from_bytes(bytes.fromhex(k)) for k in v["exponents"]["keys"]]
    assert ["%032x" % t for t in scheme.hash_pubkeys_to_exponents(BN254, keys)] == v["exponents"]["t"]
