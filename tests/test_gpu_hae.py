"""GPU tier: hashed aggregation exponents and multiplicities through the C ABI and the host
mirror -- golden fixtures (tests/golden/hae_*.json, Python oracle), the C oracle on fresh random inputs, and the
reference's own scheme tests (bgls/blsHAE_test.go:14-82, bgls/blsKosk_test.go:66-94) re-read through the mirror."""
import ctypes
import json
import os
import random
import secrets

import pytest

from oracle import coracle
from bgls_amd import Altbn128, Bls12, AggregatePoints
from bgls_amd.bgls import (AggregateSignaturesWithHAE, KeyGen, KoskSign, KoskVerifyMultiSignatureWithMultiplicity, Sign,
                           VerifyAggregateSignatureWithHAE, VerifyMultiSignatureWithHAE, hashPubKeysToExponents)
from bgls_amd.curves import Point, G1, G2, ScalePoints

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
curves = [Altbn128, Bls12]
B = lambda b: (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


@pytest.fixture(autouse=True)
def _init(gpu_lib):
    return gpu_lib


def load(curve):
    return json.load(open(os.path.join(HERE, "golden", "hae_%s.json" % curve.Name())))


def pts(curve, group, hexes):
    return [Point(curve, group, bytes.fromhex(h)) for h in hexes]


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_hae_golden(curve):
    v = load(curve)
    keys = pts(curve, G2, v["exponents"]["keys"])
    assert ["%032x" % t for t in hashPubKeysToExponents(keys)] == v["exponents"]["t"]
    a = v["aggregate_signatures"]
    assert AggregateSignaturesWithHAE(pts(curve, G1, a["sigs"]), pts(curve, G2, a["keys"])).raw.hex() == a["out"]
    assert AggregateSignaturesWithHAE(pts(curve, G1, a["sigs"]), pts(curve, G2, a["keys"])[:-1]) is None     # blsHAE.go:40-42
    for case in v["multi_cases"]:
        got = VerifyMultiSignatureWithHAE(curve, Point(curve, G1, bytes.fromhex(case["sig"])), pts(curve, G2, case["keys"]),
                                          bytes.fromhex(case["msg"]))
        assert got == case["expect"], case["name"]
    for case in v["aggregate_cases"]:
        got = VerifyAggregateSignatureWithHAE(curve, Point(curve, G1, bytes.fromhex(case["sig"])), pts(curve, G2, case["keys"]),
                                              [bytes.fromhex(m) for m in case["msgs"]])
        assert got == case["expect"], case["name"]
    for case in v["multiplicity_cases"]:
        got = KoskVerifyMultiSignatureWithMultiplicity(curve, Point(curve, G1, bytes.fromhex(case["sig"])), pts(curve, G2, case["keys"]),
                                                       case["mult"], bytes.fromhex(case["msg"]))
        assert got == case["expect"], case["name"]


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_hae_exponents_against_oracle(gpu_lib, curve):
    """Expansion-node boundaries (4 exponents per 64-byte node) and the one-block / multi-block root: every n gives the
    oracle's bytes.  Keys need not be valid points for the hash (it sees bytes), so random bytes probe the XOF alone."""
    rnd = random.Random(99)
    g2b = len(curve.GetG2().raw)
    for n in (1, 2, 3, 4, 5, 8, 9, 63, 64, 65, 1000, 4097):
        keys = rnd.randbytes(n * g2b)
        o = (ctypes.c_uint8 * (16 * n))()
        assert gpu_lib.bgls_hae_exponents(curve.id, B(keys), n, o) == 0
        assert bytes(o) == coracle.blake2xb(keys, 16 * n), n


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_aggregation_with_hae(curve):
    """bgls/blsHAE_test.go:14-56 TestAggregationWithHAE"""
    N, Size = 5, 32
    msgs, sigs, pubkeys = [], [], []
    for _ in range(N):
        m = secrets.token_bytes(Size)
        sk, vk, _ = KeyGen(curve)
        msgs.append(m); pubkeys.append(vk); sigs.append(Sign(curve, sk, m))
    aggSig = AggregateSignaturesWithHAE(sigs[:N], pubkeys[:N])
    assert VerifyAggregateSignatureWithHAE(curve, aggSig, pubkeys[:N], msgs[:N])
    assert not VerifyAggregateSignatureWithHAE(curve, aggSig, pubkeys[:N - 1], msgs[:N])
    assert AggregateSignaturesWithHAE(sigs[:N], pubkeys[:N - 1]) is None
    skf, vkf, _ = KeyGen(curve)
    pubkeys.append(vkf); msgs.append(msgs[0]); sigs.append(Sign(curve, skf, msgs[N]))
    aggSig = AggregateSignaturesWithHAE(sigs, pubkeys)
    assert VerifyAggregateSignatureWithHAE(curve, aggSig, pubkeys, msgs)                    # duplicate messages are fine here
    assert not VerifyAggregateSignatureWithHAE(curve, aggSig, pubkeys[:N], msgs[:N])
    msgs[0], msgs[1] = msgs[1], msgs[N]
    aggSig = AggregatePoints(sigs[:N])
    assert not VerifyAggregateSignatureWithHAE(curve, aggSig, pubkeys[:N], msgs[:N])


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_multisig_with_hae(curve):
    """bgls/blsHAE_test.go:58-82 TestMultiSigWithHAE (2 trials instead of 5), each verdict also asked of the C oracle"""
    for _ in range(2):
        msg = secrets.token_bytes(32)
        signers, sigs = [], []
        for _ in range(8):
            sk, vk, _ = KeyGen(curve)
            sigs.append(Sign(curve, sk, msg)); signers.append(vk)
        aggSig = AggregateSignaturesWithHAE(sigs, signers)
        kb = b"".join(k.raw for k in signers)
        assert aggSig.raw == coracle.aggregate_signatures_hae(curve.id, b"".join(s.raw for s in sigs), kb, 8)
        assert VerifyMultiSignatureWithHAE(curve, aggSig, signers, msg)
        assert coracle.verify_multi_hae(curve.id, aggSig.raw, kb, 8, msg) == 1
        assert not VerifyMultiSignatureWithHAE(curve, aggSig, signers, secrets.token_bytes(32))
        _, vkf, _ = KeyGen(curve)
        signers[0] = vkf
        assert not VerifyMultiSignatureWithHAE(curve, aggSig, signers, msg)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_kosk_multisig_with_multiplicity(curve):
    """bgls/blsKosk_test.go:66-94 shape: each signature included multiplicity[i] times (plus a negative and a zero factor,
    which curves/curve.go:190-214 defines through Point.Mul)."""
    rnd = random.Random(5 + curve.id)
    msg = secrets.token_bytes(32)
    n = 7
    mult = [rnd.randrange(1, 6) for _ in range(n)]
    mult[2], mult[4] = -3, 0
    signers, sigs = [], []
    for _ in range(n):
        sk, vk, _ = KeyGen(curve)
        sigs.append(KoskSign(curve, sk, msg)); signers.append(vk)
    aggSig = AggregatePoints(ScalePoints(sigs, mult))
    assert KoskVerifyMultiSignatureWithMultiplicity(curve, aggSig, signers, mult, msg)
    assert coracle.verify_multi_multiplicity(curve.id, aggSig.raw, b"".join(k.raw for k in signers), n, mult, msg) == 1
    bad = list(mult); bad[0] += 1
    assert not KoskVerifyMultiSignatureWithMultiplicity(curve, aggSig, signers, bad, msg)
    assert not KoskVerifyMultiSignatureWithMultiplicity(curve, aggSig, signers, mult[:-1], msg)
    assert not KoskVerifyMultiSignatureWithMultiplicity(curve, aggSig, signers, None, msg)        # plain Kosk needs the plain sum
    assert KoskVerifyMultiSignatureWithMultiplicity(curve, AggregatePoints(sigs), signers, None, msg)


def test_hae_larger_batch_matches_plain_composition(gpu_lib):
    """Size-independent property at n = 3000: the fused weighted key sum equals AggregatePoints(ScalePoints(keys, t)) built
    from the existing entry points, and the verdict follows."""
    curve = Altbn128
    n = 3000
    rnd = random.Random(31)
    g2 = curve.GetG2()
    sks = [rnd.randrange(1, curve.GetG1Order()) for _ in range(n)]
    keys = ScalePoints([g2] * n, sks)
    t = hashPubKeysToExponents(keys)
    assert t == coracle.hae_exponents(curve.id, b"".join(k.raw for k in keys), n)
    msg = b"hae batch"
    h = curve.HashToG1(msg)
    sigma = h.Mul(sum(s * e for s, e in zip(sks, t)) % curve.GetG1Order())       # = sum t_i sk_i H(m)
    assert VerifyMultiSignatureWithHAE(curve, sigma, keys, msg)
    apk = AggregatePoints(ScalePoints(keys, t))
    from bgls_amd.bgls import VerifySingleSignature
    assert VerifySingleSignature(curve, sigma, apk, msg)
    keys[n // 2] = g2
    assert not VerifyMultiSignatureWithHAE(curve, sigma, keys, msg)
