"""GPU tier: the reference's own scheme-level tests, re-read through the host mirror
(bgls_amd.curves / bgls_amd.bgls == the Go packages `curves` / `bgls` on this path)."""
import secrets

import pytest

from bgls_amd import Altbn128, Bls12, AggregatePoints, ScalePoints
from bgls_amd.bgls import (AggregateKeys, AggregateSignatures, KeyGen, KoskSign, KoskVerifyAggregateSignature,
                           KoskVerifyMultiSignature, KoskVerifySingleSignature, Sign, VerifyAggregateSignature,
                           VerifySingleSignature)

pytestmark = pytest.mark.gpu
curves = [Altbn128, Bls12]


@pytest.fixture(autouse=True)
def _init(gpu_lib):
    return gpu_lib


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_single_signer(curve):
    """bgls/bgls_test.go:19-38 TestSingleSigner"""
    sk, vk, err = KeyGen(curve)
    assert err is None
    d = secrets.token_bytes(64)
    sig = Sign(curve, sk, d)
    assert VerifySingleSignature(curve, sig, vk, d)
    sig2, _ = sig.Copy().Add(curve.GetG1())
    assert not VerifySingleSignature(curve, sig2, vk, d)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_aggregation(curve):
    """bgls/bgls_test.go:40-77 TestAggregation"""
    N, Size = 6, 32
    msgs, sigs, pubkeys = [], [], []
    for _ in range(N):
        m = secrets.token_bytes(Size)
        sk, vk, _ = KeyGen(curve)
        msgs.append(m); pubkeys.append(vk); sigs.append(Sign(curve, sk, m))
    aggSig = AggregateSignatures(sigs[:N])
    assert VerifyAggregateSignature(curve, aggSig, pubkeys[:N], msgs[:N])
    assert not VerifyAggregateSignature(curve, aggSig, pubkeys[:N - 1], msgs[:N])
    skf, vkf, _ = KeyGen(curve)
    pubkeys.append(vkf); sigs.append(Sign(curve, skf, msgs[0])); msgs.append(msgs[0])
    aggSig = AggregateSignatures(sigs)
    assert not VerifyAggregateSignature(curve, aggSig, pubkeys, msgs)            # duplicate messages
    assert not VerifyAggregateSignature(curve, aggSig, pubkeys[:N], msgs[:N])    # invalid signature
    msgs[0], msgs[1] = msgs[1], msgs[N]
    aggSig = AggregateSignatures(sigs[:N])
    assert not VerifyAggregateSignature(curve, aggSig, pubkeys[:N], msgs[:N])    # messages 0 and 1 switched


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_kosk_multisig(curve):
    """bgls/blsKosk_test.go:35-64 TestKoskMultiSig (2 trials instead of 5)"""
    for _ in range(2):
        msg = secrets.token_bytes(32)
        signers, sigs = [], []
        for _ in range(8):
            sk, vk, _ = KeyGen(curve)
            sigs.append(KoskSign(curve, sk, msg)); signers.append(vk)
        aggsig = AggregateSignatures(sigs)
        assert KoskVerifyMultiSignature(curve, aggsig, signers, msg)
        assert not KoskVerifyMultiSignature(curve, aggsig, signers, secrets.token_bytes(32))
        _, vkf, _ = KeyGen(curve)
        assert KoskVerifySingleSignature(curve, aggsig, AggregateKeys(signers), msg)
        signers[0] = vkf
        assert not KoskVerifyMultiSignature(curve, aggsig, signers, msg)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_kosk_aggregation_allows_duplicates(curve):
    """bgls/blsKosk_test.go:96-133 TestKoskAggregation: duplicates are allowed under Kosk"""
    msgs, sigs, keys = [], [], []
    m0 = secrets.token_bytes(32)
    for i in range(4):
        m = m0 if i < 2 else secrets.token_bytes(32)
        sk, vk, _ = KeyGen(curve)
        msgs.append(m); keys.append(vk); sigs.append(KoskSign(curve, sk, m))
    agg = AggregateSignatures(sigs)
    assert KoskVerifyAggregateSignature(curve, agg, keys, msgs)
    assert not VerifyAggregateSignature(curve, agg, keys, msgs)
    assert not KoskVerifyAggregateSignature(curve, agg, keys[:3], msgs)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_curve_interface(curve):
    """curves/curve_test.go: TestMul :120-141, TestAggregation :167-186, TestScaling :188-208, TestPairingProd :143-165"""
    g1, g2 = curve.GetG1(), curve.GetG2()
    order = curve.GetG1Order()
    for k in (0, 1, 12345, order - 1):
        s, ok = g1.Mul(k).Add(g1.Mul(-k))
        assert ok and s.Equals(curve.GetG1Infinity())
    for N in (2, 4, 6, 8):
        xs = [secrets.randbelow(order) for _ in range(N)]
        assert AggregatePoints([g2.Mul(x) for x in xs]).Equals(g2.Mul(sum(xs) % order))
    pts = [g1.Mul(i + 2) for i in range(4)]
    fs = [3, None, -5, 0]
    scaled = ScalePoints(pts, fs)
    assert scaled[0].Equals(g1.Mul(6)) and scaled[1].Equals(pts[1]) and scaled[2].Equals(g1.Mul(-20)) and scaled[3].Equals(curve.GetG1Infinity())
    assert ScalePoints(pts, None) is pts and ScalePoints(pts, [1]) is None
    a, b = secrets.randbelow(order), secrets.randbelow(order)
    e1, ok1 = curve.Pair(g1.Mul(a), g2.Mul(b))
    e2, ok2 = curve.Pair(g1.Mul(a * b % order), g2)
    assert ok1 and ok2 and e1.Equals(e2) and not e1.Equals(curve.GetGTIdentity())
    idt, _ = curve.Pair(g1, curve.GetG2Infinity())
    assert idt.Equals(curve.GetGTIdentity())                  # altbn128.go:478 / bls12_381.go:341
    assert curve.PairingProduct([g1], [g2, g2]) == (None, False)
    assert curve.Pair(g2, g1) == (None, False)                # type mismatch => nil,false
    p, ok = curve.MakeG1Point([1, 1], True)
    assert not ok and p is None
    pt, ok = curve.UnmarshalG1(g1.MarshalUncompressed())
    assert ok and pt.Equals(g1)
    assert curve.UnmarshalG1(b"\x00" * 5) == (None, False)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_batch_keygen_and_sign_match_per_point_calls_and_oracle(curve):
    """bgls_scale_generator / bgls_sign_batch (LoadPublicKey, Sign, KoskSign of bgls/bgls.go:40-56 over a batch): same
    bytes as the per-point mirror calls and as the C oracle; the signatures aggregate and verify."""
    import random
    from oracle import coracle
    from bgls_amd.bgls import LoadPublicKeys, SignBatch, LoadPublicKey
    rnd = random.Random(17 + curve.id)
    n = 70
    sks = [rnd.randrange(1, curve.GetG1Order()) for _ in range(n)]
    sks[3] = 1
    msgs = [rnd.randbytes(rnd.choice((8, 32, 64))) for _ in range(n)]
    pks = LoadPublicKeys(curve, sks)
    sigs = SignBatch(curve, sks, msgs)
    ksigs = SignBatch(curve, sks, msgs, kosk=True)
    for i in (0, 3, 11, n - 1):
        assert pks[i].Equals(LoadPublicKey(curve, sks[i])) and sigs[i].Equals(Sign(curve, sks[i], msgs[i]))
        assert ksigs[i].Equals(KoskSign(curve, sks[i], msgs[i]))
        assert pks[i].raw == coracle.scale_point(curve.id, 2, curve.GetG2().raw, sks[i])
        assert sigs[i].raw == coracle.scale_point(curve.id, 1, coracle.hash_to_g1(curve.id, msgs[i]), sks[i])
    assert pks[3].Equals(curve.GetG2())
    assert VerifyAggregateSignature(curve, AggregateSignatures(sigs), pks, msgs)
    assert KoskVerifyAggregateSignature(curve, AggregateSignatures(ksigs), pks, msgs)
    assert not VerifyAggregateSignature(curve, AggregateSignatures(ksigs), pks, msgs)
    assert SignBatch(curve, sks, msgs[:-1]) is None and LoadPublicKeys(curve, []) == []
