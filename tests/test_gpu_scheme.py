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


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_distinct_message_scheme(curve):
    """bgls/blsDistinctMessage_test.go:14-60 TestDistinctMsgSingleSigner + TestDistinctMsgAggregation"""
    from bgls_amd.bgls import DistinctMsgSign, DistinctMsgVerifySingleSignature, DistinctMsgVerifyAggregateSignature
    sk, vk, _ = KeyGen(curve)
    msg = secrets.token_bytes(64)
    sig = DistinctMsgSign(curve, sk, msg)
    assert DistinctMsgVerifySingleSignature(curve, sig, vk, msg)
    sig2, _ = sig.Copy().Add(curve.GetG1())
    assert not DistinctMsgVerifySingleSignature(curve, sig2, vk, msg)
    N = 6
    msgs, sigs, pubkeys = [], [], []
    for _ in range(N):
        m = secrets.token_bytes(32)
        sk, vk, _ = KeyGen(curve)
        msgs.append(m); pubkeys.append(vk); sigs.append(DistinctMsgSign(curve, sk, m))
    aggSig = AggregatePoints(sigs)
    assert DistinctMsgVerifyAggregateSignature(curve, aggSig, pubkeys, msgs)
    assert not DistinctMsgVerifyAggregateSignature(curve, aggSig, pubkeys[:N - 1], msgs)
    same = [msgs[0]] * N                                      # identical payloads are fine: the key prefix makes them distinct
    s2 = [DistinctMsgSign(curve, sk_, m_) for sk_, m_ in zip([KeyGen(curve)[0] for _ in range(N)], same)]
    msgs[0] = msgs[1]
    assert not VerifyAggregateSignature(curve, aggSig, pubkeys, msgs)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_authentication_and_batch_multisig(curve):
    """bgls/blsKosk.go:44-69 (Authenticate / CheckAuthentication) and :126-133 (KoskVerifyBatchMultiSignature),
    bgls/blsHAE.go:62-72 (VerifyBatchMultiSignatureWithHAE)"""
    from bgls_amd.bgls import Authenticate, CheckAuthentication, KoskVerifyBatchMultiSignature, VerifyBatchMultiSignatureWithHAE
    sk, vk, _ = KeyGen(curve)
    auth = Authenticate(curve, sk)
    assert CheckAuthentication(curve, vk, auth)
    _, vk2, _ = KeyGen(curve)
    assert not CheckAuthentication(curve, vk2, auth)
    groups, aggsigs, msgs = [], [], []
    for g in range(3):
        m = secrets.token_bytes(32)
        ks, ss = [], []
        for _ in range(4):
            s_, v_, _ = KeyGen(curve)
            ks.append(v_); ss.append(KoskSign(curve, s_, m))
        groups.append(ks); aggsigs.append(AggregateSignatures(ss)); msgs.append(m)
    from bgls_amd.bgls import KoskVerifyBatchMultiSignatureStepwise
    assert KoskVerifyBatchMultiSignature(curve, aggsigs, groups, msgs)
    assert KoskVerifyBatchMultiSignatureStepwise(curve, aggsigs, groups, msgs)
    assert not KoskVerifyBatchMultiSignature(curve, aggsigs, groups, msgs[::-1])
    assert not KoskVerifyBatchMultiSignature(curve, aggsigs, groups[1:] + groups[:1], msgs)
    plain = [AggregateSignatures([Sign(curve, s_, m) for s_ in sks_]) for sks_, m in
             (([11, 12], b"a" * 32), ([13, 14, 15], b"b" * 32))]
    apks = [AggregateKeys([curve.GetG2().Mul(s_) for s_ in sks_]) for sks_ in ([11, 12], [13, 14, 15])]
    for dups in (False, True):
        assert VerifyBatchMultiSignatureWithHAE(curve, plain, apks, [b"a" * 32, b"b" * 32], dups)
    assert not VerifyBatchMultiSignatureWithHAE(curve, plain, apks, [b"a" * 32, b"c" * 32], False)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_ams_consistency(curve):
    """bgls/blsAsmSigs_test.go:15-60 TestAmsConsistency (6 keys / 4 signers instead of 15 / 8)"""
    from bgls_amd.bgls import (AmsAggregateMembershipKeyShares, AmsCombineSignatureShares, AmsCreateMembershipKeyShares,
                               AmsCreateMembershipKeySharesKnownExp, AmsCreateSignatureShare, AmsVerifySignature,
                               AmsVerifySignatureWithSetCheck, hashPubKeysToExponents, _ams_h2)
    numKeys, numSigners = 6, 4
    sks, pubkeys = [], []
    for _ in range(numKeys):
        s_, v_, _ = KeyGen(curve)
        sks.append(s_); pubkeys.append(v_)
    exps = hashPubKeysToExponents(pubkeys)
    apk = AggregatePoints(ScalePoints(pubkeys, exps))
    mk = []
    for i in range(numKeys):
        shares = AmsCreateMembershipKeyShares(curve, sks[i], i, pubkeys)
        known = AmsCreateMembershipKeySharesKnownExp(curve, sks[i], apk, exps[i], numKeys)
        assert all(a.Equals(b) for a, b in zip(shares, known))
        mk.append(shares)
    membership = [AmsAggregateMembershipKeyShares(curve, [mk[j][i] for j in range(numKeys)]) for i in range(numKeys)]
    for i in (0, numKeys - 1):
        p1, ok1 = curve.Pair(membership[i], curve.GetG2())
        p2, ok2 = curve.Pair(_ams_h2(curve, apk, str(i).encode()), apk)
        assert ok1 and ok2 and p1.Equals(p2)
    msg = secrets.token_bytes(64)
    shares = [AmsCreateSignatureShare(curve, sks[i], membership[i], msg) for i in range(numSigners)]
    aggKey, aggSig = AmsCombineSignatureShares(pubkeys[:numSigners], shares)
    signer_set = list(range(numSigners))
    assert AmsVerifySignature(curve, apk, signer_set, aggKey, aggSig, msg)
    assert AmsVerifySignatureWithSetCheck(curve, lambda s: len(s) > 3, apk, signer_set, aggKey, aggSig, msg)
    assert not AmsVerifySignatureWithSetCheck(curve, lambda s: len(s) > 5, apk, signer_set, aggKey, aggSig, msg)
    assert not AmsVerifySignature(curve, apk, signer_set, aggKey, aggSig, secrets.token_bytes(64))
    assert not AmsVerifySignature(curve, apk, signer_set[:-1], aggKey, aggSig, msg)


@pytest.mark.parametrize("curve", curves, ids=lambda c: c.Name())
def test_key_set_through_the_scheme_mirror(curve):
    """bgls.KeySet in place of the []Point of public keys: VerifyAggregateSignature / KoskVerifyMultiSignature give the
    same answers (bgls/bgls_test.go:40-77, blsKosk_test.go:35-64 re-read through a resident key set); PointT.Mul."""
    import os
    from bgls_amd import bgls
    N = 12
    sks = [bgls.KeyGen(curve)[0] for _ in range(N)]
    keys = bgls.LoadPublicKeys(curve, sks)
    msgs = [os.urandom(20 + i) for i in range(N)]
    agg = bgls.AggregateSignatures(bgls.SignBatch(curve, sks, msgs))
    for devices in ([0], [0, 0, 0]):
        ks = bgls.KeySet(curve, keys, devices)
        assert len(ks) == N
        assert bgls.VerifyAggregateSignature(curve, agg, ks, msgs) is True
        assert bgls.VerifyAggregateSignature(curve, agg, ks, msgs[:-1]) is False
        assert bgls.VerifyAggregateSignature(curve, agg, ks, msgs[1:] + msgs[:1]) is False
        assert bgls.VerifyAggregateSignature(curve, None, ks, msgs) is False
        m = os.urandom(32)
        kagg = bgls.AggregateSignatures(bgls.SignBatch(curve, sks, [m] * N, kosk=True))
        assert bgls.KoskVerifyMultiSignature(curve, kagg, ks, m) is True
        assert bgls.KoskVerifyMultiSignature(curve, kagg, ks, m + b"!") is False
        assert bgls.KoskVerifyMultiSignature(curve, None, keys, m) is False       # nil aggsig: false, not an exception
        ks.free()
    with pytest.raises(TypeError):
        bgls.hashPubKeysToExponents([curve.GetG1()])                               # G1 points are not public keys
    e, ok = curve.Pair(curve.GetG1(), curve.GetG2())
    k = sks[0]
    ek, _ = curve.Pair(curve.GetG1().Mul(k), curve.GetG2())
    assert ok and e.Mul(k).Equals(ek) and e.Mul(-k).Add(ek)[0].Equals(curve.GetGTIdentity())
    big = k + 5 * curve.GetG1Order() + (1 << 300) * 0                              # > 2^256 scalars act modulo the order
    assert curve.GetG2().Mul(k + 7 * curve.GetG1Order()).Equals(curve.GetG2().Mul(k))
