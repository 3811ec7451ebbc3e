"""BGLS scheme flows on top of the oracle arithmetic (CPU oracle -- TEST INFRASTRUCTURE ONLY).

Restates bgls/bgls.go: LoadPublicKey :40-43, Sign :46-56, VerifySingleSignature :59-70,
VerifyAggregateSignature :82-84 / verifyAggSig :94-119, verifyMultiSignature :89-92,
AggregateSignatures/Keys :123-131, containsDuplicateMessage :139-150; and
bgls/blsKosk.go: KoskSign (0x01 prefix), KoskVerifyMultiSignature :117-120,
KoskVerifyAggregateSignature :100-106.
"""
from .params import CURVES
from .pairing import Pairing
from .h2c import hash_to_g1

_PAIR = {}


def pairing_for(curve):
    if curve.name not in _PAIR:
        _PAIR[curve.name] = Pairing(curve)
    return _PAIR[curve.name]


def load_public_key(curve, sk):
    return pairing_for(curve).G.g2_mul(curve.g2, sk)


def sign(curve, sk, msg):
    return pairing_for(curve).G.g1_mul(hash_to_g1(curve, msg), sk)


def kosk_sign(curve, sk, msg):
    return sign(curve, sk, b"\x01" + msg)


def contains_duplicate(msgs):
    return len(set(bytes(m) for m in msgs)) != len(msgs)


def verify_single(curve, sig, pk, msg):
    PR = pairing_for(curve)
    h = PR.G.g1_neg(hash_to_g1(curve, msg))
    return PR.T.f12_is_one(PR.pairing_product([h, sig], [pk, curve.g2]))


def verify_agg(curve, aggsig, keys, msgs, allow_duplicates=False):
    PR = pairing_for(curve)
    if len(keys) != len(msgs):
        return False
    if not allow_duplicates and contains_duplicate(msgs):
        return False
    p1 = [hash_to_g1(curve, m) for m in msgs] + [PR.G.g1_neg(aggsig)]
    p2 = list(keys) + [curve.g2]
    return PR.T.f12_is_one(PR.pairing_product(p1, p2))


def verify_aggregate_signature(curve, aggsig, keys, msgs):
    return verify_agg(curve, aggsig, keys, msgs, False)


def verify_multi_signature(curve, aggsig, keys, msg):
    PR = pairing_for(curve)
    return verify_single(curve, aggsig, PR.G.g2_sum(keys), msg)


def kosk_verify_multi_signature(curve, aggsig, keys, msg):
    return verify_multi_signature(curve, aggsig, keys, b"\x01" + msg)


def kosk_verify_aggregate_signature(curve, aggsig, keys, msgs):
    return verify_agg(curve, aggsig, keys, [b"\x01" + m for m in msgs], True)
