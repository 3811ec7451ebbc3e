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


def kosk_verify_batch_multi_signature(curve, aggsigs, pubkeys, msgs):
    """bgls/blsKosk.go:126-133: AggregateSignatures, one AggregateKeys per set, KoskVerifyAggregateSignature"""
    PR = pairing_for(curve)
    aggsig = PR.G.g1_sum(aggsigs)
    keys = [PR.G.g2_sum(ks) for ks in pubkeys]
    return kosk_verify_aggregate_signature(curve, aggsig, keys, msgs)


# ---- hashed aggregation exponents (bgls/blsHAE.go) and multiplicities (bgls/blsKosk.go:137-150) ---------------
def hash_pubkeys_to_exponents(curve, keys):
    """blsHAE.go:80-93: t_i = i-th 16-byte big-endian chunk of BLAKE2Xb(MarshalUncompressed(pk_0) || ... , 16 n)."""
    from .hashes import blake2xb
    G = pairing_for(curve).G
    raw = blake2xb(b"".join(G.g2_bytes(k) for k in keys), 16 * len(keys))
    return [int.from_bytes(raw[16 * i:16 * i + 16], "big") for i in range(len(keys))]


def aggregate_signatures_hae(curve, sigs, keys):
    """blsHAE.go:39-46 (nil on a length mismatch)."""
    if len(sigs) != len(keys):
        return None
    G = pairing_for(curve).G
    t = hash_pubkeys_to_exponents(curve, keys)
    return G.g1_sum([G.g1_mul(s, k) for s, k in zip(sigs, t)])


def verify_aggregate_signature_hae(curve, aggsig, keys, msgs):
    """blsHAE.go:49-53: keys scaled by their exponents, duplicates allowed."""
    G = pairing_for(curve).G
    t = hash_pubkeys_to_exponents(curve, keys)
    return verify_agg(curve, aggsig, [G.g2_mul(k, e) for k, e in zip(keys, t)], msgs, True)


def verify_multi_signature_hae(curve, aggsig, keys, msg):
    """blsHAE.go:56-58,74-77."""
    G = pairing_for(curve).G
    t = hash_pubkeys_to_exponents(curve, keys)
    return verify_single(curve, aggsig, G.g2_sum([G.g2_mul(k, e) for k, e in zip(keys, t)]), msg)


def kosk_verify_multi_signature_with_multiplicity(curve, aggsig, keys, multiplicity, msg):
    """blsKosk.go:137-150; ScalePoints with a negative factor negates the point first (curves/curve.go:190-214)."""
    if multiplicity is None:
        return kosk_verify_multi_signature(curve, aggsig, keys, msg)
    if len(keys) != len(multiplicity):
        return False
    G = pairing_for(curve).G
    scaled = [G.g2_mul(G.g2_neg(k), -m) if m < 0 else G.g2_mul(k, m) for k, m in zip(keys, multiplicity)]
    return kosk_verify_multi_signature(curve, aggsig, scaled, msg)
