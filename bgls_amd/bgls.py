"""Host-side mirror of the reference's `bgls` scheme functions on the hot path
(bgls/bgls.go, bgls/blsKosk.go), same names and semantics, each routed to ONE batch C call."""
import ctypes
import secrets
from . import _lib
from .curves import _reduce_scalar, AggregatePoints, ScalePoints, Point, G1, G2


def KeyGen(curve):                                   # bgls/bgls.go:30-37
    x = secrets.randbelow(curve.GetG1Order())
    return x, LoadPublicKey(curve, x), None


def LoadPublicKey(curve, sk):                        # bgls/bgls.go:40-43
    return curve.GetG2().Mul(sk)


def Sign(curve, sk, msg):                            # bgls/bgls.go:46-56
    return curve.HashToG1(msg).Mul(sk)


def KoskSign(curve, sk, msg):                        # bgls/blsKosk.go:73-77
    return Sign(curve, sk, b"\x01" + bytes(msg))


def AggregateSignatures(sigs):                       # bgls/bgls.go:123-125
    return AggregatePoints(sigs)


def AggregateKeys(keys):                             # bgls/bgls.go:129-131
    return AggregatePoints(keys)


class KeySet:
    """n public keys resident on the GPU(s) behind a bgls_keys_t handle (include/bgls_hip.h): what a []Point of public
    keys becomes in the Go shim.  Accepted wherever the Verify* functions take `keys`.  check = the construction-time
    validation of MakeG2Point / UnmarshalG2 (subgroup membership included); devices = one entry per shard."""

    def __init__(self, curve, keys, devices=None, check=True):
        raw = b"".join(k.raw for k in keys) if keys and isinstance(keys[0], Point) else bytes(keys)
        n = len(raw) // (4 * curve.fp_bytes)
        h = ctypes.c_uint64()
        if devices:
            devs = list(devices)
            rc = _lib.load().bgls_keys_upload(curve.id, _lib.buf(raw), n, (ctypes.c_int * len(devs))(*devs), len(devs), 1 if check else 0,
                                              ctypes.byref(h))
        else:                                         # NULL + one shard: the process' default device (bgls_init / bgls_select_device), as in the C++ and Go mirrors
            rc = _lib.load().bgls_keys_upload(curve.id, _lib.buf(raw), n, None, 1, 1 if check else 0, ctypes.byref(h))
        if rc != 0:
            raise ValueError("bgls_keys_upload: %d %s" % (rc, _lib.last_error()))
        self.curve, self.n, self.handle = curve, n, h.value

    def __len__(self):
        return self.n

    def free(self):
        if self.handle:
            _lib.load().bgls_keys_free(self.handle)
            self.handle = 0

    def __del__(self):                                # the Go shim's runtime.SetFinalizer
        try:
            self.free()
        except Exception:
            pass


def _offsets(msgs):
    off = (ctypes.c_uint64 * (len(msgs) + 1))()
    acc = 0
    for i, m in enumerate(msgs):
        off[i] = acc
        acc += len(m)
    off[len(msgs)] = acc
    return off


def _verify_agg(curve, aggsig, keys, msgs, allow_duplicates):
    if len(keys) != len(msgs):                       # bgls/bgls.go:95-97
        return False
    n = len(keys)
    if not (isinstance(aggsig, Point) and aggsig.curve is curve and aggsig.group == G1):
        return False
    if isinstance(keys, KeySet):
        if keys.curve is not curve:
            return False
        return _lib.load().bgls_verify_aggregate_h(keys.handle, _lib.buf(aggsig.raw), _lib.buf(b"".join(bytes(m) for m in msgs)),
                                                   _offsets(msgs), n, 1 if allow_duplicates else 0) == 1
    for k in keys:
        if not (isinstance(k, Point) and k.curve is curve and k.group == G2):
            return False
    off = _offsets(msgs)
    rc = _lib.load().bgls_verify_aggregate(curve.id, _lib.buf(aggsig.raw), _lib.buf(b"".join(k.raw for k in keys)),
                                           _lib.buf(b"".join(bytes(m) for m in msgs)), off, n, 1 if allow_duplicates else 0)
    return rc == 1                                   # every failure collapses to false (bgls.go:115-118)


def VerifyAggregateSignature(curve, aggsig, keys, msgs):      # bgls/bgls.go:82-84
    return _verify_agg(curve, aggsig, keys, msgs, False)


def KoskVerifyAggregateSignature(curve, aggsig, keys, msgs):  # bgls/blsKosk.go:100-106
    return _verify_agg(curve, aggsig, keys, [b"\x01" + bytes(m) for m in msgs], True)


def _verify_multi(curve, aggsig, keys, msg):                  # bgls/bgls.go:89-92
    if isinstance(keys, KeySet):
        if keys.curve is not curve or not (isinstance(aggsig, Point) and aggsig.curve is curve and aggsig.group == G1):
            return False
        return _lib.load().bgls_verify_multi_h(keys.handle, _lib.buf(aggsig.raw), _lib.buf(msg), len(msg)) == 1
    if not (isinstance(aggsig, Point) and aggsig.curve is curve and aggsig.group == G1):
        return False                                 # a nil / foreign aggsig is `false`, never an exception
    for k in keys:
        if not (isinstance(k, Point) and k.curve is curve and k.group == G2):
            return False
    rc = _lib.load().bgls_verify_multi(curve.id, _lib.buf(aggsig.raw), _lib.buf(b"".join(k.raw for k in keys)), len(keys),
                                       _lib.buf(msg), len(msg))
    return rc == 1


def VerifySingleSignature(curve, sig, pubkey, msg):           # bgls/bgls.go:59-70
    return _verify_multi(curve, sig, [pubkey], bytes(msg))


def KoskVerifySingleSignature(curve, sig, pubkey, msg):       # bgls/blsKosk.go:86-90
    return VerifySingleSignature(curve, sig, pubkey, b"\x01" + bytes(msg))


def KoskVerifyMultiSignature(curve, aggsig, keys, msg):       # bgls/blsKosk.go:117-120
    return _verify_multi(curve, aggsig, keys, b"\x01" + bytes(msg))


def KoskVerifyBatchMultiSignature(curve, aggsigs, pubkeys, msgs):      # bgls/blsKosk.go:126-133
    """aggsigs: one multi-signature per message, pubkeys[i]: the signers of message i.  One call: the key sums of all sets
    in one launch, then ONE aggregate verification over len(msgs) pairs (the reference: AggregateSignatures, len(msgs) x
    AggregateKeys, KoskVerifyAggregateSignature)."""
    if len(aggsigs) != len(pubkeys) or len(pubkeys) != len(msgs) or not msgs:
        return False
    if any(s.curve is not curve or s.group != G1 for s in aggsigs):
        return False
    if any(k.curve is not curve or k.group != G2 for ks in pubkeys for k in ks):
        return False
    koff = (ctypes.c_uint64 * (len(pubkeys) + 1))()
    for i, ks in enumerate(pubkeys):
        koff[i + 1] = koff[i] + len(ks)
    pm = [b"\x01" + bytes(m) for m in msgs]
    moff = (ctypes.c_uint64 * (len(pm) + 1))()
    for i, m in enumerate(pm):
        moff[i + 1] = moff[i] + len(m)
    rc = _lib.load().bgls_verify_multi_batch(curve.id, _lib.buf(b"".join(s.raw for s in aggsigs)), _lib.buf(b"".join(k.raw for ks in pubkeys for k in ks)),
                                             koff, len(msgs), _lib.buf(b"".join(pm)), moff, 1)
    return rc == 1


class AggSig:                                                 # bgls/bgls.go:22-26,73-75
    def __init__(self, keys, msgs, sig):
        self.keys, self.msgs, self.sig = keys, msgs, sig

    def Verify(self, curve):
        return VerifyAggregateSignature(curve, self.sig, self.keys, self.msgs)


class MultiSig:                                               # bgls/bgls.go:15-19, blsKosk.go:110-112
    def __init__(self, keys, sig, msg):
        self.keys, self.sig, self.msg = keys, sig, msg

    def Verify(self, curve):
        return KoskVerifyMultiSignature(curve, self.sig, self.keys, self.msg)


# ---- hashed aggregation exponents (bgls/blsHAE.go) -----------------------------------------------------------
def _g2_keys_ok(curve, keys):
    return all(isinstance(k, Point) and k.curve is curve and k.group == G2 for k in keys)


def hashPubKeysToExponents(pubkeys):                          # bgls/blsHAE.go:80-93
    if not pubkeys:
        return []
    curve = pubkeys[0].curve if isinstance(pubkeys[0], Point) else None
    if curve is None or not _g2_keys_ok(curve, pubkeys):      # the C side reads n * g2_size bytes: G2 points of ONE curve only
        raise TypeError("hashPubKeysToExponents takes G2 public keys of one curve")
    o = _lib.out(16 * len(pubkeys))
    rc = _lib.load().bgls_hae_exponents(curve.id, _lib.buf(b"".join(k.raw for k in pubkeys)), len(pubkeys), o)
    if rc != 0:
        raise RuntimeError("bgls_hae_exponents: %s" % _lib.last_error())
    raw = bytes(o)
    return [int.from_bytes(raw[16 * i:16 * i + 16], "big") for i in range(len(pubkeys))]


def AggregateSignaturesWithHAE(sigs, pubkeys):                # bgls/blsHAE.go:39-46
    if len(pubkeys) != len(sigs):
        return None
    if not sigs:
        return AggregatePoints(sigs)
    curve = sigs[0].curve
    if not _g2_keys_ok(curve, pubkeys) or not all(isinstance(s, Point) and s.curve is curve and s.group == G1 for s in sigs):
        return None
    o = _lib.out(len(sigs[0].raw))
    rc = _lib.load().bgls_aggregate_signatures_hae(curve.id, _lib.buf(b"".join(s.raw for s in sigs)),
                                                   _lib.buf(b"".join(k.raw for k in pubkeys)), len(sigs), o)
    return Point(curve, G1, bytes(o)) if rc == 0 else None


def VerifyAggregateSignatureWithHAE(curve, aggsig, pubkeys, msgs):   # bgls/blsHAE.go:49-53
    if len(pubkeys) != len(msgs) or not _g2_keys_ok(curve, pubkeys):
        return False
    if not (isinstance(aggsig, Point) and aggsig.curve is curve and aggsig.group == G1):
        return False
    n = len(pubkeys)
    off = (ctypes.c_uint64 * (n + 1))()
    acc = 0
    for i, m in enumerate(msgs):
        off[i] = acc
        acc += len(m)
    off[n] = acc
    rc = _lib.load().bgls_verify_aggregate_hae(curve.id, _lib.buf(aggsig.raw), _lib.buf(b"".join(k.raw for k in pubkeys)),
                                               _lib.buf(b"".join(bytes(m) for m in msgs)), off, n)
    return rc == 1


def VerifyMultiSignatureWithHAE(curve, aggsig, pubkeys, msg):        # bgls/blsHAE.go:56-58
    if not _g2_keys_ok(curve, pubkeys) or not (isinstance(aggsig, Point) and aggsig.curve is curve and aggsig.group == G1):
        return False
    rc = _lib.load().bgls_verify_multi_hae(curve.id, _lib.buf(aggsig.raw), _lib.buf(b"".join(k.raw for k in pubkeys)), len(pubkeys),
                                           _lib.buf(msg), len(msg))
    return rc == 1


def KoskVerifyMultiSignatureWithMultiplicity(curve, aggsig, keys, multiplicity, msg):   # bgls/blsKosk.go:137-150
    if multiplicity is None:
        return KoskVerifyMultiSignature(curve, aggsig, keys, msg)
    if len(keys) != len(multiplicity):
        return False
    if not _g2_keys_ok(curve, keys) or not (isinstance(aggsig, Point) and aggsig.curve is curve and aggsig.group == G1):
        return False
    m2 = b"\x01" + bytes(msg)
    mult = (ctypes.c_int64 * max(1, len(keys)))(*[int(m) for m in multiplicity])
    rc = _lib.load().bgls_verify_multi_multiplicity(curve.id, _lib.buf(aggsig.raw), _lib.buf(b"".join(k.raw for k in keys)), mult,
                                                    len(keys), _lib.buf(m2), len(m2))
    return rc == 1


# ---- batch forms of KeyGen / Sign -------------------------------------------------------------
def LoadPublicKeys(curve, sks):
    """LoadPublicKey (bgls/bgls.go:40-43) for a list of secret keys in one call."""
    n = len(sks)
    if n == 0:
        return []
    size = len(curve.GetG2().raw)
    o = _lib.out(n * size)
    rc = _lib.load().bgls_scale_generator(curve.id, G2, _lib.buf(b"".join(_reduce_scalar(curve, sk if sk >= 0 else sk % curve.GetG1Order()).to_bytes(32, "big") for sk in sks)), n, o)
    if rc != 0:
        raise RuntimeError("bgls_scale_generator: %s" % _lib.last_error())
    raw = bytes(o)
    return [Point(curve, G2, raw[i * size:(i + 1) * size]) for i in range(n)]


def SignBatch(curve, sks, msgs, kosk=False):
    """Sign / KoskSign (bgls/bgls.go:46-56, bgls/blsKosk.go:73-77) for n (secret key, message) pairs in one call."""
    n = len(sks)
    if n != len(msgs):
        return None
    if n == 0:
        return []
    ms = [(b"\x01" + bytes(m)) if kosk else bytes(m) for m in msgs]
    off = (ctypes.c_uint64 * (n + 1))()
    acc = 0
    for i, m in enumerate(ms):
        off[i] = acc
        acc += len(m)
    off[n] = acc
    size = len(curve.GetG1().raw)
    o = _lib.out(n * size)
    rc = _lib.load().bgls_sign_batch(curve.id, _lib.buf(b"".join(_reduce_scalar(curve, sk if sk >= 0 else sk % curve.GetG1Order()).to_bytes(32, "big") for sk in sks)),
                                     _lib.buf(b"".join(ms)), off, n, o)
    if rc != 0:
        raise RuntimeError("bgls_sign_batch: %s" % _lib.last_error())
    raw = bytes(o)
    return [Point(curve, G1, raw[i * size:(i + 1) * size]) for i in range(n)]


# ---- wrappers that are byte concatenation in front of the same path ---------------------------------------------
def DistinctMsgSign(curve, sk, msg):                          # bgls/blsDistinctMessage.go:23-34
    return Sign(curve, sk, LoadPublicKey(curve, sk).MarshalUncompressed() + bytes(msg))


def DistinctMsgVerifySingleSignature(curve, sig, pubkey, msg):   # bgls/blsDistinctMessage.go:37-40
    return VerifySingleSignature(curve, sig, pubkey, pubkey.MarshalUncompressed() + bytes(msg))


def DistinctMsgVerifyAggregateSignature(curve, aggsig, keys, msgs):   # bgls/blsDistinctMessage.go:45-57
    if len(keys) != len(msgs):
        return False
    return _verify_agg(curve, aggsig, keys, [k.MarshalUncompressed() + bytes(m) for k, m in zip(keys, msgs)], True)


def Authenticate(curve, sk):                                  # bgls/blsKosk.go:44-55: a signature on the marshalled key
    return Sign(curve, sk, LoadPublicKey(curve, sk).Marshal())


def CheckAuthentication(curve, pubkey, authentication):      # bgls/blsKosk.go:59-69
    return VerifySingleSignature(curve, authentication, pubkey, pubkey.Marshal())


def KoskVerifyBatchMultiSignatureStepwise(curve, aggsigs, pubkeys, msgs):    # bgls/blsKosk.go:126-133, call by call as the reference writes it
    aggsig = AggregateSignatures(aggsigs)
    keys = [AggregateKeys(ks) for ks in pubkeys]
    return KoskVerifyAggregateSignature(curve, aggsig, keys, msgs)


def VerifyBatchMultiSignatureWithHAE(curve, aggsigs, aggpubkeys, msgs, allowDups):   # bgls/blsHAE.go:62-72
    """As in the reference, the random factors of the allowDups branch are applied to a throw-away copy (ScalePoints
    returns new points and the result is discarded, blsHAE.go:64-69), so both branches verify the plain aggregate."""
    if allowDups:
        ScalePoints(aggsigs, [secrets.randbelow(curve.GetG1Order()) for _ in aggsigs])
    return _verify_agg(curve, AggregateSignatures(aggsigs), aggpubkeys, msgs, True)


# ---- accountable-subgroup multisignatures (bgls/blsAsmSigs.go) -----------------------------------------------------
def _ams_h0(curve, msg):                                      # getAmsH0, blsAsmSigs.go:73-78
    return curve.HashToG1(b"\x00" + bytes(msg))


def _ams_h2(curve, apk, msg):                                 # getAmsH2, blsAsmSigs.go:80-86
    return curve.HashToG1(b"\x01" + apk.MarshalUncompressed() + bytes(msg))


def AmsCreateMembershipKeySharesKnownExp(curve, sk, apk, exp, numSigners):   # blsAsmSigs.go:23-30
    return [_ams_h2(curve, apk, str(i).encode()).Mul(sk).Mul(exp) for i in range(numSigners)]


def AmsCreateMembershipKeyShares(curve, sk, curIndex, pubkeys):              # blsAsmSigs.go:17-21
    t = hashPubKeysToExponents(pubkeys)
    apk = AggregatePoints(ScalePoints(pubkeys, t))
    return AmsCreateMembershipKeySharesKnownExp(curve, sk, apk, t[curIndex], len(pubkeys))


def AmsAggregateMembershipKeyShares(curve, shares):           # blsAsmSigs.go:32-34
    return AggregatePoints(shares)


def AmsCreateSignatureShare(curve, sk, membershipKey, msg):   # blsAsmSigs.go:36-40
    sig, _ = _ams_h0(curve, msg).Mul(sk).Add(membershipKey)
    return sig


def AmsCombineSignatureShares(pubkeys, sigs):                 # blsAsmSigs.go:42-46
    return AggregatePoints(pubkeys), AggregateSignatures(sigs)


def AmsVerifySignature(curve, apk, signers, aggKey, aggSig, msg):   # blsAsmSigs.go:48-59: a three-pairing product
    aggMsg = AggregatePoints([_ams_h2(curve, apk, str(i).encode()) for i in signers])
    pt, ok = curve.PairingProduct([_ams_h0(curve, msg), aggMsg, aggSig.Mul(-1)], [aggKey, apk, curve.GetG2()])
    return ok and pt.Equals(curve.GetGTIdentity())


def AmsVerifySignatureWithSetCheck(curve, check, apk, signers, aggKey, aggSig, msg):   # blsAsmSigs.go:61-66
    if not check(signers):
        return False
    return AmsVerifySignature(curve, apk, signers, aggKey, aggSig, msg)
