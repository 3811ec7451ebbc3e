"""ctypes wrapper of the C oracle (oracle/c/liboracle.so) -- TEST INFRASTRUCTURE ONLY.
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "liboracle.so")
_lib = None
u8p = ctypes.POINTER(ctypes.c_uint8)
FP = {0: 32, 1: 48}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "c")], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _b(x):
    x = bytes(x)
    return (ctypes.c_uint8 * max(1, len(x))).from_buffer_copy(x if x else b"\0")


def _offsets(msgs):
    off = (ctypes.c_uint64 * (len(msgs) + 1))()
    acc = 0
    for i, m in enumerate(msgs):
        off[i] = acc
        acc += len(m)
    off[len(msgs)] = acc
    return off


def hash_to_g1(curve, msg):
    o = (ctypes.c_uint8 * (2 * FP[curve]))()
    rc = lib().oracle_hash_to_g1(curve, _b(msg), ctypes.c_size_t(len(msg)), o)
    assert rc == 0, rc
    return bytes(o)


def miller(curve, g1, g2):
    o = (ctypes.c_uint8 * (12 * FP[curve]))()
    assert lib().oracle_miller(curve, _b(g1), _b(g2), o) == 0
    return bytes(o)


def final_exp(curve, gt):
    o = (ctypes.c_uint8 * (12 * FP[curve]))()
    assert lib().oracle_final_exp(curve, _b(gt), o) == 0
    return bytes(o)


def pairing_product(curve, g1s, g2s, n, threads=1, faithful=0):
    o = (ctypes.c_uint8 * (12 * FP[curve]))()
    rc = lib().oracle_pairing_product(curve, _b(g1s), _b(g2s), ctypes.c_size_t(n), o, threads, faithful)
    assert rc == 0, rc
    return bytes(o)


def verify_aggregate(curve, sig, keys, msgs, allow_dups=False, threads=1, faithful=0):
    return lib().oracle_verify_aggregate(curve, _b(sig), _b(keys), _b(b"".join(msgs)), _offsets(msgs),
                                         ctypes.c_size_t(len(msgs)), 1 if allow_dups else 0, threads, faithful)


def verify_multi(curve, sig, keys, n, msg, faithful=0):
    return lib().oracle_verify_multi(curve, _b(sig), _b(keys), ctypes.c_size_t(n), _b(msg), ctypes.c_size_t(len(msg)), faithful)


def aggregate_points(curve, group, pts, n):
    size = (2 if group == 1 else 4) * FP[curve]
    o = (ctypes.c_uint8 * size)()
    assert lib().oracle_aggregate_points(curve, group, _b(pts), ctypes.c_size_t(n), o) == 0
    return bytes(o)


def scale_point(curve, group, pt, k):
    size = (2 if group == 1 else 4) * FP[curve]
    o = (ctypes.c_uint8 * size)()
    assert lib().oracle_scale_point(curve, group, _b(pt), _b(abs(k).to_bytes(32, "big")), 1 if k < 0 else 0, o) == 0
    return bytes(o)


def g2_in_subgroup(curve, pt):
    """1 = on the twist and [r]Q = infinity, 0 = not, -2 = non-canonical coordinates"""
    return lib().oracle_g2_in_subgroup(curve, _b(pt))


def g1_in_subgroup(curve, pt):
    """1 = on the curve and [r]P = infinity, 0 = not, -2 = non-canonical coordinates"""
    return lib().oracle_g1_in_subgroup(curve, _b(pt))


def miller_product(curve, g1s, g2s, n, threads=1):
    o = (ctypes.c_uint8 * (12 * FP[curve]))()
    rc = lib().oracle_miller_product(curve, _b(g1s), _b(g2s), ctypes.c_size_t(n), o, threads)
    assert rc == 0, rc
    return bytes(o)


def gt_mul(curve, a, b):
    o = (ctypes.c_uint8 * (12 * FP[curve]))()
    assert lib().oracle_gt_mul(curve, _b(a), _b(b), o) == 0
    return bytes(o)


# ---- hashed aggregation exponents / multiplicities (bgls/blsHAE.go, bgls/blsKosk.go:137-150), composed from the C pieces ----
def blake2xb(data, out_len):
    o = (ctypes.c_uint8 * max(1, out_len))()
    assert lib().oracle_blake2xb(_b(data), ctypes.c_size_t(len(data)), ctypes.c_uint32(out_len), o) == 0
    return bytes(o)[:out_len]


def hae_exponents(curve, keys, n):
    """blsHAE.go:80-93 over the uncompressed G2 wire bytes of the keys."""
    raw = blake2xb(bytes(keys), 16 * n)
    return [int.from_bytes(raw[16 * i:16 * i + 16], "big") for i in range(n)]


def _scaled(curve, group, pts, n, ks):
    size = (2 if group == 1 else 4) * FP[curve]
    return b"".join(scale_point(curve, group, pts[i * size:(i + 1) * size], ks[i]) for i in range(n))


def aggregate_signatures_hae(curve, sigs, keys, n):
    return aggregate_points(curve, 1, _scaled(curve, 1, bytes(sigs), n, hae_exponents(curve, keys, n)), n)


def verify_multi_hae(curve, sig, keys, n, msg):
    apk = aggregate_points(curve, 2, _scaled(curve, 2, bytes(keys), n, hae_exponents(curve, keys, n)), n)
    return verify_multi(curve, sig, apk, 1, msg)


def verify_aggregate_hae(curve, sig, keys, msgs):
    n = len(msgs)
    return verify_aggregate(curve, sig, _scaled(curve, 2, bytes(keys), n, hae_exponents(curve, keys, n)), msgs, allow_dups=True)


def verify_multi_multiplicity(curve, sig, keys, n, mult, msg):
    return verify_multi(curve, sig, _scaled(curve, 2, bytes(keys), n, list(mult)), n, b"\x01" + msg)
