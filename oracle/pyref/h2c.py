"""Hash-to-G1 for both curves (CPU oracle -- TEST INFRASTRUCTURE ONLY).

alt-bn128 : HashToG1 (curves/altbn128.go:509-513) -> AltbnKeccak3 (:494-497) ->
            tryAndIncrementEvm (curves/hash.go:53-77) with g1XToYSquared (altbn128.go:409-414),
            calcQuadRes (hash.go:178-190), EthereumSum256 (altbn128.go:517-522).
BLS12-381 : HashToG1 (curves/bls12_381.go:349-351) -> hashToG1BlindingAbstracted(msg,false)
            (:361-376) -> bls12Blake2b (:397-400) -> bls12FouqueTibouchi (:378-393) ->
            fouqueTibouchiG1 (curves/hash.go:86-93) -> sw (:97-167), chkPoint/isQuadRes
            (:234-265), parity (:169-172).
Pinned by the reference's own vectors: curves/testcases/{altbn128,bls12}G1Hash.dat,
curves/altbn128_test.go:16-21, curves/bls12_test.go:27-67 (tests/test_oracle_h2c.py).
"""
from .params import BN254, BLS381
from .hashes import keccak256_legacy, blake2b512
from .groups import Groups


def calc_quad_res(a, q):            # hash.go:178-190 (q = 3 mod 4)
    return pow(a, (q + 1) // 4, q)


def is_quad_res(a, q):              # hash.go:254-265 (0 counts as a square)
    if a % q == 0:
        return True
    return pow(a, (q - 1) // 2, q) == 1


def parity(x, q):                   # hash.go:169-172
    return x > q - x


def altbn_hash_to_g1(msg: bytes):
    """tryAndIncrementEvm.  Returns (x, y, tries)."""
    q = BN254.p
    c = 0
    tries = 0
    while True:
        h = keccak256_legacy(bytes([c]) + msg)
        c = (c + 1) & 0xFF
        tries += 1
        x = int.from_bytes(h, "big") % q
        y2 = pow(x, 3, q) + 3          # NOT reduced, as in altbn128.go:409-414
        root = calc_quad_res(y2, q)
        if root * root % q == y2:
            y = root
            s = keccak256_legacy(b"\xff" + msg)[31] % 2
            if s == 1:
                y = q - y
            return x, y, tries


_G_BLS = Groups(BLS381)


def bls_sw_encode(t):
    """sw(curve, t, blind=False) (hash.go:97-167): returns affine (x, y) and the index of
    the candidate that was taken."""
    q, b = BLS381.p, BLS381.b
    w = (t * t + 1 + b) % q
    w = pow(w, q - 2, q)              # ModInverse
    w = w * t % q
    w = w * BLS381.sqrt_m3 % q
    x0 = (BLS381.z_sw - t * w) % q
    if is_quad_res((pow(x0, 3, q) + b) % q, q):
        x, i = x0, 0
    else:
        x1 = (-x0 - 1) % q
        if is_quad_res((pow(x1, 3, q) + b) % q, q):
            x, i = x1, 1
        else:
            x, i = (pow(w * w % q, q - 2, q) + 1) % q, 2
    y = calc_quad_res((pow(x, 3, q) + b) % q, q)
    if parity(y, q) != parity(t, q):
        y = q - y
    return (x, y), i


def bls_fouque_tibouchi(t_bytes: bytes):
    """bls12FouqueTibouchi (bls12_381.go:378-393)."""
    q = BLS381.p
    t = int.from_bytes(t_bytes, "big") % q
    if t == 0:
        return None
    if t == BLS381.ft_root1:
        return BLS381.g1
    if t == BLS381.ft_root2:
        return _G_BLS.g1_neg(BLS381.g1)
    pt, _ = bls_sw_encode(t)
    return _G_BLS.g1_mul(pt, BLS381.cofactor)


def bls_hash_to_g1(msg: bytes):
    p1 = bls_fouque_tibouchi(blake2b512(msg + b"G1_0"))
    p2 = bls_fouque_tibouchi(blake2b512(msg + b"G1_1"))
    return _G_BLS.g1_add(p1, p2)


def hash_to_g1(curve, msg: bytes):
    if curve.name == "altbn128":
        x, y, _ = altbn_hash_to_g1(msg)
        return (x, y)
    return bls_hash_to_g1(msg)
