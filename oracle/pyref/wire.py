"""Compressed wire formats of alt-bn128 points (CPU oracle -- TEST INFRASTRUCTURE ONLY).

The compressed forms are defined by the reference's own code, not by an upstream library, so this file follows it
statement by statement, quirks included:

  Marshal        G1  curves/altbn128.go:81-89     x (32-byte BE), top bit of byte 0 set iff 2y > q
                 G2  curves/altbn128.go:203-221   x_im || x_re, top bit of each set iff 2 y_im > q / 2 y_re > q
  Unmarshal*     G1  curves/altbn128.go:296-327   compressed branch (len 32)
                 G2  curves/altbn128.go:329-376   compressed branch (len 64), square root by calcComplexQuadRes
                                                  (curves/hash.go:196-223, "Algorithm 18")
  final check        MakeG1Point / MakeG2Point -> upstream Unmarshal (curves/altbn128.go:42-57,157-179): canonical
                     coordinates (< q), curve membership and -- G2 -- membership in the order-r subgroup (upstream
                     bn256's twistPoint.IsOnCurve multiplies by the group order; oracle/pyref/subgroup.py).
                     decompress_g2(..., subgroup=False) stops before that last test (what wire.hpp's g2_decompress
                     computes; the kernel applies g2_in_subgroup to its result).

BLS12-381 (second half of this file): Marshal / UnmarshalG1 / UnmarshalG2 of the reference hand the bytes to the un-vendored
dis2/bls12 (curves/bls12_381.go:54-62,115-123,242-264) and say what layout they are meant to have -- "TODO Make this match
ebfull/pairing marshalling" (:54,59,115,120).  ebfull/pairing's layout (the "ZCash" serialisation, also Appendix C of
draft-irtf-cfrg-pairing-friendly-curves) is a published format, so THAT is what is restated here:
  compressed G1  48 bytes: x big-endian; byte 0 bit 7 = 1 (compressed), bit 6 = infinity (all other bits zero), bit 5 = y is the
                 lexicographically larger of {y, -y} (y > (p - 1) / 2)
  compressed G2  96 bytes: x.c1 || x.c0 with the same three flag bits in byte 0; "larger" compares (c1, c0) lexicographically
  decoding       flag checks, x < p, y = sqrt(x^3 + b) selected by the sort flag, then Check() = subgroup membership
                 (curves/bls12_381.go:248,259).
PARITY UNPINNED against dis2/bls12 itself (absent, no vector in the reference); pinned instead by the format's public
known-answer values: the generators' encodings (tests/test_wire.py).
"""
from .params import BN254, BLS381

Q = BN254.p
_B2 = None


def _b2():
    global _B2
    if _B2 is None:
        from .groups import Groups
        _B2 = Groups(BN254).b2          # (re, im) of 3 / (9 + i), altbnG2BRe / altbnG2BIm (altbn128.go:462-464)
    return _B2


def calc_quad_res(a):                   # curves/hash.go:178-190, q = 3 mod 4
    return pow(a, (Q + 1) // 4, Q)


def is_quad_res(a):                     # curves/hash.go:254-265
    return a % Q == 0 or pow(a, (Q - 1) // 2, Q) == 1


def _cmul(a, b):                        # complexNum.Mul, (re, im)
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[1] * b[0] + a[0] * b[1]) % Q)


def calc_complex_quad_res(y2):
    """curves/hash.go:196-223; y2 = (re, im).  Returns (re, im) or None where the reference would dereference the nil of
    ModInverse(0) (treated as "no point")."""
    re, im = y2
    if im == 0:
        return (calc_quad_res(re), 0)
    lam = calc_quad_res((re * re + im * im) % Q)
    inv2 = pow(2, -1, Q)
    delta = (re + lam) % Q * inv2 % Q
    if not is_quad_res(delta):
        delta = (re - lam) * inv2 % Q
    rre = calc_quad_res(delta)
    if rre == 0:
        return None
    rim = pow(rre, -1, Q) * inv2 % Q * im % Q
    return (rre % Q, rim)


def g1_on_curve(x, y):
    return (y * y - x * x * x - 3) % Q == 0


def g2_on_curve(x, y):                  # x, y = (re, im)
    x3 = _cmul(_cmul(x, x), x)
    b = _b2()
    y2 = _cmul(y, y)
    return y2 == ((x3[0] + b[0]) % Q, (x3[1] + b[1]) % Q)


# ---- Marshal ---------------------------------------------------------------------------------------------------
def compress_g1(P):
    """P = (x, y) or None (infinity: ToAffineCoords gives (0, 0))."""
    x, y = (0, 0) if P is None else P
    b = bytearray(x.to_bytes(32, "big"))
    if 2 * y > Q:
        b[0] += 128
    return bytes(b)


def compress_g2(P):
    """P = ((x_re, x_im), (y_re, y_im)) or None."""
    (xr, xi), (yr, yi) = ((0, 0), (0, 0)) if P is None else P
    bi, br = bytearray(xi.to_bytes(32, "big")), bytearray(xr.to_bytes(32, "big"))
    if 2 * yi > Q:
        bi[0] += 128
    if 2 * yr > Q:
        br[0] += 128
    return bytes(bi) + bytes(br)


# ---- Unmarshal, compressed branch -------------------------------------------------------------------------------
def decompress_g1(data):
    """-> (point or None for infinity, ok)."""
    assert len(data) == 32
    d = bytearray(data)
    ysgn = d[0] >= 128
    if ysgn:
        d[0] -= 128
    x = int.from_bytes(d, "big")
    if x == 0:
        return None, True
    y = calc_quad_res((pow(x, 3, Q) + 3) % Q)
    cmp2 = (2 * y > Q) - (2 * y < Q)
    if ysgn and cmp2 == -1:
        y = Q - y
    elif (not ysgn) and cmp2 == 1:
        y = Q - y
    if x >= Q or y >= Q or not g1_on_curve(x, y):          # MakeG1Point -> upstream Unmarshal
        return None, False
    return (x, y), True


def decompress_g2(data, subgroup=True):
    assert len(data) == 64
    di, dr = bytearray(data[:32]), bytearray(data[32:])
    yisgn, yrsgn = di[0] >= 128, dr[0] >= 128
    if yisgn:
        di[0] -= 128
    if yrsgn:
        dr[0] -= 128
    xi, xr = int.from_bytes(di, "big"), int.from_bytes(dr, "big")
    if xi == 0 and xr == 0:
        return None, True
    x = (xr % Q, xi % Q)
    x3 = _cmul(_cmul(x, x), x)
    b = _b2()
    y = calc_complex_quad_res(((x3[0] + b[0]) % Q, (x3[1] + b[1]) % Q))
    if y is None:
        return None, False
    yr, yi = y
    ci = (2 * yi > Q) - (2 * yi < Q)
    cr = (2 * yr > Q) - (2 * yr < Q)
    if yisgn and ci == -1:
        yi = Q - yi
    elif (not yisgn) and ci == 1:
        yi = Q - yi
    if yrsgn and cr == -1:
        yr = Q - yr
    elif (not yrsgn) and cr == 1:
        yr = Q - yr
    if xi >= Q or xr >= Q or yi >= Q or yr >= Q or not g2_on_curve((xr, xi), (yr, yi)):   # MakeG2Point -> upstream Unmarshal
        return None, False
    if subgroup:
        from .subgroup import Subgroup
        if not Subgroup(BN254).in_subgroup(((xr, xi), (yr, yi))):
            return None, False
    return ((xr, xi), (yr, yi)), True


# ================================================================================================================
# BLS12-381: ebfull/pairing ("ZCash") layout -- see the header.  Points are Groups(BLS381) tuples, None = infinity.
# ================================================================================================================
P381 = BLS381.p
_HALF381 = (P381 - 1) // 2


def _sqrt381(a):
    """square root in Fp (p = 3 mod 4) or None"""
    a %= P381
    r = pow(a, (P381 + 1) // 4, P381)
    return r if r * r % P381 == a else None


def _f2_sqrt381(a):
    """square root in Fp2 = Fp[i] / (i^2 + 1) by the complex method, or None; which of the two roots is irrelevant (the
    sort flag selects)."""
    re, im = a[0] % P381, a[1] % P381
    if im == 0:
        r = _sqrt381(re)
        if r is not None:
            return (r, 0)
        r = _sqrt381(-re)                                   # re a non-residue: the root is purely imaginary
        return None if r is None else (0, r)
    lam = _sqrt381(re * re + im * im)                       # the norm must be a square in Fp
    if lam is None:
        return None
    inv2 = pow(2, -1, P381)
    x0 = _sqrt381((re + lam) * inv2)
    if x0 is None:
        x0 = _sqrt381((re - lam) * inv2)
    if x0 is None or x0 == 0:
        return None
    x1 = im * pow(2 * x0, -1, P381) % P381
    ok = ((x0 * x0 - x1 * x1) % P381, 2 * x0 * x1 % P381) == (re, im)
    return (x0, x1) if ok else None


def _larger_fp(y):
    return y > _HALF381


def _larger_fp2(y):                                         # (re, im): compare c1 = im first, then c0
    return _larger_fp(y[1]) if y[1] != 0 else _larger_fp(y[0])


def bls_compress_g1(P):
    if P is None:
        return bytes([0xC0]) + bytes(47)
    b = bytearray(P[0].to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if _larger_fp(P[1]) else 0)
    return bytes(b)


def bls_compress_g2(Q):
    if Q is None:
        return bytes([0xC0]) + bytes(95)
    (xr, xi), y = Q
    b = bytearray(xi.to_bytes(48, "big") + xr.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if _larger_fp2(y) else 0)
    return bytes(b)


def _bls_flags(data):
    d = bytearray(data)
    c, inf, srt = d[0] >> 7, (d[0] >> 6) & 1, (d[0] >> 5) & 1
    d[0] &= 0x1F
    return c, inf, srt, d


def bls_decompress_g1(data, subgroup=True):
    """-> (point or None for infinity, ok): UnmarshalG1 on 48 bytes (curves/bls12_381.go:242-251)."""
    assert len(data) == 48
    c, inf, srt, d = _bls_flags(data)
    if not c:
        return None, False                                  # 48 bytes that do not claim to be compressed
    if inf:
        return None, (srt == 0 and not any(d))              # infinity: every other bit must be zero
    x = int.from_bytes(d, "big")
    if x >= P381:
        return None, False
    y = _sqrt381(x * x * x + BLS381.b)
    if y is None:
        return None, False
    if _larger_fp(y) != bool(srt):
        y = P381 - y
    P = (x, y)
    if subgroup:
        from .groups import Groups
        if Groups(BLS381).g1_mul(P, BLS381.r) is not None:  # Check(): [r]P = infinity
            return None, False
    return P, True


def bls_decompress_g2(data, subgroup=True):
    """UnmarshalG2 on 96 bytes (curves/bls12_381.go:253-262)."""
    assert len(data) == 96
    c, inf, srt, d = _bls_flags(data)
    if not c:
        return None, False
    if inf:
        return None, (srt == 0 and not any(d))
    xi, xr = int.from_bytes(d[:48], "big"), int.from_bytes(d[48:], "big")
    if xi >= P381 or xr >= P381:
        return None, False
    from .groups import Groups
    G = Groups(BLS381)
    T = G.T
    x = (xr, xi)
    y = _f2_sqrt381(T.f2_add(T.f2_mul(T.f2_sqr(x), x), G.b2))
    if y is None:
        return None, False
    if _larger_fp2(y) != bool(srt):
        y = T.f2_neg(y)
    Q = (x, y)
    if subgroup:
        from .subgroup import Subgroup
        if not Subgroup(BLS381).in_subgroup(Q):
            return None, False
    return Q, True
