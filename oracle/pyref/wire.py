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

BLS12-381's compressed encodings come from the un-vendored dis2/bls12 and carry TODOs in the reference
(curves/bls12_381.go:55,60,116,121): unpinned, not restated.
"""
from .params import BN254

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
