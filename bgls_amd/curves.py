"""Host-side mirror of the reference's `curves` package boundary (curves/curve.go:12-70):
CurveSystem / Point / PointT with the same method names, argument meaning and error behaviour
((value, ok) pairs, never exceptions for bad data), every method backed by the HIP library
through the C ABI.  This is the binding a maintainer would write in Go with cgo
(INTEGRATION.md shows that shim); the Python form exists so the parity tests can read like
curves/curve_test.go and bgls/bgls_test.go.

All arithmetic happens on the GPU.  Nothing here computes field or group operations.
"""
import ctypes
from . import _lib

ALTBN128, BLS12_381 = 0, 1
G1, G2 = 1, 2


def _reduce_scalar(curve, mag):
    """The ABI takes 32-byte magnitudes.  The reference multiplies by the exact big.Int; on points of order r (every
    validated Point, every GT element) a magnitude of 2^256 or more acts as its residue modulo the group order, so larger
    factors are reduced modulo GetG1Order() -- in every path, never modulo 2^256."""
    return mag if mag < 1 << 256 else mag % curve.GetG1Order()


class Point:
    """curves.Point (curves/curve.go:51-59) over uncompressed wire bytes."""

    __slots__ = ("curve", "group", "raw")

    def __init__(self, curve, group, raw):
        self.curve, self.group, self.raw = curve, group, bytes(raw)

    def Add(self, other):
        if not isinstance(other, Point) or other.curve is not self.curve or other.group != self.group:
            return None, False                      # type mismatch => nil,false (altbn128.go:60-65)
        o = _lib.out(len(self.raw))
        rc = _lib.load().bgls_point_add(self.curve.id, self.group, _lib.buf(self.raw), _lib.buf(other.raw), o)
        if rc != 0:
            return None, False
        return Point(self.curve, self.group, bytes(o)), True

    def Copy(self):
        return Point(self.curve, self.group, self.raw)

    def Equals(self, other):
        return isinstance(other, Point) and other.curve is self.curve and other.group == self.group and other.raw == self.raw

    def MarshalUncompressed(self):
        return self.raw

    def Marshal(self):
        """Point.Marshal: the compressed form -- alt-bn128's own 32 / 64-byte format (curves/altbn128.go:81-89,203-221);
        BLS12-381: 48 / 96 bytes in the ebfull/pairing layout the reference names as its target (curves/bls12_381.go:54-62,
        115-123; unpinned against the un-vendored dis2/bls12, see include/bgls_hip.h)."""
        o = _lib.out(len(self.raw) // 2)
        rc = _lib.load().bgls_compress_points(self.curve.id, self.group, _lib.buf(self.raw), 1, o)
        if rc != 0:
            raise RuntimeError("bgls_compress_points: %d %s" % (rc, _lib.last_error()))
        return bytes(o)

    def Mul(self, scalar):
        """Point.Mul (altbn128.go:107-121,235-249; bls12_381.go:65-76,126-137): negative scalars
        negate then multiply, zero gives infinity.  The caller's scalar is NOT mutated."""
        sign = 1 if scalar < 0 else 0
        mag = _reduce_scalar(self.curve, -scalar if scalar < 0 else scalar)
        o = _lib.out(len(self.raw))
        rc = _lib.load().bgls_scale_points(self.curve.id, self.group, _lib.buf(self.raw), _lib.buf(mag.to_bytes(32, "big")),
                                           _lib.buf(bytes([sign])), 1, o)
        if rc != 0:
            raise RuntimeError("bgls_scale_points: %d %s" % (rc, _lib.last_error()))
        return Point(self.curve, self.group, bytes(o))

    def ToAffineCoords(self):
        n = self.curve.fp_bytes
        v = [int.from_bytes(self.raw[i * n:(i + 1) * n], "big") for i in range(len(self.raw) // n)]
        return v                                    # G1: [x, y]; G2: [x_im, x_re, y_im, y_re] (altbn128.go:251-262)


class PointT:
    """curves.PointT (curves/curve.go:62-70): an element of GT."""

    __slots__ = ("curve", "raw")

    def __init__(self, curve, raw):
        self.curve, self.raw = curve, bytes(raw)

    def Add(self, other):
        if not isinstance(other, PointT) or other.curve is not self.curve:
            return None, False
        o = _lib.out(len(self.raw))
        rc = _lib.load().bgls_gt_mul(self.curve.id, _lib.buf(self.raw), _lib.buf(other.raw), o)
        if rc != 0:
            return None, False
        return PointT(self.curve, bytes(o)), True

    def Copy(self):
        return PointT(self.curve, self.raw)

    def Equals(self, other):                         # bytes compare, as altbn128.go:283-288
        return isinstance(other, PointT) and other.curve is self.curve and other.raw == self.raw

    def Marshal(self):
        return self.raw

    def Mul(self, scalar):
        """PointT.Mul (curves/altbn128.go:273-281, curves/bls12_381.go:170-173): exponentiation in GT."""
        sign = 1 if scalar < 0 else 0
        mag = _reduce_scalar(self.curve, -scalar if scalar < 0 else scalar)
        o = _lib.out(len(self.raw))
        rc = _lib.load().bgls_gt_pow(self.curve.id, _lib.buf(self.raw), _lib.buf(mag.to_bytes(32, "big")), sign, o)
        if rc != 0:
            return None
        return PointT(self.curve, bytes(o))


class CurveSystem:
    """curves.CurveSystem (curves/curve.go:12-49)."""

    def __init__(self, cid, name, q, order):
        self.id, self._name, self._q, self._order = cid, name, q, order
        self.fp_bytes = 32 if cid == ALTBN128 else 48

    def Name(self):
        return self._name

    def GetG1Q(self):
        return self._q

    def GetG1Order(self):
        return self._order

    def _pt_size(self, group):
        return (2 if group == G1 else 4) * self.fp_bytes

    def _make(self, group, coords, check):
        cnt = 2 if group == G1 else 4
        if len(coords) != cnt:
            return None, False
        if any(c < 0 or c >= 1 << (8 * self.fp_bytes) for c in coords):
            return None, False
        raw = b"".join(int(c).to_bytes(self.fp_bytes, "big") for c in coords)
        # upstream always validates on unmarshal (altbn128.go:39-41,157-160); bls12 only with check
        if check or self.id == ALTBN128:
            if _lib.load().bgls_point_check(self.id, group, _lib.buf(raw)) != 1:
                return None, False
        return Point(self, group, raw), True

    def MakeG1Point(self, coords, check=True):
        return self._make(G1, coords, check)

    def MakeG2Point(self, coords, check=True):
        return self._make(G2, coords, check)

    def _unmarshal(self, group, data):
        if data is not None and len(data) * 2 == self._pt_size(group):
            # compressed branch of UnmarshalG1 / UnmarshalG2 (curves/altbn128.go:296-376; curves/bls12_381.go:242-264 on 48 / 96
            # bytes); the caller's bytes are NOT mutated (the reference clears the sign bit in place, altbn128.go:306-309,344-349)
            o, ok = _lib.out(self._pt_size(group)), _lib.out(1)
            rc = _lib.load().bgls_decompress_points(self.id, group, _lib.buf(data), 1, o, ok)
            if rc != 0 or ok[0] != 1:
                return None, False
            return Point(self, group, bytes(o)), True
        if data is None or len(data) != self._pt_size(group):
            return None, False
        if _lib.load().bgls_point_check(self.id, group, _lib.buf(data)) != 1:
            return None, False
        return Point(self, group, data), True

    def UnmarshalG1(self, data):
        return self._unmarshal(G1, data)

    def UnmarshalG2(self, data):
        return self._unmarshal(G2, data)

    def UnmarshalGT(self, data):
        if data is None or len(data) != 12 * self.fp_bytes:
            return None, False
        return PointT(self, data), True

    def _gen(self, group):
        o = _lib.out(self._pt_size(group))
        rc = _lib.load().bgls_generator(self.id, group, o)
        if rc != 0:
            raise RuntimeError("bgls_generator: %d %s" % (rc, _lib.last_error()))
        return Point(self, group, bytes(o))

    def GetG1(self):
        return self._gen(G1)

    def GetG2(self):
        return self._gen(G2)

    def GetGT(self):
        pt, _ = self.Pair(self.GetG1(), self.GetG2())
        return pt

    def GetG1Infinity(self):
        return Point(self, G1, bytes(self._pt_size(G1)))

    def GetG2Infinity(self):
        return Point(self, G2, bytes(self._pt_size(G2)))

    def GetGTIdentity(self):
        o = _lib.out(12 * self.fp_bytes)
        _lib.load().bgls_gt_identity(self.id, o)
        return PointT(self, bytes(o))

    def HashToG1(self, message):
        pts = self.HashToG1Batch([message])
        return pts[0]

    def HashToG1Batch(self, messages):
        n = len(messages)
        blob = b"".join(bytes(m) for m in messages)
        off = (ctypes.c_uint64 * (n + 1))()
        acc = 0
        for i, m in enumerate(messages):
            off[i] = acc
            acc += len(m)
        off[n] = acc
        o = _lib.out(n * self._pt_size(G1))
        rc = _lib.load().bgls_hash_to_g1(self.id, _lib.buf(blob), off, n, o)
        if rc != 0:
            raise RuntimeError("bgls_hash_to_g1: %d %s" % (rc, _lib.last_error()))
        s = self._pt_size(G1)
        raw = bytes(o)
        return [Point(self, G1, raw[i * s:(i + 1) * s]) for i in range(n)]

    def Pair(self, p1, p2):
        return self.PairingProduct([p1], [p2])

    def PairingProduct(self, pts1, pts2):
        """One C call for the whole slice (replaces concurrentPairingProduct, curve.go:125-170)."""
        if len(pts1) != len(pts2):
            return None, False
        for a, b in zip(pts1, pts2):
            if not (isinstance(a, Point) and isinstance(b, Point) and a.curve is self and b.curve is self
                    and a.group == G1 and b.group == G2):
                return None, False
        o = _lib.out(12 * self.fp_bytes)
        rc = _lib.load().bgls_pairing_product(self.id, _lib.buf(b"".join(p.raw for p in pts1)),
                                              _lib.buf(b"".join(p.raw for p in pts2)), len(pts1), o)
        if rc != 0:
            return None, False
        return PointT(self, bytes(o)), True


Altbn128 = CurveSystem(ALTBN128, "altbn128",
                       21888242871839275222246405745257275088696311157297823662689037894645226208583,
                       21888242871839275222246405745257275088548364400416034343698204186575808495617)
Bls12 = CurveSystem(BLS12_381, "bls12",
                    0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
                    52435875175126190479447740508185965837690552500527637822603658699938581184513)


def AggregatePoints(points):
    """curves.AggregatePoints (curves/curve.go:73-121) as one device call."""
    if not points:
        raise ValueError("AggregatePoints of an empty slice (the reference never returns here, curve.go:94-108)")
    c, g = points[0].curve, points[0].group
    o = _lib.out(len(points[0].raw))
    rc = _lib.load().bgls_aggregate_points(c.id, g, _lib.buf(b"".join(p.raw for p in points)), len(points), o)
    if rc != 0:
        raise RuntimeError("bgls_aggregate_points: %d %s" % (rc, _lib.last_error()))
    return Point(c, g, bytes(o))


def ScalePoints(pts, factors):
    """curves.ScalePoints (curves/curve.go:190-214): nil factors -> pts; length mismatch -> nil."""
    if factors is None:
        return pts
    if len(pts) != len(factors):
        return None
    if not pts:
        return []
    c, g = pts[0].curve, pts[0].group
    signs = bytes(2 if f is None else (1 if f < 0 else 0) for f in factors)
    mags = b"".join((0 if f is None else _reduce_scalar(c, abs(f))).to_bytes(32, "big") for f in factors)
    s = len(pts[0].raw)
    o = _lib.out(len(pts) * s)
    rc = _lib.load().bgls_scale_points(c.id, g, _lib.buf(b"".join(p.raw for p in pts)), _lib.buf(mags), _lib.buf(signs), len(pts), o)
    if rc != 0:
        raise RuntimeError("bgls_scale_points: %d %s" % (rc, _lib.last_error()))
    raw = bytes(o)
    return [Point(c, g, raw[i * s:(i + 1) * s]) for i in range(len(pts))]
