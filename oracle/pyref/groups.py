"""G1 / G2 affine group law over Python ints (CPU oracle -- TEST INFRASTRUCTURE ONLY).

Points are (x, y) or None for infinity.  G1 coordinates are ints, G2 coordinates are Fp2
tuples (re, im).  Mirrors what the reference gets from upstream Add / ScalarMult
(curves/altbn128.go:59-66,107-128,181-188,235-249; curves/bls12_381.go:33-41,65-83,94-102,126-137)
and AggregatePoints (curves/curve.go:73-121).
"""
from .tower import Tower


class Groups:
    def __init__(self, curve):
        self.c = curve
        self.p = curve.p
        self.T = Tower(curve)
        T = self.T
        if curve.twist == "D":
            self.b2 = T.f2_mul((curve.b, 0), T.f2_inv(curve.xi))
        else:
            self.b2 = T.f2_mul((curve.b, 0), curve.xi)

    # ---- G1 ----
    def g1_on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return (y * y - x * x * x - self.c.b) % self.p == 0

    def g1_neg(self, P):
        return None if P is None else (P[0], (-P[1]) % self.p)

    def g1_add(self, P, Q):
        p = self.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            m = 3 * x1 * x1 * pow(2 * y1, p - 2, p) % p
        else:
            m = (y2 - y1) * pow(x2 - x1, p - 2, p) % p
        x3 = (m * m - x1 - x2) % p
        return (x3, (m * (x1 - x3) - y1) % p)

    def g1_mul(self, P, k):
        if k < 0:
            return self.g1_mul(self.g1_neg(P), -k)
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.g1_add(R, R)
            if bit == "1":
                R = self.g1_add(R, P)
        return R

    # ---- G2 (on the twist) ----
    def g2_on_curve(self, Q):
        if Q is None:
            return True
        T = self.T
        x, y = Q
        return T.f2_sub(T.f2_sqr(y), T.f2_add(T.f2_mul(T.f2_sqr(x), x), self.b2)) == (0, 0)

    def g2_neg(self, Q):
        return None if Q is None else (Q[0], self.T.f2_neg(Q[1]))

    def g2_add(self, P, Q):
        T = self.T
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if T.f2_add(y1, y2) == (0, 0):
                return None
            m = T.f2_mul(T.f2_muls(T.f2_sqr(x1), 3), T.f2_inv(T.f2_add(y1, y1)))
        else:
            m = T.f2_mul(T.f2_sub(y2, y1), T.f2_inv(T.f2_sub(x2, x1)))
        x3 = T.f2_sub(T.f2_sub(T.f2_sqr(m), x1), x2)
        return (x3, T.f2_sub(T.f2_mul(m, T.f2_sub(x1, x3)), y1))

    def g2_mul(self, Q, k):
        if k < 0:
            return self.g2_mul(self.g2_neg(Q), -k)
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.g2_add(R, R)
            if bit == "1":
                R = self.g2_add(R, Q)
        return R

    def g1_sum(self, pts):
        R = None
        for P in pts:
            R = self.g1_add(R, P)
        return R

    def g2_sum(self, pts):
        R = None
        for P in pts:
            R = self.g2_add(R, P)
        return R

    # ---- wire formats at the seam (uncompressed) ----
    # alt-bn128: G1 = x||y 32-byte BE, infinity = zeros (curves/altbn128.go:42-57,431-434);
    #            G2 = x_im||x_re||y_im||y_re (curves/altbn128.go:157-179, altbn128_test.go:26-38)
    # BLS12-381: G1 = x||y 48-byte BE (curves/testcases/bls12G1Hash.dat);
    #            G2 = x_c1||x_c0||y_c1||y_c0 (curves/bls12_381.go:147-158,209-226)
    def g1_bytes(self, P):
        n = self.c.fp_bytes
        if P is None:
            return bytes(2 * n)
        return P[0].to_bytes(n, "big") + P[1].to_bytes(n, "big")

    def g1_from_bytes(self, b):
        n = self.c.fp_bytes
        assert len(b) == 2 * n
        if b == bytes(2 * n):
            return None
        return (int.from_bytes(b[:n], "big"), int.from_bytes(b[n:], "big"))

    def g2_bytes(self, Q):
        n = self.c.fp_bytes
        if Q is None:
            return bytes(4 * n)
        (x0, x1), (y0, y1) = Q
        return b"".join(v.to_bytes(n, "big") for v in (x1, x0, y1, y0))

    def g2_from_bytes(self, b):
        n = self.c.fp_bytes
        assert len(b) == 4 * n
        if b == bytes(4 * n):
            return None
        v = [int.from_bytes(b[i * n:(i + 1) * n], "big") for i in range(4)]
        return ((v[1], v[0]), (v[3], v[2]))
