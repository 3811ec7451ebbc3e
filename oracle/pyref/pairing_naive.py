"""Textbook optimal-ate pairing over a FLAT polynomial Fp12 (CPU oracle -- TEST INFRASTRUCTURE ONLY).

Purpose: an independent, obviously-correct (and slow) definition of the reduced optimal-ate
pairing value that `pairing.py` (the tower / projective-twist restatement that the C oracle and
the HIP kernels mirror) is pinned against.  Nothing here shares code with tower.py.

The reference's Pair (curves/altbn128.go:130-141 -> bn256.Pair, curves/bls12_381.go:228-236 ->
bls12 GT.Pair) lives in absent third-party modules; this restates the published algorithm:
  BN   : f_{6u+2,Q}(P) * l_{[6u+2]Q,pi(Q)}(P) * l_{[6u+2]Q+pi(Q),-pi^2(Q)}(P), then ^((p^12-1)/r)
  BLS12: f_{|x|,Q}(P), inverted because x<0, then ^((p^12-1)/r)
Fp12 = Fp[w]/(w^12 - 2*re(xi)*w^6 + |xi|^2)  (w^6 = xi = a + i  =>  i = w^6 - a).
"""


class PolyField:
    def __init__(self, curve):
        self.p = curve.p
        a, bcoef = curve.xi
        assert bcoef == 1
        self.a = a
        # w^12 = 2a w^6 - (a^2+1)
        self.m6 = 2 * a
        self.m0 = -(a * a + 1)

    def one(self):
        return [1] + [0] * 11

    def from_fp(self, x):
        return [x % self.p] + [0] * 11

    def from_fp2_wk(self, c, k):
        """(c0 + c1*i) * w^k, k < 6."""
        out = [0] * 12
        out[k] = (c[0] - self.a * c[1]) % self.p
        out[k + 6] = c[1] % self.p
        return out

    def add(self, x, y):
        return [(s + t) % self.p for s, t in zip(x, y)]

    def sub(self, x, y):
        return [(s - t) % self.p for s, t in zip(x, y)]

    def mul(self, x, y):
        p = self.p
        t = [0] * 23
        for i, xi in enumerate(x):
            if xi:
                for j, yj in enumerate(y):
                    t[i + j] += xi * yj
        for k in range(22, 11, -1):
            c = t[k]
            if c:
                t[k - 6] += c * self.m6
                t[k - 12] += c * self.m0
        return [v % p for v in t[:12]]

    def inv(self, x):
        """Solve M x^-1 = 1 by Gauss-Jordan on the multiplication matrix."""
        p = self.p
        cols = []
        basis = self.one()
        cur = list(x)
        wk = [0] * 12
        wk[1] = 1
        for _ in range(12):
            cols.append(cur)
            cur = self.mul(cur, wk)
        # matrix A[r][c] = cols[c][r]; augmented with e0
        A = [[cols[c][r] for c in range(12)] + [basis[r]] for r in range(12)]
        for c in range(12):
            piv = next(r for r in range(c, 12) if A[r][c] % p)
            A[c], A[piv] = A[piv], A[c]
            iv = pow(A[c][c], p - 2, p)
            A[c] = [v * iv % p for v in A[c]]
            for r in range(12):
                if r != c and A[r][c]:
                    f = A[r][c]
                    A[r] = [(v - f * w) % p for v, w in zip(A[r], A[c])]
        return [A[r][12] for r in range(12)]

    def pow(self, x, e):
        r = self.one()
        for bit in bin(e)[2:]:
            r = self.mul(r, r)
            if bit == "1":
                r = self.mul(r, x)
        return r


def _line(F, P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    elif y1 == y2:
        three_x2 = F.mul(F.from_fp(3), F.mul(x1, x1))
        m = F.mul(three_x2, F.inv(F.add(y1, y1)))
    else:
        return F.sub(xt, x1)
    return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))


def _add(F, P1, P2):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    x1, y1 = P1
    x2, y2 = P2
    if x1 == x2:
        if y1 != y2:
            return None
        m = F.mul(F.mul(F.from_fp(3), F.mul(x1, x1)), F.inv(F.add(y1, y1)))
    else:
        m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(m, m), x1), x2)
    y3 = F.sub(F.mul(m, F.sub(x1, x3)), y1)
    return (x3, y3)


def untwist(curve, F, Q):
    (x, y) = Q
    if curve.twist == "D":
        return (F.from_fp2_wk(x, 2), F.from_fp2_wk(y, 3))
    w2i = F.inv(F.from_fp2_wk((1, 0), 2))
    w3i = F.inv(F.from_fp2_wk((1, 0), 3))
    return (F.mul(F.from_fp2_wk(x, 0), w2i), F.mul(F.from_fp2_wk(y, 0), w3i))


def pairing_naive(curve, P, Q):
    """P = (x, y) ints on E(Fp); Q = ((x0,x1),(y0,y1)) on the twist; returns flat Fp12 list."""
    F = PolyField(curve)
    if P is None or Q is None:
        return F.one()
    p = curve.p
    Q12 = untwist(curve, F, Q)
    P12 = (F.from_fp(P[0]), F.from_fp(P[1]))
    R = Q12
    f = F.one()
    for bit in bin(curve.loop)[3:]:
        f = F.mul(F.mul(f, f), _line(F, R, R, P12))
        R = _add(F, R, R)
        if bit == "1":
            f = F.mul(f, _line(F, R, Q12, P12))
            R = _add(F, R, Q12)
    if curve.name == "altbn128":
        Q1 = (F.pow(Q12[0], p), F.pow(Q12[1], p))
        nQ2 = (F.pow(Q1[0], p), F.sub(F.from_fp(0), F.pow(Q1[1], p)))
        f = F.mul(f, _line(F, R, Q1, P12))
        R = _add(F, R, Q1)
        f = F.mul(f, _line(F, R, nQ2, P12))
    e = F.pow(f, (p**12 - 1) // curve.r)
    if curve.name == "bls12":
        e = F.inv(e)
    return e


def tower_to_flat(curve, T, a):
    """Convert a tower.py Fp12 element to the flat representation."""
    F = PolyField(curve)
    out = [0] * 12
    for k, c in enumerate(T.f12_to_w(a)):
        out = F.add(out, F.from_fp2_wk(c, k))
    return out
