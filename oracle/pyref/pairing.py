"""Optimal-ate pairing, tower / projective-twist restatement (CPU oracle -- TEST INFRASTRUCTURE ONLY).

This is the step-by-step algorithm the C oracle (oracle/c) and the HIP kernels follow; it is
pinned against pairing_naive.py (independent flat-Fp12 textbook definition) by
tests/test_oracle_pairing.py.  PARITY UNPINNED for GT *bytes* vs the reference's upstream
libraries (no GT vector exists in the reference); the `bool` of every Verify* is
pinned by bilinearity + non-degeneracy, which the same test checks.

Reference call sites restated: Pair (curves/altbn128.go:130-141, curves/bls12_381.go:228-236),
PairingProduct -> concurrentPairingProduct (curves/curve.go:125-170), GT Add = Fp12 multiply
(curves/altbn128.go:264-271, curves/bls12_381.go:160-168), GetGTIdentity/Equals
(curves/altbn128.go:283-288,478; curves/bls12_381.go:175-180,341).

Miller loop: homogeneous projective (X,Y,Z) on the twist E': y^2 = x^3 + b'.
 doubling:  A=XY/2 B=Y^2 C=Z^2 E=3b'C F=3E G=(B+F)/2 H=(Y+Z)^2-(B+C) I=E-B J=X^2
            X3=A(B-F) Y3=G^2-3E^2 Z3=BH ; line coefficients (-H, 3J, I)
 mixed add: th=Y-yQ Z, la=X-xQ Z, C=th^2 D=la^2 E=la D F=Z C G=X D Hh=E+F-2G
            X3=la Hh, Y3=th(G-Hh)-E Y, Z3=Z E ; line coefficients (la, -th, th xQ - la yQ)
 D-type (alt-bn128): line = c0*yP + c1*xP*w + c2*w^3 ; M-type (BLS12-381): c2 + c1*xP*w^2 + c0*yP*w^3
"""
from .tower import Tower
from .groups import Groups


def naf(k):
    out = []
    while k:
        if k & 1:
            d = 2 - (k % 4)
            k -= d
        else:
            d = 0
        out.append(d)
        k >>= 1
    return out[::-1]          # most-significant first


class Pairing:
    def __init__(self, curve):
        self.c = curve
        self.T = Tower(curve)
        self.G = Groups(curve)
        self.p = curve.p
        self.b2 = self.G.b2
        self.half = pow(2, curve.p - 2, curve.p)
        self.digits = naf(curve.loop)
        assert self.digits[0] == 1

    # ---- Miller loop building blocks ----
    def dbl_step(self, R):
        T = self.T
        X, Y, Z = R
        A = T.f2_muls(T.f2_mul(X, Y), self.half)
        B = T.f2_sqr(Y)
        C = T.f2_sqr(Z)
        E = T.f2_mul(self.b2, T.f2_muls(C, 3))
        F = T.f2_muls(E, 3)
        Gv = T.f2_muls(T.f2_add(B, F), self.half)
        H = T.f2_sub(T.f2_sqr(T.f2_add(Y, Z)), T.f2_add(B, C))
        I = T.f2_sub(E, B)
        J = T.f2_sqr(X)
        X3 = T.f2_mul(A, T.f2_sub(B, F))
        Y3 = T.f2_sub(T.f2_sqr(Gv), T.f2_muls(T.f2_sqr(E), 3))
        Z3 = T.f2_mul(B, H)
        return (X3, Y3, Z3), (T.f2_neg(H), T.f2_muls(J, 3), I)

    def add_step(self, R, Q):
        T = self.T
        X, Y, Z = R
        xq, yq = Q
        th = T.f2_sub(Y, T.f2_mul(yq, Z))
        la = T.f2_sub(X, T.f2_mul(xq, Z))
        C = T.f2_sqr(th)
        D = T.f2_sqr(la)
        E = T.f2_mul(la, D)
        F = T.f2_mul(Z, C)
        Gv = T.f2_mul(X, D)
        Hh = T.f2_sub(T.f2_add(E, F), T.f2_add(Gv, Gv))
        X3 = T.f2_mul(la, Hh)
        Y3 = T.f2_sub(T.f2_mul(th, T.f2_sub(Gv, Hh)), T.f2_mul(E, Y))
        Z3 = T.f2_mul(Z, E)
        j = T.f2_sub(T.f2_mul(th, xq), T.f2_mul(la, yq))
        return (X3, Y3, Z3), (la, T.f2_neg(th), j)

    def line_to_sparse(self, coeffs, P):
        T = self.T
        c0, c1, c2 = coeffs
        xP, yP = P
        if self.c.twist == "D":
            return {0: T.f2_muls(c0, yP), 1: T.f2_muls(c1, xP), 3: c2}
        return {0: c2, 2: T.f2_muls(c1, xP), 3: T.f2_muls(c0, yP)}

    def miller(self, P, Q):
        """Miller value (pre final exponentiation) for affine P in G1, Q in G2; 1 if either is infinity."""
        T = self.T
        f = T.F12_ONE
        if P is None or Q is None:
            return f
        R = (Q[0], Q[1], (1, 0))
        nQ = self.G.g2_neg(Q)
        for d in self.digits[1:]:
            R, co = self.dbl_step(R)
            f = T.f12_mul_sparse(T.f12_sqr(f), self.line_to_sparse(co, P))
            if d:
                R, co = self.add_step(R, Q if d > 0 else nQ)
                f = T.f12_mul_sparse(f, self.line_to_sparse(co, P))
        if self.c.name == "altbn128":
            g1, g2 = T.gamma[1], T.gamma[2]
            Q1 = (T.f2_mul(T.f2_conj(Q[0]), g1[2]), T.f2_mul(T.f2_conj(Q[1]), g1[3]))
            nQ2 = (T.f2_mul(Q[0], g2[2]), T.f2_neg(T.f2_mul(Q[1], g2[3])))
            R, co = self.add_step(R, Q1)
            f = T.f12_mul_sparse(f, self.line_to_sparse(co, P))
            R, co = self.add_step(R, nQ2)
            f = T.f12_mul_sparse(f, self.line_to_sparse(co, P))
        else:
            f = T.f12_conj(f)          # x < 0
        return f

    # ---- final exponentiation: exponent exactly (p^12-1)/r ----
    def _exp_abs(self, a, e):
        return self.T.f12_pow(a, e)

    def final_exp(self, f):
        T = self.T
        # easy part: (p^6-1)(p^2+1)
        f = T.f12_mul(T.f12_conj(f), T.f12_inv(f))
        f = T.f12_mul(T.f12_frob(f, 2), f)
        if self.c.name == "altbn128":
            return self._hard_bn(f)
        return self._hard_bls(f)

    def _hard_bn(self, f):
        """f^((p^4-p^2+1)/r) = f^(l0 + l1 p + l2 p^2 + p^3) via the y0..y6 vectorial chain."""
        T = self.T
        u = self.c.u
        mul, sqr, conj, fr = T.f12_mul, T.f12_sqr, T.f12_conj, T.f12_frob
        ft1 = self._exp_abs(f, u)
        ft2 = self._exp_abs(ft1, u)
        ft3 = self._exp_abs(ft2, u)
        y0 = mul(mul(fr(f, 1), fr(f, 2)), fr(f, 3))
        y1 = conj(f)
        y2 = fr(ft2, 2)
        y3 = conj(fr(ft1, 1))
        y4 = conj(mul(ft1, fr(ft2, 1)))
        y5 = conj(ft2)
        y6 = conj(mul(ft3, fr(ft3, 1)))
        t0 = mul(mul(sqr(y6), y4), y5)
        t1 = mul(mul(y3, y5), t0)
        t0 = mul(t0, y2)
        t1 = sqr(mul(sqr(t1), t0))
        t0 = mul(t1, y1)
        t1 = mul(t1, y0)
        t0 = sqr(t0)
        return mul(t1, t0)

    def _hard_bls(self, f):
        """(p^4-p^2+1)/r = c*(x+p)*(x^2+p^2-1) + 1 with c = (x-1)^2/3 (exact integer)."""
        T = self.T
        mul, conj, fr = T.f12_mul, T.f12_conj, T.f12_frob
        ax = -self.c.x

        def exp_x(a):                     # a^x, x negative, a unitary
            return conj(self._exp_abs(a, ax))

        c = (self.c.x - 1) ** 2 // 3
        a = self._exp_abs(f, c)
        b = mul(exp_x(a), fr(a, 1))
        d = mul(mul(exp_x(exp_x(b)), fr(b, 2)), conj(b))
        return mul(d, f)

    def pair(self, P, Q):
        return self.final_exp(self.miller(P, Q))

    def pairing_product(self, Ps, Qs):
        """prod_i e(P_i, Q_i) with ONE shared final exponentiation (identical GT value to the
        reference's per-pair final exponentiations, curves/curve.go:132-170)."""
        T = self.T
        f = T.F12_ONE
        for P, Q in zip(Ps, Qs):
            f = T.f12_mul(f, self.miller(P, Q))
        return self.final_exp(f)

    # ---- GT wire format (UNPINNED vs upstream; layout modelled on bn256 GT.Marshal:
    #      coefficients from the highest tower position down, imaginary part first) ----
    def gt_bytes(self, a):
        n = self.c.fp_bytes
        g, h = a
        out = []
        for six in (h, g):
            for k in (2, 1, 0):
                out.append(six[k][1].to_bytes(n, "big"))
                out.append(six[k][0].to_bytes(n, "big"))
        return b"".join(out)

    def gt_from_bytes(self, b):
        n = self.c.fp_bytes
        v = [int.from_bytes(b[i * n:(i + 1) * n], "big") for i in range(12)]
        sixes = []
        for s in range(2):
            o = s * 6
            sixes.append(((v[o + 5], v[o + 4]), (v[o + 3], v[o + 2]), (v[o + 1], v[o + 0])))
        return (sixes[1], sixes[0])
