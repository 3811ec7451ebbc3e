"""Fp2 / Fp6 / Fp12 tower over Python big ints (CPU oracle -- TEST INFRASTRUCTURE ONLY).

Tower (both curves): Fp2 = Fp[i]/(i^2+1); Fp6 = Fp2[v]/(v^3 - xi); Fp12 = Fp6[w]/(w^2 - v).
The reference has no tower of its own (it delegates to bn256/cloudflare and dis2/bls12,
curves/altbn128.go:11, curves/bls12_381.go:11); the only Fp2 code it owns is
curves/complexNum.go:12-93, whose convention (i^2 = -1) this follows.

Element shapes: Fp2 = (c0, c1); Fp6 = (a0, a1, a2) of Fp2; Fp12 = (g, h) of Fp6 = g + h*w.
"w-basis" view used by the sparse line multiplications: Fp12 = sum_{k<6} e_k w^k with
e_k in Fp2 and w^6 = xi; (g, h) <-> e = (g0, h0, g1, h1, g2, h2).
"""


class Tower:
    def __init__(self, curve):
        self.c = curve
        self.p = curve.p
        self.xi = curve.xi
        p = self.p
        # Frobenius constants gamma[j][k] = xi^(k*(p^j-1)/6), j=1..3, k=0..5
        self.gamma = {}
        for j in (1, 2, 3):
            e = (p**j - 1) // 6
            g1 = self.f2_pow(self.xi, e)
            row = [(1, 0)]
            for _ in range(5):
                row.append(self.f2_mul(row[-1], g1))
            self.gamma[j] = row

    # ---------------- Fp2 ----------------
    def f2_add(self, a, b):
        p = self.p
        return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)

    def f2_sub(self, a, b):
        p = self.p
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)

    def f2_neg(self, a):
        p = self.p
        return ((-a[0]) % p, (-a[1]) % p)

    def f2_mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def f2_sqr(self, a):
        return self.f2_mul(a, a)

    def f2_muls(self, a, s):          # Fp2 * Fp scalar
        p = self.p
        return (a[0] * s % p, a[1] * s % p)

    def f2_conj(self, a):
        return (a[0], (-a[1]) % self.p)

    def f2_inv(self, a):
        p = self.p
        n = pow(a[0] * a[0] + a[1] * a[1], p - 2, p)
        return (a[0] * n % p, (-a[1]) * n % p)

    def f2_mulxi(self, a):
        return self.f2_mul(a, self.xi)

    def f2_pow(self, a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = self.f2_mul(r, a)
            a = self.f2_mul(a, a)
            e >>= 1
        return r

    def f2_is_zero(self, a):
        return a[0] % self.p == 0 and a[1] % self.p == 0

    def f2_sqrt(self, a):
        """Square root in Fp2 (p = 3 mod 4), or None. Used by the oracle only for sampling
        G2 points; follows the complex method of curves/hash.go:196-223 in spirit."""
        p = self.p
        if self.f2_is_zero(a):
            return (0, 0)
        a1 = self.f2_pow(a, (p - 3) // 4)
        alpha = self.f2_mul(a1, self.f2_mul(a1, a))
        x0 = self.f2_mul(a1, a)
        if alpha == ((p - 1) % p, 0):
            r = self.f2_mul((0, 1), x0)
        else:
            bb = self.f2_pow(self.f2_add((1, 0), alpha), (p - 1) // 2)
            r = self.f2_mul(bb, x0)
        return r if self.f2_sqr(r) == (a[0] % p, a[1] % p) else None

    # ---------------- Fp6 ----------------
    F6_ZERO = ((0, 0), (0, 0), (0, 0))
    F6_ONE = ((1, 0), (0, 0), (0, 0))

    def f6_add(self, a, b):
        return tuple(self.f2_add(x, y) for x, y in zip(a, b))

    def f6_sub(self, a, b):
        return tuple(self.f2_sub(x, y) for x, y in zip(a, b))

    def f6_neg(self, a):
        return tuple(self.f2_neg(x) for x in a)

    def f6_mul(self, a, b):
        m, ad, xi = self.f2_mul, self.f2_add, self.f2_mulxi
        a0, a1, a2 = a
        b0, b1, b2 = b
        c0 = ad(m(a0, b0), xi(ad(m(a1, b2), m(a2, b1))))
        c1 = ad(ad(m(a0, b1), m(a1, b0)), xi(m(a2, b2)))
        c2 = ad(ad(m(a0, b2), m(a1, b1)), m(a2, b0))
        return (c0, c1, c2)

    def f6_mulv(self, a):             # multiply by v
        return (self.f2_mulxi(a[2]), a[0], a[1])

    def f6_inv(self, a):
        m, sq, sb, ad, xi = self.f2_mul, self.f2_sqr, self.f2_sub, self.f2_add, self.f2_mulxi
        a0, a1, a2 = a
        t0 = sb(sq(a0), xi(m(a1, a2)))
        t1 = sb(xi(sq(a2)), m(a0, a1))
        t2 = sb(sq(a1), m(a0, a2))
        d = ad(m(a0, t0), xi(ad(m(a2, t1), m(a1, t2))))
        di = self.f2_inv(d)
        return (m(t0, di), m(t1, di), m(t2, di))

    # ---------------- Fp12 ----------------
    @property
    def F12_ONE(self):
        return (self.F6_ONE, self.F6_ZERO)

    def f12_mul(self, a, b):
        g0, h0 = a
        g1, h1 = b
        gg = self.f6_mul(g0, g1)
        hh = self.f6_mul(h0, h1)
        c0 = self.f6_add(gg, self.f6_mulv(hh))
        c1 = self.f6_add(self.f6_mul(g0, h1), self.f6_mul(h0, g1))
        return (c0, c1)

    def f12_sqr(self, a):
        return self.f12_mul(a, a)

    def f12_conj(self, a):            # = a^(p^6)
        return (a[0], self.f6_neg(a[1]))

    def f12_inv(self, a):
        g, h = a
        d = self.f6_sub(self.f6_mul(g, g), self.f6_mulv(self.f6_mul(h, h)))
        di = self.f6_inv(d)
        return (self.f6_mul(g, di), self.f6_neg(self.f6_mul(h, di)))

    def f12_pow(self, a, e):
        r = self.F12_ONE
        for bit in bin(e)[2:]:
            r = self.f12_sqr(r)
            if bit == "1":
                r = self.f12_mul(r, a)
        return r

    def f12_to_w(self, a):
        g, h = a
        return [g[0], h[0], g[1], h[1], g[2], h[2]]

    def f12_from_w(self, e):
        return ((e[0], e[2], e[4]), (e[1], e[3], e[5]))

    def f12_frob(self, a, j=1):
        """a^(p^j) for j in 1..3 via e_k -> conj^j(e_k) * gamma_j[k]."""
        e = self.f12_to_w(a)
        out = []
        for k in range(6):
            x = self.f2_conj(e[k]) if (j & 1) else e[k]
            out.append(self.f2_mul(x, self.gamma[j][k]))
        return self.f12_from_w(out)

    def f12_eq(self, a, b):
        return self.f12_to_w(a) == self.f12_to_w(b)

    def f12_is_one(self, a):
        return self.f12_eq(a, self.F12_ONE)

    def f12_mul_sparse(self, a, line):
        """a * (sum_k line[k] w^k) where line is a dict {k: Fp2}; same value as f12_mul."""
        e = [(0, 0)] * 6
        for k, v in line.items():
            e[k] = v
        return self.f12_mul(a, self.f12_from_w(e))
