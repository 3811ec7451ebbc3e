"""G2 subgroup membership (CPU oracle -- TEST INFRASTRUCTURE ONLY).

The reference validates G2 inputs when a Point is constructed: alt-bn128 MakeG2Point / UnmarshalG2
(curves/altbn128.go:157-179,329-376) go through upstream bn256's G2.Unmarshal, which rejects twist points outside the
order-r subgroup; BLS12-381 has Check() (curves/bls12_381.go:242-264).  The upstream sources are absent, so the
DEFINITION is restated -- Q is on the twist and [r]Q = infinity (`in_subgroup`) -- next to the endomorphism criterion
the HIP kernels use (`in_subgroup_fast`):
   alt-bn128   [u+1]Q + psi([u]Q) + psi^2([u]Q) = psi^3([2u]Q)
   BLS12-381   psi(Q) = [x]Q
with psi = twist o Frobenius o untwist, which acts on G2 as multiplication by p.  `criterion_is_exact` proves, for the
actual curve, that both accept the same points: the criterion is a group endomorphism vanishing on G2, so it is exact iff
it kills no point of prime order in the cofactor part of E'(Fp2), which is checked component by component.
"""
import math
import random

from .groups import Groups


class Subgroup:
    def __init__(self, curve):
        self.c = curve
        self.G = Groups(curve)
        T = self.G.T
        p = curve.p
        gx, gy = T.f2_pow(curve.xi, (p - 1) // 3), T.f2_pow(curve.xi, (p - 1) // 2)
        if curve.twist == "M":
            gx, gy = T.f2_inv(gx), T.f2_inv(gy)
        self.gx, self.gy = gx, gy
        self.x = curve.u if curve.name == "altbn128" else curve.x

    def psi(self, Q):
        if Q is None:
            return None
        T = self.G.T
        return (T.f2_mul(T.f2_conj(Q[0]), self.gx), T.f2_mul(T.f2_conj(Q[1]), self.gy))

    def in_subgroup(self, Q):
        """the definition: on the twist and killed by r"""
        return self.G.g2_on_curve(Q) and self.G.g2_mul(Q, self.c.r) is None

    def in_subgroup_fast(self, Q):
        G = self.G
        if Q is None:
            return True
        if not G.g2_on_curve(Q):
            return False
        xQ = G.g2_mul(Q, self.x)
        if self.c.name == "altbn128":
            lhs = G.g2_add(G.g2_add(G.g2_add(xQ, Q), self.psi(xQ)), self.psi(self.psi(xQ)))
            return lhs == self.psi(self.psi(self.psi(G.g2_add(xQ, xQ))))
        return self.psi(Q) == xQ

    # ---- structure of E'(Fp2) ----
    def twist_order(self):
        c, G = self.c, self.G
        p = c.p
        t = p + 1 - c.r * c.cofactor                      # trace over Fp (#E(Fp) = cofactor * r)
        t2 = t * t - 2 * p                                # trace over Fp2
        f = math.isqrt((4 * p * p - t2 * t2) // 3)
        assert 3 * f * f == 4 * p * p - t2 * t2
        for n in (p * p + 1 - (t2 + 3 * f) // 2, p * p + 1 - (t2 - 3 * f) // 2, p * p + 1 + (t2 + 3 * f) // 2, p * p + 1 + (t2 - 3 * f) // 2):
            if n % c.r == 0 and G.g2_mul(c.g2, n) is None and G.g2_mul(self.random_twist_point(random.Random(1)), n) is None:
                return n
        raise AssertionError("twist order not found")

    def random_twist_point(self, rnd):
        T, p = self.G.T, self.c.p
        while True:
            x = (rnd.randrange(p), rnd.randrange(p))
            rhs = T.f2_add(T.f2_mul(T.f2_sqr(x), x), self.G.b2)
            y = T.f2_sqrt(rhs)
            if y is not None and T.f2_sqr(y) == rhs:
                return (x, y)

    def small_order_point(self, rnd, N, q, e=1):
        """a point of exact order q in the q-part of E'(Fp2) (q^e || cofactor)"""
        while True:
            P = self.G.g2_mul(self.random_twist_point(rnd), N // q ** e)
            while P is not None and self.G.g2_mul(P, q) is not None:
                P = self.G.g2_mul(P, q)
            if P is not None:
                return P


def _is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _rho(n, rnd):
    if n % 2 == 0:
        return 2
    while True:
        c, x = rnd.randrange(1, n), rnd.randrange(n)
        y, d = x, 1
        while d == 1:
            x = (x * x + c) % n
            y = (y * y + c) % n
            y = (y * y + c) % n
            d = math.gcd(abs(x - y), n)
        if d != n:
            return d


def factor(n, rnd=None):
    rnd = rnd or random.Random(5)
    out, todo = [], [n]
    while todo:
        m = todo.pop()
        if m == 1:
            continue
        if _is_prime(m):
            out.append(m)
        else:
            d = _rho(m, rnd)
            todo += [d, m // d]
    return sorted(out)


def criterion_is_exact(curve, seed=11):
    """True iff in_subgroup_fast accepts exactly G2: no point of prime order in the cofactor part passes it.  Cyclic
    q-parts need one generator; where the full q-torsion is rational (q^2 | cofactor, non-cyclic) every one of the q + 1
    lines of E'[q] is tried."""
    S = Subgroup(curve)
    G = S.G
    rnd = random.Random(seed)
    N = S.twist_order()
    fs = factor(N // curve.r, rnd)
    report = []
    for q in sorted(set(fs)):
        e = fs.count(q)
        P1 = S.small_order_point(rnd, N, q, e)
        cands = [P1]
        if e > 1:
            # a second, independent point of order q (if there is none within a few tries the q-part is cyclic)
            for _ in range(20):
                P2 = S.small_order_point(rnd, N, q, e)
                if all(G.g2_mul(P1, k) != P2 for k in range(1, q)):
                    cands = [P2] + [G.g2_add(P1, G.g2_mul(P2, k)) for k in range(q)]      # the q + 1 lines of E'[q]
                    break
        for P in cands:
            assert G.g2_mul(P, q) is None and not S.in_subgroup(P)
            if S.in_subgroup_fast(P):
                return False, report
        report.append((q if q < 1 << 40 else "prime of %d bits" % q.bit_length(), e, len(cands)))
    return True, report
