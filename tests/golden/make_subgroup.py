#!/usr/bin/env python3
"""Generates tests/golden/subgroup_{altbn128,bls12}.json from the Python oracle (oracle/pyref/subgroup.py): G2 wire-format
points with the verdict of the DEFINITION of subgroup membership (on the twist and [r]Q = infinity).  Run from the repo
root: python tests/golden/make_subgroup.py"""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyref.params import BN254, BLS381
from oracle.pyref.subgroup import Subgroup, factor

for cv in (BN254, BLS381):
    S = Subgroup(cv)
    G = S.G
    rnd = random.Random(0x5B6 + len(cv.name))
    N = S.twist_order()
    fs = factor(N // cv.r)
    rows = []

    def add(Q, note):
        rows.append({"pt": G.g2_bytes(Q).hex(), "on_twist": G.g2_on_curve(Q), "in_subgroup": S.in_subgroup(Q), "note": note})

    add(None, "infinity")
    for k in (1, 2, 3, 0xDEADBEEF, cv.r - 1, rnd.randrange(cv.r)):
        add(G.g2_mul(cv.g2, k), "[k]g2")
    for _ in range(4):
        add(S.random_twist_point(rnd), "random point of E'(Fp2)")
    for q in sorted(set(fs)):
        e = fs.count(q)
        P = S.small_order_point(rnd, N, q, e)
        add(P, "point of order %s" % (q if q < 1 << 40 else "a %d-bit prime" % q.bit_length()))
        add(G.g2_add(G.g2_mul(cv.g2, rnd.randrange(1, cv.r)), P), "subgroup point + point of order %s" % (q if q < 1 << 40 else "a %d-bit prime" % q.bit_length()))
    add(G.g2_mul(S.random_twist_point(rnd), cv.r), "[r] random: in the cofactor part")
    add(G.g2_mul(S.random_twist_point(rnd), N // cv.r), "[cofactor] random: back in G2")
    bad = bytearray(G.g2_bytes(G.g2_mul(cv.g2, 5))); bad[-1] ^= 1
    rows.append({"pt": bytes(bad).hex(), "on_twist": False, "in_subgroup": False, "note": "off the twist"})
    assert any(r["in_subgroup"] for r in rows) and any(r["on_twist"] and not r["in_subgroup"] for r in rows)
    # does the Miller loop's walk over this key degenerate (running point's Z becomes 0: T = +-Q in an addition, a 2-torsion point or
    # infinity in a doubling)?  Walked with the Python oracle's own point steps (oracle/pyref/pairing.py, no field shortcuts); the GPU
    # tier reads the answer from the fixture (the Python oracle stays in the build container).
    from oracle.pyref import pairing
    pr = pairing.Pairing(cv)

    def degenerates(q):
        if q is None:
            return False
        rr, nq = (q[0], q[1], (1, 0)), pr.G.g2_neg(q)
        for d in pr.digits[1:]:
            rr, _ = pr.dbl_step(rr)
            if d:
                rr, _ = pr.add_step(rr, q if d > 0 else nq)
        if cv.name == "altbn128":
            t = pr.T
            g1, g2 = t.gamma[1], t.gamma[2]
            rr, _ = pr.add_step(rr, (t.f2_mul(t.f2_conj(q[0]), g1[2]), t.f2_mul(t.f2_conj(q[1]), g1[3])))
            rr, _ = pr.add_step(rr, (t.f2_mul(q[0], g2[2]), t.f2_neg(t.f2_mul(q[1], g2[3]))))
        return rr[2] == (0, 0)
    for r in rows:
        r["miller_degenerates"] = bool(r["on_twist"] and degenerates(G.g2_from_bytes(bytes.fromhex(r["pt"]))))
    for r in rows:      # the endomorphism criterion agrees with the definition on every row
        if r["on_twist"]:
            assert S.in_subgroup_fast(G.g2_from_bytes(bytes.fromhex(r["pt"]))) == r["in_subgroup"], r["note"]
    # G1: the same question on E(Fp).  alt-bn128 has cofactor 1 (every curve point is a member); BLS12-381 has cofactor
    # (x-1)^2/3 and the reference checks G1 points like G2 points (curves/bls12_381.go:196-264)
    g1rows = []
    p = cv.p

    def g1_random_curve_point():
        while True:
            x = rnd.randrange(p)
            y2 = (x * x * x + cv.b) % p
            y = pow(y2, (p + 1) // 4, p)
            if y * y % p == y2:
                return (x, y if rnd.random() < 0.5 else p - y)

    def add1(P, note):
        on = G.g1_on_curve(P)
        g1rows.append({"pt": G.g1_bytes(P).hex(), "on_curve": on, "in_subgroup": bool(on and G.g1_mul(P, cv.r) is None), "note": note})

    add1(None, "infinity")
    for k in (1, 2, 0xC0FFEE, cv.r - 1, rnd.randrange(cv.r)):
        add1(G.g1_mul(cv.g1, k), "[k]g1")
    for _ in range(3):
        add1(g1_random_curve_point(), "random point of E(Fp)")
    for _ in range(3):
        T = G.g1_mul(g1_random_curve_point(), cv.r)
        add1(T, "[r] random: in the cofactor part")
        add1(G.g1_add(G.g1_mul(cv.g1, rnd.randrange(1, cv.r)), T), "subgroup point + cofactor-part point")
    bad = bytearray(G.g1_bytes(G.g1_mul(cv.g1, 7))); bad[-1] ^= 1
    g1rows.append({"pt": bytes(bad).hex(), "on_curve": False, "in_subgroup": False, "note": "off the curve"})
    if cv.name == "bls12":
        assert any(r["on_curve"] and not r["in_subgroup"] for r in g1rows)
    else:
        assert all(r["in_subgroup"] == r["on_curve"] for r in g1rows)
    json.dump({"curve": cv.name, "twist_order_bits": N.bit_length(), "cofactor_factors": [str(q) for q in fs], "points": rows, "g1_points": g1rows},
              open(os.path.join(ROOT, "tests", "golden", "subgroup_%s.json" % cv.name), "w"), indent=1)
    print(cv.name, len(rows), "points;", sum(r["in_subgroup"] for r in rows), "in the subgroup")
