"""CPU tier: pin the oracle (Python big-int restatement and its C twin) against the reference's
own known-answer vectors and against an independent textbook pairing."""
import random

import pytest

from oracle import coracle
from oracle.pyref import h2c, scheme
from oracle.pyref.hashes import keccak256_legacy
from oracle.pyref.pairing import Pairing
from oracle.pyref.pairing_naive import pairing_naive, tower_to_flat
from oracle.pyref.params import CURVES


def test_keccak_legacy_known_answers():
    # Keccak-256 (0x01 padding) of "" and "abc": published test vectors of the original Keccak submission
    assert keccak256_legacy(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak256_legacy(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_reference_h2c_kats_pyref(kat):
    """curves/curve_test.go:210-244 (TestG1HashVectors), altbn128_test.go:13-22, bls12_test.go:57-67."""
    for name in ("altbn128", "bls12"):
        c = CURVES[name]
        G = Pairing(c).G
        assert len(kat[name]) == 11
        for row in kat[name]:
            assert G.g1_bytes(h2c.hash_to_g1(c, bytes.fromhex(row["msg"]))).hex() == row["point"]


def test_reference_h2c_kats_coracle(kat):
    for cid, name in ((0, "altbn128"), (1, "bls12")):
        for row in kat[name]:
            assert coracle.hash_to_g1(cid, bytes.fromhex(row["msg"])).hex() == row["point"]


def test_altbn_g2_generator_matches_reference(kat):
    """curves/altbn128_test.go:26-38: G2 generator coordinates in [xi, xr, yi, yr] order."""
    c = CURVES["altbn128"]
    assert Pairing(c).G.g2_bytes(c.g2).hex() == kat["altbn128_g2_generator"]


def test_bls_sw_degenerate():
    """curves/bls12_test.go:27-54 (TestG1SwEncodeDegenerate)."""
    c = CURVES["bls12"]
    G = Pairing(c).G
    assert h2c.bls_fouque_tibouchi(b"") is None
    s5 = h2c.calc_quad_res(c.p - 5, c.p)
    for t in (s5, c.p - s5):
        pt = h2c.bls_fouque_tibouchi(t.to_bytes(48, "big"))
        assert h2c.parity(pt[1], c.p) == h2c.parity(t, c.p)
        assert pt[0] == c.g1[0]
    assert h2c.bls_fouque_tibouchi(s5.to_bytes(48, "big")) == G.g1_neg(c.g1)
    assert h2c.bls_fouque_tibouchi((c.p - s5).to_bytes(48, "big")) == c.g1


@pytest.mark.parametrize("name", ["altbn128", "bls12"])
def test_pairing_matches_textbook_definition(name):
    c = CURVES[name]
    PR = Pairing(c)
    rnd = random.Random(11)
    a, b = rnd.randrange(1, c.r), rnd.randrange(1, c.r)
    P, Q = PR.G.g1_mul(c.g1, a), PR.G.g2_mul(c.g2, b)
    e = PR.pair(P, Q)
    assert tower_to_flat(c, PR.T, e) == pairing_naive(c, P, Q)
    # bilinear, non-degenerate, order r: pins the bool of every Verify*
    e0 = PR.pair(c.g1, c.g2)
    assert PR.T.f12_eq(e, PR.T.f12_pow(e0, a * b % c.r))
    assert not PR.T.f12_is_one(e0) and PR.T.f12_is_one(PR.T.f12_pow(e0, c.r))
    m = PR.miller(P, Q)
    assert PR.T.f12_eq(PR.final_exp(m), PR.T.f12_pow(m, (c.p**12 - 1) // c.r))


def test_golden_vectors_pyref(curve):
    """The committed fixtures are what the oracle produces today."""
    c = CURVES[curve["name"]]
    PR = Pairing(c)
    G = PR.G
    v = curve["vec"]
    for row in v["pairings"]:
        P, Q = G.g1_from_bytes(bytes.fromhex(row["g1"])), G.g2_from_bytes(bytes.fromhex(row["g2"]))
        assert PR.gt_bytes(PR.miller(P, Q)).hex() == row["miller"]
        assert PR.gt_bytes(PR.pair(P, Q)).hex() == row["gt"]
    for row in v["h2c"]:
        assert G.g1_bytes(h2c.hash_to_g1(c, bytes.fromhex(row["msg"]))).hex() == row["point"]


def test_golden_vectors_coracle(curve):
    cid, v = curve["id"], curve["vec"]
    for row in v["pairings"]:
        m = coracle.miller(cid, bytes.fromhex(row["g1"]), bytes.fromhex(row["g2"]))
        assert m.hex() == row["miller"]
        assert coracle.final_exp(cid, m).hex() == row["gt"]
    pp = v["pairing_product"]
    g1s, g2s = b"".join(map(bytes.fromhex, pp["g1s"])), b"".join(map(bytes.fromhex, pp["g2s"]))
    for threads, faithful in ((1, 0), (3, 0), (2, 1)):
        assert coracle.pairing_product(cid, g1s, g2s, len(pp["g1s"]), threads, faithful).hex() == pp["gt"]
    for row in v["h2c"]:
        assert coracle.hash_to_g1(cid, bytes.fromhex(row["msg"])).hex() == row["point"]
    for grp, key in ((1, "sum_g1"), (2, "sum_g2")):
        pts = b"".join(map(bytes.fromhex, v[key]["pts"]))
        assert coracle.aggregate_points(cid, grp, pts, len(v[key]["pts"])).hex() == v[key]["sum"]
    for grp, key in ((1, "scale_g1"), (2, "scale_g2")):
        for row in v[key]:
            k = int(row["k"])
            if abs(k) < 1 << 256:
                assert coracle.scale_point(cid, grp, bytes.fromhex(row["pt"]), k).hex() == row["out"]
    for case in v["aggregate_cases"]:
        got = coracle.verify_aggregate(cid, bytes.fromhex(case["sig"]), b"".join(map(bytes.fromhex, case["keys"])),
                                       [bytes.fromhex(m) for m in case["msgs"]], case["allow_dups"], threads=2) \
            if len(case["keys"]) == len(case["msgs"]) else 0
        assert bool(got == 1) == case["expect"], case["name"]
    for case in v["multi_cases"]:
        got = coracle.verify_multi(cid, bytes.fromhex(case["sig"]), b"".join(map(bytes.fromhex, case["keys"])), len(case["keys"]),
                                   bytes.fromhex(case["msg"]))
        assert bool(got == 1) == case["expect"], case["name"]


def test_scheme_accept_reject_pyref():
    """bgls/bgls_test.go:40-77 on the Python oracle (small n)."""
    c = CURVES["altbn128"]
    G = Pairing(c).G
    rnd = random.Random(5)
    sks = [rnd.randrange(1, c.r) for _ in range(3)]
    msgs = [rnd.randbytes(32) for _ in range(3)]
    keys = [scheme.load_public_key(c, s) for s in sks]
    agg = G.g1_sum([scheme.sign(c, s, m) for s, m in zip(sks, msgs)])
    assert scheme.verify_aggregate_signature(c, agg, keys, msgs)
    assert not scheme.verify_aggregate_signature(c, agg, keys[:2], msgs)
    assert not scheme.verify_aggregate_signature(c, agg, keys, [msgs[1], msgs[0], msgs[2]])
    assert not scheme.verify_aggregate_signature(c, agg, keys, [msgs[0], msgs[0], msgs[2]])


def test_subgroup_fixture_and_criterion():
    """G2 subgroup membership: the C oracle's definition ([r]Q = infinity) on the committed fixture, and the proof that the
    endomorphism criterion used by the HIP kernels accepts exactly G2 on both curves (every prime-order component of the
    twist's cofactor part is rejected; oracle/pyref/subgroup.py)."""
    from oracle import coracle
    from oracle.pyref.params import BN254, BLS381
    from oracle.pyref.subgroup import criterion_is_exact
    from tests.conftest import load_golden
    for cid, cv in ((0, BN254), (1, BLS381)):
        fx = load_golden("subgroup_%s.json" % cv.name)
        for r in fx["points"]:
            assert coracle.g2_in_subgroup(cid, bytes.fromhex(r["pt"])) == (1 if r["in_subgroup"] else 0), r["note"]
        exact, report = criterion_is_exact(cv)
        assert exact, report
        assert sorted(str(q) for q in fx["cofactor_factors"]) == sorted(fx["cofactor_factors"]) and len(report) == len(set(fx["cofactor_factors"]))


def test_g1_subgroup_fixture(curve):
    """G1 rows of tests/golden/make_subgroup.py (definition: on the curve and [r]P = infinity) against the C oracle and the
    Python oracle: BLS12-381 has cofactor-order points on E(Fp) that must be refused (curves/bls12_381.go:196-264 Check());
    on alt-bn128 every curve point is a member."""
    from oracle.pyref.groups import Groups
    from tests.conftest import load_golden
    cid = curve["id"]
    c = CURVES[curve["name"]]
    G = Groups(c)
    rows = load_golden("subgroup_%s.json" % curve["name"])["g1_points"]
    assert any(r["in_subgroup"] for r in rows) and any(not r["on_curve"] for r in rows)
    if cid == 1:
        assert sum(1 for r in rows if r["on_curve"] and not r["in_subgroup"]) >= 6
    for r in rows:
        pt = bytes.fromhex(r["pt"])
        assert coracle.g1_in_subgroup(cid, pt) == (1 if r["in_subgroup"] else 0), r["note"]
        P = G.g1_from_bytes(pt)
        on = G.g1_on_curve(P)
        assert on == r["on_curve"] and bool(on and G.g1_mul(P, c.r) is None) == r["in_subgroup"], r["note"]
