"""GPU tier for k_miller_x60 (bgls_amd/csrc/miller_x.hpp: the Miller loop on carry-free 28-bit limbs, lane-pair point steps),
forced for every batch size with bgls_set_miller_shape(4, mode) -- mode bit 16 selects the 64-pairing block form (a seventh
line in four of a block's ten groups; 1024 resident blocks = 2^16 pairings), clear the 60-pairing form:

  * PairingProduct (curves/curve.go:125-170) against the C oracle's GT bytes at sizes around the kernel's tile boundaries
    (30 pairings per producer wave, 60 per block, 6 per accumulator group), with points at infinity among the inputs;
  * the same batches through the 32-bit fused kernels (shape 5): identical GT bytes, also for thousands of pairings;
  * every role / priority mode gives the same bytes; an off-curve key is reported, not folded."""
import ctypes
import random

import pytest

from oracle import coracle

pytestmark = pytest.mark.gpu

# group orders (curves/altbn128.go:480, curves/bls12_381.go:339), as in tests/test_gpu_configs.py
ORDER = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
         1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def out(n):
    return (ctypes.c_uint8 * max(1, n))()


@pytest.fixture()
def shape(gpu_lib):
    def set_shape(s, arg=8):
        assert gpu_lib.bgls_set_miller_shape(s, arg if s == 4 else 6) == 0
    yield set_shape
    assert gpu_lib.bgls_set_miller_shape(0, 6) == 0


def random_points(curve, rnd, n):
    cid = curve["id"]
    g1 = bytes.fromhex(curve["vec"]["pairings"][3]["g1"])
    g2 = bytes.fromhex(curve["vec"]["pairings"][3]["g2"])
    g1s = [coracle.scale_point(cid, 1, g1, rnd.randrange(1, 1 << 250)) for _ in range(n)]
    g2s = [coracle.scale_point(cid, 2, g2, rnd.randrange(1, 1 << 250)) for _ in range(n)]
    return g1s, g2s


def test_pairing_product_equals_oracle_at_tile_boundaries(gpu_lib, curve, shape):
    cid, n_fp = curve["id"], curve["fp"]
    rnd = random.Random(60 + cid)
    for n in (1, 2, 29, 30, 31, 59, 60, 61, 63, 64, 65, 66, 120, 121, 127, 128, 129, 187, 193):
        g1s, g2s = random_points(curve, rnd, n)
        if n >= 30:                                    # points at infinity contribute the factor 1 (curves/altbn128.go:478, curves/bls12_381.go:341)
            g1s[n // 3] = bytes(2 * n_fp)
            g2s[n // 2] = bytes(4 * n_fp)
        if n >= 63:                                    # ... also in the 64-form's seventh-line slots (pairings 60..63 of a block)
            g2s[61] = bytes(4 * n_fp)
        a, b = b"".join(g1s), b"".join(g2s)
        want = coracle.pairing_product(cid, a, b, n, threads=8)
        for form, mode in (("60", 8), ("64", 16 + 8)):
            shape(4, mode)
            o = out(12 * n_fp)
            assert gpu_lib.bgls_pairing_product(cid, B(a), B(b), n, o) == 0
            assert bytes(o) == want, "k_miller_x60, %s pairings per block, n = %d" % (form, n)
        # (the 32-bit kernel k_miller_ab64 -- shape 5 -- left the shipped library in round 6: tests/test_gpu_legacy_paths.py runs it against the
        # same oracle on the LEGACY=1 build)


def test_large_batches_same_bytes_in_every_block_form_and_role_mode(gpu_lib, curve, shape):
    cid, n_fp = curve["id"], curve["fp"]
    rnd = random.Random(77 + cid)
    n = 7000
    g2 = out(4 * n_fp)
    g1 = out(2 * n_fp)
    gpu_lib.bgls_generator(cid, 2, g2)
    gpu_lib.bgls_generator(cid, 1, g1)
    k1 = b"".join(rnd.randrange(1, 1 << 250).to_bytes(32, "big") for _ in range(n))
    k2 = b"".join(rnd.randrange(1, 1 << 250).to_bytes(32, "big") for _ in range(n))
    g1s, g2s = out(n * 2 * n_fp), out(n * 4 * n_fp)
    assert gpu_lib.bgls_scale_points(cid, 1, B(bytes(g1) * n), B(k1), None, n, g1s) == 0
    assert gpu_lib.bgls_scale_points(cid, 2, B(bytes(g2) * n), B(k2), None, n, g2s) == 0
    shape(4, 8)                                        # reference: the default form; the VALUE is pinned by bilinearity below (and, for the 32-bit
    ref = out(12 * n_fp)                               # kernels of the LEGACY=1 build, by tests/test_gpu_legacy_paths.py at this size)
    assert gpu_lib.bgls_pairing_product(cid, g1s, g2s, n, ref) == 0
    for mode in (0, 1, 2, 9, 4, 16, 16 + 8, 16 + 1, 16 + 2 + 4):
        shape(4, mode)
        o = out(12 * n_fp)
        assert gpu_lib.bgls_pairing_product(cid, g1s, g2s, n, o) == 0
        assert bytes(o) == bytes(ref), "mode %d" % mode
    # bilinearity pins the value itself: prod e(a_i g1, b_i g2) = e(g1, g2)^(sum a_i b_i)
    shape(4)
    r = ORDER[cid]
    if r:
        s = sum(int.from_bytes(k1[32 * i:32 * i + 32], "big") * int.from_bytes(k2[32 * i:32 * i + 32], "big") for i in range(n)) % r
        one = out(2 * n_fp)
        assert gpu_lib.bgls_scale_points(cid, 1, g1, B(s.to_bytes(32, "big")), None, 1, one) == 0
        e = out(12 * n_fp)
        assert gpu_lib.bgls_pairing_product(cid, one, g2, 1, e) == 0
        assert bytes(e) == bytes(ref)


def test_off_curve_key_is_reported(gpu_lib, curve, shape):
    cid, n_fp = curve["id"], curve["fp"]
    rnd = random.Random(5)
    n = 200
    g1s, g2s = random_points(curve, rnd, n)
    bad = bytearray(g2s[137])
    bad[-1] ^= 1
    g2s[137] = bytes(bad)
    for mode in (8, 16 + 8):
        shape(4, mode)
        o = out(12 * n_fp)
        assert gpu_lib.bgls_pairing_product(cid, B(b"".join(g1s)), B(b"".join(g2s)), n, o) < 0
    assert gpu_lib.bgls_set_miller_shape(4, 3) < 0 and gpu_lib.bgls_set_miller_shape(4, 32) < 0      # mode words are validated


def test_degenerate_point_step_is_an_encoding_error(gpu_lib, curve, shape):
    """A twist point outside G2 handed to the Miller producers WITHOUT the subgroup check (the reference cannot construct one:
    curves/bls12_381.go:196-264, curves/altbn128.go:157-179).  Where its point steps degenerate (the fixture's point of order
    13 on BLS12-381: T = +-Q after six steps) every producer of the shipped library -- latency form, k_miller_x60 in both block forms --
    reports BGLS_ERR_ENCODING instead of an unspecified verdict; where they do not (no on-curve point of alt-bn128's twist
    degenerates: its smallest cofactor order is 10069) the value is the oracle's Miller formula, byte for byte."""
    from tests.conftest import load_golden
    cid, n_fp = curve["id"], curve["fp"]
    rows = [r for r in load_golden("subgroup_%s.json" % curve["name"])["points"] if r["on_twist"] and not r["in_subgroup"]]
    rnd = random.Random(13 + cid)
    seen_degenerate = 0
    for r in rows:
        bad = bytes.fromhex(r["pt"])
        deg = r["miller_degenerates"]       # walked with the Python oracle's point steps when the fixture was made (tests/golden/make_subgroup.py)
        seen_degenerate += deg
        for n, shapes in ((3, (0,)), (200, (4, 64))):       # <= 128 pairings: k_miller_latx; above: k_miller_x60 (both block forms)
            g1s, g2s = random_points(curve, rnd, n)
            g2s[n // 2] = bad
            a, b = b"".join(g1s), b"".join(g2s)
            for s in shapes:
                if s == 64:
                    shape(4, 16 + 8)
                else:
                    shape(s)
                o = out(12 * n_fp)
                rc = gpu_lib.bgls_pairing_product(cid, B(a), B(b), n, o)
                if deg:
                    assert rc == -2, (r["note"], n, s, rc)              # BGLS_ERR_ENCODING
                else:
                    assert rc == 0, (r["note"], n, s, rc)
                    if n == 3 or s in (4, 64):
                        assert bytes(o) == coracle.pairing_product(cid, a, b, n, threads=8), (r["note"], n, s)
    assert seen_degenerate == (1 if cid == 1 else 0)
    # the verification door: one such key among valid ones
    if cid == 1:
        bad = bytes.fromhex(next(r["pt"] for r in rows if r["note"] == "point of order 13"))
        v = next(c for c in curve["vec"]["aggregate_cases"] if c["expect"] and len(c["keys"]) >= 2)
        keys = [bytes.fromhex(k) for k in v["keys"]]
        msgs = [bytes.fromhex(m) for m in v["msgs"]]
        keys[len(keys) // 2] = bad
        off = (ctypes.c_uint64 * (len(msgs) + 1))()
        acc = 0
        for i, m in enumerate(msgs):
            off[i] = acc
            acc += len(m)
        off[len(msgs)] = acc
        shape(0)
        assert gpu_lib.bgls_verify_aggregate(cid, B(bytes.fromhex(v["sig"])), B(b"".join(keys)), B(b"".join(msgs)), off, len(msgs), 0) == -2


def test_one_round_of_64_pairing_blocks_is_the_default_for_a_lone_2_16_batch(gpu_lib, curve, shape):
    """BASELINE configs 2 / 3: 2^16 pairings with the machine to itself = 1024 blocks of the 64-form (automatic shape).  Same GT
    bytes as the 60-form and as the 32-bit kernels (round 3's choice for this size), and the value itself by bilinearity."""
    cid, n_fp = curve["id"], curve["fp"]
    rnd = random.Random(2 ** 16 + cid)
    n = 1 << 16
    g1, g2 = out(2 * n_fp), out(4 * n_fp)
    gpu_lib.bgls_generator(cid, 1, g1)
    gpu_lib.bgls_generator(cid, 2, g2)
    k1 = b"".join(rnd.randrange(1, 1 << 250).to_bytes(32, "big") for _ in range(n))
    k2 = b"".join(rnd.randrange(1, 1 << 250).to_bytes(32, "big") for _ in range(n))
    g1s, g2s = out(n * 2 * n_fp), out(n * 4 * n_fp)
    assert gpu_lib.bgls_scale_points(cid, 1, B(bytes(g1) * n), B(k1), None, n, g1s) == 0
    assert gpu_lib.bgls_scale_points(cid, 2, B(bytes(g2) * n), B(k2), None, n, g2s) == 0
    got = {}
    for name, args in (("auto", (0,)), ("x60", (4, 8)), ("x64", (4, 16 + 8))):              # (k_miller_ab64, shape 5: LEGACY=1 builds only since round 6)
        shape(*args)
        o = out(12 * n_fp)
        assert gpu_lib.bgls_pairing_product(cid, g1s, g2s, n, o) == 0, name
        got[name] = bytes(o)
    assert got["auto"] == got["x60"] == got["x64"]
    r = ORDER[cid]
    s = sum(int.from_bytes(k1[32 * i:32 * i + 32], "big") * int.from_bytes(k2[32 * i:32 * i + 32], "big") for i in range(n)) % r
    one = out(2 * n_fp)
    assert gpu_lib.bgls_scale_points(cid, 1, g1, B(s.to_bytes(32, "big")), None, 1, one) == 0
    e = out(12 * n_fp)
    shape(0)
    assert gpu_lib.bgls_pairing_product(cid, one, g2, 1, e) == 0
    assert bytes(e) == got["auto"]
