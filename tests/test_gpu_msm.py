"""GPU tier: the windowed / bucketed scalar multiplications (k_msm.hip) against the oracle.

  * bgls_weighted_sum_dev = getAggregatePubKey (bgls/blsHAE.go:74-77; curves/curve.go:73-121,190-214): the bucket method
    and the per-point double-and-add form give the oracle's bytes (sum of oracle scalar multiplications) on ragged sizes,
    edge weights (0, 1, 2^128 - 1), repeated and opposite points and the point at infinity; at 2^16 / 2^18 points the
    oracle checks the closed form (sum w_i s_i) g for points s_i g, and both GPU forms agree byte for byte;
  * skewed weights (what multiplicities look like) take the per-point form and still give the same bytes;
  * bgls_scale_generator = LoadPublicKey over a batch (bgls/bgls.go:40-43) from the fixed-base table: edge scalars and
    random ones against the oracle's double-and-add."""
import ctypes
import random

import pytest

from oracle import coracle

pytestmark = pytest.mark.gpu

ORDER = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
         1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}
FP = {0: 32, 1: 48}
SIZE_MAX = ctypes.c_size_t(-1).value


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def out(n):
    return (ctypes.c_uint8 * max(1, n))()


def gen_points(lib, cid, group, scalars):
    n = len(scalars)
    size = (2 if group == 1 else 4) * FP[cid]
    o = out(n * size)
    assert lib.bgls_scale_generator(cid, group, B(b"".join(s.to_bytes(32, "big") for s in scalars)), n, o) == 0
    return bytes(o)


def wsum(lib, cid, group, pts, weights, msm_min):
    import torch
    n = len(weights)
    size = (2 if group == 1 else 4) * FP[cid]
    dev = torch.device("cuda:0")
    t_p = torch.frombuffer(bytearray(pts if pts else b"\0"), dtype=torch.uint8).to(dev)
    t_w = torch.frombuffer(bytearray(b"".join(w.to_bytes(16, "big") for w in weights) or b"\0"), dtype=torch.uint8).to(dev)
    t_o = torch.zeros(size, dtype=torch.uint8, device=dev)
    assert lib.bgls_set_msm_min(msm_min) == 0
    try:
        rc = lib.bgls_weighted_sum_dev(cid, group, t_p.data_ptr(), t_w.data_ptr(), n, t_o.data_ptr(), None)
    finally:
        lib.bgls_set_msm_min(32)
    assert rc == 0, rc
    return bytes(t_o.cpu().numpy())


def oracle_wsum(cid, group, pts, weights):
    size = (2 if group == 1 else 4) * FP[cid]
    n = len(weights)
    scaled = b"".join(coracle.scale_point(cid, group, pts[i * size:(i + 1) * size], weights[i]) for i in range(n))
    return coracle.aggregate_points(cid, group, scaled, n)


@pytest.mark.parametrize("cid", [0, 1], ids=["altbn128", "bls12"])
@pytest.mark.parametrize("group", [1, 2], ids=["g1", "g2"])
def test_weighted_sum_bucket_method_against_oracle(gpu_lib, cid, group):
    lib = gpu_lib
    rnd = random.Random(0x3517 + 10 * cid + group)
    size = (2 if group == 1 else 4) * FP[cid]
    for n in (1, 2, 31, 33, 100, 257, 700):
        sc = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
        w = [rnd.getrandbits(128) for _ in range(n)]
        if n >= 31:
            sc[3] = sc[2]                                        # the same point twice, with the same digit in every window
            w[3] = w[2]
            sc[5] = ORDER[cid] - sc[4]                           # a point and its opposite in the same buckets
            w[5] = w[4]
            w[6], w[7], w[8] = 0, 1, (1 << 128) - 1
            w[9] = 1 << 127
        pts = bytearray(gen_points(lib, cid, group, sc))
        if n >= 31:
            pts[10 * size:11 * size] = bytes(size)               # the point at infinity (all-zero encoding)
        pts = bytes(pts)
        want = oracle_wsum(cid, group, pts, w)
        assert wsum(lib, cid, group, pts, w, 0) == want, n              # bucket method whatever n
        assert wsum(lib, cid, group, pts, w, SIZE_MAX) == want, n       # one double-and-add per point
    # everything cancels: infinity
    sc = [rnd.randrange(1, ORDER[cid]) for _ in range(20)]
    pts = gen_points(lib, cid, group, sc + [ORDER[cid] - s for s in sc])
    w = [rnd.getrandbits(128) for _ in range(20)]
    assert wsum(lib, cid, group, pts, w + w, 0) == bytes(size)
    # an off-curve point is an encoding error on both paths
    bad = bytearray(gen_points(lib, cid, group, [5] * 40)); bad[size - 1] ^= 1
    import torch
    t_p = torch.frombuffer(bad, dtype=torch.uint8).to("cuda:0")
    t_w = torch.ones(40 * 16, dtype=torch.uint8, device="cuda:0")
    t_o = torch.zeros(size, dtype=torch.uint8, device="cuda:0")
    for m in (0, SIZE_MAX):
        lib.bgls_set_msm_min(m)
        rc = lib.bgls_weighted_sum_dev(cid, group, t_p.data_ptr(), t_w.data_ptr(), 40, t_o.data_ptr(), None)
        lib.bgls_set_msm_min(32)
        assert rc < 0


@pytest.mark.parametrize("cid,group,n", [(0, 2, 1 << 16), (1, 2, 1 << 16), (0, 1, 1 << 16), (0, 2, 1 << 18)],
                         ids=["altbn128-g2-64k", "bls12-g2-64k", "altbn128-g1-64k", "altbn128-g2-256k"])
def test_weighted_sum_large_closed_form(gpu_lib, cid, group, n):
    """P_i = s_i g  =>  sum w_i P_i = (sum w_i s_i mod r) g: one oracle scalar multiplication checks 2^16 / 2^18 terms"""
    lib = gpu_lib
    rnd = random.Random(0xA11CE + cid + n)
    r = ORDER[cid]
    sc = [rnd.randrange(1, r) for _ in range(n)]
    w = [rnd.getrandbits(128) for _ in range(n)]
    pts = gen_points(lib, cid, group, sc)
    g = out((2 if group == 1 else 4) * FP[cid])
    assert lib.bgls_generator(cid, group, g) == 0
    want = coracle.scale_point(cid, group, bytes(g), sum(a * b for a, b in zip(w, sc)) % r)
    assert wsum(lib, cid, group, pts, w, 0) == want
    if n <= 1 << 16:
        assert wsum(lib, cid, group, pts, w, SIZE_MAX) == want


def test_weighted_sum_skewed_weights_fall_back(gpu_lib):
    """small multiplicities put every point into a handful of buckets: the library takes the per-point form; same bytes"""
    lib, cid, group, n = gpu_lib, 0, 2, 5000
    rnd = random.Random(77)
    r = ORDER[cid]
    sc = [rnd.randrange(1, r) for _ in range(n)]
    w = [rnd.randrange(0, 6) for _ in range(n)]
    pts = gen_points(lib, cid, group, sc)
    g = out(4 * FP[cid]); lib.bgls_generator(cid, group, g)
    want = coracle.scale_point(cid, group, bytes(g), sum(a * b for a, b in zip(w, sc)) % r)
    assert wsum(lib, cid, group, pts, w, 0) == want
    w = [(1 << 100) + 12345] * n                                      # one weight for everybody
    want = coracle.scale_point(cid, group, bytes(g), sum(a * b for a, b in zip(w, sc)) % r)
    assert wsum(lib, cid, group, pts, w, 0) == want


@pytest.mark.parametrize("cid", [0, 1], ids=["altbn128", "bls12"])
@pytest.mark.parametrize("group", [1, 2], ids=["g1", "g2"])
def test_fixed_base_generator_multiples(gpu_lib, cid, group):
    lib = gpu_lib
    rnd = random.Random(0xF1BA + 10 * cid + group)
    r = ORDER[cid]
    size = (2 if group == 1 else 4) * FP[cid]
    sc = [0, 1, 2, 255, 256, r - 1, r, r + 1, (1 << 256) - 1, 1 << 255, 0xFF << 248, 0x0100_0000_0000_0001]
    sc += [rnd.getrandbits(256) for _ in range(60)] + [rnd.getrandbits(8 * rnd.randrange(1, 32)) for _ in range(30)]
    got = gen_points(lib, cid, group, sc)
    g = out(size); lib.bgls_generator(cid, group, g)
    for i, s in enumerate(sc):
        assert got[i * size:(i + 1) * size] == coracle.scale_point(cid, group, bytes(g), s % r), hex(s)
    assert got[:size] == bytes(size)                                  # 0 g = infinity
    assert got[6 * size:7 * size] == bytes(size)                      # r g = infinity
