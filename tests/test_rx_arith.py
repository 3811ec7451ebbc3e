"""CPU tier for the carry-free 28-bit-limb arithmetic of the Miller kernel k_miller_x60 (bgls_amd/csrc/rx.hpp,
rx_pair.hpp), compiled for the host by the test harness with every column accumulation checked for 64-bit overflow:

  * consumer dot products on RAW limbs against Python integers -- random operands and the worst-case limb patterns the
    column budget of tools/gen_constants.py is computed for (this is what proves the bounds, random values never get near);
  * the lane-pair point steps, run on two lock-stepped threads, against pairing.hpp's own steps line by line over a whole
    Miller loop (same line coefficients => the kernel's partial products are those of the 32-bit kernels)."""
import ctypes
import random

import pytest

from oracle.pyref.groups import Groups
from oracle.pyref.params import CURVES

W = 28
MASK = (1 << W) - 1


def geom(cid):
    c = CURVES["altbn128" if cid == 0 else "bls12"]
    N = 10 if cid == 0 else 14
    return c, c.p, N, 1 << (W * N)


def limbs_of(x, N):
    out = [(x >> (W * i)) & MASK for i in range(N - 1)]
    out.append(x >> (W * (N - 1)))
    assert out[-1] < (1 << 32)
    return out


def val(l):
    return sum(int(v) << (W * i) for i, v in enumerate(l))


def pack(ops, N):
    """ops: list of (re_limbs, im_limbs) -> ctypes u32 array [t][half][N]"""
    flat = []
    for re, im in ops:
        flat += list(re) + list(im)
    return (ctypes.c_uint32 * len(flat))(*flat)


def run(lib, cid, op, arg, A, Bv, N):
    o = (ctypes.c_uint32 * (2 * N))()
    ovf = lib.ht_rx_raw(cid, op, arg, pack(A, N), pack(Bv, N), o)
    return ovf, list(o[:N]), list(o[N:])


def tight(l):
    return all(v < (1 << W) for v in l[:-1])


@pytest.mark.parametrize("cid", [0, 1])
def test_conversion_round_trip(host_harness, cid):
    c, p, N, Rp = geom(cid)
    n = 32 if cid == 0 else 48
    rnd = random.Random(11)
    for x in [0, 1, p - 1, p // 2] + [rnd.randrange(p) for _ in range(40)]:
        buf = (ctypes.c_uint8 * n).from_buffer_copy(x.to_bytes(n, "big"))
        lim = (ctypes.c_uint32 * N)()
        assert host_harness.ht_rx_conv(cid, 0, buf, lim) == 0
        assert tight(list(lim)) and val(lim) % p == x * Rp % p and val(lim) < 2 * p
        back = (ctypes.c_uint8 * n)()
        assert host_harness.ht_rx_conv(cid, 1, back, lim) == 0
        assert int.from_bytes(bytes(back), "big") == x
    # from_ux accepts any tight value below 4 p
    for v in [p, 2 * p - 1, 3 * p + 12345, 4 * p - 1]:
        lim = (ctypes.c_uint32 * N)(*limbs_of(v, N))
        back = (ctypes.c_uint8 * n)()
        assert host_harness.ht_rx_conv(cid, 1, back, lim) == 0
        assert int.from_bytes(bytes(back), "big") == v * pow(Rp, -1, p) % p


def operands(rnd, cid, kind):
    """one Fp2 operand as raw limbs: 'rand' = a random residue in the lazy range, 'max' = every limb at its bound"""
    c, p, N, Rp = geom(cid)
    top_p = p >> (W * (N - 1))
    if kind == "max":
        l = [MASK] * (N - 1) + [32 * (top_p + 1) - 1]
        return (l, list(l))
    return (limbs_of(rnd.randrange(31 * p), N), limbs_of(rnd.randrange(31 * p), N))


@pytest.mark.parametrize("cid", [0, 1])
def test_consumer_dot_products_and_column_budget(host_harness, cid):
    c, p, N, Rp = geom(cid)
    Rinv = pow(Rp, -1, p)
    rnd = random.Random(5 + cid)
    for kind in ["rand"] * 12 + ["max"]:
        A = [operands(rnd, cid, kind) for _ in range(3)]
        Bv = [operands(rnd, cid, kind) for _ in range(3)]
        ovf, r0, r1 = run(host_harness, cid, 0, 0, A, Bv, N)
        assert ovf == 0, "64-bit column overflow in the three-term fold (%s operands)" % kind
        re = sum(val(a[0]) * val(b[0]) - val(a[1]) * val(b[1]) for a, b in zip(A, Bv))
        im = sum(val(a[0]) * val(b[1]) + val(a[1]) * val(b[0]) for a, b in zip(A, Bv))
        assert tight(r0) and tight(r1)
        assert val(r0) % p == re * Rinv % p and val(r1) % p == im * Rinv % p
    # operands as the kernel has them (lines below 21 p, accumulator coefficients below 32 p): outputs below 4 p, which is
    # what from_ux and the xi multiple of the next publication rely on
    for _ in range(12):
        A = [(limbs_of(rnd.randrange(21 * p), N), limbs_of(rnd.randrange(21 * p), N)) for _ in range(3)]
        Bv = [(limbs_of(rnd.randrange(31 * p), N), limbs_of(rnd.randrange(31 * p), N)) for _ in range(3)]
        ovf, r0, r1 = run(host_harness, cid, 0, 0, A, Bv, N)
        assert ovf == 0 and val(r0) < 4 * p and val(r1) < 4 * p
    # symmetric squaring rows: (doubled, doubled, doubled, unused) and (plain, doubled, doubled, plain)
    for kinds in ((2, 2, 2, 0), (1, 2, 2, 1), (2, 0, 1, 2)):
        arg = sum(k << (2 * t) for t, k in enumerate(kinds))
        for kind in ["rand"] * 6 + ["max"]:
            A = [operands(rnd, cid, kind) for _ in range(4)]
            Bv = [operands(rnd, cid, kind) for _ in range(4)]
            ovf, r0, r1 = run(host_harness, cid, 1, arg, A, Bv, N)
            assert ovf == 0, "64-bit column overflow in the squaring (%s operands, kinds %s)" % (kind, kinds)
            re = sum(k * (val(a[0]) * val(b[0]) - val(a[1]) * val(b[1])) for k, a, b in zip(kinds, A, Bv))
            im = sum(k * (val(a[0]) * val(b[1]) + val(a[1]) * val(b[0])) for k, a, b in zip(kinds, A, Bv))
            assert tight(r0) and tight(r1)
            assert val(r0) % p == re * Rinv % p and val(r1) % p == im * Rinv % p
    # xi multiple of a reduction's output (value < 2 p): tight, non-negative, below 32 p
    xi_re = 9 if cid == 0 else 1
    for _ in range(20):
        a = (limbs_of(rnd.randrange(2 * p), N), limbs_of(rnd.randrange(2 * p), N))
        ovf, r0, r1 = run(host_harness, cid, 2, 0, [a] * 3, [a] * 3, N)
        assert tight(r0) and tight(r1)
        assert val(r0) % p == (xi_re * val(a[0]) - val(a[1])) % p and val(r1) % p == (xi_re * val(a[1]) + val(a[0])) % p
        assert val(r0) < 32 * p and val(r1) < 32 * p


@pytest.mark.parametrize("cname,cid", [("altbn128", 0), ("bls12", 1)])
def test_lane_pair_point_steps_match_the_library_steps(host_harness, cname, cid):
    c = CURVES[cname]
    G = Groups(c)
    rnd = random.Random(31 + cid)
    for _ in range(3):
        k1, k2 = rnd.randrange(1, c.r), rnd.randrange(1, c.r)
        g1 = G.g1_bytes(G.g1_mul(c.g1, k1))
        g2 = G.g2_bytes(G.g2_mul(c.g2, k2))
        rc = host_harness.ht_rx_miller(cid, (ctypes.c_uint8 * len(g1)).from_buffer_copy(g1), (ctypes.c_uint8 * len(g2)).from_buffer_copy(g2))
        assert rc == 0, "lane-pair point steps differ from pairing.hpp (code %d: 1 + first differing step, -3 = column overflow)" % rc


@pytest.mark.parametrize("cname,cid", [("altbn128", 0), ("bls12", 1)])
def test_key_sum_on_carry_free_limbs(host_harness, cname, cid):
    """rx_jac.hpp (the G2 key sum of AggregatePoints, curves/curve.go:73-121, as Jacobian mixed additions on signed 28-bit limbs)
    against the library's 32-bit additions and the Python oracle: random keys, the same key twice in a row (the doubling
    branch), a key followed by its negative (the sum passes through infinity), keys at infinity, an off-curve key."""
    c = CURVES[cname]
    G = Groups(c)
    rnd = random.Random(91 + cid)
    n_fp = 32 if cid == 0 else 48
    base = [G.g2_mul(c.g2, rnd.randrange(1, c.r)) for _ in range(6)]
    cases = [
        [base[0]],
        [base[0], base[1], base[2]],
        [base[0], base[0]],                                   # P + P
        [base[0], G.g2_neg(base[0])],                         # P - P = infinity
        [base[0], G.g2_neg(base[0]), base[1]],                # ... and on from infinity
        [None, base[2], None, base[3]],                       # keys at infinity
        [base[0], base[1], G.g2_add(base[0], base[1])],       # running sum equals the next key: doubling in mid-sum
        [base[0], base[1], G.g2_neg(G.g2_add(base[0], base[1])), base[4], base[5]],
        [G.g2_mul(c.g2, rnd.randrange(1, c.r)) for _ in range(25)],
    ]
    for pts in cases:
        raw = b"".join(G.g2_bytes(p) for p in pts)
        got, ref = (ctypes.c_uint8 * (4 * n_fp))(), (ctypes.c_uint8 * (4 * n_fp))()
        rc = host_harness.ht_rx_sum(cid, (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw), len(pts), got, ref)
        assert rc == 0, rc
        want = G.g2_bytes(G.g2_sum([p for p in pts]))
        assert bytes(ref) == want and bytes(got) == want, len(pts)
        # the same sum on an emulated lane pair (rx_jacpair.hpp: the shipped key-sum kernel's arithmetic)
        gp = (ctypes.c_uint8 * (4 * n_fp))()
        rc = host_harness.ht_rx_sumpair(cid, (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw), len(pts), gp)
        assert rc == 0, rc
        assert bytes(gp) == want, len(pts)
    bad = bytearray(G.g2_bytes(base[0])); bad[-1] ^= 1
    got, ref = (ctypes.c_uint8 * (4 * n_fp))(), (ctypes.c_uint8 * (4 * n_fp))()
    assert host_harness.ht_rx_sum(cid, (ctypes.c_uint8 * len(bad)).from_buffer_copy(bytes(bad)), 1, got, ref) == -2
    assert host_harness.ht_rx_sumpair(cid, (ctypes.c_uint8 * len(bad)).from_buffer_copy(bytes(bad)), 1, got) == -2


@pytest.mark.parametrize("cid", [0, 1])
def test_sqrt_powers_on_carry_free_limbs(host_harness, cid):
    """rx_pow.hpp: the symmetric squaring on worst-case limbs stays inside the 64-bit columns; the sliding-window
    powers (W = 3, 4) give the same field elements as fp.hpp's exponentiations (hash-to-G1 square roots)."""
    c, p, N, Rp = geom(cid)
    Rinv = pow(Rp, -1, p)
    rnd = random.Random(77 + cid)
    host_harness.ht_rx_pow.restype = ctypes.c_int
    top_p = p >> (W * (N - 1))
    cases = [[MASK] * (N - 1) + [2 * top_p + 1], limbs_of(p - 1, N), limbs_of(0, N), limbs_of(1, N)]
    cases += [limbs_of(rnd.randrange(2 * p), N) for _ in range(8)]
    for l in cases:
        lim = (ctypes.c_int32 * N)(*l)
        assert host_harness.ht_rx_pow(cid, 0, None, lim) == 0, "column overflow in the squaring"
        out = list(lim)
        assert all(0 <= x <= MASK for x in out[:-1])
        v = val(l)
        got = sum(x << (W * i) for i, x in enumerate(out))
        assert got % p == v * v * Rinv % p and 0 <= got < v * v // Rp + p + 1
    n = 32 if cid == 0 else 48
    for op, e in [(1, (p + 1) // 4), (2, (p - 3) // 4)]:
        for x in [0, 1, 2, p - 1, rnd.randrange(p), rnd.randrange(p), rnd.randrange(p)]:
            buf = (ctypes.c_uint8 * n)(*x.to_bytes(n, "big"))
            rc = host_harness.ht_rx_pow(cid, op, buf, None)
            assert rc == 0, (op, x, rc)
            assert int.from_bytes(bytes(buf), "big") == pow(x, e, p)
