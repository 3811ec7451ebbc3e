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
        # the same sum on an emulated lane pair (rx_jacpair.hpp: the shipped key-sum kernel's arithmetic; alt-bn128 also on the nine
        # 29-bit limbs the kernel runs on since round 5: harness curve id 2)
        for hid in ((cid, 2) if cid == 0 else (cid,)):
            gp = (ctypes.c_uint8 * (4 * n_fp))()
            rc = host_harness.ht_rx_sumpair(hid, (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw), len(pts), gp)
            assert rc == 0, (hid, rc)
            assert bytes(gp) == want, (hid, len(pts))
    bad = bytearray(G.g2_bytes(base[0])); bad[-1] ^= 1
    got, ref = (ctypes.c_uint8 * (4 * n_fp))(), (ctypes.c_uint8 * (4 * n_fp))()
    assert host_harness.ht_rx_sum(cid, (ctypes.c_uint8 * len(bad)).from_buffer_copy(bytes(bad)), 1, got, ref) == -2
    assert host_harness.ht_rx_sumpair(cid, (ctypes.c_uint8 * len(bad)).from_buffer_copy(bytes(bad)), 1, got) == -2
    if cid == 0:
        assert host_harness.ht_rx_sumpair(2, (ctypes.c_uint8 * len(bad)).from_buffer_copy(bytes(bad)), 1, got) == -2


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


# ---------------------------------------------------------------------------------------------------------------- round 5
# alt-bn128's Miller kernel runs on NINE limbs of 29 bits (struct BN254W: 81 instead of 100 multiplier instructions per limb
# product).  A 64-bit column then has 2^6 of head-room instead of 2^8 and the Montgomery radix is only 169 p, so the form's
# rules are tighter -- three-term piles of tight operands fill 63 of the 64 units of 2^58 a column holds, values are kept below
# 4 p explicitly -- and every one of them is checked here on worst-case limb patterns (harness curve id 2).
W29, N29 = 29, 9
MASK29 = (1 << W29) - 1


def limbs29(x):
    out = [(x >> (W29 * i)) & MASK29 for i in range(N29 - 1)]
    out.append(x >> (W29 * (N29 - 1)))
    assert out[-1] < (1 << 32)
    return out


def val29(l):
    return sum(int(v) << (W29 * i) for i, v in enumerate(l))


def run29(lib, op, arg, A, Bv):
    o = (ctypes.c_uint32 * (2 * N29))()
    ovf = lib.ht_rx_raw(2, op, arg, pack(A, N29), pack(Bv, N29), o)
    return ovf, list(o[:N29]), list(o[N29:])


def tight29(l):
    return all(v < (1 << W29) for v in l[:-1]) and l[-1] < (1 << 31)


def operand29(rnd, p, kind, vb):
    top_p = p >> (W29 * (N29 - 1))
    if kind == "max":
        l = [MASK29] * (N29 - 1) + [vb * (top_p + 1) - 1]
        return (l, list(l))
    return (limbs29(rnd.randrange(vb * p)), limbs29(rnd.randrange(vb * p)))


def test_29bit_form_conversions_and_consumer_arithmetic(host_harness):
    lib = host_harness
    p = CURVES["altbn128"].p
    Rp = 1 << (W29 * N29)
    Rinv = pow(Rp, -1, p)
    rnd = random.Random(2905)
    for x in [0, 1, p - 1, p // 2] + [rnd.randrange(p) for _ in range(20)]:
        buf = (ctypes.c_uint8 * 32).from_buffer_copy(x.to_bytes(32, "big"))
        lim = (ctypes.c_uint32 * N29)()
        assert lib.ht_rx_conv(2, 0, buf, lim) == 0
        assert tight29(list(lim)) and val29(lim) % p == x * Rp % p and val29(lim) < 2 * p
        back = (ctypes.c_uint8 * 32)()
        assert lib.ht_rx_conv(2, 1, back, lim) == 0 and int.from_bytes(bytes(back), "big") == x
    for v in [p, 2 * p - 1, 3 * p + 12345, 4 * p - 1]:
        back = (ctypes.c_uint8 * 32)()
        assert lib.ht_rx_conv(2, 1, back, (ctypes.c_uint32 * N29)(*limbs29(v))) == 0
        assert int.from_bytes(bytes(back), "big") == v * Rinv % p
    # the three-term fold: worst-case limbs (every limb at 2^29 - 1, values at the form's bound of 4 p) stay inside the columns
    worst = 0
    for kind in ["rand"] * 12 + ["max"]:
        A = [operand29(rnd, p, kind, 4) for _ in range(3)]
        Bv = [operand29(rnd, p, kind, 4) for _ in range(3)]
        ovf, r0, r1 = run29(lib, 0, 0, A, Bv)
        assert ovf == 0, "64-bit column overflow in the three-term fold (%s operands)" % kind
        re = sum(val29(a[0]) * val29(b[0]) - val29(a[1]) * val29(b[1]) for a, b in zip(A, Bv))
        im = sum(val29(a[0]) * val29(b[1]) + val29(a[1]) * val29(b[0]) for a, b in zip(A, Bv))
        assert tight29(r0) and tight29(r1) and val29(r0) % p == re * Rinv % p and val29(r1) % p == im * Rinv % p
        worst = max(worst, val29(r0), val29(r1))
    assert worst < 3 * p            # what ux_mulxi accepts (the kernel's operands are smaller: lines below 3.1 p, accumulators below 3 p)
    # xi multiple with the quotient subtracted in the same pass: any input below 3 p -> tight, non-negative, below 3.001 p
    edge = [(3 * p - 1, 0), (0, 3 * p - 1), (3 * p - 1, 3 * p - 1), (0, 0), (1, 1), (p, p)]
    for it in range(200):
        v0, v1 = edge[it] if it < len(edge) else (rnd.randrange(3 * p), rnd.randrange(3 * p))
        a = (limbs29(v0), limbs29(v1))
        ovf, r0, r1 = run29(lib, 2, 0, [a] * 3, [a] * 3)
        assert tight29(r0) and tight29(r1), (it, r0, r1)
        assert val29(r0) % p == (9 * v0 - v1) % p and val29(r1) % p == (9 * v1 + v0) % p
        assert val29(r0) < 3.001 * p and val29(r1) < 3.001 * p
    # the squaring as two piles of two slots + quasi-reduction: even rows (2 d0 + p0) + (2 d1 + p1), odd rows 2 (d0 + d1) + 2 d2; worst-case limbs included
    for odd_row in (0, 1):
        for kind in ["rand"] * 6 + ["max"]:
            A = [operand29(rnd, p, kind, 2 if kind == "rand" else 4) for _ in range(4)]
            Bv = [operand29(rnd, p, kind, 3 if kind == "rand" else 4) for _ in range(4)]
            ovf, r0, r1 = run29(lib, 3, odd_row, A, Bv)
            assert ovf == 0, (kind, odd_row)
            ks = [2, 2, 2, 0] if odd_row else [2, 1, 2, 1]
            re = sum(k * (val29(a[0]) * val29(b[0]) - val29(a[1]) * val29(b[1])) for k, a, b in zip(ks, A, Bv))
            im = sum(k * (val29(a[0]) * val29(b[1]) + val29(a[1]) * val29(b[0])) for k, a, b in zip(ks, A, Bv))
            assert tight29(r0) and tight29(r1)
            assert val29(r0) % p == re * Rinv % p and val29(r1) % p == im * Rinv % p
            assert val29(r0) < 2.001 * p and val29(r1) < 2.001 * p
    # quasi-reduction ladders
    for _ in range(60):
        v0, v1 = rnd.randrange(31 * p), rnd.randrange(31 * p)
        a = (limbs29(v0), limbs29(v1))
        ovf, r0, r1 = run29(lib, 4, 0x42, [a] * 3, [a] * 3)
        assert tight29(r0) and val29(r0) % p == v0 % p and val29(r0) < 4.001 * p and val29(r1) % p == v1 % p and val29(r1) < 4.001 * p
        v0 = rnd.randrange(8 * p)
        a = (limbs29(v0), limbs29(v0))
        ovf, r0, r1 = run29(lib, 4, 0x21, [a] * 3, [a] * 3)
        assert tight29(r0) and val29(r0) % p == v0 % p and val29(r0) < 2.001 * p


def test_29bit_form_point_steps_match_the_library_steps(host_harness):
    """The lane-pair point steps on nine 29-bit limbs (one more reduction per doubling, two per addition, carry steps in front of
    every product of a sum) hand over the same line coefficients as pairing.hpp's steps over a whole Miller loop, with every
    column accumulation checked for overflow."""
    c = CURVES["altbn128"]
    G = Groups(c)
    rnd = random.Random(2931)
    for _ in range(3):
        g1 = G.g1_bytes(G.g1_mul(c.g1, rnd.randrange(1, c.r)))
        g2 = G.g2_bytes(G.g2_mul(c.g2, rnd.randrange(1, c.r)))
        rc = host_harness.ht_rx_miller(2, (ctypes.c_uint8 * len(g1)).from_buffer_copy(g1), (ctypes.c_uint8 * len(g2)).from_buffer_copy(g2))
        assert rc == 0, "29-bit lane-pair point steps differ from pairing.hpp (code %d: 1 + first differing step, -3 = column overflow)" % rc


def test_29bit_form_sqrt_powers(host_harness):
    """rx_pow.hpp on nine 29-bit limbs (alt-bn128's hash-to-G1 square root since round 5): the symmetric squaring on worst-case limbs
    stays inside the signed columns (a column holds NL products' worth of 2^58 plus the reduction's NL: 18 of 32 units), the
    sliding-window powers give fp.hpp's field elements."""
    p = CURVES["altbn128"].p
    Rp = 1 << (W29 * N29)
    Rinv = pow(Rp, -1, p)
    rnd = random.Random(2977)
    host_harness.ht_rx_pow.restype = ctypes.c_int
    top_p = p >> (W29 * (N29 - 1))
    cases = [[MASK29] * (N29 - 1) + [2 * top_p + 1], limbs29(p - 1), limbs29(0), limbs29(1)] + [limbs29(rnd.randrange(2 * p)) for _ in range(8)]
    for l in cases:
        lim = (ctypes.c_int32 * N29)(*l)
        assert host_harness.ht_rx_pow(2, 0, None, lim) == 0, "column overflow in the squaring"
        out = list(lim)
        assert all(0 <= x <= MASK29 for x in out[:-1])
        v = val29(l)
        got = val29(out)
        assert got % p == v * v * Rinv % p and 0 <= got < v * v // Rp + p + 1
    for op, e in [(1, (p + 1) // 4), (2, (p - 3) // 4)]:
        for x in [0, 1, 2, p - 1, rnd.randrange(p), rnd.randrange(p), rnd.randrange(p)]:
            buf = (ctypes.c_uint8 * 32)(*x.to_bytes(32, "big"))
            rc = host_harness.ht_rx_pow(2, op, buf, None)
            assert rc == 0, (op, x, rc)
            assert int.from_bytes(bytes(buf), "big") == pow(x, e, p)


# ---------------------------------------------------------------------------------------------------------------- round 6
# k_bls_sw_jacobi's work item on the carry-free limbs (bgls_amd/csrc/h2c_x.hpp): the same header compiled for the host, every column accumulation
# checked, against h2c.hpp's 32-bit form of curves/hash.go:97-167 (which the reference's 11 KATs pin through ht_hash_to_g1).
def test_bls_sw_item_on_carry_free_limbs(host_harness):
    host_harness.ht_bls_sw_x.restype = ctypes.c_int
    host_harness.ht_bls_sw_x.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p]
    rnd = random.Random(606)
    msgs = [b"", b"a", b"abc", bytes(range(200))] + [rnd.randbytes(rnd.randrange(1, 90)) for _ in range(120)]
    seen = set()
    for m in msgs:
        for k in (0, 1):
            out = ctypes.create_string_buffer(96)
            rc = host_harness.ht_bls_sw_x(m, len(m), k, out)
            assert rc == 3, (m, k, rc)                      # H2C_SW, same point as the 32-bit form, no column overflow
            seen.add(bytes(out.raw))
    assert len(seen) == 2 * len(msgs)


def test_bls_sw_item_degenerate_and_unreduced_digests(host_harness):
    """t = digest mod q with the digest read as a 512-bit big-endian integer: the kinds no message reaches (t = 0, t = +-sqrt(-5): curves/hash.go:109-118
    via bls12FTRoot1 / 2, curves/bls12_381.go:345-346) and digests far above q, all through the one two-product reduction of h2c_x.hpp."""
    host_harness.ht_bls_sw_x_digest.restype = ctypes.c_int
    host_harness.ht_bls_sw_x_digest.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    p = CURVES["bls12"].p
    root = pow(-5, (p + 1) // 4, p)
    assert root * root % p == p - 5
    roots = sorted([root, p - root])

    def kind(v):
        out = ctypes.create_string_buffer(96)
        return host_harness.ht_bls_sw_x_digest(v.to_bytes(64, "big"), out)

    top = (1 << 512) - 1
    assert kind(0) == 0 and kind(p) == 0 and kind(p * (top // p)) == 0                  # H2C_INF
    got = {kind(roots[0]), kind(roots[1])}
    assert got == {1, 2}                                                                # +g1 / -g1, whichever root is FT_ROOT1
    assert kind(roots[0] + 7 * p) == kind(roots[0]) and kind(roots[1] + p * ((top - roots[1]) // p)) == kind(roots[1])
    rnd = random.Random(9)
    for v in [1, 2, p - 1, p + 1, top, top - 1, (1 << 384) - 1, 1 << 384, (1 << 384) + 1, 1 << 511] + [rnd.getrandbits(512) for _ in range(40)]:
        assert kind(v) == 3, hex(v)                                                     # H2C_SW, same point as the 32-bit form, no column overflow
