"""CPU tier: the device arithmetic headers (bgls_amd/csrc/*.hpp), compiled for the host by the
test harness, diffed routine by routine against the oracle.  Catches arithmetic bugs without a GPU;
the GPU tier repeats the comparisons through the C ABI on the real kernels."""
import ctypes
import random

from oracle.pyref import h2c
from oracle.pyref.hashes import blake2b512, keccak256_legacy
from oracle.pyref.pairing import Pairing
from oracle.pyref.params import CURVES


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def out(n):
    return (ctypes.c_uint8 * n)()


def test_fp_and_fp2(host_harness, curve):
    lib, cid, n = host_harness, curve["id"], curve["fp"]
    c = CURVES[curve["name"]]
    p = c.p
    T = Pairing(c).T
    rnd = random.Random(7)
    ops = ((0, lambda a, b: a * b % p), (1, lambda a, b: a * a % p), (2, lambda a, b: (a + b) % p), (3, lambda a, b: (a - b) % p),
           (4, lambda a, b: (-a) % p), (5, lambda a, b: pow(a, p - 2, p)), (6, lambda a, b: pow(a, (p + 1) // 4, p)),
           (9, lambda a, b: pow(a, p - 2, p)))          # 5: fp_inv (batched division steps since round 4), 9: the binary Euclid it replaced
    samples = [(0, 0), (1, p - 1), (p - 1, p - 1), (p - 1, 1), (2, (p + 1) // 2)] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(24)]
    for op, fn in ops:
        for a, b in samples:
            o = out(n)
            assert lib.ht_fp_op(cid, op, B(a.to_bytes(n, "big")), B(b.to_bytes(n, "big")), o) == 0
            assert int.from_bytes(bytes(o), "big") == fn(a, b), (op, a, b)
    # the inverse on many more values: small, p - small, powers of two (long runs of division steps without a swap), random
    inv_samples = list(range(0, 40)) + [p - k for k in range(1, 40)] + [1 << k for k in range(1, p.bit_length() - 1, 7)] + [rnd.randrange(p) for _ in range(600)]
    for a in inv_samples:
        o = out(n)
        assert lib.ht_fp_op(cid, 5, B(a.to_bytes(n, "big")), B(bytes(n)), o) == 0
        assert int.from_bytes(bytes(o), "big") == pow(a, p - 2, p), a
    # Legendre symbol by the binary Jacobi algorithm == Euler criterion (hash.go:254-265), on the Montgomery
    # residue (op 7) and on the plain residue (op 8)
    for a in [0, 1, 2, 3, 4, p - 1, p - 2, (p + 1) // 2, 1 << 64, (1 << 200) + 12345] + [rnd.randrange(p) for _ in range(200)]:
        want = 0 if a == 0 else (1 if pow(a, (p - 1) // 2, p) == 1 else -1)
        for op in (7, 8):
            assert lib.ht_fp_op(cid, op, B(a.to_bytes(n, "big")), B(bytes(n)), out(n)) - 10 == want, (op, a)
    f2b = lambda x: x[0].to_bytes(n, "big") + x[1].to_bytes(n, "big")
    for op, fn in ((0, T.f2_mul), (1, lambda a, b: T.f2_sqr(a)), (2, lambda a, b: T.f2_mulxi(a)), (3, lambda a, b: T.f2_inv(a))):
        for _ in range(16):
            a, b = (rnd.randrange(p), rnd.randrange(p)), (rnd.randrange(p), rnd.randrange(p))
            o = out(2 * n)
            lib.ht_f2_op(cid, op, B(f2b(a)), B(f2b(b)), o)
            assert bytes(o) == f2b(fn(a, b)), op
    # worst-case operands for the lazy-reduction bounds
    a = (p - 1, p - 1)
    o = out(2 * n)
    lib.ht_f2_op(cid, 0, B(f2b(a)), B(f2b(a)), o)
    assert bytes(o) == f2b(T.f2_mul(a, a))


def test_fp12_tower(host_harness, curve):
    lib, cid, n = host_harness, curve["id"], curve["fp"]
    c = CURVES[curve["name"]]
    PR = Pairing(c)
    T = PR.T
    rnd = random.Random(9)
    r12 = lambda: tuple(tuple((rnd.randrange(c.p), rnd.randrange(c.p)) for _ in range(3)) for _ in range(2))
    for _ in range(3):
        a, b = r12(), r12()
        for op, fn in ((0, T.f12_mul), (1, lambda a, b: T.f12_sqr(a)), (2, lambda a, b: T.f12_inv(a)), (3, lambda a, b: T.f12_frob(a, 1)),
                       (4, lambda a, b: T.f12_frob(a, 2)), (5, lambda a, b: T.f12_frob(a, 3)), (7, lambda a, b: T.f12_conj(a))):
            o = out(12 * n)
            assert lib.ht_f12_op(cid, op, B(PR.gt_bytes(a)), B(PR.gt_bytes(b)), o) == 0
            assert bytes(o) == PR.gt_bytes(fn(a, b)), op
    u = r12()
    u = T.f12_mul(T.f12_conj(u), T.f12_inv(u))
    u = T.f12_mul(T.f12_frob(u, 2), u)
    o = out(12 * n)
    lib.ht_f12_op(cid, 6, B(PR.gt_bytes(u)), None, o)
    assert bytes(o) == PR.gt_bytes(T.f12_sqr(u))          # cyclotomic squaring on a unitary element


def test_miller_final_exp_and_groups_against_golden(host_harness, curve):
    lib, cid, n, v = host_harness, curve["id"], curve["fp"], curve["vec"]
    for row in v["pairings"]:
        m = out(12 * n)
        assert lib.ht_miller(cid, B(bytes.fromhex(row["g1"])), B(bytes.fromhex(row["g2"])), m) == 0
        assert bytes(m).hex() == row["miller"]
        g = out(12 * n)
        lib.ht_f12_op(cid, 8, m, None, g)
        assert bytes(g).hex() == row["gt"]
    for base, key, size in ((0, "scale_g1", 2 * n), (10, "scale_g2", 4 * n)):
        for row in v[key]:
            k = int(row["k"])
            if 0 <= k < 1 << 256:
                o = out(size)
                lib.ht_group_op(cid, base + 1, B(bytes.fromhex(row["pt"])), None, B(k.to_bytes(32, "big")), o)
                assert bytes(o).hex() == row["out"], k
                w = out(size)                  # the scale kernels' chain (signed radix-16 windows): same golden point
                lib.ht_group_op(cid, base + 4, B(bytes.fromhex(row["pt"])), None, B(k.to_bytes(32, "big")), w)
                assert bytes(w).hex() == row["out"], k
        # scalars that stress the recoding: runs of ones (carries across limbs), isolated top bits, all digits negative
        pt = bytes.fromhex(v[key][0]["pt"])
        rnd = random.Random(base + cid)
        ks = [(1 << 256) - 1, (1 << 255) + 1, 1 << 255, (1 << 200) - (1 << 31), 0x77777777 << 100, 0x99999999 << 64 | 0x9, 65537, (1 << 17) - 1]
        ks += [rnd.getrandbits(256) for _ in range(6)] + [rnd.getrandbits(130) for _ in range(3)]
        for k in ks:
            o, w = out(size), out(size)
            lib.ht_group_op(cid, base + 1, B(pt), None, B(k.to_bytes(32, "big")), o)
            lib.ht_group_op(cid, base + 4, B(pt), None, B(k.to_bytes(32, "big")), w)
            assert bytes(o) == bytes(w), hex(k)
    for base, key, size in ((0, "sum_g1", 2 * n), (10, "sum_g2", 4 * n)):
        pts = [bytes.fromhex(x) for x in v[key]["pts"]]
        o = out(size)
        lib.ht_group_op(cid, base, B(pts[0]), B(pts[5]), None, o)     # P + P  (doubling inside add)
        o2 = out(size)
        lib.ht_group_op(cid, base + 1, B(pts[0]), None, B((2).to_bytes(32, "big")), o2)
        assert bytes(o) == bytes(o2)
        o3 = out(size)
        neg = pts[6]
        lib.ht_group_op(cid, base, B(pts[1]), B(neg), None, o3)      # P + (-P) = infinity
        assert bytes(o3) == bytes(size)


def test_hashes_and_h2c(host_harness, curve, kat):
    lib, cid, n = host_harness, curve["id"], curve["fp"]
    rnd = random.Random(13)
    for ln in (0, 1, 64, 123, 124, 127, 128, 134, 135, 136, 137, 251, 252, 300):
        m = rnd.randbytes(ln)
        o = out(32)
        lib.ht_keccak256(B(m), ctypes.c_size_t(ln), 7, o)
        assert bytes(o) == keccak256_legacy(b"\x07" + m)
        o = out(64)
        lib.ht_blake2b(B(m), ctypes.c_size_t(ln), 1, o)
        assert bytes(o) == blake2b512(m + b"G1_1")
    for row in kat[curve["name"]] + curve["vec"]["h2c"]:
        m = bytes.fromhex(row["msg"])
        o = out(2 * n)
        assert lib.ht_hash_to_g1(cid, B(m), ctypes.c_size_t(len(m)), o) == 0
        assert bytes(o).hex() == row["point"]


def test_wire_formats_and_blake2x_node(host_harness):
    """wire.hpp (alt-bn128 compressed forms, curves/altbn128.go:81-89,203-221,296-376) and hashes.hpp's BLAKE2Xb node,
    compiled for the host, against the Python oracle: round trips, sign-bit rules, rejected encodings."""
    import ctypes, random
    from oracle.pyref import wire, hashes
    from oracle.pyref.params import BN254 as C
    from oracle.pyref.groups import Groups
    G = Groups(C)
    rnd = random.Random(41)
    B = lambda b: (ctypes.c_uint8 * len(b)).from_buffer_copy(bytes(b))

    def run(op, data, outlen):
        o = (ctypes.c_uint8 * outlen)()
        return host_harness.ht_wire(op, B(data), o), bytes(o)

    for _ in range(12):
        P = G.g1_mul(C.g1, rnd.randrange(1, C.r)); Q = G.g2_mul(C.g2, rnd.randrange(1, C.r))
        rc, c1 = run(0, G.g1_bytes(P), 32); assert rc == 1 and c1 == wire.compress_g1(P)
        rc, c2 = run(1, G.g2_bytes(Q), 64); assert rc == 1 and c2 == wire.compress_g2(Q)
        rc, u1 = run(2, c1, 64); assert rc == 1 and u1 == G.g1_bytes(P)
        rc, u2 = run(3, c2, 128); assert rc == 1 and u2 == G.g2_bytes(Q)
        f = bytearray(c1); f[0] ^= 128
        rc, u = run(2, f, 64); assert rc == 1 and u == G.g1_bytes(G.g1_neg(P))
        f = bytearray(c2); f[0] ^= 128
        assert run(3, f, 128)[0] == 0 and wire.decompress_g2(bytes(f))[1] is False        # one bit only: not a point
        f[32] ^= 128
        rc, u = run(3, f, 128); assert rc == 1 and u == G.g2_bytes(G.g2_neg(Q))
    # arbitrary x: same accept / reject decision and same bytes as the oracle
    for _ in range(40):
        d1 = bytearray(rnd.randbytes(32)); d1[0] &= rnd.choice((0x3f, 0xbf, 0xff))
        pt, ok = wire.decompress_g1(bytes(d1))
        rc, u = run(2, d1, 64)
        assert rc == (1 if ok else 0) and (not ok or u == G.g1_bytes(pt))
    for _ in range(16):
        d2 = bytearray(rnd.randbytes(64)); d2[0] &= rnd.choice((0x3f, 0xbf)); d2[32] &= rnd.choice((0x3f, 0xbf))
        pt, ok = wire.decompress_g2(bytes(d2), subgroup=False)        # wire.hpp decodes; the kernel adds the subgroup test
        rc, u = run(3, d2, 128)
        assert rc == (1 if ok else 0) and (not ok or u == G.g2_bytes(pt))
    assert run(2, bytes(32), 64) == (1, bytes(64)) and run(3, bytes(64), 128) == (1, bytes(128))
    assert run(2, b"\x80" + bytes(31), 64) == (1, bytes(64))                              # x = 0 with the flag: still infinity
    assert run(0, bytes(64), 32) == (1, bytes(32)) and run(1, bytes(128), 64) == (1, bytes(64))
    assert run(2, (C.p + 1).to_bytes(32, "big"), 64)[0] == 0                               # x >= q
    # BLAKE2Xb node
    for ln, ol in ((0, 16), (100, 64), (300, 65), (10, 1000)):
        d = rnd.randbytes(ln)
        root = __import__("hashlib").blake2b(d, digest_size=64, node_offset=ol << 32).digest()
        want = hashes.blake2xb(d, ol)
        got = b""
        for i in range((ol + 63) // 64):
            take = min(64, ol - 64 * i)
            o = (ctypes.c_uint8 * 64)()
            host_harness.ht_blake2xb_node(B(root), i, ol, take, o)
            got += bytes(o)[:take]
        assert got == want


def test_r28_consumer_arithmetic(host_harness):
    """r28.hpp (28-bit limbs, Montgomery radix 2^280) against the library's 32-bit arithmetic on the host: conversion
    round trip (shift + one Barrett step), Fp2 product, xi multiple, and a five-term dot product with one reduction."""
    import ctypes, random
    p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    rnd = random.Random(28)
    B = lambda b: (ctypes.c_uint8 * len(b)).from_buffer_copy(bytes(b))
    enc = lambda re, im: re.to_bytes(32, "big") + im.to_bytes(32, "big")
    dec = lambda b: (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 1 << 253, (1 << 222) - 1, 1 << 222]
    vals = [(rnd.choice(edge), rnd.choice(edge)) for _ in range(30)] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(200)]
    for (a0, a1), (b0, b1) in zip(vals, vals[::-1]):
        o = (ctypes.c_uint8 * 64)()
        assert host_harness.ht_r28(0, B(enc(a0, a1)), B(enc(b0, b1)), o) == 0 and dec(bytes(o)) == (a0, a1)
        assert host_harness.ht_r28(1, B(enc(a0, a1)), B(enc(b0, b1)), o) == 0
        assert dec(bytes(o)) == ((a0 * b0 - a1 * b1) % p, (a0 * b1 + a1 * b0) % p)
        assert host_harness.ht_r28(2, B(enc(a0, a1)), B(enc(b0, b1)), o) == 0
        assert dec(bytes(o)) == ((9 * a0 - a1) % p, (9 * a1 + a0) % p)
        assert host_harness.ht_r28(3, B(enc(a0, a1)), B(enc(b0, b1)), o) == 1
        assert host_harness.ht_r28(4, B(enc(a0, a1)), B(enc(b0, b1)), o) == 1


def test_device_headers_under_asan_and_ubsan():
    """the host build of the device arithmetic headers under AddressSanitizer + UndefinedBehaviorSanitizer
    (tests/harness/san_main.cpp; -fno-sanitize-recover=all: any finding aborts).  It walks the lane-pair point steps of a whole Miller
    loop on both curves and on alt-bn128's nine-limb 29-bit form, the consumer's folds / squarings / xi multiples on worst-case
    limbs, the 32-bit Miller loop, both hash maps, the key sums on the carry-free limbs and the square-root powers.  -O0: the
    optimised sanitizer build of these headers takes 12 minutes, this one a minute (it is rebuilt only when a header changes)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "harness", "san_main.cpp")
    exe = os.path.join(root, "tests", "harness", "san_main.bin")
    csrc = os.path.join(root, "bgls_amd", "csrc")
    deps = [src, os.path.join(root, "tests", "harness", "host_harness.cpp")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O0", "-g0", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-pthread", "-o", exe, src], check=True, timeout=1200)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "san_main ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
