"""Compressed wire formats: alt-bn128 (curves/altbn128.go:81-89,203-221,296-376) and, second half of this
file, BLS12-381 in the ebfull/pairing layout (curves/bls12_381.go:54-62,115-123,242-264).

CPU tier: the oracle (oracle/pyref/wire.py) and the host-compiled device routines (wire.hpp) against the committed
fixture; the reference's own TestMarshal (curves/curve_test.go:23-118) is a Marshal -> Unmarshal round trip, mirrored
here.  GPU tier: the same fixture and round trips through bgls_compress_points / bgls_decompress_points and the
Point.Marshal / UnmarshalG1 / UnmarshalG2 mirror."""
import ctypes
import json
import os
import random

import pytest

from oracle.pyref import wire
from oracle.pyref.groups import Groups
from oracle.pyref.params import BN254

HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "wire_altbn128.json")))
B = lambda b: (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def test_oracle_matches_fixture_and_round_trips():
    G = Groups(BN254)
    for row in V["g1"]:
        assert wire.compress_g1(G.g1_from_bytes(bytes.fromhex(row["pt"]))).hex() == row["compressed"]
    for row in V["g2"]:
        assert wire.compress_g2(G.g2_from_bytes(bytes.fromhex(row["pt"]))).hex() == row["compressed"]
    for row in V["g1_decode"]:
        pt, ok = wire.decompress_g1(bytes.fromhex(row["in"]))
        assert ok == row["ok"] and (not ok or G.g1_bytes(pt).hex() == row["pt"])
    for row in V["g2_decode"]:
        pt, ok = wire.decompress_g2(bytes.fromhex(row["in"]))
        assert ok == row["ok"] and (not ok or G.g2_bytes(pt).hex() == row["pt"])
        pt, dec = wire.decompress_g2(bytes.fromhex(row["in"]), subgroup=False)
        assert dec == row["decoded"] and (not dec or G.g2_bytes(pt).hex() == row["pt"])
    assert any(r["decoded"] and not r["ok"] for r in V["g2_decode"])     # on the twist, outside G2: rejected by UnmarshalG2
    # TestMarshal shape: Unmarshal(Marshal(P)) == P for random points
    rnd = random.Random(8)
    for _ in range(6):
        P = G.g1_mul(BN254.g1, rnd.randrange(1, BN254.r)); Q = G.g2_mul(BN254.g2, rnd.randrange(1, BN254.r))
        assert wire.decompress_g1(wire.compress_g1(P)) == (P, True) and wire.decompress_g2(wire.compress_g2(Q)) == (Q, True)


def test_device_routines_on_host_match_fixture(host_harness):
    def run(op, data, outlen):
        o = (ctypes.c_uint8 * outlen)()
        return host_harness.ht_wire(op, B(bytes.fromhex(data)), o), bytes(o).hex()
    for row in V["g1"]:
        assert run(0, row["pt"], 32) == (1, row["compressed"])
    for row in V["g2"]:
        assert run(1, row["pt"], 64) == (1, row["compressed"])
    for row in V["g1_decode"]:
        rc, out = run(2, row["in"], 64)
        assert rc == (1 if row["ok"] else 0) and (not row["ok"] or out == row["pt"])
    for row in V["g2_decode"]:                                # wire.hpp decodes; the subgroup test is applied by the kernel
        rc, out = run(3, row["in"], 128)
        assert rc == (1 if row["decoded"] else 0) and (not row["decoded"] or out == row["pt"])


@pytest.mark.gpu
def test_gpu_wire_fixture_and_batches(gpu_lib):
    for group, key, cb in ((1, "g1", 32), (2, "g2", 64)):
        pts = b"".join(bytes.fromhex(r["pt"]) for r in V[key])
        o = (ctypes.c_uint8 * (len(V[key]) * cb))()
        assert gpu_lib.bgls_compress_points(0, group, B(pts), len(V[key]), o) == 0
        assert bytes(o) == b"".join(bytes.fromhex(r["compressed"]) for r in V[key])
        rows = V[key + "_decode"]
        ins = b"".join(bytes.fromhex(r["in"]) for r in rows)
        out = (ctypes.c_uint8 * (len(rows) * 2 * cb))(); ok = (ctypes.c_uint8 * len(rows))()
        assert gpu_lib.bgls_decompress_points(0, group, B(ins), len(rows), out, ok) == 0
        for i, r in enumerate(rows):
            assert ok[i] == (1 if r["ok"] else 0), (key, i)
            want = bytes.fromhex(r["pt"]) if r["ok"] else bytes(2 * cb)
            assert bytes(out)[i * 2 * cb:(i + 1) * 2 * cb] == want, (key, i)
    # off-curve input to compress is an encoding error
    bad = bytearray(bytes.fromhex(V["g1"][0]["pt"])); bad[40] ^= 1
    assert gpu_lib.bgls_compress_points(0, 1, B(bad), 1, (ctypes.c_uint8 * 32)()) == -2


@pytest.mark.gpu
def test_gpu_marshal_round_trip_large_batch(gpu_lib):
    """curves/curve_test.go TestMarshal shape at batch size: Unmarshal(Marshal(P)) == P for 5000 random G1 and G2 points
    (key material as it would arrive over the wire), plus the Point / CurveSystem mirror for single points."""
    from bgls_amd import Altbn128, ScalePoints
    rnd = random.Random(13)
    n = 5000
    ks = [rnd.randrange(1, Altbn128.GetG1Order()) for _ in range(n)]
    for group, gen in ((1, Altbn128.GetG1()), (2, Altbn128.GetG2())):
        pts = ScalePoints([gen] * n, ks)
        raw = b"".join(p.raw for p in pts)
        cb = len(gen.raw) // 2
        comp = (ctypes.c_uint8 * (n * cb))()
        assert gpu_lib.bgls_compress_points(0, group, B(raw), n, comp) == 0
        back = (ctypes.c_uint8 * len(raw))(); ok = (ctypes.c_uint8 * n)()
        assert gpu_lib.bgls_decompress_points(0, group, comp, n, back, ok) == 0
        assert bytes(ok) == b"\x01" * n and bytes(back) == raw
        # spot-check against the oracle
        for i in (0, 17, n - 1):
            d = bytes(comp)[i * cb:(i + 1) * cb]
            pt, good = (wire.decompress_g1 if group == 1 else wire.decompress_g2)(d)
            G = Groups(BN254)
            assert good and (G.g1_bytes(pt) if group == 1 else G.g2_bytes(pt)) == raw[i * 2 * cb:(i + 1) * 2 * cb]
        p0 = pts[3]
        m = p0.Marshal()
        assert len(m) == cb
        q, good = (Altbn128.UnmarshalG1 if group == 1 else Altbn128.UnmarshalG2)(m)
        assert good and q.Equals(p0)
        q2, good2 = (Altbn128.UnmarshalG1 if group == 1 else Altbn128.UnmarshalG2)(p0.MarshalUncompressed())
        assert good2 and q2.Equals(p0)


# ---- BLS12-381: ebfull/pairing ("ZCash") layout -----------------------------------------------------------------------
from oracle.pyref.params import BLS381  # noqa: E402

VB = json.load(open(os.path.join(HERE, "golden", "wire_bls12.json")))
# the format's public known-answer values: the standard generators' compressed encodings (the x coordinates of
# curves/bls12_381.go's generators with the compression flag set; both y are the smaller root)
KAT_G1 = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
KAT_G2 = ("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
          "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")


def test_bls_oracle_matches_fixture_kats_and_round_trips():
    G = Groups(BLS381)
    assert wire.bls_compress_g1(BLS381.g1).hex() == KAT_G1 and wire.bls_compress_g2(BLS381.g2).hex() == KAT_G2
    assert VB["g1"][0]["compressed"] == KAT_G1 and VB["g2"][0]["compressed"] == KAT_G2
    assert wire.bls_compress_g1(None).hex() == "c0" + "00" * 47 and wire.bls_compress_g2(None).hex() == "c0" + "00" * 95
    for row in VB["g1"]:
        assert wire.bls_compress_g1(G.g1_from_bytes(bytes.fromhex(row["pt"]))).hex() == row["compressed"]
    for row in VB["g2"]:
        assert wire.bls_compress_g2(G.g2_from_bytes(bytes.fromhex(row["pt"]))).hex() == row["compressed"]
    for key, dec, tob in (("g1_decode", wire.bls_decompress_g1, G.g1_bytes), ("g2_decode", wire.bls_decompress_g2, G.g2_bytes)):
        for row in VB[key][:12] + VB[key][-6:]:             # the subgroup test is a 255-bit multiplication in Python: a sample
            pt, ok = dec(bytes.fromhex(row["in"]))
            assert ok == row["ok"], row
            pt, d = dec(bytes.fromhex(row["in"]), subgroup=False)
            assert d == row["decoded"] and (not d or tob(pt).hex() == row["pt"]), row
        assert any(r["decoded"] and not r["ok"] for r in VB[key])          # on the curve, outside the subgroup: refused by Check()
        assert any(not r["decoded"] for r in VB[key])
    rnd = random.Random(9)
    for _ in range(3):                                       # TestMarshal shape (curves/curve_test.go:73-84)
        k = rnd.randrange(1, BLS381.r)
        P, Q = G.g1_mul(BLS381.g1, k), G.g2_mul(BLS381.g2, k)
        assert wire.bls_decompress_g1(wire.bls_compress_g1(P)) == (P, True) and wire.bls_decompress_g2(wire.bls_compress_g2(Q)) == (Q, True)


def test_bls_device_routines_on_host_match_fixture(host_harness):
    def run(op, data, outlen):
        o = (ctypes.c_uint8 * outlen)()
        return host_harness.ht_wire(op, B(bytes.fromhex(data)), o), bytes(o).hex()
    for row in VB["g1"]:
        assert run(4, row["pt"], 48) == (1, row["compressed"])
    for row in VB["g2"]:
        assert run(5, row["pt"], 96) == (1, row["compressed"])
    for row in VB["g1_decode"]:                              # wire.hpp decodes; Check() is applied by the kernel
        rc, out = run(6, row["in"], 96)
        assert rc == (1 if row["decoded"] else 0) and (not row["decoded"] or out == row["pt"]), row
    for row in VB["g2_decode"]:
        rc, out = run(7, row["in"], 192)
        assert rc == (1 if row["decoded"] else 0) and (not row["decoded"] or out == row["pt"]), row


@pytest.mark.gpu
def test_gpu_bls_wire_fixture(gpu_lib):
    for group, key, cb in ((1, "g1", 48), (2, "g2", 96)):
        pts = b"".join(bytes.fromhex(r["pt"]) for r in VB[key])
        o = (ctypes.c_uint8 * (len(VB[key]) * cb))()
        assert gpu_lib.bgls_compress_points(1, group, B(pts), len(VB[key]), o) == 0
        assert bytes(o) == b"".join(bytes.fromhex(r["compressed"]) for r in VB[key])
        assert bytes(o)[:cb].hex() == (KAT_G1 if group == 1 else KAT_G2)
        rows = VB[key + "_decode"]
        ins = b"".join(bytes.fromhex(r["in"]) for r in rows)
        out = (ctypes.c_uint8 * (len(rows) * 2 * cb))(); ok = (ctypes.c_uint8 * len(rows))()
        assert gpu_lib.bgls_decompress_points(1, group, B(ins), len(rows), out, ok) == 0
        for i, r in enumerate(rows):
            assert ok[i] == (1 if r["ok"] else 0), (key, i, r.get("note"))
            want = bytes.fromhex(r["pt"]) if r["ok"] else bytes(2 * cb)
            assert bytes(out)[i * 2 * cb:(i + 1) * 2 * cb] == want, (key, i)
    bad = bytearray(bytes.fromhex(VB["g1"][0]["pt"])); bad[60] ^= 1
    assert gpu_lib.bgls_compress_points(1, 1, B(bad), 1, (ctypes.c_uint8 * 48)()) == -2


@pytest.mark.gpu
def test_gpu_bls_marshal_round_trip_large_batch(gpu_lib):
    """curves/curve_test.go TestMarshal at batch size on BLS12-381: Unmarshal(Marshal(P)) == P for 3000 random G1 and G2 points,
    spot-checked against the oracle, plus the Point / CurveSystem mirror (Marshal is 48 / 96 bytes, UnmarshalG1 / UnmarshalG2
    accept both lengths as curves/bls12_381.go:242-264 does)."""
    from bgls_amd import Bls12, ScalePoints
    rnd = random.Random(14)
    n = 3000
    G = Groups(BLS381)
    ks = [rnd.randrange(1, Bls12.GetG1Order()) for _ in range(n)]
    for group, gen in ((1, Bls12.GetG1()), (2, Bls12.GetG2())):
        pts = ScalePoints([gen] * n, ks)
        raw = b"".join(p.raw for p in pts)
        cb = len(gen.raw) // 2
        comp = (ctypes.c_uint8 * (n * cb))()
        assert gpu_lib.bgls_compress_points(1, group, B(raw), n, comp) == 0
        back = (ctypes.c_uint8 * len(raw))(); ok = (ctypes.c_uint8 * n)()
        assert gpu_lib.bgls_decompress_points(1, group, comp, n, back, ok) == 0
        assert bytes(ok) == b"\x01" * n and bytes(back) == raw
        for i in (0, 1234, n - 1):
            d = bytes(comp)[i * cb:(i + 1) * cb]
            pt, good = (wire.bls_decompress_g1 if group == 1 else wire.bls_decompress_g2)(d, subgroup=False)
            assert good and (G.g1_bytes(pt) if group == 1 else G.g2_bytes(pt)) == raw[i * 2 * cb:(i + 1) * 2 * cb]
            assert (wire.bls_compress_g1 if group == 1 else wire.bls_compress_g2)(pt) == d
        p0 = pts[5]
        m = p0.Marshal()
        assert len(m) == cb and m[0] & 0x80
        un = Bls12.UnmarshalG1 if group == 1 else Bls12.UnmarshalG2
        q, good = un(m)
        assert good and q.Equals(p0)
        q2, good2 = un(p0.MarshalUncompressed())
        assert good2 and q2.Equals(p0)
        flipped = bytearray(m); flipped[0] ^= 0x20           # the other root: -P, still a valid point
        q3, good3 = un(bytes(flipped))
        assert good3 and q3.Equals(p0.Mul(-1))
        assert un(bytes([m[0] & 0x7F]) + m[1:]) == (None, False)
