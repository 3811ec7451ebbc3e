#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/ (run in the build container only).

 h2c_kat.json        the reference's own hash-to-G1 vectors, re-encoded as hex DATA:
                     curves/testcases/altbn128G1Hash.dat, curves/testcases/bls12G1Hash.dat,
                     curves/altbn128_test.go:16-21 (Solidity point), curves/bls12_test.go:57-67,
                     curves/altbn128_test.go:26-38 (G2 generator coordinates).
 hae_<curve>.json    BLAKE2Xb outputs, hashed aggregation exponents and accept/reject cases of the HAE /
                     multiplicity flows (bgls/blsHAE.go, bgls/blsKosk.go:137-150) from the Python oracle.
 wire_bls12.json     BLS12-381 compressed encodings (ebfull/pairing layout) and UnmarshalG1 / UnmarshalG2 decisions (curves/bls12_381.go:54-62,242-264)
 wire_altbn128.json  compressed point encodings and Unmarshal decisions (curves/altbn128.go:81-89,203-221,296-376)
                     from the Python oracle (oracle/pyref/wire.py).
 vectors_<curve>.json  outputs of the Python oracle (oracle/pyref) on seeded inputs: Miller / GT
                     values, pairing products, group sums, scalar multiples, extra hash-to-G1
                     messages, end-to-end accept/reject cases mirroring bgls/bgls_test.go:40-77 and
                     bgls/blsKosk_test.go:35-64.
"""
import base64, json, os, random, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle.pyref.params import BN254, BLS381
from oracle.pyref.pairing import Pairing
from oracle.pyref import h2c, scheme

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/curves/testcases"


def kat():
    out = {}
    for name, n in (("altbn128", 32), ("bls12", 48)):
        rows = []
        for line in open(os.path.join(REF, name + "G1Hash.dat")):
            a, b = line.strip().split(",")
            rows.append({"msg": base64.b64decode(a).hex(), "point": base64.b64decode(b).hex()})
        out[name] = rows
    a = 9121282642809701931333593728297233225556711250127745709186816755779879923737
    x = 11423386531623885114587219621463106117140760157404497425836076043015227528156
    y = 20262289731964024720969923714809935701428881933342918937283877214228227624643
    out["altbn128"].append({"msg": a.to_bytes((a.bit_length() + 7) // 8, "big").hex(), "point": (x.to_bytes(32, "big") + y.to_bytes(32, "big")).hex()})
    x = 315124130825307604287835216317628428134609737854237653839182597515996444073032649481416725367158979153513345579672
    y = 3093537746211397858160667262592024570071165158580434464756577567510401504168962073691924150397172185836012224315174
    out["bls12"].append({"msg": "", "point": (x.to_bytes(48, "big") + y.to_bytes(48, "big")).hex()})
    g2 = [11559732032986387107991004021392285783925812861821192530917403151452391805634,
          10857046999023057135944570762232829481370756359578518086990519993285655852781,
          4082367875863433681332203403145435568316851327593401208105741076214120093531,
          8495653923123431417604973247489272438418190587263600148770280649306958101930]
    out["altbn128_g2_generator"] = b"".join(v.to_bytes(32, "big") for v in g2).hex()
    json.dump(out, open(os.path.join(HERE, "h2c_kat.json"), "w"), indent=1)


def vectors(c, seed):
    rnd = random.Random(seed)
    PR = Pairing(c); G = PR.G; T = PR.T
    v = {"curve": c.name}
    # pairings
    pairs = []
    for i in range(3):
        P = G.g1_mul(c.g1, rnd.randrange(1, c.r)); Q = G.g2_mul(c.g2, rnd.randrange(1, c.r))
        m = PR.miller(P, Q)
        pairs.append({"g1": G.g1_bytes(P).hex(), "g2": G.g2_bytes(Q).hex(), "miller": PR.gt_bytes(m).hex(), "gt": PR.gt_bytes(PR.final_exp(m)).hex()})
    pairs.append({"g1": G.g1_bytes(c.g1).hex(), "g2": G.g2_bytes(c.g2).hex(), "miller": PR.gt_bytes(PR.miller(c.g1, c.g2)).hex(), "gt": PR.gt_bytes(PR.pair(c.g1, c.g2)).hex()})
    one = PR.gt_bytes(T.F12_ONE).hex()
    pairs.append({"g1": G.g1_bytes(None).hex(), "g2": G.g2_bytes(c.g2).hex(), "miller": one, "gt": one})
    pairs.append({"g1": G.g1_bytes(c.g1).hex(), "g2": G.g2_bytes(None).hex(), "miller": one, "gt": one})
    v["pairings"] = pairs
    Ps = [G.g1_mul(c.g1, rnd.randrange(1, c.r)) for _ in range(5)]
    Qs = [G.g2_mul(c.g2, rnd.randrange(1, c.r)) for _ in range(5)]
    v["pairing_product"] = {"g1s": [G.g1_bytes(P).hex() for P in Ps], "g2s": [G.g2_bytes(Q).hex() for Q in Qs],
                            "gt": PR.gt_bytes(PR.pairing_product(Ps, Qs)).hex()}
    # group sums (with a repeated point and an inverse pair) and scalar multiples
    pts1 = Ps + [Ps[0], G.g1_neg(Ps[1])]
    pts2 = Qs + [Qs[0], G.g2_neg(Qs[1])]
    v["sum_g1"] = {"pts": [G.g1_bytes(P).hex() for P in pts1], "sum": G.g1_bytes(G.g1_sum(pts1)).hex()}
    v["sum_g2"] = {"pts": [G.g2_bytes(P).hex() for P in pts2], "sum": G.g2_bytes(G.g2_sum(pts2)).hex()}
    ks = [0, 1, 2, c.r - 1, c.r, rnd.randrange(c.r), -rnd.randrange(c.r), 2**64 + 12345]
    v["scale_g1"] = [{"pt": G.g1_bytes(Ps[0]).hex(), "k": str(k), "out": G.g1_bytes(G.g1_mul(Ps[0], k)).hex()} for k in ks]
    v["scale_g2"] = [{"pt": G.g2_bytes(Qs[0]).hex(), "k": str(k), "out": G.g2_bytes(G.g2_mul(Qs[0], k)).hex()} for k in ks]
    # extra hash-to-G1 messages: block-boundary lengths of Keccak (136) / BLAKE2b (128)
    hs = []
    for ln in (0, 1, 31, 32, 33, 64, 123, 124, 127, 128, 134, 135, 136, 137, 200, 271, 272, 400):
        m = rnd.randbytes(ln)
        hs.append({"msg": m.hex(), "point": G.g1_bytes(h2c.hash_to_g1(c, m)).hex()})
    v["h2c"] = hs
    # end-to-end aggregate cases (bgls/bgls_test.go:40-77)
    cases = []
    for n in (1, 2, 6):
        sks = [rnd.randrange(1, c.r) for _ in range(n + 1)]
        msgs = [rnd.randbytes(32) for _ in range(n)]
        keys = [scheme.load_public_key(c, s) for s in sks]
        sigs = [scheme.sign(c, s, m) for s, m in zip(sks, msgs)]
        agg = G.g1_sum(sigs)
        kb = [G.g2_bytes(k).hex() for k in keys]
        mh = [m.hex() for m in msgs]
        cases.append({"name": "valid_n%d" % n, "sig": G.g1_bytes(agg).hex(), "keys": kb[:n], "msgs": mh, "allow_dups": False, "expect": True})
        if n > 1:
            cases.append({"name": "missing_key_n%d" % n, "sig": G.g1_bytes(agg).hex(), "keys": kb[:n - 1], "msgs": mh, "allow_dups": False, "expect": False})
            sw = [mh[1], mh[0]] + mh[2:]
            cases.append({"name": "swapped_msgs_n%d" % n, "sig": G.g1_bytes(agg).hex(), "keys": kb[:n], "msgs": sw, "allow_dups": False, "expect": False})
            # duplicate message: extra signer signs msgs[0]
            sdup = G.g1_add(agg, scheme.sign(c, sks[n], msgs[0]))
            cases.append({"name": "duplicate_msg_n%d" % n, "sig": G.g1_bytes(sdup).hex(), "keys": kb, "msgs": mh + [mh[0]], "allow_dups": False, "expect": False})
            cases.append({"name": "duplicate_msg_allowed_n%d" % n, "sig": G.g1_bytes(sdup).hex(), "keys": kb, "msgs": mh + [mh[0]], "allow_dups": True, "expect": True})
            cases.append({"name": "wrong_sig_n%d" % n, "sig": G.g1_bytes(sdup).hex(), "keys": kb[:n], "msgs": mh, "allow_dups": False, "expect": False})
        cases.append({"name": "tampered_msg_n%d" % n, "sig": G.g1_bytes(agg).hex(), "keys": kb[:n], "msgs": mh[:-1] + [(msgs[-1] + b"!").hex()], "allow_dups": False, "expect": False})
    for c_ in cases:
        got = scheme.verify_agg(c, G.g1_from_bytes(bytes.fromhex(c_["sig"])), [G.g2_from_bytes(bytes.fromhex(k)) for k in c_["keys"]],
                                [bytes.fromhex(m) for m in c_["msgs"]], c_["allow_dups"])
        assert got == c_["expect"], c_["name"]
    v["aggregate_cases"] = cases
    # multisig cases (bgls/blsKosk_test.go:35-64); msg already carries the Kosk 0x01 prefix
    mc = []
    for n in (1, 2, 8):
        sks = [rnd.randrange(1, c.r) for _ in range(n)]
        msg = b"\x01" + rnd.randbytes(32)
        keys = [scheme.load_public_key(c, s) for s in sks]
        sig = G.g1_sum([scheme.sign(c, s, msg) for s in sks])
        kb = [G.g2_bytes(k).hex() for k in keys]
        mc.append({"name": "valid_n%d" % n, "sig": G.g1_bytes(sig).hex(), "keys": kb, "msg": msg.hex(), "expect": True})
        mc.append({"name": "wrong_msg_n%d" % n, "sig": G.g1_bytes(sig).hex(), "keys": kb, "msg": (msg + b"x").hex(), "expect": False})
        other = G.g2_bytes(scheme.load_public_key(c, rnd.randrange(1, c.r))).hex()
        mc.append({"name": "wrong_signer_n%d" % n, "sig": G.g1_bytes(sig).hex(), "keys": [other] + kb[1:], "msg": msg.hex(), "expect": False})
        mc.append({"name": "preaggregated_key_n%d" % n, "sig": G.g1_bytes(sig).hex(), "keys": [G.g2_bytes(G.g2_sum(keys)).hex()], "msg": msg.hex(), "expect": True})
    for c_ in mc:
        got = scheme.verify_multi_signature(c, G.g1_from_bytes(bytes.fromhex(c_["sig"])), [G.g2_from_bytes(bytes.fromhex(k)) for k in c_["keys"]], bytes.fromhex(c_["msg"]))
        assert got == c_["expect"], c_["name"]
    v["multi_cases"] = mc
    json.dump(v, open(os.path.join(HERE, "vectors_%s.json" % c.name), "w"), indent=0)


def hae(c, seed):
    """hae_<curve>.json: BLAKE2Xb vectors, hashPubKeysToExponents, and accept/reject cases mirroring
    bgls/blsHAE_test.go:14-82 and bgls/blsKosk_test.go:66-94, all from the Python oracle."""
    from oracle.pyref import hashes
    rnd = random.Random(seed)
    PR = Pairing(c); G = PR.G
    v = {"curve": c.name, "xof": []}
    for ln, ol in ((0, 16), (1, 1), (64, 64), (128, 65), (129, 160), (1000, 333)):
        d = rnd.randbytes(ln)
        v["xof"].append({"in": d.hex(), "out_len": ol, "out": hashes.blake2xb(d, ol).hex()})
    n = 5
    sks = [rnd.randrange(1, c.r) for _ in range(n + 1)]
    keys = [scheme.load_public_key(c, sk) for sk in sks]
    kb = lambda ks: [G.g2_bytes(k).hex() for k in ks]
    v["exponents"] = {"keys": kb(keys[:n]), "t": ["%032x" % t for t in scheme.hash_pubkeys_to_exponents(c, keys[:n])]}
    # multi-signature with HAE (TestMultiSigWithHAE)
    msg = rnd.randbytes(32)
    sigs = [scheme.sign(c, sk, msg) for sk in sks[:n]]
    agg = scheme.aggregate_signatures_hae(c, sigs, keys[:n])
    v["aggregate_signatures"] = {"sigs": [G.g1_bytes(s).hex() for s in sigs], "keys": kb(keys[:n]), "out": G.g1_bytes(agg).hex()}
    mc = [{"name": "valid", "sig": G.g1_bytes(agg).hex(), "keys": kb(keys[:n]), "msg": msg.hex(), "expect": True},
          {"name": "wrong_msg", "sig": G.g1_bytes(agg).hex(), "keys": kb(keys[:n]), "msg": rnd.randbytes(32).hex(), "expect": False},
          {"name": "wrong_signer", "sig": G.g1_bytes(agg).hex(), "keys": kb([keys[n]] + keys[1:n]), "msg": msg.hex(), "expect": False},
          {"name": "plain_aggregate_rejected", "sig": G.g1_bytes(G.g1_sum(sigs)).hex(), "keys": kb(keys[:n]), "msg": msg.hex(), "expect": False}]
    for c_ in mc:
        got = scheme.verify_multi_signature_hae(c, G.g1_from_bytes(bytes.fromhex(c_["sig"])), [G.g2_from_bytes(bytes.fromhex(k)) for k in c_["keys"]], bytes.fromhex(c_["msg"]))
        assert got == c_["expect"], c_["name"]
    v["multi_cases"] = mc
    # aggregate over distinct (and duplicated) messages with HAE (TestAggregationWithHAE)
    n2 = 4
    msgs = [rnd.randbytes(32) for _ in range(n2)]
    msgs.append(msgs[0])                                    # duplicate allowed under HAE
    sg = [scheme.sign(c, sks[i], msgs[i]) for i in range(n2 + 1)]
    a4 = scheme.aggregate_signatures_hae(c, sg[:n2], keys[:n2])
    a5 = scheme.aggregate_signatures_hae(c, sg, keys[:n2 + 1])
    ac = [{"name": "valid", "sig": G.g1_bytes(a4).hex(), "keys": kb(keys[:n2]), "msgs": [m.hex() for m in msgs[:n2]], "expect": True},
          {"name": "missing_key", "sig": G.g1_bytes(a4).hex(), "keys": kb(keys[:n2 - 1]), "msgs": [m.hex() for m in msgs[:n2]], "expect": False},
          {"name": "duplicate_message_ok", "sig": G.g1_bytes(a5).hex(), "keys": kb(keys[:n2 + 1]), "msgs": [m.hex() for m in msgs], "expect": True},
          {"name": "stale_signature", "sig": G.g1_bytes(a5).hex(), "keys": kb(keys[:n2]), "msgs": [m.hex() for m in msgs[:n2]], "expect": False},
          {"name": "swapped_messages", "sig": G.g1_bytes(a4).hex(), "keys": kb(keys[:n2]), "msgs": [m.hex() for m in [msgs[1], msgs[0]] + msgs[2:n2]], "expect": False}]
    for c_ in ac:
        got = scheme.verify_aggregate_signature_hae(c, G.g1_from_bytes(bytes.fromhex(c_["sig"])), [G.g2_from_bytes(bytes.fromhex(k)) for k in c_["keys"]], [bytes.fromhex(m) for m in c_["msgs"]])
        assert got == c_["expect"], c_["name"]
    v["aggregate_cases"] = ac
    # multiplicities (TestKoskMultiSigWithMultiplicity shape; negative and zero factors exercise curve.go:190-214)
    mult = [3, 1, -2, 0, 7]
    msg = rnd.randbytes(32)
    ks = [scheme.kosk_sign(c, sk, msg) for sk in sks[:n]]
    aggm = G.g1_sum([G.g1_mul(G.g1_neg(s), -m) if m < 0 else G.g1_mul(s, m) for s, m in zip(ks, mult)])
    pc = [{"name": "valid", "sig": G.g1_bytes(aggm).hex(), "keys": kb(keys[:n]), "mult": mult, "msg": msg.hex(), "expect": True},
          {"name": "wrong_multiplicity", "sig": G.g1_bytes(aggm).hex(), "keys": kb(keys[:n]), "mult": [3, 1, 2, 0, 7], "msg": msg.hex(), "expect": False},
          {"name": "length_mismatch", "sig": G.g1_bytes(aggm).hex(), "keys": kb(keys[:n]), "mult": mult[:n - 1], "msg": msg.hex(), "expect": False},
          {"name": "nil_multiplicity_is_plain_kosk", "sig": G.g1_bytes(G.g1_sum(ks)).hex(), "keys": kb(keys[:n]), "mult": None, "msg": msg.hex(), "expect": True}]
    for c_ in pc:
        got = scheme.kosk_verify_multi_signature_with_multiplicity(c, G.g1_from_bytes(bytes.fromhex(c_["sig"])), [G.g2_from_bytes(bytes.fromhex(k)) for k in c_["keys"]], c_["mult"], bytes.fromhex(c_["msg"]))
        assert got == c_["expect"], c_["name"]
    v["multiplicity_cases"] = pc
    json.dump(v, open(os.path.join(HERE, "hae_%s.json" % c.name), "w"), indent=0)


def wire_vectors(seed):
    """wire_altbn128.json: compressed forms (curves/altbn128.go:81-89,203-221) and Unmarshal decisions (:296-376) from the
    Python oracle (oracle/pyref/wire.py): valid points with all sign-bit patterns, infinity, random x (about half have no
    point), inconsistent G2 sign bits, non-canonical x."""
    from oracle.pyref import wire
    c = BN254
    rnd = random.Random(seed)
    G = Pairing(c).G
    v = {"g1": [], "g2": [], "g1_decode": [], "g2_decode": []}
    for i in range(8):
        P = G.g1_mul(c.g1, rnd.randrange(1, c.r)); Q = G.g2_mul(c.g2, rnd.randrange(1, c.r))
        v["g1"].append({"pt": G.g1_bytes(P).hex(), "compressed": wire.compress_g1(P).hex()})
        v["g2"].append({"pt": G.g2_bytes(Q).hex(), "compressed": wire.compress_g2(Q).hex()})
    v["g1"].append({"pt": G.g1_bytes(None).hex(), "compressed": wire.compress_g1(None).hex()})
    v["g2"].append({"pt": G.g2_bytes(None).hex(), "compressed": wire.compress_g2(None).hex()})
    def dec1(d):
        pt, ok = wire.decompress_g1(d)
        return {"in": d.hex(), "ok": ok, "pt": G.g1_bytes(pt).hex() if ok else None}
    def dec2(d):
        # ok: UnmarshalG2's answer (subgroup membership included); decoded / pt: the decoding before that last test
        pt, dec = wire.decompress_g2(d, subgroup=False)
        _, ok = wire.decompress_g2(d)
        return {"in": d.hex(), "ok": ok, "decoded": dec, "pt": G.g2_bytes(pt).hex() if dec else None}
    for row in v["g1"]:
        d = bytearray(bytes.fromhex(row["compressed"]))
        v["g1_decode"].append(dec1(bytes(d)))
        d[0] ^= 128
        v["g1_decode"].append(dec1(bytes(d)))
    for row in v["g2"]:
        d = bytearray(bytes.fromhex(row["compressed"]))
        for m0, m1 in ((0, 0), (128, 0), (0, 128), (128, 128)):
            e = bytearray(d); e[0] ^= m0; e[32] ^= m1
            v["g2_decode"].append(dec2(bytes(e)))
    for i in range(24):
        d = bytearray(rnd.randbytes(32)); d[0] &= rnd.choice((0x3f, 0xbf, 0xff))
        v["g1_decode"].append(dec1(bytes(d)))
    for i in range(12):
        d = bytearray(rnd.randbytes(64)); d[0] &= rnd.choice((0x3f, 0xbf)); d[32] &= rnd.choice((0x3f, 0xbf))
        v["g2_decode"].append(dec2(bytes(d)))
    v["g1_decode"].append(dec1((c.p + 5).to_bytes(32, "big")))
    v["g2_decode"].append(dec2((c.p + 5).to_bytes(32, "big") + (7).to_bytes(32, "big")))
    assert any(r["ok"] for r in v["g1_decode"][18:]) and any(not r["ok"] for r in v["g1_decode"][18:])
    json.dump(v, open(os.path.join(HERE, "wire_altbn128.json"), "w"), indent=0)


def wire_vectors_bls(seed):
    """wire_bls12.json: BLS12-381 compressed forms in the ebfull/pairing layout the reference names as its target
    (curves/bls12_381.go:54-62,115-123 "TODO Make this match ebfull/pairing marshalling") and the UnmarshalG1 / UnmarshalG2
    decisions on 48 / 96 bytes (:242-264), from oracle/pyref/wire.py: valid points with both sort flags, infinity, flag
    violations, non-canonical x, random x (about half have no point), points on the curve / twist outside the order-r
    subgroup (tests/golden/subgroup_bls12.json).  The generators' encodings are the format's public known-answer values."""
    from oracle.pyref import wire
    c = BLS381
    rnd = random.Random(seed)
    G = Pairing(c).G
    v = {"layout": "ebfull/pairing (ZCash): bit 7 compressed, bit 6 infinity, bit 5 y lexicographically larger; G2 = x.c1 || x.c0",
         "g1": [], "g2": [], "g1_decode": [], "g2_decode": []}
    v["g1"].append({"pt": G.g1_bytes(c.g1).hex(), "compressed": wire.bls_compress_g1(c.g1).hex(), "note": "generator"})
    v["g2"].append({"pt": G.g2_bytes(c.g2).hex(), "compressed": wire.bls_compress_g2(c.g2).hex(), "note": "generator"})
    for i in range(7):
        P = G.g1_mul(c.g1, rnd.randrange(1, c.r)); Q = G.g2_mul(c.g2, rnd.randrange(1, c.r))
        if i & 1:
            P, Q = G.g1_neg(P), G.g2_neg(Q)
        v["g1"].append({"pt": G.g1_bytes(P).hex(), "compressed": wire.bls_compress_g1(P).hex()})
        v["g2"].append({"pt": G.g2_bytes(Q).hex(), "compressed": wire.bls_compress_g2(Q).hex()})
    v["g1"].append({"pt": G.g1_bytes(None).hex(), "compressed": wire.bls_compress_g1(None).hex(), "note": "infinity"})
    v["g2"].append({"pt": G.g2_bytes(None).hex(), "compressed": wire.bls_compress_g2(None).hex(), "note": "infinity"})
    def dec(group, d, note=None):
        # ok: Unmarshal's answer (Check() included); decoded / pt: the decoding before the subgroup test
        f = wire.bls_decompress_g1 if group == 1 else wire.bls_decompress_g2
        pt, d_ok = f(d, subgroup=False)
        _, ok = f(d)
        row = {"in": d.hex(), "ok": ok, "decoded": d_ok, "pt": (G.g1_bytes(pt) if group == 1 else G.g2_bytes(pt)).hex() if d_ok else None}
        if note:
            row["note"] = note
        return row
    for group, key in ((1, "g1"), (2, "g2")):
        out = v[key + "_decode"]
        for row in v[key]:
            d = bytearray(bytes.fromhex(row["compressed"]))
            out.append(dec(group, bytes(d)))
            e = bytearray(d); e[0] ^= 0x20
            out.append(dec(group, bytes(e), "sort flag flipped"))           # the other root (or: infinity with the sort flag set)
            e = bytearray(d); e[0] &= 0x7F
            out.append(dec(group, bytes(e), "compression flag clear"))
        inf = bytearray(bytes.fromhex(v[key][-1]["compressed"])); inf[-1] = 1
        out.append(dec(group, bytes(inf), "infinity flag with a non-zero bit"))
        n = 48 * group
        for i in range(16 // group):
            d = bytearray(rnd.randbytes(n))
            d[0] = 0x80 | rnd.choice((0, 0x20)) | (d[0] & 0x0F)              # x below p (whose top byte is 0x1a)
            out.append(dec(group, bytes(d), "random x"))
        big = bytearray((c.p + 5).to_bytes(48, "big")) + (bytearray((3).to_bytes(48, "big")) if group == 2 else bytearray())
        big[0] |= 0x80
        out.append(dec(group, bytes(big), "x >= p"))
    # points on the curve / twist outside the order-r subgroup: decodable, refused by Check()
    sub = json.load(open(os.path.join(HERE, "subgroup_bls12.json")))
    for row in sub["g1_points"]:
        P = G.g1_from_bytes(bytes.fromhex(row["pt"]))
        if P is not None and not row["in_subgroup"]:
            v["g1_decode"].append(dec(1, wire.bls_compress_g1(P), "cofactor point: " + row.get("note", "")))
    for row in sub["points"]:
        Q = G.g2_from_bytes(bytes.fromhex(row["pt"]))
        if Q is not None and row["on_twist"] and not row["in_subgroup"]:
            v["g2_decode"].append(dec(2, wire.bls_compress_g2(Q), "cofactor point: " + row.get("note", "")))
    for key in ("g1_decode", "g2_decode"):
        rows = v[key]
        assert any(r["ok"] for r in rows) and any(r["decoded"] and not r["ok"] for r in rows) and any(not r["decoded"] for r in rows)
    json.dump(v, open(os.path.join(HERE, "wire_bls12.json"), "w"), indent=0)


if __name__ == "__main__":
    if "--wire-only" in sys.argv:
        wire_vectors(20261002); wire_vectors_bls(20261003); print("wire fixtures written"); sys.exit(0)
    if "--wire-bls-only" in sys.argv:
        wire_vectors_bls(20261003); print("BLS12-381 wire fixture written"); sys.exit(0)
    if "--hae-only" in sys.argv:
        hae(BN254, 20260930); hae(BLS381, 20261001)
        print("hae fixtures written"); sys.exit(0)
    kat()
    vectors(BN254, 20260928)
    vectors(BLS381, 20260929)
    hae(BN254, 20260930)
    hae(BLS381, 20261001)
    wire_vectors(20261002)
    wire_vectors_bls(20261003)
    print("golden fixtures written")
