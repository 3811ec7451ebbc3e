#!/usr/bin/env python3
"""Readiness check for a multi-GPU node: ONE process, one key set sharded over devices 0..N-1 inside the C ABI
(bgls_keys_upload with a device list), for N = 1, 2, 4, 8 (as many as the node has):

  * identical GT bytes for every N (the product of the per-device partial Miller products is independent of the cut),
  * the exchange of the partials really is RCCL (bgls_last_exchange() == 2) as soon as the devices are distinct,
  * a tampered message is rejected on every N, a duplicate straddling two shards is found.

usage: python tools/scale_selfcheck.py [--curve altbn128|bls12] [--n SIGNERS]      (exit code 0 = all good)
On a one-GPU box only N = 1 runs with distinct devices; pass --share to list device 0 for every shard (peer-copy exchange)."""
import argparse
import ctypes
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bgls_amd import _lib  # noqa: E402


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def device_count():
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def run(curve="altbn128", n=4096, share=False, shards=(1, 2, 4, 8)):
    lib = _lib.load()
    cid = 0 if curve == "altbn128" else 1
    fp = 32 if cid == 0 else 48
    ndev = device_count()
    if ndev < 1 or lib.bgls_init(0) != 0:
        raise SystemExit("no usable GPU")
    order = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
             1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}[cid]
    rnd = random.Random(0x5CA1E)
    sks = [rnd.randrange(1, order) for _ in range(n)]
    kb = B(b"".join(s.to_bytes(32, "big") for s in sks))
    keys = (ctypes.c_uint8 * (n * 4 * fp))()
    assert lib.bgls_scale_generator(cid, 2, kb, n, keys) == 0
    msgs = [rnd.randbytes(64) for _ in range(n)]
    blob = b"".join(msgs)
    off = (ctypes.c_uint64 * (n + 1))(*range(0, 64 * (n + 1), 64))
    sigs = (ctypes.c_uint8 * (n * 2 * fp))()
    assert lib.bgls_sign_batch(cid, kb, B(blob), off, n, sigs) == 0
    agg = (ctypes.c_uint8 * (2 * fp))()
    assert lib.bgls_aggregate_points(cid, 1, sigs, n, agg) == 0
    # a signature that does NOT verify, so that the GT bytes are not just the identity
    other = (ctypes.c_uint8 * (2 * fp)).from_buffer_copy(bytes(sigs)[:2 * fp])
    report, ref = [], None
    for s in shards:
        if s > 1 and not share and s > ndev:
            report.append({"shards": s, "skipped": "only %d device(s)" % ndev})
            continue
        devs = (ctypes.c_int * s)(*([0] * s if share else list(range(s))))
        h = ctypes.c_uint64()
        rc = lib.bgls_keys_upload(cid, keys, n, devs, s, 1, ctypes.byref(h))
        assert rc == 0, (rc, _lib.last_error())
        ok = lib.bgls_verify_aggregate_h(h, agg, B(blob), off, n, 0)
        exch = lib.bgls_last_exchange()
        bad = bytearray(blob); bad[64 * (n // 2) + 3] ^= 0x20
        rej = lib.bgls_verify_aggregate_h(h, agg, B(bytes(bad)), off, n, 0)
        dup = bytearray(blob); dup[64 * (n - 1):64 * n] = blob[:64]          # first and last message equal: straddles every cut
        dupv = lib.bgls_verify_aggregate_h(h, agg, B(bytes(dup)), off, n, 0)
        gt = (ctypes.c_uint8 * (12 * fp))()
        lib.bgls_verify_aggregate_h_gt(h, other, B(blob), off, n, 0, gt)
        lib.bgls_keys_free(h)
        if ref is None:
            ref = bytes(gt)
        distinct = s > 1 and not share
        row = {"shards": s, "valid": ok, "tampered": rej, "duplicate": dupv, "exchange": exch, "same_gt_bytes": bytes(gt) == ref,
               "rccl_expected": distinct}
        row["ok"] = ok == 1 and rej == 0 and dupv == 0 and row["same_gt_bytes"] and (exch == 2 if distinct else True)
        report.append(row)
    return {"curve": curve, "signers": n, "devices": ndev, "rccl_available": lib.bgls_rccl_available(), "runs": report,
            "ok": all(r.get("ok", True) for r in report)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--curve", default="altbn128", choices=["altbn128", "bls12"])
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--share", action="store_true", help="every shard on device 0 (one-GPU boxes: exercises the cut, not RCCL)")
    a = ap.parse_args()
    res = run(a.curve, a.n, a.share)
    print(json.dumps(res))
    sys.exit(0 if res["ok"] else 1)
