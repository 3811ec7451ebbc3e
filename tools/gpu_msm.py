#!/usr/bin/env python3
"""Timing of the windowed / bucketed scalar multiplications on the GPU box (development tool, not part of bench.py):
weighted sums by the bucket method vs one double-and-add per point, fixed-base key generation, HAE aggregate verify.

    python tools/gpu_msm.py [--sizes 4096,65536,1048576] [--curve altbn128]
"""
import argparse
import ctypes
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bgls_amd import _lib  # noqa: E402

ORDER = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
         1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}
SIZE_MAX = ctypes.c_size_t(-1).value


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b))


def timed(f, reps=3):
    f()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,65536,1048576")
    ap.add_argument("--curve", default="altbn128")
    a = ap.parse_args()
    lib = _lib.load()
    cid = {"altbn128": 0, "bls12": 1}[a.curve]
    fp = 32 if cid == 0 else 48
    dev = torch.device("cuda:0")
    rnd = random.Random(1)
    for n in [int(x) for x in a.sizes.split(",")]:
        sks = B(rnd.randbytes(32 * n))
        for group in (1, 2):
            size = (2 if group == 1 else 4) * fp
            o = (ctypes.c_uint8 * (n * size))()
            ms = timed(lambda: lib.bgls_scale_generator(cid, group, sks, n, o))
            print("scale_generator %s g%d n=%d: %.2f ms (host buffers) = %.2f M points/s" % (a.curve, group, n, ms, n / ms / 1e3), flush=True)
            t_p = torch.frombuffer(bytearray(bytes(o)), dtype=torch.uint8).to(dev)
            t_w = torch.frombuffer(bytearray(rnd.randbytes(16 * n)), dtype=torch.uint8).to(dev)
            t_o = torch.zeros(size, dtype=torch.uint8, device=dev)
            res = {}
            for name, m in (("buckets", 0), ("per-point", SIZE_MAX)):
                if m and n > (1 << 18):
                    continue
                lib.bgls_set_msm_min(m)
                ms = timed(lambda: lib.bgls_weighted_sum_dev(cid, group, t_p.data_ptr(), t_w.data_ptr(), n, t_o.data_ptr(), None))
                res[name] = bytes(t_o.cpu().numpy())
                print("weighted_sum %s g%d n=%d %-9s: %.2f ms = %.2f M points/s" % (a.curve, group, n, name, ms, n / ms / 1e3), flush=True)
            lib.bgls_set_msm_min(32)
            if len(res) == 2:
                assert res["buckets"] == res["per-point"]


if __name__ == "__main__":
    main()
