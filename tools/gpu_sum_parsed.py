#!/usr/bin/env python3
"""Development: key-sum main pass on a resident (parsed) key set vs wire bytes, stage timers only."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import bench
from bgls_amd import _lib
lib = _lib.load()
bench.check(lib.bgls_init(0), "init")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
cid, fp = 0, 32
inst = bench.make_instance(lib, cid, n, 0xB6150000 + 4)
keys = inst["keys"][:n * 4 * fp]
h = ctypes.c_uint64()
t0 = time.perf_counter()
bench.check(lib.bgls_keys_upload(cid, bench.B(keys), n, None, 1, 0, ctypes.byref(h)), "upload")
print("upload s", time.perf_counter() - t0)
import random
rnd = random.Random(0xB6150000 + 4)
msg = b"\x01" + rnd.randbytes(64)
off = (ctypes.c_uint64 * 2)(0, len(msg))
hh = (ctypes.c_uint8 * (2 * fp))()
bench.check(lib.bgls_hash_to_g1(cid, bench.B(msg), off, 1, hh), "h2c")
sig = (ctypes.c_uint8 * (2 * fp))()
bench.check(lib.bgls_scale_points(cid, 1, hh, bench.B((sum(inst["sks"][:n]) % bench.ORDER[cid]).to_bytes(32, "big")), None, 1, sig), "sig")
for name, call in (("handle (parsed keys)", lambda: lib.bgls_verify_multi_h(h.value, sig, bench.B(msg), len(msg))),
                   ("wire bytes (host keys: includes H2D)", None)):
    if call is None:
        continue
    assert call() == 1
    lib.bgls_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(5):
        assert call() == 1
    dt = (time.perf_counter() - t0) / 5
    print(name, "ms/call", dt * 1e3, {s: bench.stage_ms_per_call(bench.stage(lib, s)[0], 5) for s in ("sum_points", "sum_main", "miller", "final_exp")})
