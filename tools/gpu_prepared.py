#!/usr/bin/env python3
"""Development: time bgls_verify_aggregate_h on a prepared vs an unprepared key set.  usage: gpu_prepared.py [curve] [n]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from bgls_amd import _lib
import bench
curve = sys.argv[1] if len(sys.argv) > 1 else "altbn128"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
cid = 0 if curve == "altbn128" else 1
lib = _lib.load(); assert lib.bgls_init(0) == 0
inst = bench.make_instance(lib, cid, n, 77)
agg = bench.aggregate_sig(lib, inst, 0, n)
off = (ctypes.c_uint64 * (n + 1))(*range(0, 64 * (n + 1), 64))
for flags in (0, 2):
    h = ctypes.c_uint64()
    t0 = time.perf_counter()
    rc = lib.bgls_keys_upload(cid, bench.B(inst["keys"]), n, None, 1, flags, ctypes.byref(h))
    assert rc == 0, (rc, _lib.last_error())
    t_up = time.perf_counter() - t0
    mb, sb = bench.B(inst["msgs"]), bench.B(agg)
    assert lib.bgls_verify_aggregate_h(h, sb, mb, off, n, 0) == 1
    lib.bgls_profile_enable(1)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        assert lib.bgls_verify_aggregate_h(h, sb, mb, off, n, 0) == 1
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = {}
    for st in ("h2c", "miller", "reduce", "final_exp"):
        a, c = ctypes.c_double(), ctypes.c_ulonglong()
        lib.bgls_profile_get(st.encode(), ctypes.byref(a), ctypes.byref(c))
        ms[st] = round(a.value / max(c.value, 1), 3)
    lib.bgls_profile_enable(0)
    print("%s n=%d flags=%d upload %.1f ms  verify (host buffers) min %.3f ms  stages %s" % (curve, n, flags, t_up * 1e3, min(ts), ms), flush=True)
    lib.bgls_keys_free(h)
