#!/usr/bin/env python3
"""Development: time the Miller stage of one aggregate verification per shape (bgls_set_miller_shape) and check that
every shape gives the same partial-product bytes.  usage: python tools/gpu_shapes.py [altbn128|bls12] [n] [shape:ng ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from bgls_amd import _lib
import bench

curve = sys.argv[1] if len(sys.argv) > 1 else "altbn128"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
shapes = [tuple(map(int, a.split(":"))) for a in sys.argv[3:]] or [(0, 6), (1, 6), (2, 6), (3, 6), (2, 12), (3, 12), (3, 24)]
cid = 0 if curve == "altbn128" else 1
fp = 32 if cid == 0 else 48
lib = _lib.load()
assert lib.bgls_init(0) == 0
keys, msgs, sig, _ = bench.make_shard(lib, cid, n, 1234)
dev = torch.device("cuda:0")
t_keys = torch.frombuffer(bytearray(keys), dtype=torch.uint8).to(dev)
t_msgs = torch.frombuffer(bytearray(msgs), dtype=torch.uint8).to(dev)
t_sig = torch.frombuffer(bytearray(sig), dtype=torch.uint8).to(dev)
gtb = 12 * fp
ref = None
for shape, ng in shapes:
    assert lib.bgls_set_miller_shape(shape, ng) == 0
    part = torch.zeros(gtb, dtype=torch.uint8, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    def run():
        rc = lib.bgls_miller_product_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), t_msgs.data_ptr(), 64, 64, n, 1, part.data_ptr(), flags.data_ptr(), None)
        assert rc == 0, (rc, _lib.last_error())
    run(); torch.cuda.synchronize()
    v = lib.bgls_final_verify_dev(cid, part.data_ptr(), 1, flags.data_ptr(), None)
    b = bytes(part.cpu().numpy())
    if ref is None:
        ref = b
    lib.bgls_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    lib.bgls_final_verify_dev(cid, part.data_ptr(), 1, flags.data_ptr(), None)     # collects the stage events
    lib.bgls_select_context(0)
    ms = {}
    for st in ("h2c", "miller", "reduce"):
        a, c = ctypes.c_double(), ctypes.c_ulonglong()
        lib.bgls_profile_get(st.encode(), ctypes.byref(a), ctypes.byref(c))
        ms[st] = round(a.value / max(c.value, 1), 3)
    lib.bgls_profile_enable(0)
    print("shape %d ng %2d  verdict %d  same_bytes %s  total %.3f ms  stages %s" % (shape, ng, v, b == ref, dt * 1e3, ms), flush=True)
lib.bgls_set_miller_shape(0, 6)
