import ctypes, os, sys, time, random
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from bgls_amd import _lib
import bench
lib = _lib.load(); assert lib.bgls_init(0) == 0
cid, fp, n = 0, 32, 1 << 20
rnd = random.Random(1)
sks = [rnd.randrange(1, bench.ORDER[cid]) for _ in range(n)]
keys = (ctypes.c_uint8 * (n * 128))()
assert lib.bgls_scale_generator(cid, 2, bench.B(b"".join(s.to_bytes(32, "big") for s in sks)), n, keys) == 0
msg = b"\x01" + rnd.randbytes(64)
off = (ctypes.c_uint64 * 2)(0, len(msg))
h = (ctypes.c_uint8 * 64)(); lib.bgls_hash_to_g1(cid, bench.B(msg), off, 1, h)
sig = (ctypes.c_uint8 * 64)()
lib.bgls_scale_points(cid, 1, h, bench.B((sum(sks) % bench.ORDER[cid]).to_bytes(32, "big")), None, 1, sig)
dev = torch.device("cuda:0")
t_keys = torch.frombuffer(bytearray(bytes(keys)), dtype=torch.uint8).to(dev)
t_sig = torch.frombuffer(bytearray(bytes(sig)), dtype=torch.uint8).to(dev)
t_msg = torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev)
L = 16
lanes = [torch.cuda.Stream(device=dev) for _ in range(L)]
for k in range(L):
    lib.bgls_select_context(k)
    assert lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_msg.data_ptr(), len(msg), lanes[k].cuda_stream) == 1
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for k in range(L):
    lib.bgls_select_context(k)
    a = time.perf_counter()
    lib.bgls_verify_multi_submit_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_msg.data_ptr(), len(msg), lanes[k].cuda_stream)
    ts.append(time.perf_counter() - a)
t1 = time.perf_counter()
for k in range(L):
    lib.bgls_select_context(k)
    assert lib.bgls_final_verify_collect(cid) == 1
t2 = time.perf_counter()
print("submit times ms:", [round(x * 1e3, 3) for x in ts])
print("all submits %.3f ms, all collected %.3f ms -> %.3f ms per verification" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / L))
