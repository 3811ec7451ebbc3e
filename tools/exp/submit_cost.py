"""host cost of one bgls_verify_multi_submit_dev / collect pair; pipelined throughput from 1 / 2 / 4 host threads (development: is config 4 host-bound?)"""
import ctypes, os, sys, time, random, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from bench import B, check, ORDER
from bgls_amd import _lib
lib = _lib.load(); assert lib.bgls_init(0) == 0
dev = torch.device("cuda:0")
n = 1 << 20
cid, fp = 0, 32
rnd = random.Random(5)
sks = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
kb = B(b"".join(s.to_bytes(32, "big") for s in sks))
keys = (ctypes.c_uint8 * (n * 4 * fp))()
check(lib.bgls_scale_generator(cid, 2, kb, n, keys), "scale_generator")
msg = b"\x01" + rnd.randbytes(64)
off = (ctypes.c_uint64 * 2)(0, len(msg))
h = (ctypes.c_uint8 * (2 * fp))()
check(lib.bgls_hash_to_g1(cid, B(msg), off, 1, h), "hash")
sig = (ctypes.c_uint8 * (2 * fp))()
check(lib.bgls_scale_points(cid, 1, h, B((sum(sks) % ORDER[cid]).to_bytes(32, "big")), None, 1, sig), "scale")
t_keys = torch.frombuffer(bytearray(bytes(keys)), dtype=torch.uint8).to(dev)
t_sig = torch.frombuffer(bytearray(bytes(sig)), dtype=torch.uint8).to(dev)
t_msg = torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev)
L = 16
streams = [torch.cuda.Stream(device=dev) for _ in range(L)]
check(lib.bgls_set_throughput_mode(1), "tm")

def submit(k):
    check(lib.bgls_select_context(k), "sel")
    check(lib.bgls_verify_multi_submit_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_msg.data_ptr(), len(msg), streams[k].cuda_stream), "submit")

def collect(k):
    check(lib.bgls_select_context(k), "sel")
    v = check(lib.bgls_final_verify_collect(cid), "collect")
    assert v == 1

for prof in (0, 1):
    lib.bgls_profile_enable(prof)
    for k in range(L): submit(k); collect(k)
    torch.cuda.synchronize()
    ts = []
    for k in range(L):
        t0 = time.perf_counter(); submit(k); ts.append(time.perf_counter() - t0)
    for k in range(L): collect(k)
    print("prof %d: submit host cost us: min %.0f median %.0f" % (prof, min(ts) * 1e6, sorted(ts)[L // 2] * 1e6))

def run(count, ks):
    Lk = len(ks)
    for i in range(count):
        submit(ks[i % Lk])
        if i >= Lk - 1: collect(ks[(i - Lk + 1) % Lk])
    for i in range(max(0, count - Lk + 1), count): collect(ks[i % Lk])

for prof in (1, 0):
    lib.bgls_profile_enable(prof)
    for T in (1, 2, 4):
        per = L // T
        groups = [list(range(t * per, (t + 1) * per)) for t in range(T)]
        steps = 64
        run(8, list(range(L)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(steps // T, g)) for g in groups]
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("prof %d threads %d: %.3f ms/step  %.2f G signers/s" % (prof, T, dt * 1e3, n / dt / 1e9))
