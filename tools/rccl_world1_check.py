#!/usr/bin/env python3
"""The multi-GPU exchange steps of bgls_amd/sharding.py through the REAL RCCL backend with a process group of one rank (a one-GPU box cannot hold
two RCCL ranks): dtype / shape / device rules of ProcessGroupNCCL for the uint8 all-gather, the status-word gather, the all-to-all by bucket with
the library's own digest / pack / scan kernels, the one-word all-reduce and the float64 MAX of the bench's timing.  Exit code 3 = RCCL could not
initialise here (environment), 0 = every step checked, anything else = a step failed.  Run by tests/test_gpu_rccl_world1.py."""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

port = int(sys.argv[1]) if len(sys.argv) > 1 else 29533
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(port))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    probe = torch.ones(4, dtype=torch.float32, device=dev)
    dist.all_reduce(probe)                                   # the communicator is created on first use
    torch.cuda.synchronize()
except Exception as e:                                       # no usable RCCL transport on this box: an environment matter
    print("RCCL init failed: %r" % (e,))
    sys.exit(3)

from bgls_amd import _lib, sharding
lib = _lib.load()
assert lib.bgls_init(0) == 0
world, rank = 1, 0
rnd = random.Random(61)
assert not sharding._solo(world)                            # a group exists: the collectives below are real RCCL calls

# (1) the partials' all-gather: uint8, odd length
part = torch.frombuffer(bytearray(rnd.randbytes(397)), dtype=torch.uint8).to(dev)
got = sharding.all_gather_bytes(part, world)
assert got.shape == (1, 397) and torch.equal(got[0], part)
# (2) partial product + status words in one gather
flags = torch.tensor([5, 2], dtype=torch.int32, device=dev)
parts, merged = sharding.gather_partials_and_flags(part[:384], flags, world)
assert torch.equal(parts, part[:384]) and merged.tolist() == [5, 2]
# (3) all-to-all of uint8 slots
send = torch.frombuffer(bytearray(rnd.randbytes(16 * 100)), dtype=torch.uint8).to(dev)
assert torch.equal(sharding.all_to_all_bytes(send, world), send)
# (4) the digest path with the library's kernels: digests -> pack into slots -> all-to-all -> packed scan; then the exact answer
n = 5000
msgs = [rnd.randbytes(64) for _ in range(n)]
for dup in (False, True):
    ms = list(msgs)
    if dup:
        ms[4321] = ms[17]
    t_msgs = torch.frombuffer(bytearray(b"".join(ms)), dtype=torch.uint8).to(dev)
    word = torch.zeros(2, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def digest(m, n_local):
        out = torch.empty(n_local * 16, dtype=torch.uint8, device=dev)
        assert lib.bgls_message_digests_dev(m.data_ptr(), 64, 64, n_local, out.data_ptr(), stream) == 0
        return out

    def pack(dg, n_local, w, cap):
        out = torch.empty(w * cap * 16, dtype=torch.uint8, device=dev)
        assert lib.bgls_digest_pack_dev(dg.data_ptr(), n_local, w, cap, out.data_ptr(), word[1:2].data_ptr(), stream) == 0
        return out

    def probe_scan(buf, rec, count, bucket, n_buckets):
        assert lib.bgls_duplicate_scan_packed_dev(buf.data_ptr(), count, bucket, n_buckets, word[1:2].data_ptr(), stream) == 0

    def exact(buf, rec, count):
        w = torch.zeros(1, dtype=torch.int32, device=dev)
        assert lib.bgls_duplicate_scan_dev(buf.data_ptr(), rec, rec, count, w.data_ptr(), stream) == 0
        torch.cuda.synchronize()
        return bool(w.item())

    # one rank plays both buckets of a two-bucket exchange: its two send slots come back through RCCL's all-to-all as they are, each is then
    # scanned as the bucket it belongs to (the bucket rule needs two buckets or more: include/bgls_hip.h)
    cap = sharding.digest_slot_records(n, 2)
    send = pack(digest(t_msgs, n), n, 2, cap)
    recs = sharding.all_to_all_bytes(send, world)
    assert recs.numel() == 2 * cap * 16
    for bucket in (0, 1):
        probe_scan(recs[bucket * cap * 16:(bucket + 1) * cap * 16], 16, cap, bucket, 2)
    _, m2 = sharding.gather_partials_and_flags(part[:384], word, world)
    torch.cuda.synchronize()
    hit = bool(int(m2[1].item()))
    assert hit == dup, (dup, m2.tolist())
    if hit:
        assert sharding.settle_digest_hit(exact, t_msgs, n, world) is True
    # the synchronous form's one-word all-reduce
    w1 = torch.tensor([1 if hit else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(w1, op=dist.ReduceOp.MAX)
    assert bool(int(w1.item())) == dup
    # without a rank: every digest gathered, the full scan over them (rounds 3-4's path), then the exact scan on a hit
    r = sharding.global_duplicate_scan(exact, t_msgs, n, world, digest=digest, msg_len=64, probe=lambda b, rec, c: exact(b, rec, c))
    assert r is (True if dup else None), (dup, r)
# (5) the bench's timing reduction and its barrier
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.25
print("rccl world-1 ok: backend %s, nccl %s" % (dist.get_backend(), ".".join(map(str, torch.cuda.nccl.version()))))
dist.destroy_process_group()
