"""CPU tier: the N>1 path (independent signer shards, one all-gather of partial GT products,
local combine) on 2 gloo ranks.  The GPU kernels cannot run here, so each rank's partial Miller
product comes from the oracle; what is under test is bgls_amd.sharding (ranges, gather order,
sig pair on rank 0 only) -- the same code bench.py runs over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import coracle
from bgls_amd.sharding import shard_range, all_gather_bytes, gather_partials_and_flags, global_duplicate_scan


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, cid, g1s, g2s, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fp = 32 if cid == 0 else 48
    lo, hi = shard_range(n, rank, world)
    part = coracle.miller_product(cid, g1s[lo * 2 * fp:hi * 2 * fp], g2s[lo * 4 * fp:hi * 4 * fp], hi - lo)
    parts = all_gather_bytes(torch.frombuffer(bytearray(part), dtype=torch.uint8), world)
    acc = bytes(parts[0].numpy())
    for r in range(1, world):
        acc = coracle.gt_mul(cid, acc, bytes(parts[r].numpy()))
    q.put((rank, lo, hi, coracle.final_exp(cid, acc)))
    dist.destroy_process_group()


def test_two_rank_shards_match_single_rank():
    cid, n = 0, 7                                  # ragged: 7 pairs over 2 ranks
    g1 = bytes.fromhex("00" * 31 + "01" + "00" * 31 + "02")
    g1s = b"".join(coracle.scale_point(cid, 1, g1, 3 + i) for i in range(n))
    g2gen = coracle.scale_point  # noqa
    from tests.conftest import load_golden
    g2 = bytes.fromhex(load_golden("vectors_altbn128.json")["pairings"][3]["g2"])
    g2s = b"".join(coracle.scale_point(cid, 2, g2, 11 + 5 * i) for i in range(n))
    want = coracle.pairing_product(cid, g1s, g2s, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cid, g1s, g2s, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 7)]
    assert res[0][3] == want and res[1][3] == want          # every rank ends with the same canonical GT bytes


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 65537):
        for w in (1, 2, 4, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))


def _worker_flags(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (1) status words travel with the partials: only rank 1 saw a bad encoding, both ranks must end up with the bit
    part = torch.full((384,), rank + 1, dtype=torch.uint8)
    flags = torch.tensor([2 if rank == 1 else 0], dtype=torch.int32)
    parts, merged = gather_partials_and_flags(part, flags, world)
    ok1 = int(merged.item()) == 2 and parts.numel() == 2 * 384 and bytes(parts[:384].numpy()) == b"\x01" * 384 \
        and bytes(parts[384:].numpy()) == b"\x02" * 384
    # (2) a duplicate that straddles the two shards is invisible to per-shard scans and visible to the global one
    n_local, ln = 4, 64
    msgs = bytearray(b"".join(bytes([16 * rank + i]) * ln for i in range(n_local)))
    if rank == 1:
        msgs[2 * ln:3 * ln] = bytes([1]) * ln                 # equals message 1 of rank 0
    seen = {}

    def scan(buf, count):
        raw = bytes(buf.numpy())
        items = [raw[i * ln:(i + 1) * ln] for i in range(count)]
        seen["count"] = count
        seen["dup"] = len(set(items)) != len(items)           # exact, like bgls/bgls.go:139-150
        return 0

    t = torch.frombuffer(msgs, dtype=torch.uint8)
    scan(t, n_local)
    local_dup = seen["dup"]
    global_duplicate_scan(scan, t, n_local, world)
    q.put((rank, ok1, local_dup, seen["dup"], seen["count"]))
    dist.destroy_process_group()


def test_two_rank_flags_and_cross_shard_duplicates():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flags, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    for rank, ok1, local_dup, global_dup, count in res:
        assert ok1 and not local_dup and global_dup and count == 8, (rank, ok1, local_dup, global_dup, count)
