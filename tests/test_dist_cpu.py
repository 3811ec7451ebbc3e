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
from bgls_amd.sharding import (shard_range, all_gather_bytes, gather_partials_and_flags, global_duplicate_scan, digest_slot_records, all_to_all_bytes,
                               enqueue_digest_probe)


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

    def scan(buf, rl, count):
        raw = bytes(buf.numpy())
        items = [raw[i * rl:(i + 1) * rl] for i in range(count)]
        seen["count"] = count
        seen["bytes"] = len(raw)
        seen["dup"] = len(set(items)) != len(items)           # exact, like bgls/bgls.go:139-150
        return 0

    t = torch.frombuffer(msgs, dtype=torch.uint8)
    scan(t, ln, n_local)
    local_dup = seen["dup"]
    global_duplicate_scan(scan, t, n_local, world, msg_len=ln)
    full = (seen["dup"], seen["count"])
    # (3) the digest path: 16-byte digests travel instead of the messages; a hit is settled by the exact scan, no hit ends there
    import hashlib
    traffic = {"digest": 0, "probe_hits": 0, "exact_scans": 0}

    def digest(m, cnt):
        raw = bytes(m.numpy())
        out = b"".join(hashlib.blake2b(raw[i * ln:(i + 1) * ln]).digest()[:16] for i in range(cnt))    # what bgls_message_digests_dev computes
        traffic["digest"] += len(out)
        return torch.frombuffer(bytearray(out), dtype=torch.uint8)

    def probe(buf, rl, count):
        raw = bytes(buf.numpy())
        items = [raw[i * rl:(i + 1) * rl] for i in range(count)]
        hit = len(set(items)) != len(items)
        traffic["probe_hits"] += hit
        return hit

    def exact(buf, rl, count):
        traffic["exact_scans"] += 1
        return scan(buf, rl, count)

    seen["dup"] = None
    global_duplicate_scan(exact, t, n_local, world, digest=digest, msg_len=ln, probe=probe)
    with_dup = (seen["dup"], traffic["probe_hits"], traffic["exact_scans"])
    clean = bytearray(b"".join(bytes([16 * rank + i]) * ln for i in range(n_local)))
    seen["dup"] = None
    global_duplicate_scan(exact, torch.frombuffer(clean, dtype=torch.uint8), n_local, world, digest=digest, msg_len=ln, probe=probe)
    without_dup = (seen["dup"], traffic["probe_hits"], traffic["exact_scans"], traffic["digest"])
    q.put((rank, ok1, local_dup, full[0], full[1], with_dup, without_dup))
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
    for rank, ok1, local_dup, global_dup, count, with_dup, without_dup in res:
        assert ok1 and not local_dup and global_dup and count == 8, (rank, ok1, local_dup, global_dup, count)
        assert with_dup == (True, 1, 1), with_dup                      # digest hit -> exact scan over the gathered messages -> duplicate
        assert without_dup == (None, 1, 1, 2 * 4 * 16), without_dup    # no digest hit: no exact scan, no message ever gathered


def _worker_bucketed(rank, world, port, q):
    """Round 5: the digest scan itself is sharded.  Rank r scans only the digests whose first byte is r mod world (what
    bgls_duplicate_scan_bucket_dev does on the GPU), the ranks' answers are OR-ed, and the result must be what the scan of
    everything gives: a straddling duplicate is found by exactly one rank (the bucket's owner) and known to both; a forced digest
    collision between two DIFFERENT messages is settled as "no duplicate" by the exact scan; a clean batch gathers no message."""
    import hashlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, ln = 64, 64
    rnd = __import__("random").Random(977)
    all_msgs = [rnd.randbytes(ln) for _ in range(world * n_local)]
    stats = {"own_hits": 0, "inserted": 0, "exact": 0, "exact_dup": None}
    forced = {}

    def dig(m):
        return forced.get(bytes(m), hashlib.blake2b(bytes(m)).digest()[:16])

    def digest(m, cnt):
        raw = bytes(m.numpy())
        return torch.frombuffer(bytearray(b"".join(dig(raw[i * ln:(i + 1) * ln]) for i in range(cnt))), dtype=torch.uint8)

    def probe(buf, rl, count, bucket, n_buckets):
        raw = bytes(buf.numpy())
        mine = [raw[i * rl:(i + 1) * rl] for i in range(count) if raw[i * rl] % n_buckets == bucket]
        stats["inserted"] += len(mine)
        hit = len(set(mine)) != len(mine)
        stats["own_hits"] += hit
        return hit

    def exact(buf, rl, count):
        raw = bytes(buf.numpy())
        items = [raw[i * rl:(i + 1) * rl] for i in range(count)]
        stats["exact"] += 1
        stats["exact_dup"] = len(set(items)) != len(items)
        return stats["exact_dup"]

    def run(msgs):
        mine = b"".join(msgs[rank * n_local:(rank + 1) * n_local])
        return global_duplicate_scan(exact, torch.frombuffer(bytearray(mine), dtype=torch.uint8), n_local, world, digest=digest, msg_len=ln,
                                     probe=probe, rank=rank)

    clean = run(all_msgs)                                             # None: proven duplicate-free by the digests, no message gathered
    inserted_clean, exact_clean = stats["inserted"], stats["exact"]
    dup = list(all_msgs)
    dup[world * n_local - 3] = dup[5]                                 # message 5 of rank 0 again near the end of the last rank's range
    with_dup = run(dup)
    hits_dup = stats["own_hits"]
    stats["own_hits"] = 0
    forced[bytes(all_msgs[7])] = dig(all_msgs[n_local + 9])           # two DIFFERENT messages, one digest (a 2^-128 event, forced)
    collided = run(all_msgs)
    # several status words in one gather: word 0 = status, word 1 = this rank's probe word
    parts, merged = gather_partials_and_flags(torch.full((384,), rank + 1, dtype=torch.uint8), torch.tensor([0, 1 if rank == 1 else 0], dtype=torch.int32), world)
    q.put((rank, clean, inserted_clean, exact_clean, with_dup, hits_dup, collided, stats["own_hits"], stats["exact"], merged.tolist(), parts.numel()))
    dist.destroy_process_group()


def test_two_rank_bucketed_digest_scan():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bucketed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert sum(r[2] for r in res) == 2 * 64                         # every digest entered exactly one rank's table
    assert all(0 < r[2] < 2 * 64 for r in res)                      # ... and neither rank scanned them all
    assert sum(r[5] for r in res) == 1                              # the duplicate pair was found by exactly one rank (its bucket's owner)
    assert sum(r[7] for r in res) == 1                              # the forced collision likewise
    for rank, clean, inserted_clean, exact_clean, with_dup, hits_dup, collided, own_coll, exact_total, merged, nparts in res:
        assert clean is None and exact_clean == 0                   # no hit anywhere: no rank gathered a message
        assert with_dup is True                                     # both ranks learn of the duplicate, both run the exact scan
        assert collided is False                                    # equal digests of different messages: the exact scan says "no duplicate"
        assert exact_total == 2
        assert merged == [0, 1] and nparts == 2 * 384               # the probe word rides with the status word


def _worker_multisig(rank, world, port, cid, keys, sig, bad_sig, msg, n, q):
    """BASELINE config 4 over N ranks (bench.py bench_multisig_sharded's data path, SURVEY 8e multisig variant): every rank adds its
    contiguous range of the keys, ONE all-gather of the partial key sums, every rank adds the N partials and runs the
    two-pairing check -- the oracle stands in for the HIP library on this CPU tier."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fp = 32 if cid == 0 else 48
    lo, hi = shard_range(n, rank, world)
    part = coracle.aggregate_points(cid, 2, keys[lo * 4 * fp:hi * 4 * fp], hi - lo)
    parts = all_gather_bytes(torch.frombuffer(bytearray(part), dtype=torch.uint8), world)
    flat = bytes(parts.reshape(-1).numpy())
    ok = coracle.verify_multi(cid, sig, flat, world, msg)
    ok_bad = coracle.verify_multi(cid, bad_sig, flat, world, msg)
    # a rank that drops its last key changes the aggregate key: every rank must reject
    short = coracle.aggregate_points(cid, 2, keys[lo * 4 * fp:(hi - (1 if rank == world - 1 else 0)) * 4 * fp], hi - lo - (1 if rank == world - 1 else 0))
    parts2 = all_gather_bytes(torch.frombuffer(bytearray(short), dtype=torch.uint8), world)
    ok_short = coracle.verify_multi(cid, sig, bytes(parts2.reshape(-1).numpy()), world, msg)
    whole = coracle.aggregate_points(cid, 2, keys[:n * 4 * fp], n)
    summed = coracle.aggregate_points(cid, 2, flat, world)
    q.put((rank, ok, ok_bad, ok_short, whole == summed, parts.shape[0], parts.shape[1]))
    dist.destroy_process_group()


def test_two_rank_multisig_gathers_partial_key_sums():
    from tests.conftest import load_golden
    for cid, name in ((0, "altbn128"), (1, "bls12")):
        fp = 32 if cid == 0 else 48
        case = next(c for c in load_golden("vectors_%s.json" % name)["multi_cases"] if c["expect"] and len(c["keys"]) >= 4)
        bad = next(c for c in load_golden("vectors_%s.json" % name)["multi_cases"] if not c["expect"])
        keys = b"".join(bytes.fromhex(k) for k in case["keys"])
        n = len(case["keys"])
        sig, msg = bytes.fromhex(case["sig"]), bytes.fromhex(case["msg"])
        bad_sig = bytes.fromhex(bad["sig"]) if bytes.fromhex(bad["sig"]) != sig else bytes(2 * fp)
        assert coracle.verify_multi(cid, sig, keys, n, msg) == 1
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_multisig, args=(r, 2, port, cid, keys, sig, bad_sig, msg, n, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
        for rank, ok, ok_bad, ok_short, same_sum, rows, width in res:
            assert ok == 1 and ok_short == 0 and same_sum and rows == 2 and width == 4 * fp, (name, rank, ok, ok_bad, ok_short, same_sum, rows, width)
            if bad_sig != bytes(2 * fp):
                assert ok_bad == 0


def _worker_all_to_all(rank, world, port, q):
    """Round 6 (verdict r5 item 6): the digest exchange as an all-to-all by bucket.  `pack` and `probe` below do in Python what
    bgls_digest_pack_dev / bgls_duplicate_scan_packed_dev do on the GPU (slots of equal size, padding that belongs to another bucket,
    overflow = "undecided"); under test is bgls_amd.sharding: who receives what, how many bytes travel, the verdicts, and the refusals
    that must come BEFORE the first collective."""
    import hashlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local, ln = 96, 64
    rnd = __import__("random").Random(4242)
    all_msgs = [rnd.randbytes(ln) for _ in range(world * n_local)]
    stats = {"received": 0, "own": 0, "foreign_real": 0, "overflow": 0, "exact": 0, "sent_bytes": 0}
    forced = {}

    def dig(m):
        return forced.get(bytes(m), hashlib.blake2b(bytes(m)).digest()[:16])

    def digest(m, cnt):
        raw = bytes(m.numpy())
        return torch.frombuffer(bytearray(b"".join(dig(raw[i * ln:(i + 1) * ln]) for i in range(cnt))), dtype=torch.uint8)

    def pack(d, cnt, nb, cap):
        raw = bytes(d.numpy())
        slots = [[] for _ in range(nb)]
        for i in range(cnt):
            rec = raw[16 * i:16 * i + 16]
            b = rec[0] % nb
            if len(slots[b]) < cap:
                slots[b].append(rec)
            else:
                stats["overflow"] += 1
        out = b"".join(b"".join(sl) + (bytes([(b + 1) % nb]) + bytes(15)) * (cap - len(sl)) for b, sl in enumerate(slots))
        stats["sent_bytes"] += len(out)
        return torch.frombuffer(bytearray(out), dtype=torch.uint8)

    def probe(buf, rl, count, bucket, n_buckets):
        raw = bytes(buf.numpy())
        recs = [raw[i * rl:(i + 1) * rl] for i in range(count)]
        mine = [r for r in recs if r[0] % n_buckets == bucket]
        stats["received"] += count
        stats["own"] += len(mine)
        stats["foreign_real"] += sum(1 for r in recs if r[0] % n_buckets != bucket and r[1:] != bytes(15))
        return len(set(mine)) != len(mine) or stats["overflow"] > 0

    def exact(buf, rl, count):
        raw = bytes(buf.numpy())
        items = [raw[i * rl:(i + 1) * rl] for i in range(count)]
        stats["exact"] += 1
        return len(set(items)) != len(items)

    def run(msgs, **kw):
        mine = b"".join(msgs[rank * n_local:(rank + 1) * n_local])
        return global_duplicate_scan(exact, torch.frombuffer(bytearray(mine), dtype=torch.uint8), n_local, world, digest=digest, msg_len=ln,
                                     probe=probe, rank=rank, pack=pack, **kw)

    clean = run(all_msgs)
    after_clean = dict(stats)
    dup = list(all_msgs)
    dup[world * n_local - 2] = dup[3]                                 # message 3 of rank 0 again at the end of the last rank's range
    with_dup = run(dup)
    forced[bytes(all_msgs[11])] = dig(all_msgs[n_local + 17])         # two DIFFERENT messages, one digest
    collided = run(all_msgs)
    forced.clear()
    # an adversary's batch: every digest of this rank starts with the same byte -> one send slot overflows -> undecided -> exact scan -> no duplicate
    for m in all_msgs[rank * n_local:(rank + 1) * n_local]:
        forced[bytes(m)] = bytes([6]) + hashlib.blake2b(bytes(m)).digest()[1:16]
    skewed = run(all_msgs, slot_records=n_local // 2 + 8)              # slots of the size a fair share needs, not the production slack
    overflowed = stats["overflow"]
    stats["overflow"] = 0
    forced.clear()
    # refusals before any collective: were they raised after the exchange had been entered, the OTHER rank would hang in it
    errs = []
    for bad in (dict(probe=lambda b, r, c: False), dict(rank=None)):
        try:
            global_duplicate_scan(exact, torch.zeros(n_local * ln, dtype=torch.uint8), n_local, world, digest=digest, msg_len=ln,
                                  **{**dict(probe=probe, rank=rank, pack=pack), **bad})
            errs.append(None)
        except (TypeError, ValueError) as e:
            errs.append(type(e).__name__)
    try:
        enqueue_digest_probe(digest, lambda b, r, c: None, torch.zeros(n_local * ln, dtype=torch.uint8), n_local, world, rank=rank, pack=pack)
        errs.append(None)
    except TypeError:
        errs.append("TypeError")
    # the raw exchange: chunk r of the send buffer arrives as chunk `rank` of rank r's receive buffer
    got = all_to_all_bytes(torch.tensor([10 * rank + r for r in range(world) for _ in range(4)], dtype=torch.uint8), world)
    q.put((rank, clean, after_clean, with_dup, collided, skewed, overflowed, stats["exact"], errs, got.tolist()))
    dist.destroy_process_group()


def test_two_rank_digest_exchange_is_an_all_to_all_by_bucket():
    world, n_local = 2, 96
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_all_to_all, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    cap = digest_slot_records(n_local, world)
    assert cap == n_local // world + n_local // (4 * world) + 1024
    assert sum(r[2]["own"] for r in res) == world * n_local          # every digest reached exactly one rank: its bucket's owner
    for rank, clean, st, with_dup, collided, skewed, overflowed, exact_total, errs, got in res:
        assert clean is None and st["exact"] == 0                    # proven duplicate-free from the digests: no message gathered
        assert st["received"] == world * cap and st["foreign_real"] == 0       # a rank holds its own bucket and padding, nothing else
        assert st["sent_bytes"] == world * cap * 16                  # what travels per rank: slots, not world x n_local digests
        assert with_dup is True and collided is False                # found across the shard boundary; a collision settled as "no duplicate"
        assert skewed is False and overflowed > 0                    # overflow -> undecided -> the exact scan decides
        assert exact_total == 3
        assert errs == ["TypeError", "TypeError", "TypeError"]      # refused before the first collective (no hang: the run ended)
        assert got == [r * 10 + rank for r in range(world) for _ in range(4)]
