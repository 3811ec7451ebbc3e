#!/usr/bin/env python3
"""Headline benchmark: aggregate-verify signer-pairs/sec (BASELINE.json `metric`).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one complete VerifyAggregateSignature over the resident batch: duplicate-message
scan, n hash-to-G1, n+1 Miller loops, GT product, ONE final exponentiation, compare with 1 --
everything bgls/bgls.go:94-119 does, inputs already in HBM (keys as the reference's wire-format
bytes, 64-byte messages as in bgls/bgls_test.go:186-202).
N = 1 workload: BASELINE.json configs[1] (alt-bn128, 2^16 signers, 1 MI355X).
N > 1: weak scaling -- every rank verifies its own 2^16-signer shard of ONE n = N * 2^16
aggregate signature: partial Miller product per rank, one RCCL all-gather of the 384-byte
partials, local combine + final exponentiation on every rank (bgls_amd/sharding.py).
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with torch's own streams):
# with the default, two of the four verification lanes land on one queue and serialise.  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from bgls_amd import _lib  # noqa: E402
from bgls_amd.sharding import all_gather_bytes, gather_partials_and_flags, global_duplicate_scan  # noqa: E402

# Algorithmic work model (SURVEY.md 8d / DESIGN.md): 32x32->64 MACs per unit.
MAC_PER_FPMUL = {0: 136, 1: 300}                       # CIOS 2L^2+L, L = 8 / 12
MILLER_FPMUL = {0: 8250, 1: 6700}                      # Miller loop, Fp multiplications per pair
PAIR_FPMUL = {0: 9030, 1: 14650}                       # whole path per signer-pair (hash + Miller + product)
MULTISIG_FPMUL = {0: 29, 1: 29}                        # one G2 mixed addition per signer
ALGO_BYTES_PER_PAIR = {0: 128 + 64, 1: 192 + 64}       # key + message read once
ORDER = {0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
         1: 52435875175126190479447740508185965837690552500527637822603658699938581184513}


def B(b):
    return (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if b else b"\0")


def check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed: %d %s" % (what, rc, _lib.last_error()))
    return rc


def make_shard(lib, cid, n, seed):
    """n keys, n distinct 64-byte messages and the shard's partial aggregate signature, all
    produced by the engine itself on the GPU (setup, untimed)."""
    fp = 32 if cid == 0 else 48
    rnd = random.Random(seed)
    msgs = rnd.randbytes(64 * n)
    sks = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
    kb = b"".join(s.to_bytes(32, "big") for s in sks)
    g2 = (ctypes.c_uint8 * (4 * fp))()
    check(lib.bgls_generator(cid, 2, g2), "generator")
    keys = (ctypes.c_uint8 * (n * 4 * fp))()
    check(lib.bgls_scale_points(cid, 2, B(bytes(g2) * n), B(kb), None, n, keys), "scale_points(G2)")
    off = (ctypes.c_uint64 * (n + 1))(*[64 * i for i in range(n + 1)])
    hs = (ctypes.c_uint8 * (n * 2 * fp))()
    check(lib.bgls_hash_to_g1(cid, B(msgs), off, n, hs), "hash_to_g1")
    sigs = (ctypes.c_uint8 * (n * 2 * fp))()
    check(lib.bgls_scale_points(cid, 1, hs, B(kb), None, n, sigs), "scale_points(G1)")
    agg = (ctypes.c_uint8 * (2 * fp))()
    check(lib.bgls_aggregate_points(cid, 1, sigs, n, agg), "aggregate_points")
    return bytes(keys), msgs, bytes(agg), bytes(sigs)


def stage(lib, name):
    ms, cnt = ctypes.c_double(), ctypes.c_ulonglong()
    lib.bgls_profile_get(name.encode(), ctypes.byref(ms), ctypes.byref(cnt))
    return ms.value, cnt.value


def cpu_baseline(cid, keys, msgs, sigs, n, fp, lib):
    """Oracle (C restatement, oracle/c) timed on this box's host cores over a bounded sample of the
    same instance, in the reference's parallel shape: one task per hash and per FULL pairing
    (final exponentiation inside every pairing, curves/curve.go:132-134)."""
    from oracle import coracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)
    probe = min(n, 4 * cores)

    def run(cnt, faithful):
        agg = (ctypes.c_uint8 * (2 * fp))()
        check(lib.bgls_aggregate_points(cid, 1, B(sigs[:cnt * 2 * fp]), cnt, agg), "aggregate_points(sample)")
        ms = [msgs[64 * i:64 * i + 64] for i in range(cnt)]
        t0 = time.perf_counter()
        ok = coracle.verify_aggregate(cid, bytes(agg), keys[:cnt * 4 * fp], ms, False, cores, faithful)
        dt = time.perf_counter() - t0
        if ok != 1:
            raise RuntimeError("oracle rejected the GPU-generated instance (cpu_baseline sample)")
        return dt

    t_probe = run(probe, 1)
    cnt = int(min(n, max(probe, probe * 12.0 / max(t_probe, 1e-3))))     # ~12 s of CPU work
    dt = run(cnt, 1)
    dt_shared = run(min(cnt, 4096), 0)
    return {"value": cnt / dt, "unit": "signer-pairs/s", "cores": cores, "kind": "port",
            "sample": "first %d signers of the same instance, C oracle, %d threads, final exponentiation per pairing "
                      "(reference shape); with one shared final exponentiation: %.0f pairs/s" % (cnt, cores, min(cnt, 4096) / dt_shared)}


def bench_multisig(args, lib, cid, fp, n, dev, rank, world):
    """BASELINE.json config 4: n signers on ONE message -- G2 key sum (AggregatePoints,
    curves/curve.go:73-121) + hash + 2 pairings (bgls/bgls.go:59-70,89-92).  Single GPU."""
    if world != 1:
        raise SystemExit("multisig workload is single-GPU in this round")
    rnd = random.Random(0xB6150000 + 4)
    sks = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
    kb = b"".join(s.to_bytes(32, "big") for s in sks)
    g2 = (ctypes.c_uint8 * (4 * fp))()
    check(lib.bgls_generator(cid, 2, g2), "generator")
    keys = (ctypes.c_uint8 * (n * 4 * fp))()
    check(lib.bgls_scale_points(cid, 2, B(bytes(g2) * n), B(kb), None, n, keys), "scale_points(G2)")
    msg = b"\x01" + rnd.randbytes(64)
    off = (ctypes.c_uint64 * 2)(0, len(msg))
    h = (ctypes.c_uint8 * (2 * fp))()
    check(lib.bgls_hash_to_g1(cid, B(msg), off, 1, h), "hash_to_g1")
    sig = (ctypes.c_uint8 * (2 * fp))()
    check(lib.bgls_scale_points(cid, 1, h, B((sum(sks) % ORDER[cid]).to_bytes(32, "big")), None, 1, sig), "scale_points(sig)")
    t_keys = torch.frombuffer(bytearray(bytes(keys)), dtype=torch.uint8).to(dev)
    t_sig = torch.frombuffer(bytearray(bytes(sig)), dtype=torch.uint8).to(dev)
    t_msg = torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(nn=n):
        return check(lib.bgls_verify_multi_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), nn, t_msg.data_ptr(), len(msg), stream), "verify_multi_dev")

    if step() != 1 or step(n - 1) != 0:
        raise RuntimeError("multisig correctness gate failed")
    # L verifications in flight (own context and stream each): the key sum of one overlaps the serial hash / pairing /
    # final-exponentiation tail of the others; every step is a complete verification whose verdict is checked
    L = max(1, min(16, args.in_flight))
    lanes = [torch.cuda.Stream(device=dev) for _ in range(L)]
    torch.cuda.synchronize()

    def submit(k):
        check(lib.bgls_select_context(k), "select_context")
        check(lib.bgls_verify_multi_submit_dev(cid, t_sig.data_ptr(), t_keys.data_ptr(), n, t_msg.data_ptr(), len(msg), lanes[k].cuda_stream),
              "verify_multi_submit_dev")

    def collect(k):
        check(lib.bgls_select_context(k), "select_context")
        return check(lib.bgls_final_verify_collect(cid), "final_verify_collect")

    def run(count):
        for i in range(count):
            submit(i % L)
            if i >= L - 1 and collect((i - L + 1) % L) != 1:
                raise RuntimeError("verification failed inside the timed region")
        for i in range(max(0, count - L + 1), count):
            if collect(i % L) != 1:
                raise RuntimeError("verification failed inside the timed region")
        check(lib.bgls_select_context(0), "select_context")

    lib.bgls_profile_enable(1)
    for _ in range(args.warmup):
        step()
    stages_excl = {s_: stage(lib, s_) for s_ in ("sum_points", "h2c", "miller", "reduce", "final_exp")}
    run(L)
    lib.bgls_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sum_ms, sum_cnt = stage(lib, "sum_points")
    lib.bgls_profile_enable(0)
    peak = ctypes.c_double()
    check(lib.bgls_probe_mad_peak(ctypes.byref(peak)), "probe_mad_peak")
    avg_s = sum_ms / max(sum_cnt, 1) * 1e-3
    macs = n * MULTISIG_FPMUL[cid] * 3 * MAC_PER_FPMUL[cid]          # one G2 mixed addition = 29 Fp2-level products ~ 3 Fp mults each
    bytes_per_launch = n * 4 * fp
    print(json.dumps({
        "metric": "multisig-verify signers/sec", "value": n * args.steps / elapsed, "unit": "signers/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s KoskVerifyMultiSignature, %d signers on one message, keys resident in HBM" % (args.curve, n)},
        "roofline": {"bound": "valu-int32-mac", "kernel": "k_sum_first+k_sum_next", "achieved": macs / avg_s / 1e12, "peak": peak.value / 1e12,
                     "unit": "TMAC/s", "frac": macs / avg_s / peak.value, "traffic": None, "launch_ms": avg_s * 1e3,
                     "hbm_algorithmic_GBps": bytes_per_launch / avg_s / 1e9},
        "in_flight": L,
        "stage_ms_exclusive": {k: (v[0] / max(v[1], 1)) for k, v in stages_excl.items()},
    }), flush=True)


def bench_multisig_hae(args, lib, cid, fp, n, rank, world):
    """VerifyMultiSignatureWithHAE (bgls/blsHAE.go:56-58) -- exponents from BLAKE2Xb over all keys (root
    digest on the host while the keys upload, expansion on the device), apk = sum t_i pk_i as one fused weighted sum, then
    a single-signature check.  Host-buffer entry point: the hash needs the key bytes on the host, so this figure is
    PCIe- and host-hash-inclusive by construction."""
    if world != 1:
        raise SystemExit("multisig-hae workload is single-GPU in this round")
    rnd = random.Random(0xB6150000 + 6)
    sks = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
    g2 = (ctypes.c_uint8 * (4 * fp))()
    check(lib.bgls_generator(cid, 2, g2), "generator")
    keys = (ctypes.c_uint8 * (n * 4 * fp))()
    check(lib.bgls_scale_points(cid, 2, B(bytes(g2) * n), B(b"".join(s.to_bytes(32, "big") for s in sks)), None, n, keys), "scale_points(G2)")
    t_raw = (ctypes.c_uint8 * (16 * n))()
    t0 = time.perf_counter()
    check(lib.bgls_hae_exponents(cid, keys, n, t_raw), "hae_exponents")
    t_hash = time.perf_counter() - t0
    tb = bytes(t_raw)
    e = sum(sk * int.from_bytes(tb[16 * i:16 * i + 16], "big") for i, sk in enumerate(sks)) % ORDER[cid]
    msg = rnd.randbytes(64)
    off = (ctypes.c_uint64 * 2)(0, len(msg))
    h = (ctypes.c_uint8 * (2 * fp))()
    check(lib.bgls_hash_to_g1(cid, B(msg), off, 1, h), "hash_to_g1")
    sig = (ctypes.c_uint8 * (2 * fp))()
    check(lib.bgls_scale_points(cid, 1, h, B(e.to_bytes(32, "big")), None, 1, sig), "scale_points(sig)")
    mb = B(msg)

    def step(nn=n):
        return check(lib.bgls_verify_multi_hae(cid, sig, keys, nn, mb, len(msg)), "verify_multi_hae")

    if step() != 1 or step(n - 1) != 0:
        raise RuntimeError("multisig-hae correctness gate failed")
    for _ in range(args.warmup):
        step()
    lib.bgls_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if step() != 1:
            raise RuntimeError("verification failed inside the timed region")
    elapsed = time.perf_counter() - t0
    sum_ms, sum_cnt = stage(lib, "sum_points")
    lib.bgls_profile_enable(0)
    peak = ctypes.c_double()
    check(lib.bgls_probe_mad_peak(ctypes.byref(peak)), "probe_mad_peak")
    avg_s = sum_ms / args.steps * 1e-3
    fpmul_per_signer = 127 * 18 + 64 * 29                 # 128-bit double-and-add on G2: doubling ~ 18 m, mixed addition ~ 29 m
    macs = n * fpmul_per_signer * MAC_PER_FPMUL[cid]
    print(json.dumps({
        "metric": "multisig-hae-verify signers/sec", "value": n * args.steps / elapsed, "unit": "signers/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "%s VerifyMultiSignatureWithHAE, %d signers on one message, host buffers (keys cross PCIe and are "
                               "hashed on the host inside the call)" % (args.curve, n)},
        "roofline": {"bound": "valu-int32-mac", "kernel": "k_wsum_first (+k_sum_next)", "achieved": macs / avg_s / 1e12, "peak": peak.value / 1e12,
                     "unit": "TMAC/s", "frac": macs / avg_s / peak.value, "traffic": None, "launch_ms": avg_s * 1e3},
        "host_side": {"blake2xb_root_plus_expansion_ms": t_hash * 1e3, "key_bytes": n * 4 * fp,
                      "note": "the BLAKE2Xb root is one sequential compression chain over all key bytes (blsHAE.go:81-84)"},
    }), flush=True)


def bench_decompress(args, lib, cid, fp, n, rank, world):
    """UnmarshalG2 of n compressed alt-bn128 keys (curves/altbn128.go:329-376) -- the step in front of the
    hot path when keys arrive over the wire.  Host buffers in and out (64 B -> 128 B per key)."""
    if world != 1 or cid != 0:
        raise SystemExit("decompress workload: alt-bn128, single GPU")
    rnd = random.Random(0xB6150000 + 7)
    sks = [rnd.randrange(1, ORDER[cid]) for _ in range(n)]
    g2 = (ctypes.c_uint8 * 128)()
    check(lib.bgls_generator(cid, 2, g2), "generator")
    keys = (ctypes.c_uint8 * (n * 128))()
    check(lib.bgls_scale_points(cid, 2, B(bytes(g2) * n), B(b"".join(s.to_bytes(32, "big") for s in sks)), None, n, keys), "scale_points(G2)")
    comp = (ctypes.c_uint8 * (n * 64))()
    check(lib.bgls_compress_points(cid, 2, keys, n, comp), "compress_points")
    out = (ctypes.c_uint8 * (n * 128))()
    ok = (ctypes.c_uint8 * n)()

    def step():
        check(lib.bgls_decompress_points(cid, 2, comp, n, out, ok), "decompress_points")

    step()
    if bytes(out) != bytes(keys) or bytes(ok) != b"\x01" * n:
        raise RuntimeError("decompress round trip failed")
    for _ in range(args.warmup):
        step()
    lib.bgls_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    k_ms, k_cnt = stage(lib, "sum_points")
    lib.bgls_profile_enable(0)
    peak = ctypes.c_double()
    check(lib.bgls_probe_mad_peak(ctypes.byref(peak)), "probe_mad_peak")
    avg_s = k_ms / max(k_cnt, 1) * 1e-3
    fpmul_per_key = 3 * 320 + 40          # reference algorithm: two square roots and one residuosity test by exponentiation, plus x^3 etc.
    macs = n * fpmul_per_key * MAC_PER_FPMUL[cid]
    print(json.dumps({
        "metric": "G2 key decompression keys/sec", "value": n * args.steps / elapsed, "unit": "keys/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "altbn128 UnmarshalG2 (compressed), %d keys, host buffers in and out" % n},
        "roofline": {"bound": "valu-int32-mac", "kernel": "k_decompress_bn<2>", "achieved": macs / avg_s / 1e12, "peak": peak.value / 1e12,
                     "unit": "TMAC/s", "frac": macs / avg_s / peak.value, "traffic": None, "launch_ms": avg_s * 1e3,
                     "hbm_algorithmic_GBps": n * 192 / avg_s / 1e9},
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--curve", default="altbn128", choices=["altbn128", "bls12"])
    ap.add_argument("--n", "--signers", dest="n", type=int, default=1 << 16, help="signers per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-throughput-mode", action="store_true", help="keep the 64-pairing Miller kernel also when launches overlap")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="verifications kept in flight (1 = strictly sequential, max 16); default 4, multisig workload 8")
    ap.add_argument("--workload", default="aggregate", choices=["aggregate", "multisig", "multisig-hae", "decompress"],
                    help="aggregate = VerifyAggregateSignature (headline); multisig = KoskVerifyMultiSignature (BASELINE config 4)")
    args = ap.parse_args()
    if args.in_flight is None:
        args.in_flight = 8 if args.workload == "multisig" else 4

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch N > 1 with torch.distributed.run" % (world, args.gpus))
    share = os.environ.get("BGLS_BENCH_SHARE_GPU") == "1"   # development: all ranks on cuda:0 over gloo, to exercise the
    if share:                                               # N > 1 code path on a one-GPU box (numbers meaningless)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()
    check(lib.bgls_init(local_rank), "bgls_init")
    cid = 0 if args.curve == "altbn128" else 1
    fp = 32 if cid == 0 else 48
    n = args.n
    gtb = 12 * fp

    if args.workload == "multisig":
        return bench_multisig(args, lib, cid, fp, n, dev, rank, world)
    if args.workload == "multisig-hae":
        return bench_multisig_hae(args, lib, cid, fp, n, rank, world)
    if args.workload == "decompress":
        return bench_decompress(args, lib, cid, fp, n, rank, world)

    # ---- setup (untimed): resident shard + the global aggregate signature on rank 0
    keys, msgs, part_sig, sigs = make_shard(lib, cid, n, 0xB6150000 + 1 + 1000 * rank)
    t_keys = torch.frombuffer(bytearray(keys), dtype=torch.uint8).to(dev)
    t_msgs = torch.frombuffer(bytearray(msgs), dtype=torch.uint8).to(dev)
    t_psig = torch.frombuffer(bytearray(part_sig), dtype=torch.uint8).to(dev)
    all_sigs = all_gather_bytes(t_psig, world)
    agg = (ctypes.c_uint8 * (2 * fp))()
    check(lib.bgls_aggregate_points(cid, 1, B(bytes(all_sigs.cpu().numpy().tobytes())), world, agg), "aggregate_points(global)")
    t_sig = torch.frombuffer(bytearray(bytes(agg)), dtype=torch.uint8).to(dev)
    # L verifications in flight on L library contexts / streams (default 4): every step is still one complete pass (duplicate
    # scan, hash, Miller, reduce, exchange when N > 1, final exponentiation, verdict checked), but the serial latency-bound
    # stages of one step overlap the Miller launch of its neighbours.  --in-flight 1 runs them strictly one after the other.
    L = max(1, min(16, args.in_flight))
    lanes = [{"stream": torch.cuda.Stream(device=dev), "part": torch.zeros(gtb, dtype=torch.uint8, device=dev),
              "flags": torch.zeros(1, dtype=torch.int32, device=dev)} for _ in range(L)]
    torch.cuda.synchronize()

    def submit(k, msgs_t=t_msgs):
        """Enqueue one complete verification on lane k (its own library context and stream); nothing here waits for the GPU."""
        ln = lanes[k]
        h = ln["stream"].cuda_stream
        check(lib.bgls_select_context(k), "select_context")
        with torch.cuda.stream(ln["stream"]):
            ln["flags"].zero_()
            if world > 1:                 # duplicates may straddle shards: exact scan over every rank's messages
                global_duplicate_scan(lambda buf, count: check(lib.bgls_duplicate_scan_dev(buf.data_ptr(), 64, 64, count, ln["flags"].data_ptr(), h),
                                                               "duplicate_scan_dev"), msgs_t, n, world)
            check(lib.bgls_miller_product_dev(cid, t_sig.data_ptr() if rank == 0 else None, t_keys.data_ptr(), msgs_t.data_ptr(),
                                              64, 64, n, 1 if world == 1 else 0, ln["part"].data_ptr(), ln["flags"].data_ptr(), h), "miller_product_dev")
            if world == 1:
                check(lib.bgls_final_verify_submit_dev(cid, ln["part"].data_ptr(), 1, ln["flags"].data_ptr(), h), "final_verify_submit_dev")
            else:
                parts, merged = gather_partials_and_flags(ln["part"], ln["flags"], world)
                check(lib.bgls_final_verify_submit_dev(cid, parts.data_ptr(), world, merged.data_ptr(), h), "final_verify_submit_dev")

    def collect(k):
        check(lib.bgls_select_context(k), "select_context")
        return check(lib.bgls_final_verify_collect(cid), "final_verify_collect")

    def step(msgs_t=t_msgs):
        submit(0, msgs_t)
        return collect(0)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # correctness gate: the valid instance verifies, one flipped message bit rejects
    if step() != 1:
        raise RuntimeError("valid instance rejected")
    bad = t_msgs.clone()
    if rank == world - 1:
        bad[64 * (n // 2) + 3] ^= 0x20
    torch.cuda.synchronize()              # `bad` was written on torch's default stream, the lanes have their own
    if step(bad) != 0:
        raise RuntimeError("tampered instance accepted")

    pipelined = L > 1

    def run(count, overlap):
        if not overlap:
            for _ in range(count):
                if step() != 1:
                    raise RuntimeError("verification failed inside the timed region")
            return
        for i in range(count):
            submit(i % L)
            if i >= L - 1 and collect((i - L + 1) % L) != 1:
                raise RuntimeError("verification failed inside the timed region")
        for i in range(max(0, count - L + 1), count):
            if collect(i % L) != 1:
                raise RuntimeError("verification failed inside the timed region")
        check(lib.bgls_select_context(0), "select_context")

    # warm-up: W steps one at a time with the stage timers on -- these give each kernel's duration when it has the machine
    # to itself ("exclusive"); then, if overlapping, L untimed overlapped steps to prime the other contexts' workspaces
    lib.bgls_profile_enable(1)
    run(args.warmup, False)
    sync()
    stages_excl = {s: stage(lib, s) for s in ("dup_check", "h2c", "miller", "reduce", "final_exp")}
    if pipelined:
        # launches overlap from here on: alt-bn128 Miller launches may take the shape that is fastest in that regime
        # (bgls_set_throughput_mode: 60 pairings per block, see k_miller_s60); verdicts are identical in both modes
        if cid == 0 and not args.no_throughput_mode:
            check(lib.bgls_set_throughput_mode(1), "set_throughput_mode")
        run(L, True)
    lib.bgls_profile_enable(1)
    sync()
    t0 = time.perf_counter()
    run(args.steps, pipelined)
    sync()
    elapsed = time.perf_counter() - t0
    check(lib.bgls_set_throughput_mode(0), "set_throughput_mode")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stages = {s: stage(lib, s) for s in ("dup_check", "h2c", "miller", "reduce", "final_exp")}
    lib.bgls_profile_enable(0)

    if rank == 0:
        peak = ctypes.c_double()
        check(lib.bgls_probe_mad_peak(ctypes.byref(peak)), "probe_mad_peak")
        mil_ms, mil_cnt = stages["miller"]
        mil_events_s = mil_ms / max(mil_cnt, 1) * 1e-3        # HIP events around the launch on its own stream
        macs_per_launch = (n + 1) * MILLER_FPMUL[cid] * MAC_PER_FPMUL[cid]
        # With several verifications in flight the launches of the dominant kernel share the machine: the events around ONE
        # launch then also span the time it waits for and shares SIMDs with its neighbours.  The time one launch costs
        # the machine over the timed region is the region divided by the launches in it (= ms_per_step, every other
        # stage's time included, so this is the conservative reading); strictly sequential runs use the events.
        mil_avg_s = mil_events_s if L == 1 else elapsed / args.steps
        achieved = macs_per_launch / mil_avg_s / 1e12 if mil_avg_s > 0 else 0.0
        value = world * n * args.steps / elapsed
        ex_ms, ex_cnt = stages_excl["miller"]
        excl = None
        if ex_cnt:
            ex_s = ex_ms / ex_cnt * 1e-3
            excl = {"launch_ms": ex_s * 1e3, "achieved": macs_per_launch / ex_s / 1e12, "frac": macs_per_launch / ex_s / peak.value if peak.value else None,
                    "note": "the Miller launch with one verification in flight (the %d warm-up steps; k_miller_ab64, the shape the library "
                            "uses when launches do not overlap): its duration with the machine to itself.  With %d in flight consecutive "
                            "launches share the machine: launch_ms_events (HIP events around one launch) stretches, launch_ms = timed "
                            "region / launches is what a launch costs the machine." % (ex_cnt, L)}
        cname = "BN254" if cid == 0 else "BLS381"
        # the library's dispatch rule (Engine::miller_coop): 64 pairings per block, consecutive launches of 1024 blocks
        miller_kernel = "k_miller_ab64<%s>%s" % (cname, "" if (n + 63) // 64 <= 1024 else " x%d launches" % (((n + 63) // 64 + 1023) // 1024))
        if pipelined and cid == 0 and not args.no_throughput_mode:
            miller_kernel = "k_miller_s60<BN254> (timed region; the exclusive figures are k_miller_ab64<BN254>, the shape used when launches do not overlap)"
        out = {
            "metric": "aggregate-verify signer-pairs/sec", "value": value, "unit": "signer-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s VerifyAggregateSignature, %d signers per GPU (%d total), distinct 64-byte messages, "
                                   "keys and messages resident in HBM" % (args.curve, n, world * n),
                       "curve": args.curve, "signers_per_gpu": n, "in_flight": L, "parallelism": "signer-shards x%d + all-gather of GT partials" % world},
            "roofline": {"bound": "valu-int32-mac", "kernel": miller_kernel, "achieved": achieved, "peak": peak.value / 1e12,
                         "unit": "TMAC/s", "frac": achieved / (peak.value / 1e12) if peak.value else None, "traffic": None,
                         "launch_ms": mil_avg_s * 1e3, "launch_ms_events": mil_events_s * 1e3, "macs_per_launch": macs_per_launch,
                         "exclusive": excl,
                         "hbm_side": {"achieved": n * ALGO_BYTES_PER_PAIR[cid] / mil_avg_s / 1e9 if mil_avg_s > 0 else None, "peak": 8000.0,
                                      "unit": "GB/s", "note": "algorithmic bytes of one Miller launch / its duration"},
                         "note": "integer bignum path: bounded by v_mad_u64_u32 issue, not HBM or MFMA (SURVEY 8d); peak measured "
                                 "live by bgls_probe_mad_peak; HBM side: %.3f GB/s algorithmic of 8000 peak"
                                 % (value * ALGO_BYTES_PER_PAIR[cid] / world / 1e9),
                         "whole_path_frac": value / world * PAIR_FPMUL[cid] * MAC_PER_FPMUL[cid] / peak.value if peak.value else None},
            "stage_ms_per_step": {k: (v[0] / max(v[1], 1)) for k, v in stages.items()},
            "stage_ms_exclusive": {k: (v[0] / max(v[1], 1)) for k, v in stages_excl.items()},
        }
        if world == 1:
            # the same verification through the host-buffer entry point (keys + messages cross PCIe inside the call)
            off = (ctypes.c_uint64 * (n + 1))(*[64 * i for i in range(n + 1)])
            kb_, mb_, sb_ = B(keys), B(msgs), B(bytes(agg))
            check(lib.bgls_verify_aggregate(cid, sb_, kb_, mb_, off, n, 0), "verify_aggregate(host)")
            t1 = time.perf_counter()
            ok = check(lib.bgls_verify_aggregate(cid, sb_, kb_, mb_, off, n, 0), "verify_aggregate(host)")
            dt = time.perf_counter() - t1
            out["pcie_inclusive"] = {"value": n / dt, "unit": "signer-pairs/s", "ms_per_call": dt * 1e3, "verdict": ok,
                                     "note": "bgls_verify_aggregate with host buffers (pageable memory), never the headline value"}
            pmc = os.path.join(ROOT, "profiles", "r1", "pmc_traffic.json")
            if os.path.exists(pmc) and args.curve == "altbn128" and n == 1 << 16:
                det = json.load(open(pmc))          # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the evidence run (profiles/r1)
                shape = det.get("throughput_shape") if (pipelined and cid == 0 and not args.no_throughput_mode) else None
                out["roofline"]["traffic"] = (shape or det).get("bytes_per_launch_fetch_x2")     # bytes per launch, gfx950 FETCH_SIZE correction applied
                out["roofline"]["traffic_detail"] = det
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cid, keys, msgs, sigs, n, fp, lib)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
