#!/usr/bin/env python3
"""Copy a gpurun evidence run (tools/gpu_profile_round.sh <tag>) into profiles/r1 and rebuild pmc_traffic.json."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
src, dst = f"gpurun_out/{tag}", "profiles/r1"
for d in ("stats", "stats_seq", "pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    shutil.rmtree(f"{dst}/{d}", ignore_errors=True)
    os.makedirs(f"{dst}/{d}")
    for f in glob.glob(f"{src}/{d}/*.csv"):
        if not f.endswith("agent_info.csv"):
            shutil.copy(f, f"{dst}/{d}/")
for f in glob.glob(f"{src}/bench_*.json") + [f"{src}/pytest_gpu.log"]:
    shutil.copy(f, dst)
def avg(path, kern, ctr):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    return sum(v) / len(v), len(v)
def traffic(kern, label):
    f, nf = avg(glob.glob(f"{dst}/pmc_fetch/*counter_collection.csv")[0], kern, "FETCH_SIZE")
    w, nw = avg(glob.glob(f"{dst}/pmc_write/*counter_collection.csv")[0], kern, "WRITE_SIZE")
    return {"kernel": label, "fetch_size_kb_raw": f, "write_size_kb_raw": w, "launches_averaged": [nf, nw],
            "bytes_per_launch_uncorrected": (f + w) * 1024, "bytes_per_launch_fetch_x2": (2 * f + w) * 1024,
            "algorithmic_bytes_per_launch": 65536 * 192 + 64}
out = traffic("k_miller_ab64<bgls::BN254", "k_miller_ab64<BN254>")
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/r1/pmc_fetch, pmc_write); FETCH_SIZE on gfx950 "
               "under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md), other widths uncalibrated, so both readings are given. "
               "The excess over the algorithmic bytes is the producer wave's private stack (point-step and line-product temporaries "
               "beyond the 256-VGPR budget), served mostly by L2 / Infinity Cache: counted at the fabric side, not HBM-exclusive.")
try:
    out["throughput_shape"] = traffic("k_miller_s60", "k_miller_s60<BN254>")
    out["throughput_shape"]["note"] = "the 60-pairing shape has no line products in the producer and spills far less"
except Exception as e:       # no such launches in this run
    pass
json.dump(out, open(f"{dst}/pmc_traffic.json", "w"), indent=1)
print({k: v for k, v in out.items() if k not in ("note",)})
