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
kern = "k_miller_ab64"
f, nf = avg(glob.glob(f"{dst}/pmc_fetch/*counter_collection.csv")[0], kern, "FETCH_SIZE")
w, nw = avg(glob.glob(f"{dst}/pmc_write/*counter_collection.csv")[0], kern, "WRITE_SIZE")
out = {"kernel": kern + "<BN254>", "fetch_size_kb_raw": f, "write_size_kb_raw": w, "launches_averaged": [nf, nw],
       "bytes_per_launch_uncorrected": (f + w) * 1024, "bytes_per_launch_fetch_x2": (2 * f + w) * 1024,
       "algorithmic_bytes_per_launch": 65536 * 192 + 64,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/r1/pmc_fetch, pmc_write); FETCH_SIZE on gfx950 "
               "under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md), other widths uncalibrated, so both readings are given. "
               "The excess over the algorithmic bytes is the producer wave's private stack (point-step and line-product temporaries "
               "beyond the 256-VGPR budget), served mostly by L2 / Infinity Cache: counted at the fabric side, not HBM-exclusive."}
json.dump(out, open(f"{dst}/pmc_traffic.json", "w"), indent=1)
print({k: v for k, v in out.items() if k != "note"})
