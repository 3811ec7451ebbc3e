#!/usr/bin/env python3
"""Copy an evidence run (tools/gpu_profile_round.sh <tag>) from gpurun_out/<tag> into profiles/r<N> and rebuild
profiles/r<N>/pmc_traffic.json (per-launch fabric traffic of the dominant kernels, gfx950 FETCH_SIZE x2 correction applied
as MI355X_MICROARCH.md prescribes; both readings kept; since round 6 also the clock the kernel ran at, GRBM_GUI_ACTIVE per
XCD / rocprof's average launch duration, which is what `roofline.peak_at_kernel_clock` of the bench line is made from).
usage: tools/refresh_profiles.py --round 6 <tag>"""
import argparse, csv, glob, json, os, shutil, sys
_ap = argparse.ArgumentParser()
_ap.add_argument("--round", type=int, required=True)
_ap.add_argument("tag")
_args = _ap.parse_args()
RND = "r%d" % _args.round
tag = _args.tag
src, dst = "gpurun_out/%s" % tag, "profiles/%s" % RND
os.makedirs(dst, exist_ok=True)
for d in glob.glob(src + "/stats_*") + glob.glob(src + "/pmc_*"):
    if not os.path.isdir(d):
        continue
    out = os.path.join(dst, os.path.basename(d).replace(" ", "_")[:60])
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    for f in glob.glob(d + "/**/*.csv", recursive=True):
        if f.endswith("agent_info.csv") or f.endswith("kernel_trace.csv") and os.path.getsize(f) > 3_000_000:
            continue
        shutil.copy(f, out)
for f in glob.glob(src + "/bench_*.json") + glob.glob(src + "/bench_*.txt") + glob.glob(src + "/bench_default.out") + [src + "/pytest_gpu.log"]:
    if os.path.exists(f):
        shutil.copy(f, dst)


def avg(pattern, kern, ctr):
    fs = glob.glob(pattern)
    if not fs:
        return None, 0
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def sq(name, kern):
    """VALUBusy and the LDS conflict ratio of `kern` from the two SQ passes (rocprofiler's gfx9 formula: VALUBusy = sum
    SQ_ACTIVE_INST_VALU / CU_NUM / max GRBM_GUI_ACTIVE; the csv rows are summed over the 8 XCDs, so the per-XCD GRBM reading is
    the sum / 8)."""
    out = {}
    act, _ = avg("%s/pmc_%s_SQ_WAVE_CYCLES/*counter_collection.csv" % (dst, name), kern, "SQ_ACTIVE_INST_VALU")
    grbm, _ = avg("%s/pmc_%s_SQ_INSTS_SALU/*counter_collection.csv" % (dst, name), kern, "GRBM_GUI_ACTIVE")
    conf, _ = avg("%s/pmc_%s_SQ_INSTS_SALU/*counter_collection.csv" % (dst, name), kern, "SQ_LDS_BANK_CONFLICT")
    idx, _ = avg("%s/pmc_%s_SQ_INSTS_SALU/*counter_collection.csv" % (dst, name), kern, "SQ_LDS_IDX_ACTIVE")
    wait, _ = avg("%s/pmc_%s_SQ_WAVE_CYCLES/*counter_collection.csv" % (dst, name), kern, "SQ_WAIT_ANY")
    wcyc, _ = avg("%s/pmc_%s_SQ_WAVE_CYCLES/*counter_collection.csv" % (dst, name), kern, "SQ_WAVE_CYCLES")
    if act and grbm:
        out["valu_busy"] = act / 256.0 / (grbm / 8.0)
        out["valu_busy_inputs"] = {"SQ_ACTIVE_INST_VALU": act, "GRBM_GUI_ACTIVE_sum_over_8_xcds": grbm, "cus": 256}
    if conf is not None and idx:
        out["lds_conflict_ratio"] = conf / idx
        out["lds_inputs"] = {"SQ_LDS_BANK_CONFLICT": conf, "SQ_LDS_IDX_ACTIVE": idx}
    if wait and wcyc:
        out["wait_any_over_wave_cycles"] = wait / wcyc
    return out


def launch_ms(stats_name, kern):
    """rocprof's average duration of `kern` (ms) in the --kernel-trace --stats run `stats_name`"""
    for f in glob.glob("%s/stats_%s/*kernel_stats.csv" % (dst, stats_name)):
        for r in csv.DictReader(open(f)):
            if kern in r["Name"]:
                return float(r["AverageNs"]) / 1e6
    return None


def clock_ghz(name, kern):
    """the clock `kern` ran at IN the counter pass: GRBM_GUI_ACTIVE (summed over the 8 XCDs by the csv) / 8 / the dispatch's own
    End_Timestamp - Start_Timestamp, averaged over its dispatches"""
    fs = glob.glob("%s/pmc_%s_SQ_INSTS_SALU/*counter_collection.csv" % (dst, name))
    if not fs:
        return None, None
    v = [(float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(fs[0]))
         if kern in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    v = [(c, d) for c, d in v if d > 0]
    if not v:
        return None, None
    return sum(c / 8.0 / d for c, d in v) / len(v), sum(d for _, d in v) / len(v) / 1e6


def traffic(name, kern, algorithmic):
    f, nf = avg("%s/pmc_%s_FETCH_SIZE/*counter_collection.csv" % (dst, name), kern, "FETCH_SIZE")
    w, nw = avg("%s/pmc_%s_WRITE_SIZE/*counter_collection.csv" % (dst, name), kern, "WRITE_SIZE")
    if f is None or w is None:
        return None
    return {"kernel": kern, "fetch_size_kb_raw": f, "write_size_kb_raw": w, "launches_averaged": [nf, nw],
            "bytes_per_launch_uncorrected": (f + w) * 1024, "bytes_per_launch_fetch_x2": (2 * f + w) * 1024,
            "algorithmic_bytes_per_launch": algorithmic,
            "ratio_to_algorithmic": (2 * f + w) * 1024 / algorithmic}


out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/%s/pmc_*)" % RND + "; FETCH_SIZE on gfx950 under-reports wide "
               "coalesced reads by 2x (MI355X_MICROARCH.md), other widths uncalibrated, so both readings are given.  Traffic is counted "
               "at the fabric side of L2 (Infinity Cache hits included), per launch."}
for key, name, kern, algo, stats in (("k_miller_x60_altbn128", "bn_x60", "k_miller_x60<bgls::BN254", 1048576 * 192, "bn_x60_1048576"),
                                     ("k_miller_x60_bls12", "bls_x60", "k_miller_x60<bgls::BLS381", 1048576 * 256, "bls_x60_1048576"),
                                     ("k_sumpair_main_altbn128", "multisig", "k_sumpair_main", 1048576 * 128, "multisig_1048576")):
    t = traffic(name, kern, algo)
    if t:
        t.update(sq(name, kern))
        ghz, pmc_ms = clock_ghz(name, kern)
        if ghz:
            t["kernel_clock_ghz"] = ghz
            t["kernel_clock_inputs"] = {"launch_ms_in_the_counter_pass": pmc_ms, "rocprof_average_ms_stats_run": launch_ms(stats, kern)}
        t["counters_from"] = "profiles/%s (builder run)" % RND
        out[key] = t
json.dump(out, open(dst + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {"GB/launch x2": v["bytes_per_launch_fetch_x2"] / 1e9, "ratio": v["ratio_to_algorithmic"]}) for k, v in out.items() if k != "note"}, indent=1))

# VGPR / scratch / spill / LDS table of every kernel in the shipped library (tools/kernel_resources.py)
import subprocess
with open(dst + "/kernel_resources.txt", "w") as fh:
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_resources.py")], stdout=fh, stderr=subprocess.STDOUT, timeout=600)
