#!/bin/bash
# Development: LDS / wait / traffic counters of k_miller_x60 from the quick microbenchmark binaries (tools/mb_x60q.hip), one
# rocprofv3 --pmc pass per counter set (never combined with tracing).  usage: gpu_pmc_x60q.sh <tag> [n]
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-pmcq}
N=${2:-1048576}
mkdir -p $O
ROOT=$PWD
cd /tmp
for cv in bn bls; do
  BIN=$ROOT/tools/mb_x60q_60_$cv.bin
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --output-format csv -d $O/${cv}_$tag -o x -- $BIN 0 2 $N > $O/${cv}_$tag.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        d = {c: v / max(1, cnt[(k, c)]) for c, v in agg[k].items()}
        extra = ""
        if "SQ_LDS_BANK_CONFLICT" in d: extra = "  conflict/idx_active %.3f  wait_any/wave_cycles %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"])
        print(f.split("/")[-4] if len(f.split("/")) > 4 else f, k, {c: "%.4g" % v for c, v in d.items()}, extra)
PY
