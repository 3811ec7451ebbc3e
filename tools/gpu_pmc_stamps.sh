#!/bin/bash
# Development (on the GPU box): SQ / instruction-cache counters of k_miller_x60 from tools/mb_stamps*.bin, one --pmc pass per counter set.
# usage: tools/gpu_pmc_stamps.sh <out dir under gpurun_out> <binary> [n]
export TMPDIR=/tmp
O=$PWD/gpurun_out/$1
BIN=$PWD/$2
N=${3:-1048560}
mkdir -p $O
cd /tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -o p -- $BIN $N /tmp/stamps_pmc > $O/pmc_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_miller_x60" not in k: continue
        k = k[k.find("k_miller_x60"):][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k in sorted(agg):
        print(k, {c: "%.5g" % (v / max(1, cnt[(k, c)])) for c, v in agg[k].items()})
PY
