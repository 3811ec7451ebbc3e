#!/bin/bash
# Development: SQ counters of k_miller_x60 (whole / producer only / consumer only) from the microbenchmark binary.
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_x60
mkdir -p $O
BIN=$PWD/tools/mb_x60.bin
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o x60 -- $BIN x60 ${1:-61440} > $O/stats.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -o x60 -- $BIN x60 ${1:-61440} > $O/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        print(k, {c: "%.4g" % (v / max(1, cnt[(k, c)])) for c, v in agg[k].items()})
PY
