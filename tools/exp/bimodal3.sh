#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/bimodal3; mkdir -p $O
run() { tag=$1; shift
(cd /tmp && env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python $OLDPWD/bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 12 --warmup 12 > $O/$tag.log 2>&1)
python - <<P
import csv,glob
f=glob.glob("$O/$tag/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_miller_x60" in r["Kernel_Name"]]
d=[round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,2) for r in rows]
print("$tag", "slow %d of %d" % (sum(1 for x in d if x > 5.6), len(d)), d[:16])
P
}
run q1 GPU_MAX_HW_QUEUES=1
run q4 GPU_MAX_HW_QUEUES=4
run q16 GPU_MAX_HW_QUEUES=16
run np60 BGLS_X_NP=60
