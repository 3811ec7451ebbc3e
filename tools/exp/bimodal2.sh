#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/bimodal2; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $OLDPWD/bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 12 --warmup 12 > $O/t.log 2>&1)
python - <<P
import csv,glob
f=glob.glob("$O/t/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_miller_x60" in r["Kernel_Name"]]
print([round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,2) for r in rows])
P
