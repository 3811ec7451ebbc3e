#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/stagger; mkdir -p $O
run() { curve=$1; stg=$2
(cd /tmp && BGLS_X60_STAGGER=$stg rocprofv3 --kernel-trace --output-format csv -d $O/$curve$stg -o t -- python $OLDPWD/bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 10 --warmup 10 --curve $curve > $O/$curve$stg.log 2>&1)
python - <<P
import csv,glob,statistics
f=glob.glob("$O/$curve$stg/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_miller_x60" in r["Kernel_Name"]]
d=sorted(round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,2) for r in rows)
print("$curve stagger $stg: min %.2f  q1 %.2f  median %.2f  max %.2f" % (d[0], d[len(d)//4], statistics.median(d), d[-1]))
P
}
for s in 0 600 1250 2500; do run altbn128 $s; done
for s in 0 1150 2300 4600; do run bls12 $s; done
