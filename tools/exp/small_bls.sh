#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/smallbls; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $OLDPWD/bench.py --only small --n 64 --curve bls12 > $O/t.log 2>&1)
python - <<P
import csv,glob
f=glob.glob("$O/t/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>=15: print("%5s %9.1f us  %s"%(r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:80]))
P
tail -1 $O/t.log | cut -c1-200
