#!/bin/bash
# kernel times of ONE lone 2^16 verification per curve (rocprofv3 kernel trace): what the tails cost
export TMPDIR=/tmp
for c in ${1:-bls12 altbn128}; do
O=$PWD/gpurun_out/lone_$c; rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $OLDPWD/bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 10 --warmup 4 --curve $c > $O/t.log 2>&1)
python - <<P
import csv,glob
f=glob.glob("$O/t/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:18]:
    if int(r["Calls"])>=14: print("%5s %9.1f us  %s"%(r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:90]))
P
grep "^DETAIL" $O/t.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()[7:]); print('$c ms/step %.4g' % d['ms_per_step'], {k:round(v,3) for k,v in d.get('stage_ms_exclusive',{}).items()})"
done
