#!/bin/bash
# per-launch durations of the lone 64-form Miller round: consumer placement by claim (default) vs by arithmetic (BGLS_X60_CLAIM=0)
export TMPDIR=/tmp
O=$PWD/gpurun_out/bimodal; mkdir -p $O
for curve in altbn128 bls12; do for claim in 1 0; do
  (cd /tmp && BGLS_X60_CLAIM=$claim rocprofv3 --kernel-trace --output-format csv -d $O/c$claim$curve -o t -- python $OLDPWD/bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 12 --warmup 12 --curve $curve > $O/c$claim$curve.log 2>&1)
  python - <<P
import csv,glob
f=glob.glob("$O/c$claim$curve/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_miller_x60" in r["Kernel_Name"]]
print("$curve claim=$claim", [round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,2) for r in rows])
P
done; done
python -m pytest tests/test_gpu_x60.py -x -q -m gpu 2>&1 | tail -2
show() { grep "^DETAIL " | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
ex=d['roofline']['exclusive']
print('value %.5g  ms/step %.5g  seq %.5g  miller excl %.4g ms frac %.4g' % (d['value'], d['ms_per_step'], d['sequential']['ms_per_step_median'], ex['launch_ms'], ex['frac']))"; }
for claim in 1 0; do for c in altbn128 bls12; do echo "== $c 2^20 claim=$claim"; BGLS_X60_CLAIM=$claim python bench.py --only aggregate --n 1048576 --in-flight 4 --no-cpu-baseline --reps 2 --steps 8 --warmup 3 --curve $c 2>/dev/null | show; done; done
