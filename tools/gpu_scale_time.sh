#!/bin/bash
# kernel times of the per-point scalar multiplications (bench instance setup: 2^16 signatures / keys per curve) + the tests that pin them
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scheme.py tests/test_gpu_hae.py -x -q 2>&1 | tail -2
for c in altbn128 bls12; do
  (cd /tmp && rm -rf /tmp/sc_$c && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc_$c -o s -- python $GRAFT_REPO_ROOT/bench.py --only aggregate --n 65536 --curve $c --in-flight 1 --no-cpu-baseline --reps 1 --steps 2 --warmup 1 > /tmp/sc_$c.log 2>&1)
  python - <<PY
import csv,glob
f=glob.glob('/tmp/sc_$c/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_scale','k_fb_scale')): print('$c   %-60s %4s %9.3f ms'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6))
PY
done
