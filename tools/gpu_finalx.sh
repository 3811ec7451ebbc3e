#!/bin/bash
# finalx.hpp against finalexp.hpp on the GPU box: parity tier, then kernel times of a small verification
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for c in altbn128 bls12; do for fx in 0 1; do
  (cd /tmp && rm -rf /tmp/st_$c$fx && BGLS_FINALX=$fx rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$c$fx -o s -- python $GRAFT_REPO_ROOT/bench.py --only small --n 64 --curve $c > /tmp/log_$c$fx 2>&1)
  tail -1 /tmp/log_$c$fx | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$c finalx=$fx n=64 ms', d.get('ms_per_step'))"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/st_$c$fx/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_final','k_miller_lat','k_h2c','k_bls_sw','k_bls_comb')): print('   %-50s %4s %9.3f ms'%(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e6))
PY
done; done
