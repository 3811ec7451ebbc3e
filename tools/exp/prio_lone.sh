#!/bin/bash
# lone 2^16 verification: role / priority modes of k_miller_x60 (64-pairing form)
for curve in altbn128 bls12; do
for mode in 24 20 16 28; do
  echo "== $curve mode $mode"
  BGLS_MILLER_SHAPE=4 BGLS_X60_ROT=$mode python bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 6 --curve $curve 2>/tmp/err.txt | grep "^DETAIL " | python -c "
import json,sys
d=json.loads(sys.stdin.readline()[7:])
print('seq', round(d['sequential']['ms_per_step_median'],3), 'miller', round(d['stage_ms_exclusive']['miller'],3), 'excl frac', round(d['roofline']['exclusive']['frac'],3))"
  tail -2 /tmp/err.txt
done; done
