#!/bin/bash
# parity of the throughput Miller kernel + the two 2^20 records, one verification at a time and four in flight
python -m pytest tests/test_gpu_x60.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
show() { grep "^DETAIL " | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
ex=d['roofline']['exclusive']
print('value %.5g  ms/step %.5g  seq %.5g  miller excl %.4g ms frac %.4g' % (d['value'], d['ms_per_step'], d['sequential']['ms_per_step_median'], ex['launch_ms'], ex['frac']))"; }
for c in altbn128 bls12; do echo "== $c 2^20"; python bench.py --only aggregate --n 1048576 --in-flight 4 --no-cpu-baseline --reps 2 --steps 8 --warmup 3 --curve $c 2>/dev/null | show; done
