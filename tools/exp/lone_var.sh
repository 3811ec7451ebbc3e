#!/bin/bash
show() { grep "^DETAIL " | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('seq med %.4g min %.4g' % (d['sequential']['ms_per_step_median'], d['sequential']['ms_per_step_min']), {k:round(v,3) for k,v in d.get('stage_ms_exclusive',{}).items()})"; }
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for i in 1 2 3 4; do python bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 8 2>/dev/null | show; done
for i in 1 2; do python bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 8 --curve bls12 2>/dev/null | show; done
