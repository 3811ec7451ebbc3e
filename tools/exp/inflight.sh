#!/bin/bash
show() { grep "^DETAIL " | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value %.5g  ms/step %.5g' % (d['value'], d['ms_per_step']))"; }
for L in 2 3 4 6 8; do for c in altbn128 bls12; do echo -n "in flight $L $c: "; python bench.py --only aggregate --n 1048576 --in-flight $L --no-cpu-baseline --reps 2 --steps 12 --warmup 2 --curve $c 2>/dev/null | show; done; done
