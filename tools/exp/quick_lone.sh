#!/bin/bash
# lone 2^16 verification per curve, with and without the signature pair on the side stream
show() { grep "^DETAIL " | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms/step %.4g  seq med %.4g min %.4g' % (d['ms_per_step'], d['sequential']['ms_per_step_median'], d['sequential']['ms_per_step_min']), {k:round(v,3) for k,v in d.get('stage_ms_exclusive',{}).items()})"; }
for rep in 1 2; do
for c in altbn128 bls12; do for f in 1 0; do echo "== $c 2^16 lone, BGLS_SIG_FORK=$f"; BGLS_SIG_FORK=$f python bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 10 --warmup 10 --curve $c 2>/dev/null | show; done; done
done
