#!/bin/bash
# fan-in of the reduce passes that run on the 36-lane product (k_reduce_fx)
show() { grep "^DETAIL \|^{" | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms/step %.4g' % d['ms_per_step'], {k:round(v,3) for k,v in d.get('stage_ms_exclusive',{}).items() if k in ('reduce',)})"; }
for R in 6 10 12; do
echo -n "R=$R n=64: "; BGLS_REDUCEX_R=$R python bench.py --only small --n 64 2>/dev/null | show
for c in altbn128 bls12; do echo -n "R=$R $c 2^16 lone: "; BGLS_REDUCEX_R=$R python bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 10 --warmup 6 --curve $c 2>/dev/null | grep "^DETAIL" | show; done
done
