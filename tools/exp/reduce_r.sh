#!/bin/bash
show() { grep "^DETAIL \|^{" | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms/step %.4g' % d['ms_per_step'], {k:round(v,3) for k,v in d.get('stage_ms_exclusive',{}).items()})"; }
for R in 4 3 2; do
echo "== R=$R n=64"; BGLS_REDUCE_R=$R python bench.py --only small --n 64 2>/dev/null | show
echo "== R=$R 2^16 lone"; BGLS_REDUCE_R=$R python bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 6 2>/dev/null | grep "^DETAIL" | show
done
