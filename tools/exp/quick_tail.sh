#!/bin/bash
# quick look at the latency configs after a change to the tails: multisig check, lone 2^16 per curve, n = 64
python -m pytest tests/test_gpu_configs.py tests/test_gpu_keys.py tests/test_gpu_batch_multi.py tests/test_gpu_scheme.py tests/test_gpu_x60.py -x -q -m gpu 2>&1 | tail -3
show() { grep "^DETAIL \|^{" | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value %.4g  ms/step %.4g  seq %.4g' % (d['value'], d['ms_per_step'], d.get('sequential',{}).get('ms_per_step_median',0)), {k:round(v,3) for k,v in d.get('stage_ms_exclusive',{}).items()})"; }
echo "== multisig wire"; python bench.py --only multisig --n 1048576 --in-flight 16 --reps 1 --steps 32 --warmup 4 2>/dev/null | show
echo "== multisig key set"; python bench.py --only multisig --key-set --n 1048576 --in-flight 16 --reps 1 --steps 32 --warmup 4 2>/dev/null | show
for c in altbn128 bls12; do echo "== $c 2^16 lone"; python bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 6 --curve $c 2>/dev/null | show; done
echo "== n=64"; python bench.py --only small --n 64 2>/dev/null | show
