#!/bin/bash
# A/B of the key-sum kernels (BGLS_SUMX=0: 32-bit limbs, 1: carry-free on one lane, 2: carry-free on lane pairs), on the GPU box
python -m pytest tests/test_gpu_keys.py tests/test_gpu_batch_multi.py tests/test_gpu_scheme.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
for c in altbn128 bls12; do for m in 0 1 2; do
  for what in multisig multisig_batch; do
    BGLS_SUMX=$m python bench.py --only $what --n 1048576 --curve $c --steps 20 --warmup 4 --reps 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$c mode $m $what', d.get('value'), d.get('ms_per_step'), {k:v for k,v in (d.get('stages_ms') or d.get('stages') or {}).items()} if isinstance(d.get('stages_ms') or d.get('stages'), dict) else '')"
  done; done; done
