#!/bin/bash
# round 5, first GPU call: parity of k_miller_x60 on the 29-bit form, issue-rate probes (FP64 / 64-bit adds), A/B of the kernel, headline record
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_x60.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5/pytest_x60.log
cat gpurun_out/r5/pytest_x60.log
./tools/mb_x60.bin issue > gpurun_out/r5/mb_issue.txt 2>&1
grep "waves/SIMD 3" gpurun_out/r5/mb_issue.txt | grep -i "mad_u64_u32 (vcc\|fma\|add_co\|FP64\|lshl_add\|add_f64\|v_add_u32 \|mul_lo"
./tools/mb_x60.bin x60 1048560 0 > gpurun_out/r5/mb_x60_1m.txt 2>&1; cat gpurun_out/r5/mb_x60_1m.txt
./tools/mb_x60.bin x60 61440 0 > gpurun_out/r5/mb_x60_61440.txt 2>&1; cat gpurun_out/r5/mb_x60_61440.txt
show() { grep "^DETAIL " | tail -1 | sed 's/^DETAIL //' | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
ex=d['roofline']['exclusive']
print('value %.5g  ms/step %.5g  seq %.5g  miller excl %.4g ms frac %.4g' % (d['value'], d['ms_per_step'], d['sequential']['ms_per_step_median'], ex['launch_ms'], ex['frac']))"; }
for c in altbn128 bls12; do echo "== $c 2^20"; python bench.py --only aggregate --n 1048576 --in-flight 4 --no-cpu-baseline --reps 2 --steps 8 --warmup 3 --curve $c 2>gpurun_out/r5/bench_$c.err | tee gpurun_out/r5/bench_$c.out | show; done
