#!/bin/bash
# Round-2 evidence run (on the GPU box): GPU tests, the default bench line, rocprofv3 kernel stats per BASELINE config with
# one verification in flight (kernels with the machine to themselves), FETCH_SIZE / WRITE_SIZE passes of the dominant
# kernels.  Output under gpurun_out/<tag>/; tools/refresh_profiles_r2.py copies the summaries into profiles/r2/.
export TMPDIR=/tmp
R=${1:-r2}
O=$PWD/gpurun_out/$R
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
prof() {   # name, then bench arguments
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o $name -- python $OLDPWD/bench.py "$@" > $O/stats_$name.log 2>&1)
}
pmc() {    # name, counter, then bench arguments
  local name=$1; local ctr=$2; shift; shift
  (cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -o $name -- python $OLDPWD/bench.py "$@" > $O/pmc_${name}_$ctr.log 2>&1)
}
SEQ="--only aggregate --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 2"
prof bn_ab64_65536 $SEQ --n 65536
BGLS_THROUGHPUT=1 prof bn_s60_61440 $SEQ --n 61440
prof bls_ab64_65536 $SEQ --n 65536 --curve bls12
prof multisig_1048576 --only multisig --n 1048576 --in-flight 1 --reps 1 --steps 5 --warmup 2
prof default_overlapped --no-cpu-baseline --no-records --reps 1 --steps 5 --warmup 2
prof bn_small_64 --only small --n 64
# bucket-method weighted sums / fixed-base key generation (tools/gpu_msm.py): timings and kernel stats
python tools/gpu_msm.py --sizes 65536,262144,1048576 > $O/bench_msm_altbn128.txt 2>&1
python tools/gpu_msm.py --sizes 65536,1048576 --curve bls12 > $O/bench_msm_bls12.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_msm_1048576 -o msm_1048576 -- python $OLDPWD/tools/gpu_msm.py --sizes 1048576 > $O/stats_msm_1048576.log 2>&1)
# the N = 2 code path on this one GPU (two ranks over gloo; numbers meaningless, the line's shape is what is kept)
BGLS_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --signers 65536 --no-cpu-baseline --reps 1 > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err
prof bn_prepared_1048576 $SEQ --n 1048576 --prepared --steps 3 --warmup 1
prof bls_prepared_1048576 $SEQ --n 1048576 --prepared --curve bls12 --steps 3 --warmup 1
for c in FETCH_SIZE WRITE_SIZE; do
  pmc bn_ab64 $c $SEQ --n 65536 --steps 2 --warmup 1
  BGLS_THROUGHPUT=1 pmc bn_s60 $c $SEQ --n 61440 --steps 2 --warmup 1
  pmc bls_ab64 $c $SEQ --n 65536 --curve bls12 --steps 2 --warmup 1
  pmc multisig $c --only multisig --n 1048576 --in-flight 1 --reps 1 --steps 2 --warmup 1
  pmc bn_prepared $c $SEQ --n 1048576 --prepared --steps 2 --warmup 1
done
pmc bn_s60 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" $SEQ --n 61440 --steps 2 --warmup 1 2>/dev/null
find $O -name "*.csv" | wc -l; du -sh $O
