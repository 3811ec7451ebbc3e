#!/bin/bash
# A round's evidence run (on the GPU box): GPU tests, the default bench line, rocprofv3 kernel stats per BASELINE config with one
# verification in flight (kernels with the machine to themselves), FETCH_SIZE / WRITE_SIZE and SQ passes of the dominant
# kernels (one --pmc pass per counter set, never combined with tracing).  Output under gpurun_out/<tag>/;
# `tools/refresh_profiles.py --round N <tag>` copies the summaries into profiles/rN/.  (One script for every round since round 6;
# rounds 2-5 kept near-identical copies.)
# usage: tools/gpu_profile_round.sh <tag>        e.g. r6g
export TMPDIR=/tmp
R=${1:?tag}
O=$PWD/gpurun_out/$R
mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log; fi
python bench.py --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err; tail -1 $O/bench_default.out > $O/bench_default.json; tail -c 600 $O/bench_default.json; echo
prof() {   # name, then bench arguments
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o $name -- python $OLDPWD/bench.py "$@" > $O/stats_$name.log 2>&1)
}
pmc() {    # name, counters, then bench arguments
  local name=$1; local ctr=$2; shift; shift
  local tag=$(echo $ctr | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${name}_$tag -o $name -- python $OLDPWD/bench.py "$@" > $O/pmc_${name}_$tag.log 2>&1)
}
SEQ="--only aggregate --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 2"
prof bn_x60_1048576 $SEQ --n 1048576
prof bls_x60_1048576 $SEQ --n 1048576 --curve bls12
prof bn_x64_65536 $SEQ --n 65536
prof bls_x64_65536 $SEQ --n 65536 --curve bls12
prof multisig_1048576 --only multisig --n 1048576 --in-flight 1 --reps 1 --steps 5 --warmup 2
prof multisig_keyset_1048576 --only multisig --key-set --n 1048576 --in-flight 1 --reps 1 --steps 5 --warmup 2
prof default_overlapped --no-cpu-baseline --no-records --reps 1 --steps 5 --warmup 2
prof bn_small_64 --only small --n 64
prof bn_prepared_1048576 $SEQ --n 1048576 --prepared
prof bls_prepared_1048576 $SEQ --n 1048576 --curve bls12 --prepared
for c in FETCH_SIZE WRITE_SIZE; do
  pmc bn_x60 $c $SEQ --n 1048576 --steps 2 --warmup 1
  pmc bls_x60 $c $SEQ --n 1048576 --curve bls12 --steps 2 --warmup 1
  pmc multisig $c --only multisig --n 1048576 --in-flight 1 --reps 1 --steps 2 --warmup 1
done
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  pmc bn_x60 "$set" $SEQ --n 1048576 --steps 2 --warmup 1
  pmc bls_x60 "$set" $SEQ --n 1048576 --curve bls12 --steps 2 --warmup 1
  pmc multisig "$set" --only multisig --n 1048576 --in-flight 1 --reps 1 --steps 2 --warmup 1
done
BGLS_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --signers 65536 --no-cpu-baseline --reps 1 > $O/bench_two_ranks_one_gpu.out 2> $O/bench_two_ranks_one_gpu.err; tail -1 $O/bench_two_ranks_one_gpu.out > $O/bench_two_ranks_one_gpu.json
find $O -name "*.csv" | wc -l; du -sh $O
