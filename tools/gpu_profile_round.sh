#!/bin/bash
# Round evidence run (on the GPU box): tests, bench, rocprofv3 kernel stats, PMC passes.
export TMPDIR=/tmp
R=${1:-r1}
O=gpurun_out/$R
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench_altbn128.json 2> $O/bench_altbn128.err; tail -c 600 $O/bench_altbn128.json
python bench.py --curve bls12 --steps 10 --warmup 2 > $O/bench_bls12.json 2> $O/bench_bls12.err
python bench.py --in-flight 1 --no-cpu-baseline > $O/bench_altbn128_sequential.json 2>/dev/null
python bench.py --curve bls12 --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_bls12_sequential.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seq -o ${R}_seq -- python bench.py --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_seq_run.log 2>&1
python bench.py --workload multisig-hae --n 1048576 --steps 3 --warmup 1 > $O/bench_multisig_hae_altbn128_1M.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $R -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o $R -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o $R -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/pmc_sq -o $R -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_lds -o $R -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_lds.log 2>&1
find $O -name "*.csv" | head -30
du -sh $O
python bench.py --workload multisig --n 1048576 --steps 5 --warmup 2 > $O/bench_multisig_altbn128_1M.json 2> $O/bench_multisig.err; tail -c 400 $O/bench_multisig_altbn128_1M.json
python bench.py --n 1048576 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_altbn128_1M.json 2>/dev/null
python bench.py --curve bls12 --n 1048576 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_bls12_1M.json 2>/dev/null
