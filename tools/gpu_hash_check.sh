export TMPDIR=/tmp
O=$PWD/gpurun_out/hash1; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "hash or h2c or config or aggregate" 2>&1 | tail -3
SEQ="--only aggregate --in-flight 1 --no-cpu-baseline --reps 1 --steps 5 --warmup 2"
for c in altbn128 bls12; do
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$c -o s -- python $OLDPWD/bench.py $SEQ --n 1048576 --curve $c > $O/st_$c.log 2>&1)
grep -h "k_h2c_bn_finish\|k_bls_sw_jacobi\|k_miller_x60" $O/st_$c/*/*kernel_stats.csv $O/st_$c/*kernel_stats.csv 2>/dev/null | awk -F'","' '{print substr($1,1,40), $2, $4}'
tail -1 $O/st_$c.log | cut -c1-300
done
