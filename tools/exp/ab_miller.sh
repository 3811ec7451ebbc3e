#!/bin/bash
# same-box A/B of two libraries on the Miller kernel (development): tools/exp/ab_miller.sh <variant.so> ; alternates shipped / variant twice
export TMPDIR=/tmp
O=$PWD/gpurun_out/abm; mkdir -p $O
cp bgls_amd/libbgls_hip.so /tmp/ship.so
V=$1
for round in 1 2; do
for v in ship var; do
  if [ $v = ship ]; then cp /tmp/ship.so bgls_amd/libbgls_hip.so; else cp $V bgls_amd/libbgls_hip.so; fi
  for cv in altbn128 bls12; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${v}_${cv}_$round -o m -- python $OLDPWD/bench.py --only aggregate --in-flight 1 --no-cpu-baseline --reps 1 --steps 6 --warmup 2 --n 1048576 --curve $cv > $O/${v}_${cv}_$round.log 2>&1)
    echo "== $v $cv $round: $(python3 tools/rocpd_stats.py $O/${v}_${cv}_$round/m_results.db k_miller_x60 | cut -c1-40,70-140)"
  done
done
done
cp /tmp/ship.so bgls_amd/libbgls_hip.so
