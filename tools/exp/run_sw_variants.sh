#!/bin/bash
# on the GPU box: time k_bls_sw_jacobi of each variant library in tools/exp/libsw_*.so (development only)
export TMPDIR=/tmp
O=gpurun_out/swv; mkdir -p $O
cp bgls_amd/libbgls_hip.so /tmp/ship.so
for v in "$@"; do
  if [ "$v" = ship ]; then cp /tmp/ship.so bgls_amd/libbgls_hip.so; else cp tools/exp/libsw_$v.so bgls_amd/libbgls_hip.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$v -o sw -- python tools/exp_hash_time.py > $O/$v.log 2>&1
  echo "== $v"; python3 tools/rocpd_stats.py $O/$v/sw_results.db k_bls_sw
done
cp /tmp/ship.so bgls_amd/libbgls_hip.so
