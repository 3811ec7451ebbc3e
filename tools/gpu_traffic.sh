#!/bin/bash
# quick: timing + fabric traffic of the Miller kernel for the current build
export TMPDIR=/tmp
O=gpurun_out/traffic; rm -rf $O; mkdir -p $O
python bench.py --steps 5 --warmup 2 --no-cpu-baseline $@ 2>/dev/null | grep -o "\"value\": [0-9.]*\|\"stage_ms_per_step\".*" | cut -c1-220
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline $@ > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline $@ > /dev/null 2>&1
python3 - <<'PY'
import csv
for c, d in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"gpurun_out/traffic/{d}/t_counter_collection.csv")) if "k_miller" in r["Kernel_Name"] and r["Counter_Name"] == c]
    print(c, "GB/launch %.2f" % (sum(v) / len(v) / 1e6), len(v))
PY
