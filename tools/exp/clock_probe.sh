#!/bin/bash
# slow lone launches inside the real flow: more wave-cycles (everything slower) or the same wave-cycles over a longer time (late blocks)?
export TMPDIR=/tmp
O=$PWD/gpurun_out/clk; mkdir -p $O
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc2 -o t -- python $OLDPWD/bench.py --only aggregate --n 65536 --in-flight 1 --no-cpu-baseline --reps 1 --steps 12 --warmup 12 > $O/pmc2.log 2>&1)
python - <<P
import csv,glob,collections
f=glob.glob("$O/pmc2/**/*counter_collection.csv", recursive=True)[0]
d=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "k_miller_x60" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"]); d[r["Dispatch_Id"]]["dur"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k,v in list(d.items())[:30]:
    print("%.2f ms"%v["dur"], {a:round(b/1e6,1) for a,b in v.items() if a!="dur"})
P
