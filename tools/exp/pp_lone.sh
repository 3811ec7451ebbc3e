#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/pp; mkdir -p $O
for skip in 8192 4096 3072 2048; do mode=hash; export BGLS_H2C_CAP=$skip
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/s$skip -o t -- python $OLDPWD/tools/exp/pp_lone.py $mode > $O/$mode.log 2>&1)
python - <<P
import csv,glob
f=glob.glob("$O/s$skip/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
d=[round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,2) for r in rows if "k_miller_x60" in r["Kernel_Name"]]
print("cap $skip: slow %d of %d" % (sum(1 for x in d if x > 5.6), len(d)), d[:12])
P
done
