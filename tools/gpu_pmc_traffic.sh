export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_tr
mkdir -p $O
BIN=$PWD/tools/mb_x60.bin
$BIN x60 1048576 8 2>&1 | grep -v NC=2
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -o t -- $BIN x60 184320 8 > $O/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][20:60]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(c, k, "avg KB %.0f over %d launches -> per 2^20: %.1f GB" % (sum(v)/len(v), len(v), sum(v)/len(v)*1024/184320*1048576/1e9))
PY
