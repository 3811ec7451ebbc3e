#!/usr/bin/env python3
"""Per-kernel statistics out of a rocprofv3 rocpd database (the default output format): name, calls, average / min / max ms.
usage: rocpd_stats.py <results.db> [substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
nm = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, count(*), avg(end - start), min(end - start), max(end - start), sum(end - start) from kernels group by %s order by 6 desc" % (nm, nm)).fetchall()
for n, c, a, lo, hi, s in rows:
    if sub in n:
        print("%-70s %6d  avg %9.3f  min %9.3f  max %9.3f ms" % (n[:70], c, a / 1e6, lo / 1e6, hi / 1e6))
