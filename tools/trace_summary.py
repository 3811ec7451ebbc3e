#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace: per kernel calls / mean / min, and a coarse timeline of the last N ms.
usage: python tools/trace_summary.py <dir with *_kernel_trace.csv> [window_ms]"""
import csv, glob, sys, collections
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bgls::", "")[:46], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ev.sort()
agg = collections.defaultdict(list)
for s, e, n, q in ev:
    agg[n].append((e - s) / 1e6)
print("%-48s %6s %9s %9s %9s" % ("kernel", "calls", "mean ms", "min ms", "sum ms"))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-48s %6d %9.3f %9.3f %9.2f" % (n, len(v), sum(v) / len(v), min(v), sum(v)))
if win:
    t1 = ev[-1][1]
    t0 = t1 - int(win * 1e6)
    print("\ntimeline of the last %.1f ms (start offset ms, duration ms, kernel, queue):" % win)
    for s, e, n, q in ev:
        if s >= t0 and (e - s) > 20000:
            print("%9.3f %8.3f  %-46s %s" % ((s - t0) / 1e6, (e - s) / 1e6, n, q))
