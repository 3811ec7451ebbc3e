#!/usr/bin/env python3
"""Development: where does k_miller_latx spend a step?  Runs one multi-signature record and the n = 64 record on the
LATX_DBG build of the library (tools/exp/latx_dbg_build.sh writes tools/exp/libbgls_hip_latxdbg.so; see k_millerlatx.hip) and prints the 100 MHz time stamps of
the producer / consumer hand-overs of the last launch."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from bgls_amd import _lib
dbg = os.path.join(root, "tools", "exp", "libbgls_hip_latxdbg.so")
_lib.LIB_PATH = dbg
import bench  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "small"
verbose = "-v" in sys.argv
sys.argv = ["bench.py", "--only", "small", "--n", "64"] if which == "small" else \
    ["bench.py", "--only", "multisig", "--n", "1048576", "--in-flight", "1", "--reps", "1", "--steps", "3", "--warmup", "2"]
try:
    bench.main()
except SystemExit:
    pass
lib = ctypes.CDLL(dbg)
buf = (ctypes.c_ulonglong * (2 * 2 * 4 * 160 + 12 * 160))()
print("dump rc", lib.bgls_dbg_latx_dump(buf))
for blk in range(2):
    t = [[buf[(blk * 4 + k) * 160 + i] for i in range(160)] for k in range(4)]
    n = sum(1 for v in t[0] if v)
    if not n:
        continue
    t0 = t[0][0]
    c = [[buf[1280 + (blk * 4 + k) * 160 + i] for i in range(160)] for k in range(4)]
    print("block", "sig" if blk else "0", "events", n, "total us", (t[1][n - 1] - t0) / 100.0,
          "shader clocks per us", (c[1][n - 1] - c[0][0]) / ((t[1][n - 1] - t0) / 100.0))
    if not verbose:
        continue
    print("  step: producer work us | producer wait us || consumer work us | consumer wait us")
    for i in range(n):
        pw = (t[0][i] - (t[1][i - 1] if i else t0)) / 100.0
        pwait = (t[1][i] - t[0][i]) / 100.0
        cw = (t[2][i] - (t[3][i - 1] if i else t[2][0])) / 100.0 if t[2][i] else 0
        cwait = (t[3][i] - t[2][i]) / 100.0 if t[2][i] else 0
        print("  %3d  %6.2f %6.2f || %6.2f %6.2f" % (i, pw, pwait, cw, cwait))

r = [[buf[2560 + k * 160 + i] for i in range(160)] for k in range(12)]
print("doubling steps of the general block, shader clocks between the stamps 0..7 (loads | product 1 | store+sync | middle | product 2 | store+line+sync | tail):")
for i in range(160):
    if r[0][i] and r[4][i] > r[0][i] and not (r[7][i] > r[0][i] and r[5][i] > r[4][i]):
        print("  add %3d " % i + " ".join("%6d" % (r[k + 1][i] - r[k][i]) for k in range(4)), " total", r[4][i] - r[0][i])
    elif r[0][i] and r[7][i] > r[0][i]:
        print("  %3d " % i + " ".join("%6d" % (r[k + 1][i] - r[k][i]) for k in range(7)), " total", r[7][i] - r[0][i])
