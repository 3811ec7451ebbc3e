#!/usr/bin/env python3
"""Development: where does k_sum_tree_x spend its time?  One multi-signature record on the TREEX_DBG build of the library
(hipcc -DTREEX_DBG on k_sumtree.hip, linked like tools/exp/latx_dbg_build.sh does) and the shader-clock stamps of the waves that
carried sums upwards, per level: park | ticket | fetch | add."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from bgls_amd import _lib
dbg = os.path.join(root, "tools", "exp", "libbgls_hip_treexdbg.so")
_lib.LIB_PATH = dbg
import bench  # noqa: E402
sys.argv = ["bench.py", "--only", "multisig", "--n", "1048576", "--in-flight", "1", "--reps", "1", "--steps", "3", "--warmup", "2"]
try:
    bench.main()
except SystemExit:
    pass
lib = ctypes.CDLL(dbg)
buf = (ctypes.c_ulonglong * (20 * 8))()
print("dump rc", lib.bgls_dbg_treex_dump(buf))
t = [[buf[l * 8 + k] for k in range(8)] for l in range(20)]
print("leaves: loads %d, first addition %d clocks" % (t[0][1] - t[0][0], t[0][2] - t[0][1]))
for l in range(1, 19):
    if t[l][4]:
        print("level %2d: park %6d  ticket %6d  fetch %6d  add %6d   (since the level below: %d)" % (l, t[l][1] - t[l][0], t[l][2] - t[l][1], t[l][3] - t[l][2], t[l][4] - t[l][3], t[l][0] - (t[l - 1][4] if l > 1 else t[0][2])))
print("root: to 32-bit record %d, to affine bytes %d clocks; whole climb %d clocks" % (t[19][1] - t[19][0], t[19][2] - t[19][1], t[19][2] - t[0][0]))
