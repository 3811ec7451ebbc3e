#!/usr/bin/env python3
"""Development: the phases of k_finalx (alt-bn128, n = 64 record) on the FX_DBG build of the library (hipcc -DFX_DBG on k_finalx.hip,
linked like tools/exp/latx_dbg_build.sh does): shader clocks."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from bgls_amd import _lib
dbg = os.path.join(root, "tools", "exp", "libbgls_hip_fxdbg.so")
_lib.LIB_PATH = dbg
import bench  # noqa: E402
sys.argv = ["bench.py", "--only", "small", "--n", "64"]
try:
    bench.main()
except SystemExit:
    pass
lib = ctypes.CDLL(dbg)
t = (ctypes.c_ulonglong * 16)()
print("dump rc", lib.bgls_dbg_fx_dump(t))
seg = [("parse the partial", 0, 1), ("inversion: norms down to Fp2 (4 products, 2 maps)", 1, 9), ("inversion: the Fp2 inverse and N^-1", 9, 10), ("inversion: last product", 10, 2),
       ("rest of the easy part (2 products, 1 map)", 2, 3), ("f^u", 3, 4), ("f^(u^2)", 4, 5), ("f^(u^3)", 5, 6), ("hard part's chain (7 maps, 12 products deep)", 6, 7),
       ("serialise, compare", 8, 11), ("whole kernel", 0, 11)]
for nm, a, b in seg:
    print("%-52s %8d clocks  %6.1f us" % (nm, t[b] - t[a], (t[b] - t[a]) / 2400.0))
