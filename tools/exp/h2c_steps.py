#!/usr/bin/env python3
"""Development: the phases of k_h2c_bn_wide (n = 64) on the H2C_DBG build of the library (hipcc -DH2C_DBG on k_hash.hip, linked
like tools/exp/latx_dbg_build.sh does): shader clocks of Keccak | x^3 + 3 | square root | check | vote."""
import ctypes, os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from bgls_amd import _lib
dbg = os.path.join(root, "tools", "exp", "libbgls_hip_h2cdbg.so")
_lib.LIB_PATH = dbg
import bench  # noqa: E402
sys.argv = ["bench.py", "--only", "small", "--n", "64"]
try:
    bench.main()
except SystemExit:
    pass
lib = ctypes.CDLL(dbg)
buf = (ctypes.c_ulonglong * 16)()
print("dump rc", lib.bgls_dbg_h2c_dump(buf))
names = ["keccak", "x^3+3 (32-bit)", "square root", "check", "vote .. exit of the loop", "store"]
for k, nm in enumerate(names):
    print("%-28s %8d clocks" % (nm, buf[k + 1] - buf[k]))
