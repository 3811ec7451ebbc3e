import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgls_amd import _lib, Altbn128
lib = _lib.load(); lib.bgls_init(0)
g1, g2 = Altbn128.GetG1(), Altbn128.GetG2()
for mode in (0, 3):
    os.environ["BGLS_FE_MODE"] = str(mode)
for i in range(3): Altbn128.Pair(g1, g2)
buf = (ctypes.c_ulonglong * 16)()
lib.bgls_debug_fe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
print("rc", lib.bgls_debug_fe(buf))
v = list(buf)
names = ["inv(+conj)", "rest of easy", "pow1", "pow2+pow3", "chain", "16 fe_mul", "16 cyclo_sqr", "16 frob"]
for i in range(8):
    print("%-14s %10d cycles" % (names[i], v[i + 1] - v[i]))

names2 = ["entry->loaded", "3 mul_wide", "store partner", "tree level 1", "final sums+redc", "xi publish"]
for i in range(6):
    print("  fe_mul %-16s %8d cycles" % (names2[i], v[10 + i] - v[9 + i]))
