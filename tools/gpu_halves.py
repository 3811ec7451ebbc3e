"""Development tool: time the Miller kernel's producer and consumer halves separately (BGLS_MILLER_DBG=1/2 give
wrong results on purpose; this script only reads the stage timer)."""
import ctypes, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgls_amd import _lib
L = _lib.load(); assert L.bgls_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fp = 32 if cid == 0 else 48
rnd = random.Random(7)
# keys: n copies of g2 scaled would be slow to make here; any valid G2 points do for timing: use the generator
g2 = (ctypes.c_uint8 * (4 * fp))(); L.bgls_generator(cid, 2, g2)
keys = bytes(g2) * n
msgs = b"".join(rnd.randbytes(64) for _ in range(n))
off = (ctypes.c_uint64 * (n + 1))(*[64 * i for i in range(n + 1)])
g1 = (ctypes.c_uint8 * (2 * fp))(); L.bgls_generator(cid, 1, g1)
B = lambda b: (ctypes.c_uint8 * len(b)).from_buffer_copy(b)
kb, mb = B(keys), B(msgs)
for rep in range(2):
    L.bgls_verify_aggregate(cid, g1, kb, mb, off, n, 1)
L.bgls_profile_enable(1)
for rep in range(4):
    L.bgls_verify_aggregate(cid, g1, kb, mb, off, n, 1)
st = {}
for name in ("h2c", "miller", "reduce", "final_exp"):
    ms = ctypes.c_double(); cnt = ctypes.c_ulonglong()
    L.bgls_profile_get(name.encode(), ctypes.byref(ms), ctypes.byref(cnt))
    st[name] = round(ms.value / max(cnt.value, 1), 3)
print("DBG=%s n=%d curve=%d" % (os.environ.get("BGLS_MILLER_DBG", "0"), n, cid), st, flush=True)
