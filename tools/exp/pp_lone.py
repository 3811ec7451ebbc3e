"""lone 2^16 Miller launches through bgls_pairing_product, optionally with a host-synchronised bgls_hash_to_g1 of 2^16 messages before each"""
import ctypes, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bgls_amd import _lib
L = _lib.load(); assert L.bgls_init(0) == 0
B = lambda b: (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b))
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
cid = 1 if mode.endswith("mbls") else 0
hcid = 1 if mode.startswith("hbls") else 0
fp, n = (48 if cid else 32), 65536
rnd = random.Random(3)
kb = B(b"".join(rnd.randrange(1, 1 << 250).to_bytes(32, "big") for _ in range(n)))
keys = (ctypes.c_uint8 * (n * 4 * fp))(); assert L.bgls_scale_generator(cid, 2, kb, n, keys) == 0
g1s = (ctypes.c_uint8 * (n * 2 * fp))(); assert L.bgls_scale_generator(cid, 1, kb, n, g1s) == 0
gt = (ctypes.c_uint8 * (12 * fp))()
blob = B(rnd.randbytes(64 * n)); off = (ctypes.c_uint64 * (n + 1))(*range(0, 64 * (n + 1), 64)); hs = (ctypes.c_uint8 * (n * 2 * fp))()
hs = (ctypes.c_uint8 * (n * 2 * 48))()
for _ in range(24):
    if mode == "hash" or mode.startswith("hb"): L.bgls_hash_to_g1(hcid, blob, off, n, hs)
    if mode == "hash_small": assert L.bgls_hash_to_g1(cid, blob, off, 200, hs) == 0
    assert L.bgls_pairing_product(cid, g1s, keys, n, gt) == 0
print("done")
