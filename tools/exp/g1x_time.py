"""wall time of the G1 entry points at the seam (sign_batch, hash_to_g1, scale_points G1), carry-free (default) vs BGLS_G1X=0"""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bgls_amd import _lib
L = _lib.load(); assert L.bgls_init(0) == 0
B = lambda b: (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b))
n = 1 << 18
rnd = random.Random(1)
kb = B(b"".join(rnd.randrange(1, 1 << 250).to_bytes(32, "big") for _ in range(n)))
blob = B(rnd.randbytes(64 * n)); off = (ctypes.c_uint64 * (n + 1))(*range(0, 64 * (n + 1), 64))
for cid, fp in ((0, 32), (1, 48)):
    sg = (ctypes.c_uint8 * (n * 2 * fp))(); hs = (ctypes.c_uint8 * (n * 2 * fp))(); sc = (ctypes.c_uint8 * (n * 2 * fp))()
    for name, f in (("sign_batch", lambda: L.bgls_sign_batch(cid, kb, blob, off, n, sg)), ("hash_to_g1", lambda: L.bgls_hash_to_g1(cid, blob, off, n, hs)),
                    ("scale_points_g1", lambda: L.bgls_scale_points(cid, 1, hs, kb, None, n, sc))):
        assert f() == 0
        t0 = time.perf_counter(); assert f() == 0; dt = time.perf_counter() - t0
        print("curve %d %-16s n = 2^18: %.1f ms" % (cid, name, dt * 1e3))
    import hashlib
    print("  digest", hashlib.sha256(bytes(sg) + bytes(hs) + bytes(sc)).hexdigest()[:16])
