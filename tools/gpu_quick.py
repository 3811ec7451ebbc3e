"""Ad-hoc GPU sanity + timing (not a test): smoke, MAC peak probe, stage timings at a few n."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
from bgls_amd import _lib, Altbn128, Bls12, bgls, curves

t = time.time(); g.smoke(); print("smoke %.1fs" % (time.time() - t), flush=True)
L = _lib.load()
pk = ctypes.c_double()
print("probe rc", L.bgls_probe_mad_peak(ctypes.byref(pk)), "peak MAC/s %.3e" % pk.value, flush=True)

def instance(curve, n):
    import random
    rnd = random.Random(1234 + n)
    msgs = [rnd.randbytes(64) for _ in range(n)]
    sks = [rnd.randrange(1, curve.GetG1Order()) for _ in range(n)]
    g2 = curve.GetG2()
    keys = curves.ScalePoints([g2] * n, sks)
    hs = curve.HashToG1Batch(msgs)
    sigs = curves.ScalePoints(hs, sks)
    agg = curves.AggregatePoints(sigs)
    return agg, keys, msgs

for curve in (Altbn128, Bls12):
    for n in (64, 1024, 8192):
        t = time.time(); agg, keys, msgs = instance(curve, n); ts = time.time() - t
        L.bgls_profile_enable(1)
        t = time.time(); ok = bgls.VerifyAggregateSignature(curve, agg, keys, msgs); tv = time.time() - t
        t = time.time(); ok2 = bgls.VerifyAggregateSignature(curve, agg, keys, msgs); tv2 = time.time() - t
        msgs2 = list(msgs); msgs2[n // 2] = b"x" + msgs2[n // 2][1:]
        bad = bgls.VerifyAggregateSignature(curve, agg, keys, msgs2)
        st = {}
        for name in ("dup_check", "h2c", "miller", "reduce", "final_exp"):
            ms = ctypes.c_double(); cnt = ctypes.c_ulonglong()
            L.bgls_profile_get(name.encode(), ctypes.byref(ms), ctypes.byref(cnt))
            st[name] = round(ms.value / max(cnt.value, 1), 3)
        print(curve.Name(), "n=%d setup %.2fs verify %s/%s bad=%s wall %.1fms/%.1fms" % (n, ts, ok, ok2, bad, tv * 1e3, tv2 * 1e3), st, flush=True)
