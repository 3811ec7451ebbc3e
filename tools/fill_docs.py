#!/usr/bin/env python3
"""Fill the R4_* placeholders of DESIGN.md / README.md from an evidence run's bench output (the DETAIL line of
gpurun_out/<tag>/bench_default.out).  usage: python tools/fill_docs.py <tag> [--check]   (--check: only list what is left)"""
import json, re, sys
tag = sys.argv[1]
d = None
for l in open("gpurun_out/%s/bench_default.out" % tag):
    if l.startswith("DETAIL "):
        d = json.loads(l[7:])
r = d["records"]
ms, msk = r["altbn128_multisig_1048576"], r["altbn128_multisig_1048576_key_set"]
bls, b16, l16, s64 = r["bls12_1048576"], r["altbn128_65536"], r["bls12_65536"], r["altbn128_64"]
f = lambda x, n=2: ("%." + str(n) + "f") % x
v = {
    "R4TAG": tag,
    "R4_BN_V": f(d["value"] / 1e6, 1), "R4_BN_MS": f(d["ms_per_step"], 1), "R4_BN_K": f(d["roofline"]["exclusive"]["launch_ms"], 1), "R4_BN_F": f(d["roofline"]["exclusive"]["frac"]),
    "R4_BLS_V": f(bls["value"] / 1e6, 1), "R4_BLS_MS": f(bls["ms_per_step"], 1), "R4_BLS_K": f(bls["roofline"]["exclusive"]["launch_ms"], 1), "R4_BLS_F": f(bls["roofline"]["exclusive"]["frac"]),
    "R4_16MS": "%s / %s" % (f(b16["ms_per_step"], 1), f(l16["ms_per_step"], 1)), "R4_16": "%s / %s" % (f(b16["value"] / 1e6, 1), f(l16["value"] / 1e6, 1)),
    "R4_MSK_SEQ": f(msk["sequential"]["ms_per_step_median"]), "R4_MS_SEQ": f(ms["sequential"]["ms_per_step_median"]),
    "R4_MSK_V": f(msk["value"] / 1e9), "R4_MS_V": f(ms["value"] / 1e9), "R4_MS_MS": "%s / %s" % (f(ms["ms_per_step"]), f(msk["ms_per_step"])),
    "R4_MSK_K": f(msk["roofline"]["launch_ms"]), "R4_MS_K": f(ms["roofline"]["launch_ms"]), "R4_MSK_F": f(msk["roofline"]["frac"]), "R4_MS_F": f(ms["roofline"]["frac"]),
    "R4_MS_ST": f(ms["roofline"]["stage"]["stage_ms"]), "R4_MS_SF": f(ms["roofline"]["stage"]["frac"]),
    "R4_64_V": f(s64["value"] / 1e3, 1), "R4_64_MS": f(s64["ms_per_step"]),
}
for path in ("DESIGN.md", "README.md"):
    s = open(path).read()
    for k in sorted(v, key=len, reverse=True):
        s = s.replace(k, v[k])
    left = sorted(set(re.findall(r"R4_[A-Z0-9_]+|R4TAG", s)))
    print(path, "placeholders left:", left)
    if "--check" not in sys.argv:
        open(path, "w").write(s)
