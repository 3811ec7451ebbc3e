#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the shipped library, read from the gfx950 code objects' metadata.
usage: python tools/kernel_resources.py [path/to/libbgls_hip.so] [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "bgls_amd", "libbgls_hip.so")
flt = [a for a in sys.argv[1:] if not a.endswith(".so")]
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    raw = open(lib, "rb").read()
    # every clang offload bundle in the library holds one gfx950 code object (an ELF); carve them out by their ELF headers
    rows = []
    pos = 0
    k = 0
    while True:
        pos = raw.find(b"\x7fELF\x02\x01\x01\x40", pos)          # ELFOSABI_AMDGPU_HSA
        if pos < 0:
            break
        path = os.path.join(d, "co%d.elf" % k)
        open(path, "wb").write(raw[pos:])
        txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "?"])[1]
            rows.append((g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count"),
                         g("group_segment_fixed_size")))
        pos += 8
        k += 1
print("%-78s %5s %5s %5s %8s %6s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratchB", "spills", "ldsB"))
for r in sorted(set(rows)):
    name = subprocess.run([LLVM + "/llvm-cxxfilt", r[0]], capture_output=True, text=True).stdout.strip() if os.path.exists(LLVM + "/llvm-cxxfilt") else r[0]
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("bgls::", "")
    if flt and not any(f in name for f in flt):
        continue
    print("%-78s %5s %5s %5s %8s %6s %6s" % (name[:78], r[1], r[2], r[3], r[4], r[5], r[6]))
