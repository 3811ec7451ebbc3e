"""Source-level rules that cost measurable time when broken (no GPU needed).

Static LDS in front of a dynamic array: a kernel that declares `__shared__ T x;` AND uses `extern __shared__` gets the static
object at LDS offset 0 and the dynamic array behind it -- 4 bytes in for an `int` -- so every 16-byte slot access of the
finalx.hpp layout (fx_ld / fx_st: ds_read_b128 / ds_write_b128) becomes a misaligned access.  Round 4 found k_miller_latx doing
that since round 3 (Miller tail 0.45 -> 0.35 ms once the word moved into the dynamic array); the units built on that layout must
not declare static LDS."""
import os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bgls_amd", "csrc")
FX_UNITS = ["k_millerlatx.hip", "k_finalx.hip", "k_sumtree.hip", "finalx.hpp", "miller_x.hpp", "k_millerx_bn.hip", "k_millerx_bls.hip",
            "k_millerx64_bn.hip", "k_millerx64_bls.hip"]


def closure(name, seen):
    """the file and, transitively, every `#include "..."` of it that lives in csrc (a `__shared__` added in an included header
    would bring the misalignment back without touching the listed files)"""
    if name in seen:
        return
    seen.add(name)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(os.path.join(CSRC, name), encoding="utf-8").read(), flags=re.M):
        inc = os.path.basename(m.group(1))
        if os.path.exists(os.path.join(CSRC, inc)):
            closure(inc, seen)


def test_no_static_lds_in_the_units_that_read_dynamic_lds_in_16_byte_pieces():
    static_decl = re.compile(r"^\s*__shared__\s")
    files = set()
    for name in FX_UNITS:
        assert os.path.exists(os.path.join(CSRC, name)), name
        closure(name, files)
    assert {"rx_pair.hpp", "rx.hpp", "jac_coop.hpp", "points_inl.hpp"} <= files, "the include closure is not being followed"
    for name in sorted(files):
        for ln, line in enumerate(open(os.path.join(CSRC, name), encoding="utf-8"), 1):
            code = line.split("//")[0]
            assert not static_decl.search(code), "%s:%d declares static LDS in front of the dynamic array" % (name, ln)
