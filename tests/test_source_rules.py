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


def test_no_static_lds_in_the_units_that_read_dynamic_lds_in_16_byte_pieces():
    static_decl = re.compile(r"^\s*__shared__\s")
    for name in FX_UNITS:
        path = os.path.join(CSRC, name)
        assert os.path.exists(path), name
        for ln, line in enumerate(open(path, encoding="utf-8"), 1):
            code = line.split("//")[0]
            assert not static_decl.search(code), "%s:%d declares static LDS in front of the dynamic array" % (name, ln)
