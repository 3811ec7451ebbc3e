"""CPU tier: the C-ABI library loads, exports every symbol include/bgls_hip.h declares, and fails
loudly (BGLS_ERR_NO_DEVICE) instead of falling back when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "bgls_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bgls_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    from bgls_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_sizes_and_identity():
    from bgls_amd import _lib
    lib = _lib.load()
    assert lib.bgls_abi_version() == 2
    assert [lib.bgls_fp_size(c) for c in (0, 1, 7)] == [32, 48, 0]
    assert (lib.bgls_g1_size(0), lib.bgls_g2_size(0), lib.bgls_gt_size(0)) == (64, 128, 384)
    assert (lib.bgls_g1_size(1), lib.bgls_g2_size(1), lib.bgls_gt_size(1)) == (96, 192, 576)
    o = (ctypes.c_uint8 * 384)()
    assert lib.bgls_gt_identity(0, o) == 0 and bytes(o) == bytes(383) + b"\x01"


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bgls_amd import _lib
    lib = _lib.load()
    o = (ctypes.c_uint8 * 64)()
    rc = lib.bgls_generator(0, 1, o)
    assert rc == -4 and "device" in _lib.last_error().lower()
    assert lib.bgls_verify_multi(0, o, o, 0, o, 0) == -4


def test_product_never_imports_oracle():
    """The product path must not route through the checker."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bgls_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle/", "ORACLEDIR/") or "import" not in txt or \
                    not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "liboracle" not in txt, f
