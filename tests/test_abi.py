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


def test_exception_barrier_at_the_c_abi():
    """include/bgls_hip.h promises "no exceptions, no abort()" (the reference never panics: curves/curve.go:15-22).  Every
    extern "C" body is a function-try-block; a C++ exception raised inside the library -- on the calling thread or on a shard's
    host thread -- comes back as a negative code."""
    from bgls_amd import _lib
    lib = _lib.load()
    assert lib.bgls_selftest_exception_barrier(0) == -6 and "memory" in _lib.last_error()
    assert lib.bgls_selftest_exception_barrier(1) == -6
    assert lib.bgls_selftest_exception_barrier(2) == -5
    assert lib.bgls_selftest_exception_barrier(3) == -1 and "selftest" in _lib.last_error()
    assert lib.bgls_selftest_exception_barrier(4) == -1
    assert lib.bgls_selftest_exception_barrier(5) == -6          # a real allocation failure, not a thrown stand-in
    assert lib.bgls_selftest_exception_barrier(6) == -6          # raised on a second host thread, joined, reported
    assert lib.bgls_selftest_exception_barrier(7) == -6 and _lib.last_error()          # selection restored by the guard (else -1); the message survives
    assert lib.bgls_selftest_exception_barrier(99) == 0
    # an absurd batch size is refused before anything is sized by it
    o = (ctypes.c_uint8 * 64)()
    off = (ctypes.c_uint64 * 2)(0, 0)
    assert lib.bgls_verify_aggregate(0, o, o, o, off, (1 << 30) - 1 + 1, 0) == -1
    assert lib.bgls_hash_to_g1(0, o, off, 1 << 40, o) == -1


def test_every_abi_body_is_guarded():
    """Source rule: each `int bgls_*(...)` defined in engine's extern "C" block is a function-try-block closed by BGLS_ABI_GUARD
    (one-line accessors that cannot throw are exempt)."""
    src = ""
    csrc = os.path.join(ROOT, "bgls_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".inc")):
            src += open(os.path.join(csrc, f)).read()
    block = src[src.index('extern "C" {'):]
    names = [m.group(1) for m in re.finditer(r"^int (bgls_[a-z0-9_]+)\(", block, flags=re.M)]
    assert len(names) >= 55
    for m in re.finditer(r"^int (bgls_[a-z0-9_]+)\(([^{;]*)\)\s*(try\s*)?\{([^\n]*)$", block, flags=re.M):
        name, guarded, rest = m.group(1), m.group(3), m.group(4)
        one_liner = rest.strip().endswith("}")
        assert guarded or one_liner, name + " is not a function-try-block"
    assert block.count("} BGLS_ABI_GUARD") + block.count("BGLS_ABI_GUARD\n") >= 55
