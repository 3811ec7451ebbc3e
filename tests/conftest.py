import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CURVES = [(0, "altbn128"), (1, "bls12")]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat():
    return load_golden("h2c_kat.json")


@pytest.fixture(scope="session", params=CURVES, ids=[c[1] for c in CURVES])
def curve(request):
    cid, name = request.param
    return {"id": cid, "name": name, "vec": load_golden("vectors_%s.json" % name), "fp": 32 if cid == 0 else 48}


@pytest.fixture(scope="session")
def host_harness():
    """The device arithmetic headers compiled for the host (test-only build)."""
    so = os.path.join(ROOT, "tests", "harness", "libhost_harness.so")
    src = os.path.join(ROOT, "tests", "harness", "host_harness.cpp")
    csrc = os.path.join(ROOT, "bgls_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-shared", "-fPIC", "-pthread", "-o", so, src],
                       check=True, timeout=900)
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def gpu_lib():
    from bgls_amd import _lib
    lib = _lib.load()
    rc = lib.bgls_init(0)
    assert rc == 0, "bgls_init failed on the GPU box: %d %s" % (rc, _lib.last_error())
    return lib
