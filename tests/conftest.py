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
    import importlib.util
    spec = importlib.util.spec_from_file_location("build_harness", os.path.join(ROOT, "tests", "harness", "build_harness.py"))
    bh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bh)
    so = bh.build()
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def gpu_lib():
    from bgls_amd import _lib
    lib = _lib.load()
    rc = lib.bgls_init(0)
    assert rc == 0, "bgls_init failed on the GPU box: %d %s" % (rc, _lib.last_error())
    return lib
