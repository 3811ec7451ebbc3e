"""Builds tests/harness/libhost_harness.so (TEST-ONLY: the device arithmetic headers compiled for the host).
host_harness.cpp is compiled as four parts in parallel (-DHT_PART=0..3) and linked: under two minutes instead of five for the
single translation unit.  Used by tests/conftest.py and __graft_entry__.build()."""
import os, subprocess, tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"
SRC = os.path.join(HERE, "host_harness.cpp")
SO = os.path.join(HERE, "libhost_harness.so")
PARTS = 4


def deps():
    csrc = os.path.join(ROOT, "bgls_amd", "csrc")
    return [SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp")]


def stale():
    return not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps())


def build(force=False, timeout=900):
    if not force and not stale():
        return SO
    with tempfile.TemporaryDirectory(prefix="bgls_harness_") as tmp:
        objs = [os.path.join(tmp, "part%d.o" % p) for p in range(PARTS)]

        def one(p):
            subprocess.run([CLANGXX, "-std=c++17", "-O1", "-fPIC", "-pthread", "-DHT_PART=%d" % p, "-c", SRC, "-o", objs[p]], check=True, timeout=timeout)

        with ThreadPoolExecutor(max_workers=min(PARTS, os.cpu_count() or 1)) as ex:
            list(ex.map(one, range(PARTS)))
        subprocess.run([CLANGXX, "-shared", "-fPIC", "-pthread", "-o", SO] + objs, check=True, timeout=timeout)
    return SO


if __name__ == "__main__":
    print(build(force=True))
