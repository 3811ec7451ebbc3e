"""GPU tier: the C++ host mirror (include/bgls/curves.hpp, include/bgls/bgls.hpp) running the reference's
scheme tests against the HIP library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_mirror_reference_scheme_tests(gpu_lib, tmp_path):
    exe = str(tmp_path / "test_bgls")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_bgls.cpp"),
                    "-L", os.path.join(ROOT, "bgls_amd"), "-lbgls_hip", "-Wl,-rpath," + os.path.join(ROOT, "bgls_amd"), "-o", exe],
                   check=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


def test_cpp_mirror_compiles():
    """CPU tier: the header-only mirror compiles against the C ABI (no GPU needed to build)."""
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "test_bgls.cpp")], check=True, timeout=300)
