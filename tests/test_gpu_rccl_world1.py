"""GPU tier: the N-rank exchange steps of bgls_amd/sharding.py through the real RCCL backend (backend "nccl") with a process group of ONE rank.
A one-GPU box cannot hold two RCCL ranks, and the two-rank tests of tests/test_dist_cpu.py run over gloo: this is the run that shows
ProcessGroupNCCL accepts the collectives as they are issued (uint8 all-gather of odd length, int32 status words viewed as bytes, uint8 all-to-all,
one-word int32 MAX, float64 MAX, barrier) around the library's own digest / pack / scan kernels.  The check runs in a subprocess under a timeout;
an RCCL that cannot initialise on the box at all (exit code 3) is reported as a skip with its message, any later failure fails the test."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_exchange_steps_through_rccl_with_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_check.py"), str(port)], capture_output=True, text=True, timeout=300, env=env)
    if p.returncode == 3:
        pytest.skip("RCCL did not initialise on this box: " + p.stdout.strip()[-300:])
    assert p.returncode == 0 and "rccl world-1 ok" in p.stdout, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
