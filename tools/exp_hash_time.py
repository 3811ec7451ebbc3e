"""rocprof target: hash 2^20 32-byte messages to G1 on BLS12-381 (k_bls_sw_jacobi over 2^21 items), three times."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bgls_amd import curves
n = 1 << 20
rng = np.random.default_rng(7)
msgs = [bytes(m) for m in rng.integers(0, 256, size=(n, 32), dtype=np.uint8)]
c = curves.Bls12
for _ in range(3):
    out = c.HashToG1Batch(msgs)
print("ok", len(out))
