#!/usr/bin/env python3
"""Development: one multi-signature record (BASELINE config 4) per BGLS_SUM_WAVES value given on the command line, key numbers only."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for w in (sys.argv[1:] or ["3072"]):
    extra = ["--key-set"] if w.endswith("k") else []
    env = dict(os.environ, BGLS_SUM_WAVES=w.rstrip("k"))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--only", "multisig"] + extra + [ "--n", "1048576", "--in-flight", "1", "--reps", "1", "--steps", "5", "--warmup", "4"],
                         env=env, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
    r = json.loads(out[-1])
    print("waves", w, "sequential ms", round(r["sequential"]["ms_per_step_median"], 3), "stages", {k: round(v, 3) for k, v in r["stage_ms_exclusive"].items()},
          "main frac", round(r["roofline"]["frac"], 3), "stage frac", round(r["roofline"]["stage"]["frac"], 3), flush=True)
