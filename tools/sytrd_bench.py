"""Tridiagonalisation kernels on the GPU box: one workgroup against several, by matrix order."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api
import lab  # tools/lab.py: libmodalhip_lab.so
ctx = api.Context(0)
rng = np.random.default_rng(3)
for m in [96, 160, 200, 222, 240, 256]:
    a = rng.standard_normal((m, m)); a = a + a.T + 2 * m * np.eye(m)
    row = []
    for variant in (0, 1):
        lab.tridiagonalize(ctx, a, variant=variant, reps=2)
        d, e, ms = lab.tridiagonalize(ctx, a, variant=variant, reps=20)
        row.append((ms, d, e))
    diff = max(np.max(np.abs(row[0][1] - row[1][1])), np.max(np.abs(np.abs(row[0][2]) - np.abs(row[1][2]))))
    print(f"m {m:4d}  one workgroup {row[0][0]*1e3:8.1f} us   several {row[1][0]*1e3:8.1f} us   max |T0 - T1| {diff:.2e}", flush=True)
