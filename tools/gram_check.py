"""The Gram kernel at the solver's block shapes: time per X^T Y, TF/s and effective HBM rate (both panels read once).
    python tools/gram_check.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api
import lab

ctx = api.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 542253
for wa, wb in [(80, 80), (160, 96), (240, 80), (240, 240), (234, 215), (200, 200), (240, 160)]:
    ms = lab.bench_dense(ctx, 0, n, wa, wb, reps=10)
    print(f"n {n}  {wa:3d} x {wb:3d}   {ms * 1e3:8.1f} us   {2e-9 * n * wa * wb / ms:7.2f} TFLOP/s   {8e-6 * n * (wa + wb) / ms:7.1f} GB/s (panels once)", flush=True)
