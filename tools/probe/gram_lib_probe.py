import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from mesheditor_amd import api
import lab
ctx = api.Context(0)
n = 446631
for wa, wb in [(80, 80), (80, 64), (160, 80), (160, 96), (96, 96), (64, 64), (240, 80)]:
    ms = lab.bench_dense(ctx, 0, n, wa, wb, reps=10)
    print(f"n {n}  {wa:3d} x {wb:3d}   {ms * 1e3:8.1f} us   {2e-9 * n * wa * wb / ms:7.2f} TFLOP/s   {8e-6 * n * (wa + wb) / ms:7.1f} GB/s", flush=True)
