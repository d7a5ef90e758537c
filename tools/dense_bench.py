"""Gram / basis-update roofline microbenchmark on the GPU box (tall-skinny fp64 MFMA kernels of the eigensolver)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api
import lab  # tools/lab.py: libmodalhip_lab.so
ctx = api.Context(0)
n = int(os.environ.get("ROWS", 446631))
for wa, wb in [] if os.environ.get("COMBINE_ONLY") else [(16, 16), (32, 32), (48, 48), (64, 64), (75, 75), (80, 80), (75, 40), (40, 40), (128, 128), (150, 75), (225, 225)]:
    ms = lab.bench_dense(ctx, 0, n, wa, wb)
    by, fl = 8.0 * n * (wa + wb), 2.0 * n * wa * wb
    print(f"gram    {wa:4d} x {wb:4d}  {ms*1e3:8.1f} us  {by/ms/1e6:8.1f} GB/s ({100*by/ms/1e6/8000:5.1f}% HBM)  {fl/ms/1e9:7.2f} TF/s ({100*fl/ms/1e9/78.6:5.1f}% fp64 MFMA)", flush=True)
cases = [tuple(int(v) for v in c.split('+')) for c in os.environ['CASES'].split(',')] if os.environ.get('CASES') else [(75, 75), (75, 150), (160, 62), (80, 80), (96, 128), (32, 100)]
for wa, wb in cases:
    if wa + wb > 256:
        continue
    ms = lab.bench_dense(ctx, 1, n, wa, wb)
    by, fl = 8.0 * n * (2 * wa + wb), 2.0 * n * (wa + wb) * wa
    print(f"combine {wa:4d} + {wb:4d} -> {wa}  {ms*1e3:8.1f} us  {by/ms/1e6:8.1f} GB/s ({100*by/ms/1e6/8000:5.1f}% HBM)  {fl/ms/1e9:7.2f} TF/s ({100*fl/ms/1e9/78.6:5.1f}% fp64 MFMA)", flush=True)
