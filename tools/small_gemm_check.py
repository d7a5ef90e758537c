"""k_small_gemm beside the library's dgemm at the Rayleigh-Ritz step's orders (time per call, max error against numpy).
    python tools/small_gemm_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api
import lab

ctx = api.Context(0)
rng = np.random.default_rng(5)
for (M, N, K, ta) in [(240, 240, 240, False), (240, 80, 240, True), (720, 720, 720, False), (720, 240, 720, False), (240, 240, 720, True), (215, 240, 720, True)]:
    a = rng.standard_normal((K, M) if ta else (M, K)); b = rng.standard_normal((K, N))
    lab.small_gemm(ctx, a, b, ta=ta)
    c, ms = lab.small_gemm(ctx, a, b, ta=ta, reps=20)
    want = (a.T if ta else a) @ b
    print(f"{M:4d} x {N:4d} x {K:4d} {'T' if ta else 'N'}N  {ms * 1e3:7.1f} us   {2e-9 * M * N * K / (ms * 1e-3):7.1f} GFLOP/s   max err {np.abs(c - want).max():.1e}", flush=True)
