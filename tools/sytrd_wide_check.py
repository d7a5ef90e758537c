"""The wide tridiagonalisation (orders 257 .. 768, 48 workgroups over all XCDs) on the GPU box: Q^T A Q = T checked with LAPACK's own
back-transformation of the returned reflectors, eigenvalues of T against numpy's of A, and the time per order beside rocSOLVER's share
of a 215-pair solve (8.3 ms at order 720).
    python tools/sytrd_wide_check.py [orders ...]"""
import os, sys
import numpy as np
import scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api
import lab

ctx = api.Context(0)
rng = np.random.default_rng(3)
for m in [int(a) for a in sys.argv[1:]] or [64, 200, 256, 300, 480, 640, 720, 768]:
    a = rng.standard_normal((m, m)); a = a + a.T + 2 * m * np.eye(m)
    lab.tridiagonalize_full(ctx, a, variant=2, reps=1)
    d, e, refl, tau, ms = lab.tridiagonalize_full(ctx, a, variant=2, reps=10)
    t = np.diag(d) + np.diag(e, -1) + np.diag(e, 1)
    ev_err = np.abs(np.linalg.eigvalsh(t) - np.linalg.eigvalsh(a)).max() / np.abs(a).max()
    # Q from the reflectors (lower storage: reflector k has v[k+1] = 1, tail in refl[k+2:, k]), exactly as dorgtr builds it
    q = np.eye(m)
    for k in range(m - 2, -1, -1):
        v = np.zeros(m); v[k + 1] = 1.0; v[k + 2:] = refl[k + 2:, k]
        q -= tau[k] * np.outer(v, v @ q)
    resid = np.abs(q.T @ a @ q - t).max() / np.abs(a).max()
    line = f"m {m:4d}  wide {ms * 1e3:8.1f} us   max |Q^T A Q - T| / |A| {resid:.2e}   eigenvalues {ev_err:.2e}"
    if m <= 256:
        d1, e1, ms1 = lab.tridiagonalize(ctx, a, variant=1, reps=10)
        line += f"   (k_sytrd_multi: {ms1 * 1e3:.1f} us, max |d - d'| {np.abs(d - d1).max():.1e})"
    print(line, flush=True)
