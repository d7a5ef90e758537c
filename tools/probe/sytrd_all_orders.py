"""Every order 2 .. 256 through the register-resident tridiagonalisation: spectrum against LAPACK's, on a well-conditioned and on a
graded (Rayleigh-Ritz-like: entries from 1 down to 1e-12) matrix."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from scipy.linalg import eigvalsh_tridiagonal
from mesheditor_amd import api
import lab
ctx = api.Context(0)
rng = np.random.default_rng(11)
bad = []
for m in range(2, 257):
    for kind in ("random", "graded", "blockdiag"):
        b = rng.standard_normal((m, m))
        a = b + b.T
        if kind == "graded":
            s = np.logspace(0, -12, m)
            a = a * s[:, None] * s[None, :] + np.diag(np.linspace(1, 2, m))
        if kind == "blockdiag":
            h = m // 2
            a[:h, h:] = 0; a[h:, :h] = 0
        for variant in (3,):
            d, e, _ = lab.tridiagonalize(ctx, a, variant=variant)
            ok = np.isfinite(d).all() and np.isfinite(e).all()
            err = np.abs(eigvalsh_tridiagonal(d, e) - np.linalg.eigvalsh(a)).max() / np.abs(a).max() if ok else np.inf
            if not err < 1e-12 * m:
                bad.append((m, kind, variant, err))
print("bad:", bad[:40], len(bad))
