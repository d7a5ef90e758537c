"""The one-CU register-resident tridiagonalisation (variant 3) against the one-workgroup (0) and multi-workgroup (1) kernels:
spectrum of the tridiagonal form against LAPACK's, reflectors through Q^T A Q, bit-reproducibility, and time per call."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from scipy.linalg import eigvalsh_tridiagonal
from mesheditor_amd import api as mh
import lab

ctx = mh.Context(0)
rng = np.random.default_rng(5)
for m in (2, 3, 31, 32, 33, 64, 65, 127, 160, 222, 255, 256):
    b = rng.standard_normal((m, m))
    a = b + b.T + np.diag(rng.standard_normal(m) * 3)
    ref = np.linalg.eigvalsh(a)
    line = [f"m={m:3d}"]
    for variant in (0, 1, 3):
        if variant == 1 and m < 64: continue
        d, e, refl, tau, ms = lab.tridiagonalize_full(ctx, a, variant=variant, reps=20 if m >= 64 else 2)
        w = eigvalsh_tridiagonal(d, e) if m > 1 else d
        err = np.abs(w - ref).max() / np.abs(ref).max()
        # Q from the reflectors: H_0 H_1 ... ; T = Q^T A Q
        q = np.eye(m)
        for k in range(m - 2, -1, -1):
            v = np.zeros(m); v[k + 1] = 1.0; v[k + 2:] = refl[k + 2:, k]
            q -= tau[k] * np.outer(v, v @ q)
        t = q.T @ a @ q
        tt = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        qerr = np.abs(t - tt).max() / np.abs(a).max()
        d2, e2, refl2, tau2, _ = lab.tridiagonalize_full(ctx, a, variant=variant, reps=1)
        same = np.array_equal(d, d2) and np.array_equal(e, e2) and np.array_equal(refl, refl2)
        line.append(f"v{variant}: spec {err:.1e} QtAQ {qerr:.1e} {'same' if same else 'DIFFERS'} {ms*1e3:7.1f} us")
    print("  ".join(line), flush=True)

# per-phase cycle stamps (only in a build with -DMH_REGS_STAMPS)
import ctypes as C
from mesheditor_amd import _lib
L = _lib.lib() if hasattr(_lib, "lib") else None
if L is not None and hasattr(L, "mh_debug_regs_stamps"):
    names = ["sweeps p", "bar", "reduce", "bar", "pv+next", "update", "bar"]
    for m in (64, 222, 256):
        b = rng.standard_normal((m, m)); a = b + b.T
        out = (C.c_ulonglong * 8)()
        L.mh_debug_regs_stamps(out, 1)
        lab.tridiagonalize_full(ctx, a, variant=3, reps=10)
        L.mh_debug_regs_stamps(out, 0)
        tot = sum(out[:7])
        print(f"m={m}: " + "  ".join(f"{n} {out[i] / 10 / (m - 1):.0f}" for i, n in enumerate(names)) + f"   total {tot / 10 / (m - 1):.0f} ticks/step")
