"""Meshes far above the metric's size: Kuhn cubes of 0.5 M and 1 M tets, 65 pairs; time, iterations and the solver's own residual report.
    python tools/probe/big_mesh_probe.py [n ...]"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
ctx = api.Context(0)
m = meshes.MATERIALS["Steel"]
for n in [int(a) for a in sys.argv[1:]] or [44, 55]:
    pts, tets = meshes.kuhn_box(n, n, n, 0.1, 0.1, 0.1)
    t0 = time.perf_counter()
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    t_asm = time.perf_counter() - t0
    times = []
    for rep in range(2):
        t0 = time.perf_counter()
        ev, prof = s.eigs(65, -(2 * np.pi * 20.0) ** 2, 1e-6)
        times.append(time.perf_counter() - t0)
    rep = s.residual_report() if hasattr(s, "residual_report") else None
    f = np.sqrt(np.maximum(ev[6:9], 0)) / (2 * np.pi)
    print(f"{len(tets)} tets, {s.n} unknowns: assembly {t_asm * 1e3:.0f} ms, solve {times[-1] * 1e3:.0f} ms, {prof['restarts']:.0f} iterations; first elastic {f[0]:.1f} Hz (x3 multiplet spread {np.ptp(f) / f[0]:.1e}); rigid |lambda| / lambda_7 {np.abs(ev[:6]).max() / ev[6]:.1e}; report {rep}", flush=True)
    s.close(); mesh.close()
