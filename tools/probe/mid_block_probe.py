"""129-pair solves (block of 160 columns: the first width that keeps M P) on three meshes.   python tools/probe/mid_block_probe.py"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
ctx = api.Context(0)
for name in ("cube_s30k", "scan_s30k_repaired", "cube_s100k"):
    pts, tets, m, kw = meshes.workload(name)
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    s.eigs(129, -(2 * np.pi * 20.0) ** 2, 1e-6)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        ev, prof = s.eigs(129, -(2 * np.pi * 20.0) ** 2, 1e-6)
        ts.append(time.perf_counter() - t0)
    print(f"{name:20s} 129 pairs: {prof['restarts']:.0f} iterations, {1e3 * np.median(ts):.1f} ms, lambda_129 {ev[-1]:.9e}", flush=True)
    s.close(); mesh.close()
