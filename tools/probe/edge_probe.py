"""Edge requests through the C ABI: tiny and odd pair counts, tight tolerances, other shifts; each against the 65-pair solve of the same
system where they overlap.   python tools/probe/edge_probe.py"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
ctx = api.Context(0)
for name in ("cube_s10k", "ball_s10k", "scan_s30k_repaired"):
    pts, tets, m, kw = meshes.workload(name)
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    ref, _ = s.eigs(65, -(2 * np.pi * 20.0) ** 2, 1e-6)
    for k, fmin, tol in [(1, 20, 1e-6), (2, 20, 1e-6), (7, 20, 1e-6), (8, 20, 1e-6), (17, 20, 1e-6), (33, 20, 1e-6), (65, 20, 1e-9), (65, 20, 1e-11), (65, 1, 1e-6), (65, 200, 1e-6), (65, 2000, 1e-6), (129, 20, 1e-6)]:
        t0 = time.perf_counter()
        try:
            ev, prof = s.eigs(k, -(2 * np.pi * fmin) ** 2, tol)
            kk = min(k, 65)
            el = ref[:kk] > 1e-6 * ref[64]
            err = (np.abs(ev[:kk][el] - ref[:kk][el]) / ref[:kk][el]).max() if el.any() else 0.0
            print(f"{name:20s} nev {k:4d} fmin {fmin:5d} tol {tol:.0e}: {prof['restarts']:3.0f} its {1e3 * (time.perf_counter() - t0):7.1f} ms  max rel diff vs 65-pair solve {err:.1e}", flush=True)
        except Exception as e:
            print(f"{name:20s} nev {k:4d} fmin {fmin:5d} tol {tol:.0e}: FAILED {str(e)[:150]}", flush=True)
    s.close(); mesh.close()
