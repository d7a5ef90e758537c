import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
ctx = api.Context(0)
pts, tets, m, kw = meshes.workload("cube_s10k")
mesh = api.Mesh(ctx, pts, tets)
s = api.System(ctx, mesh, api.material(*m))
for k in (100, 215, 260, 300):
    t0 = time.perf_counter()
    ev, prof = s.eigs(k, -(2 * np.pi * 20.0) ** 2, 1e-6)
    dt = time.perf_counter() - t0
    ev2, _ = s.eigs(k, -(2 * np.pi * 20.0) ** 2, 1e-6)
    print(k, "pairs", len(ev), "its", prof.get("restarts"), "%.0f ms" % (dt * 1e3), "repeatable", np.array_equal(ev, ev2), "lam7 %.6e last %.6e" % (ev[6], ev[-1]), flush=True)
