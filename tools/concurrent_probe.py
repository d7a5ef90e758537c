"""Throughput of S concurrent solves on one GPU (each on its own context = stream), S100k, 65 pairs."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesheditor_amd import api, meshes

pts, tets, m, kw = meshes.workload("cube_s100k")
mat = api.material(*m)
cfg = api.default_config(num_modes=kw["num_modes"], num_fem_modes=kw["num_fem_modes"])
ex = pts[:: len(pts) // 10][:10].astype(np.float32)
for S in (1, 2, 3, 4):
    ctxs = [api.Context(0) for _ in range(S)]
    mesh = [api.Mesh(c, pts, tets) for c in ctxs]
    reps = 4
    def work(i):
        for _ in range(reps):
            r = api.mesh2modes(ctxs[i], pts, tets, mat, ex, config=cfg, mesh=mesh[i])
            assert len(r.eigenvalues) == 65
    for i in range(S):
        api.mesh2modes(ctxs[i], pts, tets, mat, ex, config=cfg, mesh=mesh[i])  # warm
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"streams {S}: {S*reps} solves in {dt:.3f} s -> {65*S*reps/dt:.1f} eigenpairs/s, {1e3*dt/reps:.1f} ms per mesh per stream", flush=True)
    for c in ctxs:
        c.close()
