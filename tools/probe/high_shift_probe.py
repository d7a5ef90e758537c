import sys, faulthandler, numpy as np
faulthandler.enable()
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
ctx = api.Context(0)
pts, tets = meshes.jittered_box(6, 4242)
m = meshes.MATERIALS["Glass"]
mesh = api.Mesh(ctx, pts, tets)
s = api.System(ctx, mesh, api.material(*m))
print("n", s.n, flush=True)
for nev, f in ((25, 20.0), (40, 20.0), (40, 2000.0), (40, 20000.0)):
    try:
        ev, prof = s.eigs(nev, -(2 * np.pi * f) ** 2, 1e-5)
        print(nev, f, "ok", prof["restarts"], ev[6], flush=True)
    except Exception as e:
        print(nev, f, "error", e, flush=True)
rng = np.random.default_rng(3)
ex = pts[rng.choice(len(pts), 12, replace=False)].astype(np.float32) + 1e-3
r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=10, num_fem_modes=40, min_mode_freq=20000.0, max_mode_freq=1e6))
print("modes", len(r.freqs), flush=True)
