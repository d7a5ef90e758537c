"""Stretched UV spheres (ellipsoids: needle quads, planar caps) filled WITHOUT the flat-cell pass -- a caller's own mesh with flat cells -- through the solver.
    python tools/probe/flat_ellipsoid_probe.py [seed] [count]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes, tets as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = api.Context(0)
mats = [meshes.MATERIALS[k] for k in meshes.MATERIAL_ORDER]
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    m = mats[trial % len(mats)]
    seg = int(rng.choice([32, 48, 64, 80] if len(sys.argv) < 4 else [24, 32, 48, 64]))
    P, F = meshes.uv_sphere_surface(0.1, seg, seg // 2)
    s = rng.uniform(0.4, 1.6, 3)
    P = P * s
    p, t, left = T.tetrahedralize(P, F, break_flat_cells=False)
    q = p[t.astype(np.int64)]
    vol6 = np.abs(np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0]))
    e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
    sh = vol6 * np.sqrt(2) / e2 ** 1.5
    ex = p[(np.arange(10) * len(P)) // 10].astype(np.float32)
    t0 = time.time()
    try:
        r = api.mesh2modes(ctx, p, t, api.material(*m), ex, config=api.default_config(num_modes=50, num_fem_modes=65))
        out = "%d pairs, %s iterations, %.0f ms" % (len(r.eigenvalues), r.profile.get("restarts"), 1e3 * (time.time() - t0))
    except Exception as e:  # noqa: BLE001
        out = "EXCEPTION " + repr(e)[:160] + " <- " + repr(e.__cause__)[:200]
    print(f"{trial} ellipsoid {seg}x{seg//2} scale {s.round(2)}: {len(t)} tets, worst shape {sh.min():.1e}, below 1e-4: {(sh < 1e-4).sum()}; {out}", flush=True)
