"""The 128 x 64 UV sphere through the front end with and without the reference's Options::Quality: size and shape
measures of the fill, and -- on a GPU -- iterations and time of the 65-pair solve.   python tools/probe/quality_sphere_probe.py [seg rings] [--solve]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mesheditor_amd import meshes, tets as T

args = [a for a in sys.argv[1:] if not a.startswith("--")]
seg, rings = (int(args[0]), int(args[1])) if len(args) >= 2 else (128, 64)
solve = "--solve" in sys.argv
P, F = meshes.uv_sphere_surface(0.15, seg, rings)
for name, kw in (("default", {}), ("quality", {"quality": True})):
    t0 = time.time()
    pts, tets, left = T.tetrahedralize(P, F, **kw)
    t_fill = time.time() - t0
    p = pts[tets.astype(np.int64)]
    vol = np.abs(np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])) / 6
    e = np.stack([np.linalg.norm(p[:, i] - p[:, j], axis=1) for i in range(4) for j in range(i + 1, 4)], 1)
    q = vol * 6 * np.sqrt(2) / np.sqrt((e ** 2).mean(1)) ** 3
    line = "uv sphere %dx%d %-8s: %d surface points -> %d points %d tets (%d left on the surface, fill %.1f s), shape min %.1e, below 1e-3: %d, pct1 %.3f pct10 %.3f" % (
        seg, rings, name, len(P), len(pts), len(tets), left, t_fill, q.min(), int((q < 1e-3).sum()), np.percentile(q, 1), np.percentile(q, 10))
    if solve:
        from mesheditor_amd import api
        ctx = api.Context(0)
        m = meshes.MATERIALS["Ceramic"]
        ex = pts[(np.arange(10) * len(P)) // 10].astype(np.float32)
        for rep in range(2):
            t0 = time.time()
            r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=50, num_fem_modes=65))
            ctx.synchronize()
            t_dev = time.time() - t0
        line += "; device: %d pairs, %s iterations, %.0f ms, f7 %.2f Hz" % (len(r.eigenvalues), r.profile.get("restarts"), 1e3 * t_dev, np.sqrt(max(r.eigenvalues[6], 0)) / 2 / np.pi if len(r.eigenvalues) > 6 else 0)
        ctx.close()
    print(line, flush=True)
