"""A THICK scanned solid (no thin walls): a bumpy ball as a level set -> marching tetrahedra -> Taubin smoothing -> the front end -> the device.
The skillet scans are thin-walled; a solid rock is the other half of the RealImpact class (interior far from every surface vertex).
    python tools/probe/solid_scan_probe.py [h ...]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mesheditor_amd import api, meshes, tets as T


def bumpy_ball(p, radius=0.1, seed=5):
    rng = np.random.Generator(np.random.MT19937(seed))
    k = rng.uniform(20.0, 60.0, (8, 3))
    ph = rng.uniform(0, 2 * np.pi, 8)
    bumps = sum(np.sin(p @ k[i] + ph[i]) for i in range(8)) * (0.012 / 8)
    return np.linalg.norm(p * np.array([1.0, 0.8, 0.65]), axis=1) - radius + bumps


def main():
    hs = [float(a) for a in sys.argv[1:]] or [0.012, 0.006]
    ctx = api.Context(0)
    for h in hs:
        v, f = meshes.marching_tets_surface(bumpy_ball, (-0.14, -0.16, -0.19), (0.14, 0.16, 0.19), h)
        v = meshes.taubin_smooth(v, f, 8)
        v, f = meshes.largest_component(v, f)
        t0 = time.time()
        pts, tets, left = T.tetrahedralize(v, f)
        t_fill = time.time() - t0
        p = pts[tets.astype(np.int64)]
        vol = np.abs(np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])) / 6
        e = np.stack([np.linalg.norm(p[:, i] - p[:, j], axis=1) for i in range(4) for j in range(i + 1, 4)], 1)
        q = vol * 6 * np.sqrt(2) / np.sqrt((e ** 2).mean(1)) ** 3
        m = meshes.MATERIALS["Ceramic"]
        ex = pts[(np.arange(10) * len(v)) // 10].astype(np.float32)
        times, r = [], None
        for _ in range(2):
            t0 = time.time()
            r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=50, num_fem_modes=65))
            ctx.synchronize()
            times.append(time.time() - t0)
        print("solid scan h=%.4f: %d surface points, %d triangles -> %d points %d tets (%d left on the surface, fill %.1f s), shape min %.1e pct1 %.3f pct10 %.3f; device: %d pairs, %s iterations, %.0f ms"
              % (h, len(v), len(f), len(pts), len(tets), left, t_fill, q.min(), np.percentile(q, 1), np.percentile(q, 10), len(r.eigenvalues), r.profile.get("restarts"), 1e3 * times[-1]), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
