"""BASELINE config 2 names a UV-sphere primitive: a UV sphere surface through the path's own front end, solved on the device and by the
oracle.  Needle fans at the poles, coplanar ring quads: the surface class that used to defeat the tetrahedraliser (round 3) and whose
cap slivers stress the eigensolver.
    python tools/probe/uv_sphere_probe.py [segments rings ...]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mesheditor_amd import api, meshes, tets as T


def main():
    from oracle import pyoracle as po
    args = [int(a) for a in sys.argv[1:]] or [24, 12, 48, 24, 64, 32]
    ctx = api.Context(0)
    for seg, rings in zip(args[::2], args[1::2]):
        P, F = meshes.uv_sphere_surface(0.15, seg, rings)
        t0 = time.time()
        pts, tets, left = T.tetrahedralize(P, F)
        t_fill = time.time() - t0
        p = pts[tets.astype(np.int64)]
        vol = np.abs(np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])) / 6
        e = np.stack([np.linalg.norm(p[:, i] - p[:, j], axis=1) for i in range(4) for j in range(i + 1, 4)], 1)
        q = vol * 6 * np.sqrt(2) / np.sqrt((e ** 2).mean(1)) ** 3
        m = meshes.MATERIALS["Ceramic"]
        ex = pts[(np.arange(10) * len(P)) // 10].astype(np.float32)
        cfg = dict(num_modes=50, num_fem_modes=65)
        t0 = time.time()
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(**cfg))
        ctx.synchronize()
        t_dev = time.time() - t0
        line = "uv sphere %dx%d: %d surface points -> %d points %d tets (%d left on the surface, fill %.2f s), shape min %.1e pct1 %.3f; device: %d pairs, %s iterations, %.0f ms" % (
            seg, rings, len(P), len(pts), len(tets), left, t_fill, q.min(), np.percentile(q, 1), len(r.eigenvalues), r.profile.get("restarts"), 1e3 * t_dev)
        if len(r.eigenvalues) and len(tets) < 60000:
            ref = po.mesh2modes(pts, tets, po.material(*m), ex, config=po.default_config(**cfg))
            el = ref.eigenvalues > 1e-6 * ref.eigenvalues[-1]
            line += "; against the oracle: max rel %.1e over %d elastic pairs" % (np.abs(r.eigenvalues[el] / ref.eigenvalues[el] - 1).max(), el.sum())
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
