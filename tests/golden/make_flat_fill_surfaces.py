"""Generates tests/golden/flat_fill_surfaces.npz: three closed surfaces of the round-6 soak (tools/probe/r06_soak.py 20 60 <seed>) whose default fills kept
a flat cell until the front end's flat-cell pass learnt the edge split and the star's Chebyshev centre (docs/LAB_NOTEBOOK.md section 13) -- one of each kind:

    torus_sliver   seed 4, surface 34: torus 16 x 12, an interior sliver of four SURFACE vertices across the tube (2e-8)
    torus_edge     seed 2, surface 52: torus 12 x 19, a recovery point a hair off a surface edge (1.4e-5)
    ellipsoid_cap  seed 3, surface 57: ellipsoid 64 x 32, caps on planar surface quads over a fan of thin cells (3.8e-7)

Inputs only (positions in double as the soak hands them over, triangle indices); the test fills them and looks at the cells.  The soak's random stream is
replayed without the solves (they draw nothing).      python tests/golden/make_flat_fill_surfaces.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mesheditor_amd import meshes  # noqa: E402


def torus(R, r, nu, nv):
    u = np.arange(nu) * 2 * np.pi / nu
    v = np.arange(nv) * 2 * np.pi / nv
    P = np.array([[(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)] for a in u for b in v], np.float32).astype(np.float64)
    F = []
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = i * nv + j, ((i + 1) % nu) * nv + j, ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv
            F += [(a, b, c), (a, c, d)]
    return P, np.array(F, np.uint32)


def soak_surface(seed, index, n_boxes=20):
    """the `index`-th surface of `r06_soak.py n_boxes <more than index> seed`"""
    rng = np.random.default_rng(seed)
    for _ in range(n_boxes):
        nx, ny, nz = (int(v) for v in rng.integers(2, 15, 3))
        if nx * ny * nz < 24:
            continue
        rng.uniform(0.6, 1.6), rng.uniform(0.6, 1.6), rng.uniform(0.6, 1.6)
        rng.uniform(0.0, 0.3)
        rng.uniform(-1, 1, ((nx + 1) * (ny + 1) * (nz + 1), 3))
        rng.choice([20, 45, 65])
    for trial in range(index + 1):
        kind = trial % 3
        if kind == 0:
            seg = int(rng.choice([16, 24, 32, 48, 64, 80]))
            scale = rng.uniform(0.4, 1.6, 3)
            if trial == index:
                P, F = meshes.uv_sphere_surface(0.1, seg, max(6, seg // 2))
                P, name = P * scale, f"ellipsoid {seg}x{max(6, seg // 2)}"
        elif kind == 1:
            nu, nv = int(rng.integers(12, 48)), int(rng.integers(6, 20))
            r = 0.1 * rng.uniform(0.15, 0.5)
            scale = rng.uniform(0.6, 1.4, 3)
            if trial == index:
                P, F = torus(0.1, r, nu, nv)
                P, name = P * scale, f"torus {nu}x{nv}"
        else:
            h = float(rng.choice([0.02, 0.016, 0.013]))
            thickness = h * rng.uniform(1.1, 1.6)
            noise = int(rng.integers(1, 1000))
            if trial == index:
                P, F = meshes.skillet_scan_surface(h, thickness, noise_seed=noise)
                name = f"scan h={h}"
        rng.choice([30, 45, 65])
    return np.ascontiguousarray(P, np.float64), np.ascontiguousarray(F, np.uint32), name


if __name__ == "__main__":
    out = {}
    for key, seed, index in (("torus_sliver", 4, 34), ("torus_edge", 2, 52), ("ellipsoid_cap", 3, 57)):
        P, F, name = soak_surface(seed, index)
        print(key, name, len(P), "points", len(F), "triangles")
        out[key + "_P"], out[key + "_F"] = P, F
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "flat_fill_surfaces.npz"), **out)
