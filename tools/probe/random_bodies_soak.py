"""Robustness of the cycle shapes chosen at the end of round 5: many random bodies (jittered Kuhn boxes of random proportions and jitter, random
materials, 20-65 pairs), every solve must converge without the spectral-bound retry; prints the cycle class, iterations and the worst count."""
import sys, os, io, contextlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes
ctx = api.Context(0)
rng = np.random.default_rng(2026)
mats = [meshes.MATERIALS[k] for k in meshes.MATERIAL_ORDER]
worst, fails, n = 0, [], 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    nx, ny, nz = (int(v) for v in rng.integers(2, 15, 3))
    if nx * ny * nz < 24: continue
    lx, ly, lz = 0.02 * nx * rng.uniform(0.6, 1.6), 0.02 * ny * rng.uniform(0.6, 1.6), 0.02 * nz * rng.uniform(0.6, 1.6)
    pts, tets = meshes.kuhn_box(nx, ny, nz, lx, ly, lz)
    jitter = rng.uniform(0.0, 0.3)
    h = min(lx / nx, ly / ny, lz / nz)
    interior = np.ones(len(pts), bool)
    pts = pts + (rng.uniform(-1, 1, pts.shape) * jitter * h * 0.5)
    m = mats[trial % len(mats)]
    pairs = int(rng.choice([20, 45, 65]))
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    dofs = 0
    try:
        ev, prof = s.eigs(pairs, residual_tol=1e-5)
        worst = max(worst, prof["restarts"]); n += 1
        print(f"{trial:3d} box {nx}x{ny}x{nz} jitter {jitter:.2f} tets {len(tets)} per point {len(tets) / len(pts):.2f} pairs {pairs} iterations {prof['restarts']} selfcheck {prof['rr_selfcheck']:.1e}", flush=True)
    except Exception as e:
        fails.append((trial, nx, ny, nz, jitter, str(e)[:160]))
        print(f"{trial:3d} box {nx}x{ny}x{nz} jitter {jitter:.2f} FAILED {str(e)[:160]}", flush=True)
    s.close(); mesh.close()
print("solved", n, "worst iterations", worst, "failures", fails)
