"""The round-6 soak's surface kinds at 3-10 times its sizes (40-200 k tets): stretched fine UV spheres, fine tori, finer scan-like skillets -- through the front end's default options and
mesh2modes with the default config.     python tools/probe/r06_soak_large.py [seed] [count]"""
import sys, os, time, importlib.util
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
from mesheditor_amd import api, meshes, tets as front_end
spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_flat_fill_surfaces.py"))
mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(900 + seed)
ctx = api.Context(0)
mats = [meshes.MATERIALS[k] for k in meshes.MATERIAL_ORDER]
fails, its = [], []
for trial in range(count):
    kind = trial % 3
    t0 = time.time()
    if kind == 0:
        seg = int(rng.choice([96, 112, 128]))
        P, F = meshes.uv_sphere_surface(0.1, seg, seg // 2)
        P = P * rng.uniform(0.5, 1.5, 3); name = f"ellipsoid {seg}x{seg // 2}"
    elif kind == 1:
        nu, nv = int(rng.integers(64, 128)), int(rng.integers(20, 40))
        P, F = mk.torus(0.1, 0.1 * rng.uniform(0.2, 0.5), nu, nv)
        P = P * rng.uniform(0.7, 1.3, 3); name = f"torus {nu}x{nv}"
    else:
        h = float(rng.choice([0.009, 0.0075]))
        P, F = meshes.skillet_scan_surface(h, h * rng.uniform(1.1, 1.6), noise_seed=int(rng.integers(1, 1000))); name = f"scan h={h}"
    try:
        pts, tets, left = front_end.tetrahedralize(P, F)
    except RuntimeError as e:
        print(f"surf {trial} {name}: front end: {str(e)[:160]}", flush=True); fails.append((trial, "fill", str(e)[:60])); continue
    t1 = time.time()
    q = pts[tets.astype(np.int64)]
    vol6 = np.abs(np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0]))
    e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
    smin = float((vol6 * np.sqrt(2) / e2 ** 1.5).min())
    pairs = 65
    ex = pts[(np.arange(10) * len(P)) // 10].astype(np.float32)
    try:
        r = api.mesh2modes(ctx, pts, tets, api.material(*mats[trial % len(mats)]), ex, config=api.default_config(num_modes=50, num_fem_modes=pairs))
    except Exception as e:  # noqa: BLE001
        print(f"surf {trial} {name}: {len(tets)} tets, worst shape {smin:.1e}: EXCEPTION {e!r} <- {e.__cause__!r}"[:400], flush=True); fails.append((trial, "exception")); continue
    ctx.synchronize()
    its.append(r.profile.get("restarts", 0))
    print(f"surf {trial} {name}: {len(P)} -> {len(pts)} points {len(tets)} tets ({left} on the surface, fill {t1 - t0:.1f} s), worst shape {smin:.1e}; {len(r.eigenvalues)} of {pairs} pairs, {r.profile.get('restarts')} iterations, {1e3 * (time.time() - t1):.0f} ms, at floor {r.profile.get('pairs_at_floor')}", flush=True)
    if len(r.eigenvalues) != pairs: fails.append((trial, "pairs"))
    if smin < 1e-3: fails.append((trial, "shape", smin))
print("solves", len(its), "iterations min / median / max", min(its), int(np.median(its)), max(its), "failures", fails)
