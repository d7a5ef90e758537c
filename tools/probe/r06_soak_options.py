"""Round-6 soak of the SOLVER on whatever the front end's options can make (GPU): the surfaces of tools/probe/r06_soak.py (tests/golden/make_flat_fill_surfaces.py: soak_surface),
filled with random Quality / MaxVolume / InteriorShell / RepairSlivers / BreakFlatCells -- a third of them with the sliver repair OFF, i.e. raw Delaunay fills with cells flat to
rounding, the worst a caller's own TetMesh can look like -- then mesh2modes with the default config.  Every solve must return all its pairs; the log says which fall-back, if any,
each one needed.     python tools/probe/r06_soak_options.py <seed> <count> [first]"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from mesheditor_amd import api, meshes, tets as front_end  # noqa: E402

spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_flat_fill_surfaces.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)

seed, count = int(sys.argv[1]), int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = api.Context(0)
mats = [meshes.MATERIALS[k] for k in meshes.MATERIAL_ORDER]
fails, its = [], []
for index in range(first, first + count):
    rng = np.random.default_rng(700000 * seed + index)
    P, F, name = mk.soak_surface(seed, index)
    if len(P) > 1300:
        continue
    opts = dict(quality=bool(rng.random() < 0.2), max_volume=0.0, interior_shell=str(rng.choice(["when_flat", "never", "always"])), repair_slivers=bool(rng.random() < 0.65),
                break_flat_cells=bool(rng.random() < 0.7))
    if rng.random() < 0.2:
        a, b, c = P[F[:, 0].astype(np.int64)], P[F[:, 1].astype(np.int64)], P[F[:, 2].astype(np.int64)]
        opts["max_volume"] = float(abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6 / rng.integers(2000, 12000))
    try:
        pts, tets, left = front_end.tetrahedralize(P, F, **opts)
    except RuntimeError as e:
        print(f"{seed}/{index} {name} {opts}: front end: {str(e)[:120]}", flush=True)
        fails.append((index, "fill", str(e)[:60]))
        continue
    q = pts[tets.astype(np.int64)]
    vol6 = np.abs(np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0]))
    e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
    shape = vol6 * np.sqrt(2) / e2 ** 1.5
    pairs = int(rng.choice([30, 45, 65]))
    m = mats[index % len(mats)]
    ex = pts[(np.arange(10) * len(P)) // 10].astype(np.float32)
    t1 = time.time()
    tag = f"{seed}/{index} {name} repair={int(opts['repair_slivers'])} flat_pass={int(opts['break_flat_cells'])} shell={opts['interior_shell']} q={int(opts['quality'])} maxvol={opts['max_volume']:.1e}: {len(tets)} tets, worst shape {shape.min():.1e}, {int((shape < 1e-4).sum())} cells below 1e-4"
    try:
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=pairs - 15, num_fem_modes=pairs))
    except Exception as e:  # noqa: BLE001
        print(f"{tag}: EXCEPTION {e!r} <- {e.__cause__!r}"[:500], flush=True)
        fails.append((index, "exception", repr(e.__cause__)[:100]))
        continue
    ctx.synchronize()
    its.append(r.profile.get("restarts", 0))
    print(f"{tag}; {len(r.eigenvalues)} of {pairs} pairs, {r.profile.get('restarts')} iterations, {1e3 * (time.time() - t1):.0f} ms{', %d pairs at the rounding floor' % r.profile['pairs_at_floor'] if r.profile.get('pairs_at_floor') else ''}", flush=True)
    if len(r.eigenvalues) != pairs: fails.append((index, "pairs", len(r.eigenvalues)))
print("solves", len(its), "iterations min / median / max", min(its), int(np.median(its)), max(its), "failures", fails)
