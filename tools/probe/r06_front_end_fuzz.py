"""CPU-only fuzz of the tet front end over its OPTIONS (round 6): the soak's random closed surfaces (tests/golden/make_flat_fill_surfaces.py: soak_surface), some decimated
by SimplifySurface first, through tetra::Tetrahedralize with random Quality / MaxVolume / InteriorShell / RepairSlivers / InteriorSteiner -- and what every fill must be:
input vertices untouched, every tetrahedron positively oriented, every face on at most two tetrahedra, the boundary the input's own triangulation when no point is left on the
surface, every point in a tetrahedron, the volume the surface encloses (where its winding is consistent), MaxVolume kept; with the default-quality options (repair on) no cell below 1e-3.
    python tools/probe/r06_front_end_fuzz.py <seed> <count> [first]"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from mesheditor_amd import tets as front_end  # noqa: E402

spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_flat_fill_surfaces.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)

seed, count = int(sys.argv[1]), int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fails = []
for index in range(first, first + count):
    rng = np.random.default_rng(100000 * seed + index)
    P, F, name = mk.soak_surface(seed, index)
    if len(P) > 2500:
        continue
    opts = dict(quality=bool(rng.random() < 0.3), max_volume=0.0, interior_shell=str(rng.choice(["when_flat", "never", "always"])), repair_slivers=bool(rng.random() < 0.8),
                interior_steiner=bool(rng.random() < 0.85), break_flat_cells=bool(rng.random() < 0.85))
    a, b, c = P[F[:, 0].astype(np.int64)], P[F[:, 1].astype(np.int64)], P[F[:, 2].astype(np.int64)]
    if rng.random() < 0.3:
        opts["max_volume"] = float(abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6 / rng.integers(2000, 20000))
    decimated = rng.random() < 0.25
    if decimated:
        P32, F = front_end.simplify_surface(P.astype(np.float32), F, float(rng.uniform(0.3, 0.8)))
        P = P32.astype(np.float64)
        a, b, c = P[F[:, 0].astype(np.int64)], P[F[:, 1].astype(np.int64)], P[F[:, 2].astype(np.int64)]
    enclosed = abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6
    directed = np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]).astype(np.int64)
    oriented = len(np.unique(directed[:, 0] * (len(P) + 1) + directed[:, 1])) == len(directed)  # (the scan surfaces come out of marching tetrahedra with mixed winding: no signed volume)
    tag = f"{seed}/{index} {name}{' decimated' if decimated else ''} {len(P)} pts {opts}"
    t0 = time.time()
    try:
        p, t, left = front_end.tetrahedralize(P, F, **opts)
    except RuntimeError as e:
        print(f"{tag}: ERROR {str(e)[:160]}", flush=True)
        fails.append((index, "error", str(e)[:80]))
        continue
    dt = time.time() - t0
    q = p[t.astype(np.int64)]
    vol6 = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0])
    e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
    shape = float((vol6 * np.sqrt(2) / e2 ** 1.5).min())
    faces = np.sort(np.concatenate([t[:, [1, 2, 3]], t[:, [0, 2, 3]], t[:, [0, 1, 3]], t[:, [0, 1, 2]]]), axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    bad = []
    if not np.array_equal(p[: len(P)], P): bad.append("input vertices moved")
    # (the library's own last check is the EXACT orientation; in rounded arithmetic a cell of a raw fill -- RepairSlivers off -- may come out at -4e-18 of its size)
    if not vol6.min() > (0 if opts["repair_slivers"] else -1e-12 * e2.max() ** 1.5): bad.append(f"{int((vol6 <= 0).sum())} tetrahedra not positive")
    if counts.max() > 2: bad.append("a face on three tetrahedra")
    if len(np.unique(t)) != len(p): bad.append(f"{len(p) - len(np.unique(t))} points in no tetrahedron")
    if oriented and abs(vol6.sum() / 6 - enclosed) > 1e-9 * enclosed: bad.append(f"volume {vol6.sum() / 6:.6e} vs {enclosed:.6e}")
    if left == 0 and {tuple(r) for r in uniq[counts == 1]} != {tuple(sorted(r)) for r in F.tolist()}: bad.append("boundary is not the input triangulation")
    if opts["max_volume"] > 0 and vol6.max() / 6 > opts["max_volume"] * 1.0000001: bad.append(f"a tetrahedron of {vol6.max() / 6:.3e} above MaxVolume {opts['max_volume']:.3e}")
    if opts["repair_slivers"] and opts["break_flat_cells"] and opts["interior_steiner"] and shape < 1e-3: bad.append(f"flat cell {shape:.1e}")
    print(f"{tag}: {len(p)} points {len(t)} tets, {left} on the surface, worst shape {shape:.1e}, {dt:.1f} s{' <-- ' + '; '.join(bad) if bad else ''}", flush=True)
    if bad: fails.append((index, bad))
print("failures", fails)
