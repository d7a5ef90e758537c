"""A TetMesh of TWO disjoint bodies (twelve rigid-body modes, six of them not in the cold start's block) and of a body touching itself at one vertex, against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()
ctx = api.Context(0)
m = meshes.MATERIALS["Ceramic"]
for case in ("two boxes", "two jittered boxes far apart", "three bodies"):
    p1, t1 = meshes.kuhn_box(6, 5, 4, 0.12, 0.1, 0.08)
    p2, t2 = meshes.kuhn_box(4, 4, 7, 0.05, 0.05, 0.09, origin=(0.3, 0.0, 0.0) if case == "two boxes" else (5.0, 2.0, -3.0))
    parts = [(p1, t1), (p2, t2)]
    if case == "three bodies":
        parts.append(meshes.kuhn_box(3, 3, 3, 0.04, 0.04, 0.04, origin=(0.0, 0.4, 0.0)))
    rng = np.random.default_rng(5)
    pts, tets, off = [], [], 0
    for p, t in parts:
        if "jittered" in case: p = p + rng.uniform(-1, 1, p.shape) * 0.002
        pts.append(p); tets.append(t + off); off += len(p)
    pts, tets = np.concatenate(pts), np.concatenate(tets).astype(np.uint32)
    pairs = 45
    ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
    t0 = time.time()
    try:
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=pairs - 15, num_fem_modes=pairs))
        ev = r.eigenvalues
        msg = f"{len(ev)} pairs, {r.profile['restarts']} iterations, {1e3 * (time.time() - t0):.0f} ms, modes kept {len(r.freqs)}"
    except Exception as e:  # noqa: BLE001
        ev, msg = None, f"EXCEPTION {e!r} <- {e.__cause__!r}"[:300]
    evo, _, _ = pyoracle.System(pts, tets, pyoracle.material(*m)).eigs(pairs)
    nrigid = int((np.abs(evo) < 1e-6 * evo[-1]).sum())
    print(case, len(tets), "tets:", msg, "| oracle rigid", nrigid, flush=True)
    if ev is not None and len(ev) == len(evo):
        el = evo > 1e-6 * evo[-1]
        print("    max rel (elastic)", (np.abs(ev[el] - evo[el]) / evo[el]).max(), "rigid dev", np.abs(ev[~el]).max() / evo[el][0])
