"""Bodies joined at ONE vertex / along ONE edge (mechanisms: 3 / 1 zero-energy modes beyond the six rigid-body ones, not in the cold start's block) against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()
ctx = api.Context(0)
m = meshes.MATERIALS["Ceramic"]
def weld(parts):
    pts = np.concatenate([p for p, _ in parts]); off = np.cumsum([0] + [len(p) for p, _ in parts[:-1]])
    tets = np.concatenate([t + o for (_, t), o in zip(parts, off)]).astype(np.uint32)
    key = np.round(pts * 1e9).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    return pts[first], inv.reshape(-1)[tets].astype(np.uint32)
a = meshes.kuhn_box(4, 4, 4, 0.08, 0.08, 0.08)
cases = {"joined at one vertex": weld([a, meshes.kuhn_box(4, 4, 4, 0.08, 0.08, 0.08, origin=(0.08, 0.08, 0.08))]),
         "joined along one edge": weld([a, meshes.kuhn_box(4, 4, 4, 0.08, 0.08, 0.08, origin=(0.08, 0.08, 0.0))]),
         "joined on one face (control)": weld([a, meshes.kuhn_box(4, 4, 4, 0.08, 0.08, 0.08, origin=(0.08, 0.0, 0.0))])}
for name, (pts, tets) in cases.items():
    pairs = 45
    ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
    t0 = time.time()
    try:
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=30, num_fem_modes=pairs))
        ev, msg = r.eigenvalues, f"{len(r.eigenvalues)} pairs, {r.profile.get('restarts')} iterations, {1e3 * (time.time() - t0):.0f} ms"
    except Exception as e:  # noqa: BLE001
        ev, msg = None, f"EXCEPTION {e!r} <- {e.__cause__!r}"[:300]
    evo, _, _ = pyoracle.System(pts, tets, pyoracle.material(*m)).eigs(pairs)
    el = evo > 1e-6 * evo[-1]
    print(f"{name}: {len(pts)} points {len(tets)} tets: {msg} | oracle zero modes {int((~el).sum())}", flush=True)
    if ev is not None and len(ev) == pairs:
        print("    max rel (elastic)", (np.abs(ev[el] - evo[el]) / evo[el]).max(), "zero-mode dev", np.abs(ev[~el]).max() / evo[el][0])
