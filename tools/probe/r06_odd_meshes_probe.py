"""Odd but legal TetMesh inputs through mesh2modes against the oracle: inverted tetrahedra, unused points, duplicate tetrahedra, a body far from the origin, millimetre and
kilometre scales, a soft and a very stiff material, one tetrahedron."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()
ctx = api.Context(0)
base_p, base_t = meshes.kuhn_box(7, 6, 5, 0.14, 0.12, 0.1)
rng = np.random.default_rng(9)
base_p = base_p + rng.uniform(-1, 1, base_p.shape) * 0.003
cer = meshes.MATERIALS["Ceramic"]
cases = []
t = base_t.copy(); flip = rng.random(len(t)) < 0.3; t[flip] = t[flip][:, [1, 0, 2, 3]]
cases.append(("30 % of the tetrahedra inverted", base_p, t, cer))
cases.append(("200 unused points", np.vstack([base_p, rng.uniform(-1, 1, (200, 3))]), base_t, cer))
cases.append(("every tenth tetrahedron twice", base_p, np.vstack([base_t, base_t[::10]]), cer))
cases.append(("1 km from the origin", base_p + np.array([1000.0, -500.0, 250.0]), base_t, cer))
cases.append(("millimetre scale", base_p * 1e-3, base_t, cer))
cases.append(("100 m scale", base_p * 1e3, base_t, cer))
cases.append(("soft rubber (E = 1e6, nu = 0.49)", base_p, base_t, (1100.0, 1e6, 0.49, 30.0, 1e-6)))
cases.append(("stiff (E = 1e12, nu = 0.05)", base_p, base_t, (3500.0, 1e12, 0.05, 1.0, 1e-9)))
cases.append(("one tetrahedron", np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1]], float), np.array([[0, 1, 2, 3]], np.uint32), cer))
cases.append(("two tetrahedra", np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1], [0.1, 0.1, 0.1]], float), np.array([[0, 1, 2, 3], [1, 2, 3, 4]], np.uint32), cer))
for name, pts, tets, m in cases:
    pairs = 45 if len(tets) > 10 else 12
    ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
    cfg = api.default_config(num_modes=max(1, pairs - 15), num_fem_modes=pairs)
    t0 = time.time()
    try:
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=cfg)
        ev, msg = r.eigenvalues, f"{len(r.eigenvalues)} pairs, {r.profile.get('restarts')} iterations, {1e3 * (time.time() - t0):.0f} ms, modes kept {len(r.freqs)}, mass {r.mass:.6g}"
    except Exception as e:  # noqa: BLE001
        ev, msg = None, f"EXCEPTION {e!r} <- {e.__cause__!r}"[:260]
    try:
        so = pyoracle.System(pts, tets, pyoracle.material(*m))
        evo, _, _ = so.eigs(pairs)
        omsg = f"oracle {len(evo)} pairs"
    except Exception as e:  # noqa: BLE001
        evo, omsg = None, f"oracle EXCEPTION {e!r}"[:200]
    print(f"{name}: {msg} | {omsg}", flush=True)
    if ev is not None and evo is not None and len(ev) == len(evo) and len(ev):
        el = evo > 1e-6 * evo[-1]
        print("    max rel (elastic)", (np.abs(ev[el] - evo[el]) / evo[el]).max(), "rigid dev", np.abs(ev[~el]).max() / evo[el][0] if (~el).any() else None, "oracle rigid", int((~el).sum()))
