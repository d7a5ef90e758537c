"""Two findings of tools/probe/r06_odd_meshes_probe.py, looked at: a body 1 km from the origin (device: no pairs), and a millimetre-scale body (device and oracle 4e-5 apart:
which of the two is right is decided by the scaling law lambda(s x) = lambda(x) / s^2 against the metre-scale solve)."""
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
m = meshes.MATERIALS["Ceramic"]
pairs = 45
cfg = api.default_config(num_modes=30, num_fem_modes=pairs)
def device(pts):
    ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
    try:
        r = api.mesh2modes(ctx, pts, base_t, api.material(*m), ex, config=cfg)
        return r.eigenvalues, r.profile
    except Exception as e:  # noqa: BLE001
        print("   device EXCEPTION", repr(e.__cause__)[:300]); return None, None
def oracle(pts):
    ev, _, _ = pyoracle.System(pts, base_t, pyoracle.material(*m)).eigs(pairs); return ev
ev0, _ = device(base_p); evo0 = oracle(base_p)
el = evo0 > 1e-6 * evo0[-1]
print("metre scale: device vs oracle", (np.abs(ev0[el] - evo0[el]) / evo0[el]).max())
for s in (1e-1, 1e-2, 1e-3, 1e-4):
    ev, prof = device(base_p * s); evo = oracle(base_p * s)
    law = evo0[el] / s ** 2
    print(f"scale {s:g}: device vs law {(np.abs(ev[el] - law) / law).max() if ev is not None and len(ev) == pairs else None}, oracle vs law {(np.abs(evo[el] - law) / law).max()}, iterations {prof and prof.get('restarts')}", flush=True)
for d in (1.0, 10.0, 100.0, 1000.0, 1e5):
    off = np.array([d, -0.5 * d, 0.25 * d])
    ev, prof = device(base_p + off); evo = oracle(base_p + off)
    print(f"offset {d:g} m: device pairs {None if ev is None else len(ev)}, device vs metre-scale {(np.abs(ev[el] - evo0[el]) / evo0[el]).max() if ev is not None and len(ev) == pairs else None}, oracle vs metre-scale {(np.abs(evo[el] - evo0[el]) / evo0[el]).max()}, iterations {prof and prof.get('restarts')}", flush=True)
