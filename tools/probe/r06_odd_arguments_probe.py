"""Odd ARGUMENTS of mesh2modes, device and oracle side by side (same call, same inputs): what each returns -- pair counts, kept modes, or the error.  Each device call in this process,
the whole script under the caller's `timeout`."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()
ctx = api.Context(0)
pts, tets = meshes.kuhn_box(6, 5, 4, 0.3, 0.25, 0.2)
pts = pts + np.random.default_rng(3).uniform(-1, 1, pts.shape) * 0.004
cer = meshes.MATERIALS["Ceramic"]
ex10 = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
def both(name, p=pts, t=tets, m=cer, ex=ex10, **cfg):
    out = []
    for side, mod in (("device", api), ("oracle", pyoracle)):
        t0 = time.time()
        try:
            if side == "device":
                r = api.mesh2modes(ctx, p, t, api.material(*m), ex, config=api.default_config(**cfg))
            else:
                r = pyoracle.mesh2modes(p, t, pyoracle.material(*m), ex, config=pyoracle.default_config(**cfg))
            ev = np.asarray(r.eigenvalues)
            f = np.asarray(r.freqs)
            out.append((side, f"{len(ev)} pairs, {len(f)} modes kept, f0 {f[0] if len(f) else None}, {1e3 * (time.time() - t0):.0f} ms", ev, f))
        except Exception as e:  # noqa: BLE001
            out.append((side, f"ERROR {str(e)[:110]} <- {str(getattr(e, '__cause__', ''))[:110]}", None, None))
    agree = ""
    if out[0][2] is not None and out[1][2] is not None and len(out[0][2]) == len(out[1][2]) and len(out[0][2]):
        evo = out[1][2]; el = evo > 1e-6 * max(evo[-1], 1e-300)
        if el.any(): agree = f" | elastic eigenvalues agree to {(np.abs(out[0][2][el] - evo[el]) / evo[el]).max():.1e}"
        if len(out[0][3]) == len(out[1][3]) and len(out[0][3]): agree += f", freqs to {(np.abs(out[0][3] - out[1][3]) / out[1][3]).max():.1e}"
    print(f"{name}:\n    device: {out[0][1]}\n    oracle: {out[1][1]}{agree}", flush=True)
both("defaults")
both("seven pairs (one elastic)", num_fem_modes=7, num_modes=1)
both("six pairs (none elastic)", num_fem_modes=6, num_modes=1)
both("num_modes > num_fem_modes", num_fem_modes=20, num_modes=50)
both("num_modes = 0", num_fem_modes=30, num_modes=0)
both("min freq above max freq", min_mode_freq=5000.0, max_mode_freq=100.0)
both("a band with no mode in it", min_mode_freq=20.0, max_mode_freq=50.0)
both("tolerance 1e-2", tolerance=1e-2)
both("tolerance 1e-12", tolerance=1e-12)
both("no excitation position", ex=np.zeros((0, 3), np.float32))
both("excitation positions far away and NaN", ex=np.array([[1e6, 0, 0], [np.nan, 0, 0], [0.1, 0.1, 0.1]], np.float32))
both("Poisson ratio 0.4999", m=(1100.0, 1e7, 0.4999, 5.0, 1e-7))
both("Poisson ratio 0.5", m=(1100.0, 1e7, 0.5, 5.0, 1e-7))
both("zero density", m=(0.0, 7e10, 0.2, 5.0, 1e-7))
both("negative Young modulus", m=(2700.0, -7e10, 0.2, 5.0, 1e-7))
q = pts.copy(); q[17, 1] = np.nan
both("a NaN coordinate", p=q)
q = pts.copy(); q[17, 1] = np.inf
both("an infinite coordinate", p=q)
both("fundamental_freq given", fundamental_freq=440.0)
