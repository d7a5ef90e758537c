"""Odd WARM-START seeds (the reference's SubspaceIterate branch takes the previous solve's basis as it is): garbage, zeros, NaN, a basis of another body, too few columns."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()
ctx = api.Context(0)
pts, tets = meshes.kuhn_box(6, 5, 4, 0.3, 0.25, 0.2)
pts = pts + np.random.default_rng(3).uniform(-1, 1, pts.shape) * 0.004
m = meshes.MATERIALS["Ceramic"]
ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
cfg = api.default_config(num_modes=30, num_fem_modes=45)
cold = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=cfg, keep_basis=True)
basis = np.array(cold.basis, np.float32)
n = basis.shape[0]
evo = pyoracle.mesh2modes(pts, tets, pyoracle.material(*m), ex, config=pyoracle.default_config(num_modes=30, num_fem_modes=45)).eigenvalues
el = evo > 1e-6 * evo[-1]
print("cold:", len(cold.eigenvalues), "pairs,", cold.profile["restarts"], "iterations; basis", basis.shape)
rng = np.random.default_rng(1)
other_pts = pts * np.array([1.3, 0.8, 1.1])
other = np.array(api.mesh2modes(ctx, other_pts, tets, api.material(*m), ex, config=cfg, keep_basis=True).basis, np.float32)
seeds = {"its own basis": basis, "its own basis, columns reversed": basis[:, ::-1], "the basis of a stretched body": other, "Gaussian noise": rng.standard_normal(basis.shape).astype(np.float32),
         "zeros": np.zeros_like(basis), "one NaN": np.where(np.arange(basis.size).reshape(basis.shape) == 12345, np.nan, basis).astype(np.float32), "all columns equal": np.repeat(basis[:, 7:8], basis.shape[1], 1),
         "huge values (1e30)": basis * 1e30, "too few columns (ignored: cold)": basis[:, :20], "wrong row count (ignored: cold)": basis[:-3]}
for name, seed in seeds.items():
    t0 = time.time()
    try:
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=cfg, seed_basis=seed)
        ev = r.eigenvalues
        ok = len(ev) == len(evo) and (np.abs(ev[el] - evo[el]) / evo[el]).max()
        print(f"{name}: {len(ev)} pairs, {r.profile.get('restarts')} iterations, {1e3 * (time.time() - t0):.0f} ms, vs oracle {ok}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name}: ERROR {str(e)[:100]} <- {str(e.__cause__)[:160]}", flush=True)
