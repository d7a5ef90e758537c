"""Eigenvalues (and a checksum of the eigenvectors) of a fixed set of solves, for bit-for-bit before/after comparisons of a
refactoring: python tools/dump_eigs.py <tag> writes gpurun_out/dump_<tag>.npz."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api, meshes
tag = sys.argv[1]
ctx = api.Context(0)
out = {}
sig = -(2 * np.pi * 20.0) ** 2
for name, nev, tol in [("cube_small", 20, 1e-6), ("bar_square", 30, 1e-6), ("bar_thin", 24, 1e-6), ("cube_s10k", 65, 1e-5), ("cube_s30k", 65, 1e-5), ("ball_s10k", 65, 1e-5)]:
    p, t, m, kw = meshes.workload(name)
    s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*m))
    ev, prof = s.eigs(nev, sig, tol)
    vec = s.eigenvectors(nev)
    out[name + "_ev"] = np.asarray(ev)
    out[name + "_vsum"] = np.array([np.abs(vec).sum(), (vec * vec).sum()])
    out[name + "_iters"] = np.array([prof["restarts"]])
    if name == "cube_s10k":  # warm start from the converged vectors with a little noise
        rng = np.random.default_rng(3)
        seed = (vec + 1e-3 * np.abs(vec).max() * rng.standard_normal(vec.shape)).astype(np.float32)
        ev2, prof2 = s.eigs(nev, sig, tol, seed_basis=seed)
        out["warm_ev"] = np.asarray(ev2)
        out["warm_iters"] = np.array([prof2["restarts"]])
    s.close()
os.makedirs("gpurun_out", exist_ok=True)
np.savez(f"gpurun_out/dump_{tag}.npz", **out)
print({k: (v.tolist() if v.size < 3 else float(v[-1])) for k, v in out.items() if k.endswith("iters") or k.endswith("_ev")})
