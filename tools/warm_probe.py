import sys, time, numpy as np
sys.path.insert(0, '.')
from mesheditor_amd import api, meshes
ctx = api.Context(0)
p, t, m, kw = meshes.workload("cube_s100k")
mesh = api.Mesh(ctx, p, t)
s = api.System(ctx, mesh, api.material(*m))
sig = -(2 * np.pi * 20.0) ** 2
for rep in range(2):
    t0 = time.perf_counter(); ev, prof = s.eigs(65, sig, 1e-5); ctx.synchronize(); dt = time.perf_counter() - t0
    print("cold", rep, "%.1f ms" % (1e3 * dt), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in prof.items()} if isinstance(prof, dict) else prof)
basis = s.eigenvectors(65).astype(np.float32)  # n x 65, reference numbering
rng = np.random.default_rng(1)
for noise in (0.0, 1e-3, 1e-2, 1e-1):
    seed = basis + noise * np.abs(basis).max() * rng.standard_normal(basis.shape).astype(np.float32)
    t0 = time.perf_counter(); ev2, prof2 = s.eigs(65, sig, 1e-5, seed_basis=seed); ctx.synchronize(); dt = time.perf_counter() - t0
    print("warm noise", noise, "%.1f ms" % (1e3 * dt), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in prof2.items()} if isinstance(prof2, dict) else prof2)
