import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
from make_flat_fill_surfaces import soak_surface
from mesheditor_amd import api, tets as fe, meshes
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()
ctx = api.Context(0)
for seed, index in ((33, 31), (31, 3)):
    rng = np.random.default_rng(700000 * seed + index)
    P, F, name = soak_surface(seed, index)
    opts = dict(quality=bool(rng.random() < 0.2), max_volume=0.0, interior_shell=str(rng.choice(["when_flat", "never", "always"])), repair_slivers=bool(rng.random() < 0.65), break_flat_cells=bool(rng.random() < 0.7))
    if rng.random() < 0.2:
        a, b, c = P[F[:, 0].astype(np.int64)], P[F[:, 1].astype(np.int64)], P[F[:, 2].astype(np.int64)]
        opts["max_volume"] = float(abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6 / rng.integers(2000, 12000))
    p, t, left = fe.tetrahedralize(P, F, **opts)
    pairs = int(rng.choice([30, 45, 65]))
    m = meshes.MATERIALS[meshes.MATERIAL_ORDER[index % len(meshes.MATERIAL_ORDER)]]
    ex = p[(np.arange(10) * len(P)) // 10].astype(np.float32)
    r = api.mesh2modes(ctx, p, t, api.material(*m), ex, config=api.default_config(num_modes=pairs - 15, num_fem_modes=pairs))
    s = pyoracle.System(p, t, pyoracle.material(*m)); ev, _, _ = s.eigs(pairs)
    el = ev > 1e-6 * ev[-1]
    rel = np.abs(r.eigenvalues[el] - ev[el]) / ev[el]
    print(seed, index, name, opts, len(t), "pairs", len(r.eigenvalues), "at floor", r.profile.get("pairs_at_floor"), "its", r.profile["restarts"], "elastic", el.sum(), "max rel", rel.max(), "rigid dev", np.abs(r.eigenvalues[~el]).max() / ev[el][0], "oracle rigid", np.abs(ev[~el]).max() / ev[el][0])
    print("   worst five:", np.sort(rel)[-5:])
