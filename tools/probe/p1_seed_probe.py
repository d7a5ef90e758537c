"""Experiment: how many LOBPCG iterations does the S100k solve need when it starts from the P1 (linear-tet Galerkin)
eigenvectors prolonged to the P2 space, instead of noise?  (MH_VERBOSE=1 prints the history.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from mesheditor_amd import api, meshes
name = sys.argv[1] if len(sys.argv) > 1 else "cube_s100k"
p, t, m, kw = meshes.workload(name)
ctx = api.Context(0)
s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*m))
sigma = -(2 * np.pi * 20.0) ** 2
nev = 65
t0 = time.time(); ev_cold, prof = s.eigs(nev, sigma, 1e-5); print("cold", prof["restarts"], time.time() - t0, flush=True)
t0 = time.time(); ev_cold, prof = s.eigs(nev, sigma, 1e-5); print("cold", prof["restarts"], time.time() - t0, flush=True)
K2, M2 = s.to_scipy()
en = s.element_nodes().astype(np.int64)
nn = s.node_count
verts = np.unique(en[:, :4])
vid = -np.ones(nn, np.int64); vid[verts] = np.arange(len(verts))
rows, cols, vals = [verts], [vid[verts]], [np.ones(len(verts))]
edges = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
mids = {}
for e, (a, b) in enumerate(edges):
    mid = en[:, 4 + e]; pa = vid[en[:, a]]; pb = vid[en[:, b]]
    u, first = np.unique(mid, return_index=True)
    rows += [u, u]; cols += [pa[first], pb[first]]; vals += [np.full(len(u), .5), np.full(len(u), .5)]
Pn = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nn, len(verts))).tocsr()
# mid nodes appear in several edge slots: average duplicates away
Pn.sum_duplicates()
rs = np.asarray(Pn.sum(axis=1)).ravel(); Pn = sp.diags(1.0 / rs) @ Pn
P = sp.kron(Pn, sp.identity(3), format="csr")
K1 = (P.T @ K2 @ P).tocsc(); M1 = (P.T @ M2 @ P).tocsc()
print("P1 dofs", K1.shape[0], "check rigid translation in null space:", abs(K2 @ (P @ np.tile([1.0, 0, 0], len(verts)))).max() / abs(K2).max(), flush=True)
t0 = time.time()
k = 80
w1, V1 = spla.eigsh(K1, k=k, M=M1, sigma=sigma, which="LM")
print("P1 eigsh", time.time() - t0, flush=True)
order = np.argsort(w1); w1, V1 = w1[order], V1[:, order]
print("P1 vs P2 eigenvalue ratio (modes 7, 20, 40, 65):", [float(w1[i] / ev_cold[i]) for i in (6, 19, 39, 64)], flush=True)
seed = np.asfortranarray(P @ V1, dtype=np.float32)
for tol_p1 in (None,):
    t0 = time.time(); ev_w, prof = s.eigs(nev, sigma, 1e-5, seed_basis=seed); print("seeded", prof["restarts"], time.time() - t0, flush=True)
print("eigenvalue agreement", np.abs(ev_w[6:] / ev_cold[6:] - 1).max())
# a sloppy P1 solve: perturb the P1 vectors by 1e-2 relative noise
rng = np.random.default_rng(0)
V1n = V1 + 1e-2 * np.linalg.norm(V1, axis=0) / np.sqrt(V1.shape[0]) * rng.standard_normal(V1.shape)
seed = np.asfortranarray(P @ V1n, dtype=np.float32)
ev_w, prof = s.eigs(nev, sigma, 1e-5, seed_basis=seed); print("seeded, 1e-2 noise", prof["restarts"], flush=True)
