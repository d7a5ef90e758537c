"""Prototype: would eigenvectors of the LINEAR-tet (P1) discretisation, interpolated to the quadratic nodes, cut the iteration count of
the quadratic solve when used as the start block?  P1 pairs from scipy (shift-invert) on the host -- the prototype asks about the
iteration count only, not about the cost of getting the seeds.
    python tools/proto/p1_seed.py cube_s10k [cube_s30k ...]"""
import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes

def linear_tet_system(pts, tets, E, nu, rho):
    n = len(pts)
    lam, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    D = np.zeros((6, 6)); D[:3, :3] = lam; D[np.arange(3), np.arange(3)] += 2 * mu; D[3:, 3:] = np.eye(3) * mu
    p = pts[tets]                                     # T x 4 x 3
    J = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=1)  # rows = edges
    vol = np.abs(np.linalg.det(J)) / 6
    Jinv = np.linalg.inv(J)                           # columns = gradients of lambda1..3
    g = np.zeros((len(tets), 4, 3)); g[:, 1:] = np.transpose(Jinv, (0, 2, 1)); g[:, 0] = -g[:, 1:].sum(1)
    B = np.zeros((len(tets), 6, 12))
    for a in range(4):
        gx, gy, gz = g[:, a, 0], g[:, a, 1], g[:, a, 2]
        B[:, 0, 3 * a] = gx; B[:, 1, 3 * a + 1] = gy; B[:, 2, 3 * a + 2] = gz
        B[:, 3, 3 * a] = gy; B[:, 3, 3 * a + 1] = gx
        B[:, 4, 3 * a + 1] = gz; B[:, 4, 3 * a + 2] = gy
        B[:, 5, 3 * a] = gz; B[:, 5, 3 * a + 2] = gx
    Ke = np.einsum("tji,jk,tkl,t->til", B, D, B, vol)
    Mloc = (np.ones((4, 4)) + np.eye(4)) / 20.0
    Me = np.einsum("ab,t->tab", Mloc, rho * vol)
    dof = (3 * tets[:, :, None] + np.arange(3)).reshape(len(tets), 12)
    rows, cols = np.repeat(dof, 12, axis=1).ravel(), np.tile(dof, (1, 12)).ravel()
    K = sp.csc_matrix((Ke.ravel(), (rows, cols)), shape=(3 * n, 3 * n))
    mr, mc = np.repeat(tets, 4, axis=1).ravel(), np.tile(tets, (1, 4)).ravel()
    Mn = sp.csc_matrix((Me.ravel(), (mr, mc)), shape=(n, n))
    M = sp.kron(Mn, sp.identity(3), format="csc")
    return K, M

ctx = api.Context(0)
for name in sys.argv[1:] or ["cube_s10k"]:
    pts, tets, m, kw = meshes.workload(name)
    rho, E, nu = m[0], m[1], m[2]
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    nev, sigma = 65, -(2 * np.pi * 20.0) ** 2
    cold, pc = s.eigs(nev, sigma, 1e-6)
    t0 = time.perf_counter()
    K, M = linear_tet_system(np.asarray(pts, float), np.asarray(tets, np.int64), E, nu, rho)
    b = 80
    vals, vecs = spla.eigsh(K, k=b, M=M, sigma=sigma, which="LM")
    order = np.argsort(vals); vals, vecs = vals[order], vecs[:, order]
    t_host = time.perf_counter() - t0
    en = s.element_nodes().astype(np.int64)          # T' x 10: corners then the six edge midpoints (reference order)
    kept = en[:, :4]
    n2 = int(en.max()) + 1
    U1 = vecs.reshape(len(pts), 3, b)
    # quadratic node values: corners copy, midpoints average their two corners (edge order of the element: find it from coordinates)
    corner_of = np.full(n2, -1, np.int64)
    # the element's corner NODE ids map to mesh points through the tets the system kept: recover by position
    P2 = np.zeros((n2, 3, b))
    # mesh points of each kept tet: the system may have dropped degenerate tets; match kept elements to tets by order (none dropped on these workloads)
    assert len(en) == len(tets)
    corner_of[kept.ravel()] = np.asarray(tets, np.int64).ravel()
    P2[kept.ravel()] = U1[np.asarray(tets, np.int64).ravel()]
    edges = [(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)]
    # which local edge each of the six midpoint slots belongs to: decide once from the node positions of element 0 through the P1 map
    xyz = np.zeros((n2, 3)); xyz[kept.ravel()] = np.asarray(pts, float)[np.asarray(tets, np.int64).ravel()]
    filled = corner_of >= 0
    # midpoint coordinates are not exported; assume the reference's edge order and verify on the spectrum below (a wrong order would ruin the seeds)
    for slot, (a, c) in enumerate(edges):
        mid = en[:, 4 + slot]
        P2[mid] = 0.5 * (U1[np.asarray(tets, np.int64)[:, a]] + U1[np.asarray(tets, np.int64)[:, c]])
    seed = P2.reshape(3 * n2, b).astype(np.float32)
    warm, pw = s.eigs(nev, sigma, 1e-6, seed_basis=seed)
    el = cold > 1e-6 * cold[-1]
    print(f"{name}: cold {pc['restarts']:.0f} iterations, P1-seeded {pw['restarts']:.0f} iterations; max rel diff of the eigenvalues {np.abs(warm[el] - cold[el]).max() / cold[el].max():.1e}; "
          f"P1 eigenvalue 7 / P2 eigenvalue 7 = {vals[6] / cold[6]:.3f}, pair 65: {vals[64] / cold[64]:.3f}  (host P1 solve {t_host:.1f} s)", flush=True)
    s.close(); mesh.close()
