"""Design prototype (CPU, scipy), round 4: a coarse space for thin-walled scan meshes.

The P2 -> P1 two-grid pair is the weak link on one-element-thick walls.  Compared here, as preconditioners of PCG on
A x = b (A = K - sigma M, P2 level) with the SAME level-2 smoother (sliver patches + Jacobi, Chebyshev), coarse problems solved
exactly so that only the coarse SPACE is judged:
   p1        the P1 subspace (what the device cycle has)
   agg       rigid-body modes of graph-grown aggregates of P2 nodes, tentative prolongator (unsmoothed)
   sa        the same, prolongator smoothed once:  P = (I - w D^-1 A) T   (smoothed aggregation)
   p1+agg    both corrections, P1 first (multiplicative)
    python tools/proto/thinwall.py scan 0.020 0.026 | scan 0.011 0.015 | cube 10
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fem
from smoothers import ChebyM, PatchInverse, pcg, tet_quality


def node_graph(A, nnod):
    """Adjacency of the nodes (3x3 blocks of A)."""
    A = A.tocoo()
    g = sp.coo_matrix((np.ones(len(A.row)), (A.row // 3, A.col // 3)), shape=(nnod, nnod)).tocsr()
    g.data[:] = 1
    return g


def greedy_aggregates(g, passes=1):
    """Root + all free neighbours; leftovers join the neighbouring aggregate they touch most (the device's graph_aggregates rule)."""
    n = g.shape[0]
    agg = -np.ones(n, np.int64)
    indptr, indices = g.indptr, g.indices
    na = 0
    for i in range(n):
        nb = indices[indptr[i]:indptr[i + 1]]
        if agg[i] < 0 and np.all(agg[nb] < 0):
            agg[nb] = na
            agg[i] = na
            na += 1
    for i in np.where(agg < 0)[0]:
        nb = indices[indptr[i]:indptr[i + 1]]
        cand = agg[nb][agg[nb] >= 0]
        if len(cand):
            agg[i] = np.bincount(cand).argmax()
        else:
            agg[i] = na
            na += 1
    for _ in range(passes - 1):  # pairwise merging: coarsen further
        pair = -np.ones(na, np.int64)
        coarse = sp.coo_matrix((np.ones(g.nnz), (agg[g.tocoo().row], agg[g.tocoo().col])), shape=(na, na)).tocsr()
        new = 0
        for a in range(na):
            if pair[a] >= 0:
                continue
            nb = coarse.indices[coarse.indptr[a]:coarse.indptr[a + 1]]
            w = coarse.data[coarse.indptr[a]:coarse.indptr[a + 1]]
            best, bw = -1, 0
            for b_, w_ in zip(nb, w):
                if b_ != a and pair[b_] < 0 and w_ > bw:
                    best, bw = b_, w_
            pair[a] = new
            if best >= 0:
                pair[best] = new
            new += 1
        agg, na = pair[agg], new
    return agg, na


def rbm_prolongator(xyz, agg, na):
    n = len(xyz)
    cnt = np.bincount(agg, minlength=na)
    cent = np.stack([np.bincount(agg, weights=xyz[:, d], minlength=na) / cnt for d in range(3)], 1)
    r = xyz - cent[agg]
    rows, cols, vals = [], [], []
    for p in range(3):
        rows.append(3 * np.arange(n) + p); cols.append(6 * agg + p); vals.append(np.ones(n))
    eye = np.eye(3)
    for q in range(3):
        u = np.cross(np.broadcast_to(eye[q], r.shape), r)
        for p in range(3):
            rows.append(3 * np.arange(n) + p); cols.append(6 * agg + 3 + q); vals.append(u[:, p])
    T = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(3 * n, 6 * na)).tocsr()
    cn = np.sqrt(np.array(T.multiply(T).sum(0)).ravel())
    cn[cn == 0] = 1
    return T @ sp.diags(1 / cn)


class TwoLevel:
    """pre-smooth, exact corrections in the given coarse spaces (in order), post-smooth (mirrored order: symmetric)."""
    def __init__(self, A, S, spaces):
        self.A, self.S = A, S
        self.spaces = []
        for P in spaces:
            Ac = (P.T @ A @ P).tocsc()
            Ac = Ac + 1e-10 * sp.diags(Ac.diagonal())
            self.spaces.append((P, spla.splu(Ac)))

    def __call__(self, r):
        x = self.S.apply(r)
        for P, lu in self.spaces:
            x = x + P @ lu.solve(P.T @ (r - self.A @ x))
        for P, lu in reversed(self.spaces[:-1]):
            x = x + P @ lu.solve(P.T @ (r - self.A @ x))
        return self.S.apply(r, x)


def main():
    kind = sys.argv[1]
    mat = (8000, 2.1e11, 0.28)
    if kind == "scan":
        from mesheditor_amd import meshes
        pts, tets = meshes.skillet_scan_tets(float(sys.argv[2]), float(sys.argv[3]))
    else:
        n_ = int(sys.argv[2])
        pts, tets = fem.kuhn_box(n_, n_, n_, 0.3, 0.3, 0.3)
    tets = tets.astype(np.int64)
    K, M, nodes, nnod = fem.assemble_p2(pts, tets, *mat)
    sigma = -(2 * np.pi * 20.0) ** 2
    A = (K - sigma * M).tocsr()
    P21 = fem.p2_to_p1_prolongation(len(pts), nodes, nnod)
    xyz = fem.node_coords(pts, nodes, nnod)
    q = tet_quality(pts, tets)
    n = A.shape[0]
    print(f"tets {len(tets)} pts {len(pts)} P2 nodes {nnod} dof {n}  quality pct 1/10/50 {np.percentile(q, [1, 10, 50]).round(4)}", flush=True)
    b = np.random.default_rng(0).standard_normal(n)

    def dofs(nodeset):
        nodeset = np.asarray(nodeset)
        return (3 * nodeset[:, None] + np.arange(3)[None, :]).ravel()

    bad = np.where(q < 0.02)[0]
    minv = PatchInverse(A, [dofs(nodes[e]) for e in bad], jacobi="all") if len(bad) else None
    d = 1.0 / A.diagonal()
    apply_minv = minv if minv is not None else (lambda r: d * r)
    g = node_graph(A, nnod)
    agg1, na1 = greedy_aggregates(g, 1)
    agg2, na2 = greedy_aggregates(g, 2)
    print(f"aggregates: {na1} of {nnod / na1:.1f} nodes, {na2} of {nnod / na2:.1f} nodes; sliver patches {len(bad)}", flush=True)
    # smoothed prolongators
    lam = ChebyM(A, lambda r: d * r, 1).lmax / 1.1
    Dinv = sp.diags(d)

    def smoothed(T, omega=4.0 / 3.0):
        return (T - (omega / lam) * (Dinv @ (A @ T))).tocsr()

    T1, T2 = rbm_prolongator(xyz, agg1, na1), rbm_prolongator(xyz, agg2, na2)
    for deg, ratio in ((2, 8.0), (3, 16.0), (5, 60.0)):
        S = ChebyM(A, apply_minv, deg, ratio)
        for name, spaces in (("p1", [P21]), ("agg fine", [T1]), ("agg coarse", [T2]), ("sa fine", [smoothed(T1)]), ("sa coarse", [smoothed(T2)]),
                             ("p1 + agg coarse", [P21, T2]), ("p1 + sa coarse", [P21, smoothed(T2)])):
            t0 = time.time()
            cyc = TwoLevel(A, S, spaces)
            it, kappa = pcg(A, b, cyc)
            dims = "+".join(str(P.shape[1]) for P in spaces)
            print(f"Cheb({deg}) r{ratio:<4.0f} {name:18s} coarse dofs {dims:>12s}  pcg its {it:4d}  kappa {kappa:8.1f}  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
