"""Design prototype (CPU, scipy), round 5: the level BELOW P1.  The device cycle solves the P1 problem by gamma = 3 cycles of Chebyshev(5)-
Jacobi around an exact solve on rigid-body modes of graph aggregates (~30 corner nodes each, unsmoothed prolongator).  How good is one
such cycle as a preconditioner of the P1 operator, and what would a SMOOTHED prolongator (smoothed aggregation, P = (I - w D^-1 A) T)
or smaller aggregates buy?      python tools/proto/p1_sa.py scan 0.011 0.015 | cube 16"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fem
from smoothers import ChebyM, pcg
from thinwall import node_graph, greedy_aggregates, rbm_prolongator


class Cycle1:
    def __init__(self, A, S, P, gamma):
        self.A, self.S, self.P, self.gamma = A, S, P, gamma
        A0 = (P.T @ A @ P).toarray()
        self.c0 = sla.cho_factor(A0 + 1e-12 * np.diag(np.diag(A0)))

    def __call__(self, r):
        x = None
        for _ in range(self.gamma):
            x = self.S.apply(r, x)
            x = x + self.P @ sla.cho_solve(self.c0, self.P.T @ (r - self.A @ x))
            x = self.S.apply(r, x)
        return x


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
    K, M = fem.assemble_p1(pts, tets, *mat)[:2]
    sigma = -(2 * np.pi * 20.0) ** 2
    A = (K - sigma * M).tocsr()
    n = A.shape[0]
    nnod = n // 3
    print(f"P1: {nnod} nodes, {n} dof", flush=True)
    b = np.random.default_rng(0).standard_normal(n)
    d = 1.0 / A.diagonal()
    g = node_graph(A, nnod)
    lam = ChebyM(A, lambda r: d * r, 1).lmax / 1.1
    Dinv = sp.diags(d)
    for passes in (1, 2):
        agg, na = greedy_aggregates(g, passes)
        T = rbm_prolongator(pts, agg, na)
        Ts = (T - (4.0 / 3.0 / lam) * (Dinv @ (A @ T))).tocsr()
        for deg in (2, 5):
            S = ChebyM(A, lambda r: d * r, deg, 8.0)
            for name, P in (("plain", T), ("smoothed", Ts)):
                for gamma in (1, 3):
                    t0 = time.time()
                    it, kappa = pcg(A, b, Cycle1(A, S, P, gamma))
                    print(f"aggregates {na:5d} of {nnod / na:5.1f} nodes  Cheb({deg})  {name:9s} gamma {gamma}: pcg its {it:4d}  kappa {kappa:8.1f}  nnz(P) {P.nnz}  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
