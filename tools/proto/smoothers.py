"""Design prototype (CPU, scipy): which level-2 smoother makes the P2 -> P1 -> aggregates cycle robust on a sliver-rich
unstructured mesh?  PCG on A x = b with the cycle as preconditioner; iterations to 1e-8 and the condition estimate.
    python tools/proto/smoothers.py scan 0.016 0.022   |   cube 8
"""
import sys, time, os
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fem
from solver import rbm_aggregates, morton_order


def tet_quality(pts, tets):
    p = pts[tets.astype(np.int64)]
    vol = np.abs(np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])) / 6
    e = np.stack([np.linalg.norm(p[:, i] - p[:, j], axis=1) for i in range(4) for j in range(i + 1, 4)], 1)
    return vol * 6 * np.sqrt(2) / np.sqrt((e ** 2).mean(1)) ** 3


class PatchInverse:
    """M^-1 = sum over patches R^T inv(A[p,p]) R (+ optional Jacobi on the dofs no patch covers / everywhere)."""
    def __init__(self, A, patches, jacobi="uncovered", weight=None):
        n = A.shape[0]
        A = A.tocsr()
        rows, cols, vals = [], [], []
        covered = np.zeros(n, bool)
        for p in patches:
            sub = A[p][:, p].toarray()
            inv = np.linalg.inv(sub)
            rows.append(np.repeat(p, len(p))); cols.append(np.tile(p, len(p))); vals.append(inv.ravel())
            covered[p] = True
        d = 1.0 / A.diagonal()
        if jacobi == "uncovered":
            d = np.where(covered, 0.0, d)
        elif jacobi == "none":
            d = np.zeros(n)
        rows.append(np.arange(n)); cols.append(np.arange(n)); vals.append(d)
        self.B = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
        self.nnz = self.B.nnz

    def __call__(self, r):
        return self.B @ r


class ChebyM:
    """Chebyshev on M^-1 A with a general SPD M^-1 (callable)."""
    def __init__(self, A, minv, degree, ratio=8.0):
        self.A, self.minv, self.degree = A, minv, degree
        rng = np.random.default_rng(1)
        v = rng.standard_normal(A.shape[0])
        for _ in range(20):
            v = minv(A @ v)
            lm = np.linalg.norm(v)
            v /= lm
        self.lmax, self.lmin = 1.1 * lm, 1.1 * lm / ratio

    def apply(self, b, x=None):
        A = self.A
        theta, delta = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        if x is None:
            x = np.zeros_like(b); r = b.copy()
        else:
            r = b - A @ x
        d = self.minv(r) / theta
        for k in range(self.degree):
            x = x + d
            if k == self.degree - 1:
                break
            r = r - A @ d
            rho_new = 1.0 / (2 * sigma - rho)
            d = rho_new * rho * d + (2 * rho_new / delta) * self.minv(r)
            rho = rho_new
        return x


class Cycle:
    def __init__(self, A2, A1, P, T, S2, deg1=4, gamma=3):
        self.A2, self.A1, self.P, self.T, self.S2, self.gamma = A2, A1, P, T, S2, gamma
        d1 = 1.0 / A1.diagonal()
        self.S1 = ChebyM(A1, lambda r: d1 * r, deg1)
        A0 = (T.T @ A1 @ T).toarray()
        self.c0 = sla.cho_factor(A0 + 1e-12 * np.diag(np.diag(A0)))

    def __call__(self, r):
        x2 = self.S2.apply(r)
        r1 = self.P.T @ (r - self.A2 @ x2)
        x1 = None
        for g in range(self.gamma):
            x1 = self.S1.apply(r1, x1)
            x1 = x1 + self.T @ sla.cho_solve(self.c0, self.T.T @ (r1 - self.A1 @ x1))
            x1 = self.S1.apply(r1, x1)
        x2 = x2 + self.P @ x1
        return self.S2.apply(r, x2)


def pcg(A, b, prec, tol=1e-8, maxit=400):
    x = np.zeros_like(b); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z
    alphas, betas = [], []
    r0 = np.linalg.norm(r)
    for it in range(1, maxit + 1):
        Ap = A @ p
        a = rz / (p @ Ap)
        x += a * p; r -= a * Ap
        if np.linalg.norm(r) < tol * r0:
            break
        z = prec(r); rz_new = r @ z; beta = rz_new / rz; rz = rz_new
        p = z + beta * p
        alphas.append(a); betas.append(beta)
    # Lanczos tridiagonal from the CG coefficients -> extreme eigenvalues of the preconditioned operator
    m = len(alphas)
    Tm = np.zeros((m, m))
    for i in range(m):
        Tm[i, i] = 1 / alphas[i] + (betas[i - 1] / alphas[i - 1] if i else 0)
        if i + 1 < m:
            Tm[i, i + 1] = Tm[i + 1, i] = np.sqrt(betas[i]) / alphas[i]
    ev = np.linalg.eigvalsh(Tm) if m else np.array([1.0])
    return it, ev.max() / ev.min()


def main():
    kind = sys.argv[1]
    if kind == "scan":
        from mesheditor_amd import meshes
        h, th = float(sys.argv[2]), float(sys.argv[3])
        pts, tets = meshes.skillet_scan_tets(h, th)
        mat = (8000, 2.1e11, 0.28)
    else:
        nn_ = int(sys.argv[2])
        pts, tets = fem.kuhn_box(nn_, nn_, nn_, 0.3, 0.3, 0.3); mat = (8000, 2.1e11, 0.28)
    tets = tets.astype(np.int64)
    # positive orientation
    p = pts[tets]
    det = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])
    K, M, nodes, nnod = fem.assemble_p2(pts, tets, *mat)
    sigma = -(2 * np.pi * 20.0) ** 2
    A2 = (K - sigma * M).tocsr()
    K1, M1 = fem.assemble_p1(pts, tets, *mat)
    A1 = (K1 - sigma * M1).tocsr()
    P21 = fem.p2_to_p1_prolongation(len(pts), nodes, nnod)
    T, nagg = rbm_aggregates(pts, 32)
    q = tet_quality(pts, tets)
    print(f"tets {len(tets)} pts {len(pts)} P2 nodes {nnod} dof {A2.shape[0]}  quality pct 1/10/50 {np.percentile(q, [1, 10, 50]).round(4)}", flush=True)
    n = A2.shape[0]
    rng = np.random.default_rng(0)
    b = rng.standard_normal(n)
    d2 = 1.0 / A2.diagonal()
    xyz = fem.node_coords(pts, nodes, nnod)
    order = morton_order(xyz)

    def dofs(nodeset):
        nodeset = np.asarray(nodeset)
        return (3 * nodeset[:, None] + np.arange(3)[None, :]).ravel()

    def run(name, S2):
        t0 = time.time()
        cyc = Cycle(A2, A1, P21, T, S2)
        it, kappa = pcg(A2, b, cyc)
        print(f"{name:58s} pcg its {it:4d}  kappa {kappa:9.1f}  ({time.time() - t0:.0f}s)", flush=True)

    run("point Jacobi, Chebyshev(2), ratio 8  [current]", ChebyM(A2, lambda r: d2 * r, 2, 8))
    run("point Jacobi, Chebyshev(6), ratio 30", ChebyM(A2, lambda r: d2 * r, 6, 30))
    # 3x3 node blocks
    nb = PatchInverse(A2, [dofs([i]) for i in range(nnod)], jacobi="none")
    run("3x3 node-block Jacobi, Chebyshev(2), ratio 8", ChebyM(A2, nb, 2, 8))
    # Morton runs of 42 nodes (126 dof), non-overlapping
    for size in (42, 170):
        runs = [dofs(order[i:i + size]) for i in range(0, nnod, size)]
        pm = PatchInverse(A2, runs, jacobi="none")
        run(f"Morton-run blocks of {size} nodes, Chebyshev(2), ratio 8", ChebyM(A2, pm, 2, 8))
        shifted = [dofs(order[max(0, i):i + size]) for i in range(-size // 2, nnod, size)]
        pm2 = PatchInverse(A2, runs + shifted, jacobi="none")
        run(f"two shifted partitions of {size}-node runs (additive), Cheb(2)", ChebyM(A2, pm2, 2, 8))
    # element patches on the worst elements + Jacobi on the rest
    for frac in (0.1, 0.3, 1.0):
        thr = np.quantile(q, frac) if frac < 1 else np.inf
        bad = np.where(q <= thr)[0]
        ep = PatchInverse(A2, [dofs(nodes[e]) for e in bad], jacobi="uncovered" if frac < 1 else "none")
        run(f"element patches on the worst {int(100 * frac)}% tets + Jacobi elsewhere, Cheb(2)", ChebyM(A2, ep, 2, 8))
        ep2 = PatchInverse(A2, [dofs(nodes[e]) for e in bad], jacobi="all")
        run(f"element patches on the worst {int(100 * frac)}% tets + Jacobi everywhere, Cheb(2)", ChebyM(A2, ep2, 2, 8))


if __name__ == "__main__":
    main()


class CycleExact1(Cycle):
    """Level 1 solved exactly: what the two-grid P2/P1 method alone can do."""
    def __init__(self, A2, A1, P, S2):
        self.A2, self.A1, self.P, self.S2 = A2, A1, P, S2
        self.lu = spla.splu(A1.tocsc())

    def __call__(self, r):
        x2 = self.S2.apply(r)
        x2 = x2 + self.P @ self.lu.solve(self.P.T @ (r - self.A2 @ x2))
        return self.S2.apply(r, x2)
