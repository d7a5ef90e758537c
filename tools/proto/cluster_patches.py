"""Design prototype (CPU, scipy), round 6: exact patches over CLUSTERS of badly shaped elements against the device's weighted per-element patches.
Two-grid method (P1 level solved exactly, so that only the level-2 smoother is judged), PCG on A x = b, iterations to 1e-8 and condition estimate.
    python tools/proto/cluster_patches.py scan 0.016 0.022 | sphere 64 32 | sphere 96 48
"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fem
from smoothers import ChebyM, pcg, tet_quality
from thinwall import TwoLevel


class Patches:
    """M^-1 = D^-1 + sum_p w_p R_p^T inv(A[p,p]) R_p"""
    def __init__(self, A, patches, weights=None, jacobi=True):
        n = A.shape[0]
        A = A.tocsr()
        rows, cols, vals = [], [], []
        covered = np.zeros(n, bool)
        for k, p in enumerate(patches):
            sub = A[p][:, p].toarray()
            inv = np.linalg.inv(sub) * (1.0 if weights is None else weights[k])
            rows.append(np.repeat(p, len(p))); cols.append(np.tile(p, len(p))); vals.append(inv.ravel())
            covered[p] = True
        if jacobi:
            d = 1.0 / A.diagonal()
            if jacobi == "uncovered":
                d = np.where(covered, 0.0, d)
            rows.append(np.arange(n)); cols.append(np.arange(n)); vals.append(d)
        self.B = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()

    def __call__(self, r):
        return self.B @ r


class PatchesChol:
    """The same operator applied through Cholesky factors (backward stable) instead of explicit inverses."""
    def __init__(self, A, patches, jacobi=True):
        import scipy.linalg as sla
        A = A.tocsr()
        self.f = [(p, sla.cho_factor(A[p][:, p].toarray())) for p in patches]
        self.d = 1.0 / A.diagonal() if jacobi else np.zeros(A.shape[0])
        self.sla = sla

    def __call__(self, r):
        y = self.d * r
        for p, c in self.f:
            y[p] += self.sla.cho_solve(c, r[p])
        return y


def dofs(nodeset):
    nodeset = np.asarray(nodeset)
    return (3 * nodeset[:, None] + np.arange(3)[None, :]).ravel()


def clusters_of(bad, nodes, nnod, cap):
    """Connected components of the bad elements (sharing a node), each the union of its elements' nodes; components larger than `cap` nodes are cut
    into chunks in breadth-first element order."""
    if len(bad) == 0:
        return []
    owner = {}
    parent = list(range(len(bad)))
    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for k, e in enumerate(bad):
        for v in nodes[e]:
            if v in owner:
                a, b = find(owner[v]), find(k)
                if a != b:
                    parent[a] = b
            else:
                owner[v] = k
    comps = {}
    for k in range(len(bad)):
        comps.setdefault(find(k), []).append(k)
    out = []
    for members in comps.values():
        cur = []
        seen = set()
        for k in members:
            new = [v for v in nodes[bad[k]] if v not in seen]
            if cur and len(cur) + len(new) > cap:
                out.append(np.array(cur)); cur = []; seen = set(); new = list(nodes[bad[k]])
            cur += new; seen.update(new)
        if cur:
            out.append(np.array(cur))
    return out


def main():
    kind = sys.argv[1]
    from mesheditor_amd import meshes
    if kind == "scan":
        pts, tets = meshes.skillet_scan_tets(float(sys.argv[2]), float(sys.argv[3]))
        mat = (8000, 2.1e11, 0.28)
    else:
        from mesheditor_amd import tets as T
        # (the numbers recorded in docs/LAB_NOTEBOOK.md section 13 were made on the fill the front end of early round 6 gave with break_flat_cells=False: 172 caps
        # at 1e-8 on the 96 x 48 sphere; the final front end fills these spheres without flat cells, so the flat cells are now made: 60 interior points moved to
        # 1e-6 of their height over a face)
        P, F = meshes.uv_sphere_surface(0.15, int(sys.argv[2]), int(sys.argv[3]))
        pts, tets, _ = T.tetrahedralize(P, F)
        pts, _ = meshes.with_flat_cells(pts, tets, len(P), count=60, eps=1e-6, seed=int(sys.argv[2]))
        mat = (2700, 7.2e10, 0.19)
    tets = tets.astype(np.int64)
    K, M, nodes, nnod = fem.assemble_p2(pts, tets, *mat)
    sigma = -(2 * np.pi * 20.0) ** 2
    A = (K - sigma * M).tocsr()
    P21 = fem.p2_to_p1_prolongation(len(pts), nodes, nnod)
    q = tet_quality(pts, tets)
    n = A.shape[0]
    print(f"tets {len(tets)} pts {len(pts)} P2 nodes {nnod} dof {n}  quality min {q.min():.1e} pct 1/10 {np.percentile(q, [1, 10]).round(4)}", flush=True)
    b = np.random.default_rng(0).standard_normal(n)
    thr = float(os.environ.get("THR", "0.02"))
    bad = np.where(q < thr)[0]
    # the device's weights: (largest number of patches at any of a patch's nodes)^-0.35
    cover = np.zeros(nnod)
    for e in bad:
        cover[nodes[e]] += 1
    w = np.array([cover[nodes[e]].max() ** -0.35 for e in bad]) if len(bad) else None
    variants = [("jacobi only", lambda: Patches(A, [], None)),
                ("element patches, weighted (device)", lambda: Patches(A, [dofs(nodes[e]) for e in bad], w))]
    if os.environ.get("SKIP_BASE"):
        variants = []
    for cap in [int(c) for c in os.environ.get("CAPS", "64,128,256").split(",")]:
        cl = clusters_of(bad, nodes, nnod, cap)
        sizes = np.array([len(c) for c in cl]) if cl else np.array([0])
        if os.environ.get("CHOL"):
            variants.append((f"cluster patches cap {cap} through Cholesky factors: {len(cl)} patches, nodes max {sizes.max()}", (lambda cl=cl: PatchesChol(A, [dofs(c) for c in cl]))))
        for jac in (True,):
            variants.append((f"cluster patches cap {cap} jacobi {jac}: {len(cl)} patches, nodes max {sizes.max()} mean {sizes.mean():.0f}, {int((3*sizes)**2 @ np.ones(len(sizes)) * 8 / 1e6)} MB", (lambda cl=cl, jac=jac: Patches(A, [dofs(c) for c in cl], None, jac))))
    print(f"bad elements (q < {thr}): {len(bad)}", flush=True)
    for deg, ratio in ((5, 60.0),):
        for name, make in variants:
            t0 = time.time()
            S = ChebyM(A, make(), deg, ratio)
            cyc = TwoLevel(A, S, [P21])
            it, kappa = pcg(A, b, cyc, maxit=300)
            print(f"Cheb({deg}) r{ratio:<4.0f} {name:100s} lmax {S.lmax:7.2f} pcg its {it:4d}  kappa {kappa:10.1f}  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
