"""Design prototype (CPU, scipy), round 5: does a 3x3 NODAL block Jacobi in the level-2 smoother help on thin-walled scan meshes?
Same frame as thinwall.py: PCG on A x = b (A = K - sigma M, P2), two-level cycle with the P1 subspace solved exactly, level-2 smoother =
Chebyshev on M^-1 A with M^-1 = point Jacobi / nodal-block Jacobi, each with and without the sliver patches.
    python tools/proto/blockjacobi.py scan 0.020 0.026 | scan 0.011 0.015 | cube 10"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fem
from smoothers import ChebyM, PatchInverse, pcg, tet_quality
from thinwall import TwoLevel


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
    q = tet_quality(pts, tets)
    n = A.shape[0]
    print(f"tets {len(tets)} pts {len(pts)} P2 nodes {nnod} dof {n}  quality pct 1/10/50 {np.percentile(q, [1, 10, 50]).round(4)}", flush=True)
    b = np.random.default_rng(0).standard_normal(n)
    dofs = lambda ns: (3 * np.asarray(ns)[:, None] + np.arange(3)[None, :]).ravel()
    bad = np.where(q < 0.02)[0]
    d = 1.0 / A.diagonal()
    # nodal 3x3 blocks
    Ab = A.tobsr(blocksize=(3, 3))
    blocks = np.zeros((nnod, 3, 3))
    for i in range(nnod):
        for p in range(Ab.indptr[i], Ab.indptr[i + 1]):
            if Ab.indices[p] == i:
                blocks[i] = Ab.data[p]
    inv = np.linalg.inv(blocks)
    Binv = sp.bsr_matrix((inv, np.arange(nnod), np.arange(nnod + 1)), shape=(n, n)).tocsr()
    cond = np.array([np.linalg.cond(bk) for bk in blocks])
    print(f"nodal blocks: condition pct 50/90/99/max {np.percentile(cond, [50, 90, 99, 100]).round(1)}; sliver patches {len(bad)}", flush=True)
    patches = PatchInverse(A, [dofs(nodes[e]) for e in bad], jacobi="none") if len(bad) else None
    variants = {
        "point Jacobi": lambda r: d * r,
        "block Jacobi": lambda r: Binv @ r,
    }
    if patches is not None:
        variants["point Jacobi + patches"] = lambda r: d * r + patches(r)
        variants["block Jacobi + patches"] = lambda r: Binv @ r + patches(r)
    for deg, ratio in ((2, 8.0), (5, 60.0)):
        for name, minv in variants.items():
            t0 = time.time()
            S = ChebyM(A, minv, deg, ratio)
            cyc = TwoLevel(A, S, [P21])
            it, kappa = pcg(A, b, cyc)
            print(f"Cheb({deg}) r{ratio:<4.0f} {name:26s} lmax {S.lmax / 1.1:7.3f}  pcg its {it:4d}  kappa {kappa:8.1f}  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
