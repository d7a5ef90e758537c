"""Design prototype: LOBPCG + 3-level preconditioner (P2 -> P1 -> rigid-body aggregates, dense coarse).

Throw-away numerical testbed for the GPU eigensolver design.  Usage:
    python tools/proto/solver.py cube 12 65
"""
import sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import scipy.linalg as sla
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import fem


def morton_order(xyz, bits=10):
    lo, hi = xyz.min(0), xyz.max(0)
    q = ((xyz - lo) / np.maximum(hi - lo, 1e-300).max() * ((1 << bits) - 1)).astype(np.uint64)
    def spread(v):
        out = np.zeros_like(v)
        for b in range(bits):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out
    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def rbm_aggregates(pts, agg_size):
    """Tentative prolongator T (3*npts x 6*nagg) from Morton-run aggregates."""
    order = morton_order(pts)
    npts = len(pts)
    nagg = max(1, npts // agg_size)
    agg = np.empty(npts, dtype=np.int64)
    agg[order] = np.minimum(np.arange(npts) // agg_size, nagg - 1)
    cent = np.zeros((nagg, 3))
    cnt = np.bincount(agg, minlength=nagg)
    for d in range(3):
        cent[:, d] = np.bincount(agg, weights=pts[:, d], minlength=nagg) / cnt
    r = pts - cent[agg]
    rows, cols, vals = [], [], []
    for p in range(3):
        rows.append(3 * np.arange(npts) + p); cols.append(6 * agg + p); vals.append(np.ones(npts))
    # rotation columns: u = w x r ; w = e_q -> u = e_q x r
    eye = np.eye(3)
    for q in range(3):
        u = np.cross(np.broadcast_to(eye[q], r.shape), r)
        for p in range(3):
            rows.append(3 * np.arange(npts) + p); cols.append(6 * agg + 3 + q); vals.append(u[:, p])
    T = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(3 * npts, 6 * nagg)).tocsr()
    # scale rotation columns to unit-ish norm
    cn = np.sqrt(np.array(T.multiply(T).sum(0)).ravel())
    cn[cn == 0] = 1
    return T @ sp.diags(1 / cn), nagg


class Cheby:
    """Chebyshev smoother on D^-1 A for eigenvalues in [lmax/ratio, lmax]; x0 = 0 start variant."""
    def __init__(self, A, degree, ratio=8.0, lmax=None):
        self.A = A
        self.dinv = 1.0 / A.diagonal()
        if lmax is None:
            # power iteration estimate of rho(D^-1 A)
            rng = np.random.default_rng(1)
            v = rng.standard_normal(A.shape[0])
            for _ in range(15):
                v = self.dinv * (A @ v)
                lm = np.linalg.norm(v)
                v /= lm
            lmax = 1.1 * lm
        self.lmax, self.lmin, self.degree = lmax, lmax / ratio, degree

    def apply(self, b, x=None):
        A, dinv = self.A, self.dinv[:, None] if b.ndim == 2 else self.dinv
        theta, delta = 0.5 * (self.lmax + self.lmin), 0.5 * (self.lmax - self.lmin)
        sigma = theta / delta
        rho = 1.0 / sigma
        if x is None:
            x = np.zeros_like(b); r = b.copy()
        else:
            r = b - A @ x
        d = dinv * r / theta
        for k in range(self.degree):
            x = x + d
            if k == self.degree - 1:
                break
            r = r - A @ d
            rho_new = 1.0 / (2 * sigma - rho)
            d = rho_new * rho * d + (2 * rho_new / delta) * (dinv * r)
            rho = rho_new
        return x


class ThreeLevel:
    def __init__(self, A2, A1, P21, T, deg2=2, deg1=2, ratio=8.0, gamma=1):
        self.A2, self.A1, self.P, self.T = A2, A1, P21, T
        self.gamma = gamma
        self.S2 = Cheby(A2, deg2, ratio)
        self.S1 = Cheby(A1, deg1, ratio)
        A0 = (T.T @ A1 @ T).toarray()
        self.A0 = A0
        self.c0 = sla.cho_factor(A0 + 1e-14 * np.diag(np.diag(A0)))
        self.napply = 0

    def apply(self, R):
        self.napply += 1
        x2 = self.S2.apply(R)
        r1 = self.P.T @ (R - self.A2 @ x2)
        x1 = None
        for g in range(self.gamma):
            x1 = self.S1.apply(r1, x1)
            r0 = self.T.T @ (r1 - self.A1 @ x1)
            x0 = sla.cho_solve(self.c0, r0)
            x1 = x1 + self.T @ x0
            x1 = self.S1.apply(r1, x1)
        x2 = x2 + self.P @ x1
        x2 = self.S2.apply(R, x2)
        return x2


def svqb(V, MV, drop=1e-12):
    """M-orthonormalise V (SVQB, Stathopoulos-Wu): returns transform Q (k x k') with V@Q M-orthonormal."""
    G = V.T @ MV
    G = 0.5 * (G + G.T)
    d = np.sqrt(np.maximum(np.diag(G), 1e-300))
    Gs = G / d[:, None] / d[None, :]
    w, U = np.linalg.eigh(Gs)
    keep = w > drop * w.max()
    Q = (U[:, keep] / np.sqrt(w[keep])[None, :]) / d[:, None]
    return Q


def lobpcg(A, M, prec, X, nev, tol=1e-7, maxit=60, verbose=True):
    """Soft-locking LOBPCG on (A, M).  Basis S = [X, W, P] kept M-orthonormal; the new search
    directions P are formed in coefficient space (Hetmaniuk-Lehoucq 'ortho' variant)."""
    n, b = X.shape
    MX = M @ X
    Q = svqb(X, MX); X = X @ Q; MX = MX @ Q
    b = X.shape[1]
    AX = A @ X
    th, C = sla.eigh(X.T @ AX)
    X, AX, MX = X @ C, AX @ C, MX @ C
    P = AP = MP = None
    hist = []
    for it in range(maxit):
        R = AX - MX * th
        rn = np.linalg.norm(R, axis=0) / (np.abs(th) * np.linalg.norm(MX, axis=0))
        conv = rn < tol
        nconv = int(np.sum(conv[:nev]))
        hist.append((it, nconv, rn[:nev].max()))
        if verbose:
            print(f"  it {it:3d} conv {nconv:3d}/{nev} max_res {rn[:nev].max():.2e} act {int((~conv).sum())}")
        if nconv >= nev:
            break
        act = ~conv
        W = prec(R[:, act])
        for _ in range(2):
            W = W - X @ (MX.T @ W)
            if P is not None:
                W = W - P @ (MP.T @ W)
            MW = M @ W
            Q = svqb(W, MW); W = W @ Q; MW = MW @ Q
        AW = A @ W
        if P is not None:
            S, AS, MS = np.hstack([X, W, P]), np.hstack([AX, AW, AP]), np.hstack([MX, MW, MP])
        else:
            S, AS, MS = np.hstack([X, W]), np.hstack([AX, AW]), np.hstack([MX, MW])
        gA = S.T @ AS; gA = 0.5 * (gA + gA.T)
        gM = S.T @ MS; gM = 0.5 * (gM + gM.T)
        ev, C = sla.eigh(gA, gM)
        Cx = C[:, :b]
        th = ev[:b]
        # P coefficients: the [W,P] part of the active new Ritz vectors, gM-orthonormalised against Cx
        Cp = Cx[:, act].copy(); Cp[:b, :] = 0
        Cp = Cp - Cx @ (Cx.T @ (gM @ Cp))
        Gp = Cp.T @ gM @ Cp; Gp = 0.5 * (Gp + Gp.T)
        d = np.sqrt(np.maximum(np.diag(Gp), 1e-300))
        w, U = np.linalg.eigh(Gp / d[:, None] / d[None, :])
        keep = w > 1e-10 * w.max()
        Cp = Cp @ ((U[:, keep] / np.sqrt(w[keep])[None, :]) / d[:, None])
        P, AP, MP = S @ Cp, AS @ Cp, MS @ Cp
        X, AX, MX = S @ Cx, AS @ Cx, MS @ Cx
    return th, X, hist


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "cube"
    nn_ = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nev = int(sys.argv[3]) if len(sys.argv) > 3 else 45
    agg = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    deg2 = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    deg1 = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    if shape == "cube":
        pts, tets = fem.kuhn_box(nn_, nn_, nn_, 0.3, 0.3, 0.3); mat = (2700, 7.2e10, 0.19)
    elif shape == "plate":
        pts, tets = fem.kuhn_box(nn_, nn_, 1, 0.26, 0.26, 0.012); mat = (8000, 2.1e11, 0.28)
    elif shape == "plate2":
        pts, tets = fem.kuhn_box(nn_, nn_, 2, 0.26, 0.26, 0.012); mat = (8000, 2.1e11, 0.28)
    elif shape == "bar":
        pts, tets = fem.kuhn_box(20, 4, 4, 0.3, 0.05, 0.05); mat = (1000, 1e7, 0.0)
    t0 = time.time()
    K, M, nodes, nnod = fem.assemble_p2(pts, tets, *mat)
    print(f"tets {len(tets)} pts {len(pts)} nodes {nnod} dof {K.shape[0]} nnz {K.nnz} assemble {time.time()-t0:.1f}s")
    sigma = -(2 * np.pi * 20.0) ** 2
    A2 = (K - sigma * M).tocsr()
    K1, M1 = fem.assemble_p1(pts, tets, *mat)
    A1 = (K1 - sigma * M1).tocsr()
    P21 = fem.p2_to_p1_prolongation(len(pts), nodes, nnod)
    G = (P21.T @ A2 @ P21 - A1)
    print("galerkin check |P'A2P - A1|/|A1| =", abs(G).max() / abs(A1).max())
    T, nagg = fem_rbm = rbm_aggregates(pts, agg)
    print(f"aggregates {nagg} coarse dof {6*nagg}")
    t0 = time.time()
    ml = ThreeLevel(A2, A1, P21, T, deg2, deg1)
    print(f"setup {time.time()-t0:.1f}s  lmax2 {ml.S2.lmax:.3f} lmax1 {ml.S1.lmax:.3f}")
    b = nev + 10
    rng = np.random.default_rng(0)
    X0 = rng.standard_normal((K.shape[0], b))
    t0 = time.time()
    th, X, hist = lobpcg(A2, M, ml.apply, X0, nev, tol=1e-7, maxit=80)
    lam = th[:nev] + sigma
    print(f"lobpcg {time.time()-t0:.1f}s iters {len(hist)} prec applies {ml.napply}")
    if "--ref" in sys.argv:
        t0 = time.time()
        ref = spla.eigsh(K, k=nev, M=M, sigma=sigma, which="LM", tol=1e-10)[0]
        ref.sort()
        print(f"eigsh {time.time()-t0:.1f}s")
        rel = np.abs(lam - ref) / np.maximum(np.abs(ref), abs(sigma))
        print("max rel eigenvalue error vs eigsh:", rel.max())
    f = np.sqrt(np.maximum(lam, 0)) / (2 * np.pi)
    print("freqs:", np.round(f[:16], 2))




class TwoLevel:
    def __init__(self, A1, T, deg1=2, ratio=8.0):
        self.A1, self.T = A1, T
        self.S1 = Cheby(A1, deg1, ratio)
        A0 = (T.T @ A1 @ T).toarray()
        self.c0 = sla.cho_factor(A0 + 1e-14 * np.diag(np.diag(A0)))
        self.napply = 0
    def apply(self, r1):
        self.napply += 1
        x1 = self.S1.apply(r1)
        r0 = self.T.T @ (r1 - self.A1 @ x1)
        x1 = x1 + self.T @ sla.cho_solve(self.c0, r0)
        return self.S1.apply(r1, x1)


def sweep():
    import itertools
    shape, nn_, nev = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    if shape == "cube":
        pts, tets = fem.kuhn_box(nn_, nn_, nn_, 0.3, 0.3, 0.3); mat = (2700, 7.2e10, 0.19)
    else:
        pts, tets = fem.kuhn_box(nn_, nn_, 1, 0.26, 0.26, 0.012); mat = (8000, 2.1e11, 0.28)
    K, M, nodes, nnod = fem.assemble_p2(pts, tets, *mat)
    sigma = -(2 * np.pi * 20.0) ** 2
    A2 = (K - sigma * M).tocsr()
    K1, M1 = fem.assemble_p1(pts, tets, *mat)
    A1 = (K1 - sigma * M1).tocsr()
    P21 = fem.p2_to_p1_prolongation(len(pts), nodes, nnod)
    b = nev + 10
    for agg, deg2, deg1, ratio, nested, gamma in [(8, 2, 2, 8, 1, 1), (8, 2, 2, 8, 1, 2), (8, 2, 3, 8, 1, 2), (8, 2, 2, 8, 1, 3), (8, 1, 2, 8, 1, 2), (8, 2, 4, 8, 1, 2), (16, 2, 3, 8, 1, 3), (8, 2, 2, 8, 0, 2)]:
        T, nagg = rbm_aggregates(pts, agg)
        ml = ThreeLevel(A2, A1, P21, T, deg2, deg1, ratio, gamma)
        rng = np.random.default_rng(0)
        it1 = 0
        if nested:
            tl = TwoLevel(A1, T, deg1, ratio)
            X1 = rng.standard_normal((A1.shape[0], b))
            th1, X1, h1 = lobpcg(A1, M1, tl.apply, X1, nev, tol=1e-3, maxit=60, verbose=False)
            it1 = len(h1)
            X0 = P21 @ X1
        else:
            X0 = rng.standard_normal((K.shape[0], b))
        t0 = time.time()
        th, X, hist = lobpcg(A2, M, ml.apply, X0, nev, tol=1e-6, maxit=80, verbose=False)
        print(f"gamma {gamma} agg {agg} (coarse {6*nagg}) deg2 {deg2} deg1 {deg1} ratio {ratio} nested {nested}: P1 its {it1}, P2 its {len(hist)} ({time.time()-t0:.0f}s) first res {hist[0][2]:.2e}", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "sweep":
    sweep()

if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "sweep"):
    main()
