"""Design prototype (numpy/scipy): P2 tet FEM assembly used to validate the GPU eigensolver design.

Not product code and not the oracle: a throw-away numerical testbed kept as design evidence
(DESIGN.md cites the iteration counts measured with it).  Follows the algorithm described in
SURVEY.md section 8a rows A3-A6 (reference src/audio/mesh2modes.cpp:137-327).
"""
import math
import numpy as np
import scipy.sparse as sp

EDGE_CORNERS = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def kuhn_box(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0):
    vx, vy, vz = nx + 1, ny + 1, nz + 1
    i, j, k = np.meshgrid(np.arange(vx), np.arange(vy), np.arange(vz), indexing="ij")
    pts = np.stack([lx * i / nx, ly * j / ny, lz * k / nz], -1).reshape(-1, 3).astype(np.float64)
    def vid(i, j, k):
        return (i * vy + j) * vz + k
    ci, cj, ck = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    ci, cj, ck = ci.ravel(), cj.ravel(), ck.ravel()
    c = [vid(ci, cj, ck), vid(ci + 1, cj, ck), vid(ci, cj + 1, ck), vid(ci + 1, cj + 1, ck),
         vid(ci, cj, ck + 1), vid(ci + 1, cj, ck + 1), vid(ci, cj + 1, ck + 1), vid(ci + 1, cj + 1, ck + 1)]
    corners = [(0, 1, 3, 7), (0, 3, 2, 7), (0, 2, 6, 7), (0, 6, 4, 7), (0, 4, 5, 7), (0, 5, 1, 7)]
    tets = np.stack([np.stack([c[a] for a in t], -1) for t in corners], 1).reshape(-1, 4)
    return pts, tets.astype(np.uint32)


def quad_basis():
    """Mass[10][10], Grad[10][4][10][4]: exact unit-volume integrals via the factorial formula."""
    fact = [1, 1, 2, 6, 24, 120, 720, 5040]
    def unit_integral(p):
        return sum(c * 6 * fact[e[0]] * fact[e[1]] * fact[e[2]] * fact[e[3]] / fact[sum(e) + 3] for c, e in p)
    def mul(a, b):
        return [(ca * cb, tuple(x + y for x, y in zip(ea, eb))) for ca, ea in a for cb, eb in b]
    def unit(i):
        return tuple(1 if k == i else 0 for k in range(4))
    n = [None] * 10
    dn = [[[] for _ in range(4)] for _ in range(10)]
    for i in range(4):
        n[i] = [(2, tuple(2 * x for x in unit(i))), (-1, unit(i))]
        dn[i][i] = [(4, unit(i)), (-1, (0, 0, 0, 0))]
    for e, (i, j) in enumerate(EDGE_CORNERS):
        n[4 + e] = [(4, tuple(a + b for a, b in zip(unit(i), unit(j))))]
        dn[4 + e][i] = [(4, unit(j))]
        dn[4 + e][j] = [(4, unit(i))]
    mass = np.zeros((10, 10))
    grad = np.zeros((10, 4, 10, 4))
    for a in range(10):
        for c in range(10):
            mass[a, c] = unit_integral(mul(n[a], n[c]))
            for k in range(4):
                for l in range(4):
                    if dn[a][k] and dn[c][l]:
                        grad[a, k, c, l] = unit_integral(mul(dn[a][k], dn[c][l]))
    return mass, grad


def build_quad_mesh(npts, tets):
    """Midside ids in first-encounter order (reference mesh2modes.cpp:246-264)."""
    T = len(tets)
    a = tets[:, [e[0] for e in EDGE_CORNERS]].astype(np.int64)
    b = tets[:, [e[1] for e in EDGE_CORNERS]].astype(np.int64)
    key = (np.minimum(a, b) << 32) | np.maximum(a, b)
    flat = key.ravel()
    uniq, first, inv = np.unique(flat, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    mids = (npts + rank[inv]).reshape(T, 6)
    nodes = np.concatenate([tets.astype(np.int64), mids], 1)
    return nodes, npts + len(uniq)


def element_bases(pts, tets):
    p = pts[tets.astype(np.int64)]  # T,4,3
    m = np.concatenate([np.ones((len(tets), 4, 1)), p], -1)  # T,4,4 rows [1 x y z]
    minv = np.linalg.inv(m)  # columns = coefficients of lambda_i: lambda_i = minv[0,i] + minv[1:,i].x
    phig = np.transpose(minv[:, 1:, :], (0, 2, 1))  # T,4(i),3
    det = np.einsum("ti,ti->t", p[:, 3] - p[:, 0], np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]))
    vol = np.abs(det / 6)
    return vol, phig


def assemble_p2(pts, tets, rho, E, nu):
    lam = nu * E / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    nodes, nn = build_quad_mesh(len(pts), tets)
    mass, grad = quad_basis()
    vol, phig = element_bases(pts, tets)
    T = len(tets)
    # G[t,a,c,p,q] = sum_kl grad[a,k,c,l] phig[t,k,p] phig[t,l,q]
    G = np.einsum("akcl,tkp,tlq->tacpq", grad, phig, phig, optimize=True)
    tr = np.einsum("tacpp->tac", G)
    Ke = vol[:, None, None, None, None] * (lam * G + mu * np.transpose(G, (0, 1, 2, 4, 3))
                                            + mu * tr[..., None, None] * np.eye(3))
    rows = (3 * nodes[:, :, None, None, None] + np.arange(3)[None, None, None, :, None])
    cols = (3 * nodes[:, None, :, None, None] + np.arange(3)[None, None, None, None, :])
    rows = np.broadcast_to(rows, Ke.shape).ravel()
    cols = np.broadcast_to(cols, Ke.shape).ravel()
    n = 3 * nn
    K = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    Me = rho * vol[:, None, None] * mass[None]
    mr = np.broadcast_to(nodes[:, :, None], Me.shape).ravel()
    mc = np.broadcast_to(nodes[:, None, :], Me.shape).ravel()
    Mn = sp.coo_matrix((Me.ravel(), (mr, mc)), shape=(nn, nn)).tocsr()
    M = sp.kron(Mn, sp.identity(3), format="csr")
    return K, M, nodes, nn


def assemble_p1(pts, tets, rho, E, nu):
    lam = nu * E / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    vol, phig = element_bases(pts, tets)
    G = np.einsum("tkp,tlq->tklpq", phig, phig)
    tr = np.einsum("tklpp->tkl", G)
    Ke = vol[:, None, None, None, None] * (lam * G + mu * np.transpose(G, (0, 1, 2, 4, 3))
                                            + mu * tr[..., None, None] * np.eye(3))
    nodes = tets.astype(np.int64)
    rows = np.broadcast_to(3 * nodes[:, :, None, None, None] + np.arange(3)[None, None, None, :, None], Ke.shape).ravel()
    cols = np.broadcast_to(3 * nodes[:, None, :, None, None] + np.arange(3)[None, None, None, None, :], Ke.shape).ravel()
    n = 3 * len(pts)
    K = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    m1 = (np.ones((4, 4)) + np.eye(4)) / 20.0
    Me = rho * vol[:, None, None] * m1[None]
    Mn = sp.coo_matrix((Me.ravel(), (np.broadcast_to(nodes[:, :, None], Me.shape).ravel(),
                                     np.broadcast_to(nodes[:, None, :], Me.shape).ravel())), shape=(len(pts),) * 2).tocsr()
    return K, sp.kron(Mn, sp.identity(3), format="csr")


def p2_to_p1_prolongation(npts, nodes, nn):
    """P (3nn x 3npts): identity on corners, 1/2-1/2 on midside nodes."""
    rows, cols, vals = [np.arange(npts)], [np.arange(npts)], [np.ones(npts)]
    for e, (i, j) in enumerate(EDGE_CORNERS):
        mid = nodes[:, 4 + e]
        for c in (i, j):
            rows.append(mid); cols.append(nodes[:, c]); vals.append(np.full(len(mid), 0.5))
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    P = sp.coo_matrix((v, (r, c)), shape=(nn, npts)).tocsr()
    P.data[:] = np.where(P.data >= 1.0, 1.0, 0.5)  # duplicates summed -> reset
    P = P.tocsr()
    # duplicates: the same (mid, corner) pair appears once per incident element; rebuild with unique pairs
    P = sp.csr_matrix((np.ones_like(P.data), P.indices, P.indptr), shape=P.shape)
    d = np.array(P.sum(1)).ravel()
    P = sp.diags(1.0 / d) @ P
    return sp.kron(P, sp.identity(3), format="csr")


def node_coords(pts, nodes, nn):
    xyz = np.zeros((nn, 3))
    xyz[: len(pts)] = pts
    for e, (i, j) in enumerate(EDGE_CORNERS):
        xyz[nodes[:, 4 + e]] = 0.5 * (pts[nodes[:, i]] + pts[nodes[:, j]])
    return xyz
