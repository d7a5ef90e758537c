"""Shared test helpers: the mode-family classifier of the reference's bar tests, restated.

Follows tests/ModalSolverTest.cpp:83-114 (Classify) and :130-138 (BendingTheory) of the reference.
"""
import numpy as np

BENDING_BL = (4.73004074, 7.85320462, 10.9956078)


def classify(positions, shapes, mode, length, width, thickness, nx):
    """'longitudinal' | 'torsional' | 'bending' | 'bending_y' | 'bending_z' | 'other' from shape energy fractions."""
    u = shapes[:, mode, :].astype(np.float64)
    p = positions.astype(np.float64)
    ry, rz = p[:, 1] - width / 2, p[:, 2] - thickness / 2
    axial, lat_y, lat_z = (u[:, 0] ** 2).sum(), (u[:, 1] ** 2).sum(), (u[:, 2] ** 2).sum()
    total = (u ** 2).sum()
    if total <= 0:
        return "other"
    sl = np.rint(p[:, 0] * nx / length).astype(int)
    circ = np.bincount(sl - sl.min(), weights=ry * u[:, 2] - rz * u[:, 1])
    r2 = np.bincount(sl - sl.min(), weights=ry * ry + rz * rz)
    rotation = float((circ[r2 > 0] ** 2 / r2[r2 > 0]).sum())
    if axial / total > 0.85:
        return "longitudinal"
    if rotation / total > 0.85:
        return "torsional"
    lateral = lat_y + lat_z
    if lateral / total > 0.6 and rotation / total < 0.5:
        if lat_y / lateral > 0.8:
            return "bending_y"
        if lat_z / lateral > 0.8:
            return "bending_z"
        return "bending"
    return "other"


def families(freqs, positions, shapes, length, width, thickness, nx):
    fam = {}
    for k in range(len(freqs)):
        fam.setdefault(classify(positions, shapes, k, length, width, thickness, nx), []).append(float(freqs[k]))
    return fam


def bending_theory(E, rho, length, thickness, per_root):
    rg = thickness / np.sqrt(12.0)
    base = np.sqrt(E / rho) * rg / (2 * np.pi * length * length)
    return [bl * bl * base for bl in BENDING_BL for _ in range(per_root)]


def check_family(fem, theory, tol, min_count=2):
    n = min(len(fem), len(theory))
    assert n >= min_count, (fem, theory)
    for a, b in zip(fem[:n], theory[:n]):
        assert abs(a / b - 1.0) < tol, (fem[:n], theory[:n])


def subspace_angle_sin(A, B, M=None):
    """Largest principal-angle sine between span(A) and span(B) (columns), optionally in the M inner product."""
    import scipy.linalg as sla
    if M is not None:
        # M-orthonormalise via Cholesky of the Gram matrices
        def orth(X):
            G = X.T @ (M @ X)
            L = np.linalg.cholesky(0.5 * (G + G.T))
            return sla.solve_triangular(L, X.T, lower=True).T
        QA, QB = orth(A), orth(B)
        C = QA.T @ (M @ QB)
    else:
        QA, _ = np.linalg.qr(A)
        QB, _ = np.linalg.qr(B)
        C = QA.T @ QB
    s = np.linalg.svd(C, compute_uv=False)
    return float(np.sqrt(max(0.0, 1.0 - min(1.0, s.min()) ** 2)))
