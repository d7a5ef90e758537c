"""Synthetic tetrahedral meshes for tests and benches.

The reference has no tet generator that can run here (its tetrahedraliser is out of scope), and the RealImpact
scans are absent, so workloads are structured Kuhn meshes: every grid cell split into six tetrahedra around its
main diagonal -- the same construction as the reference's test helper MakeBarTets
(tests/ModalSolverTest.cpp:38-70), restated with numpy.
"""
import numpy as np

# Acoustic material table: src/audio/AcousticMaterial.h:31-38 (density, Young, Poisson, alpha, beta)
MATERIALS = {
    "Ceramic": (2700.0, 7.2e10, 0.19, 6.0, 1e-7),
    "Glass": (2600.0, 6.2e10, 0.20, 1.0, 1e-7),
    "Wood": (750.0, 1.1e10, 0.25, 60.0, 2e-6),
    "Plastic": (1070.0, 1.4e9, 0.35, 30.0, 1e-6),
    "Iron": (8000.0, 2.1e11, 0.28, 5.0, 1e-7),
    "Polycarbonate": (1190.0, 2.4e9, 0.37, 0.5, 4e-7),
    "Steel": (7850.0, 2.0e11, 0.29, 5.0, 3e-8),
}
MATERIAL_ORDER = ["Ceramic", "Glass", "Wood", "Plastic", "Iron", "Polycarbonate", "Steel"]


def kuhn_box(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0, origin=(0.0, 0.0, 0.0)):
    """(points float64 [V,3], tets uint32 [T,4]); vertex id = (i*(ny+1) + j)*(nz+1) + k."""
    vx, vy, vz = nx + 1, ny + 1, nz + 1
    i, j, k = np.meshgrid(np.arange(vx), np.arange(vy), np.arange(vz), indexing="ij")
    pts = np.stack([lx * i / nx + origin[0], ly * j / ny + origin[1], lz * k / nz + origin[2]], -1).reshape(-1, 3).astype(np.float64)

    def vid(a, b, c):
        return (a * vy + b) * vz + c

    ci, cj, ck = (a.ravel() for a in np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"))
    c = [vid(ci, cj, ck), vid(ci + 1, cj, ck), vid(ci, cj + 1, ck), vid(ci + 1, cj + 1, ck),
         vid(ci, cj, ck + 1), vid(ci + 1, cj, ck + 1), vid(ci, cj + 1, ck + 1), vid(ci + 1, cj + 1, ck + 1)]
    corners = [(0, 1, 3, 7), (0, 3, 2, 7), (0, 2, 6, 7), (0, 6, 4, 7), (0, 4, 5, 7), (0, 5, 1, 7)]
    tets = np.stack([np.stack([c[a] for a in t], -1) for t in corners], 1).reshape(-1, 4)
    return pts, np.ascontiguousarray(tets, dtype=np.uint32)


def ball(n, radius):
    """Kuhn cube [-1,1]^3 pushed through the radial cube->ball map p <- p*|p|_inf/|p|_2*R."""
    pts, tets = kuhn_box(n, n, n, 2.0, 2.0, 2.0, origin=(-1.0, -1.0, -1.0))
    linf = np.abs(pts).max(1)
    l2 = np.linalg.norm(pts, axis=1)
    scale = np.where(l2 > 0, linf / np.where(l2 > 0, l2, 1.0), 0.0) * radius
    return pts * scale[:, None], tets


def jittered_box(n, seed, base=0.2):
    """One mesh of the 64-mesh batch: extents scaled by U(0.8,1.25) per axis, interior
    points displaced by U(-0.15h, 0.15h)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    ext = base * rng.uniform(0.8, 1.25, 3)
    pts, tets = kuhn_box(n, n, n, *ext)
    h = ext / n
    grid = np.stack(np.meshgrid(np.arange(n + 1), np.arange(n + 1), np.arange(n + 1), indexing="ij"), -1).reshape(-1, 3)
    interior = np.all((grid > 0) & (grid < n), axis=1)
    disp = rng.uniform(-0.15, 0.15, pts.shape) * h
    pts[interior] += disp[interior]
    return pts, tets


def workload(name):
    """Named workloads: returns (points, tets, material tuple, solver kwargs)."""
    if name == "bar_square":  # tests/ModalSolverTest.cpp:228-245
        p, t = kuhn_box(20, 4, 4, 0.3, 0.05, 0.05)
        return p, t, (1000.0, 1e7, 0.0, 0.0, 0.0), {}
    if name == "bar_thin":  # tests/ModalSolverTest.cpp:249-261
        p, t = kuhn_box(30, 5, 1, 0.3, 0.05, 0.01)
        return p, t, (1000.0, 1e9, 0.0, 0.0, 0.0), {}
    if name == "cube_small":
        p, t = kuhn_box(4, 4, 4, 0.1, 0.1, 0.1)
        return p, t, MATERIALS["Ceramic"], {}
    if name == "ball_s10k":  # BASELINE.json configs[1]
        p, t = ball(12, 0.15)
        return p, t, MATERIALS["Ceramic"], {"num_modes": 50, "num_fem_modes": 65}
    if name == "cube_s10k":
        p, t = kuhn_box(12, 12, 12, 0.3, 0.3, 0.3)
        return p, t, MATERIALS["Ceramic"], {"num_modes": 50, "num_fem_modes": 65}
    if name == "cube_s30k":
        p, t = kuhn_box(17, 17, 17, 0.3, 0.3, 0.3)
        return p, t, MATERIALS["Iron"], {"num_modes": 50, "num_fem_modes": 65}
    if name == "cube_s100k":  # the metric's "100k-tet mesh", 50 modes
        p, t = kuhn_box(26, 26, 26, 0.3, 0.3, 0.3)
        return p, t, MATERIALS["Iron"], {"num_modes": 50, "num_fem_modes": 65}
    if name == "skillet_s100k":  # BASELINE.json configs[2]: thin iron disc-like plate, 200 modes
        p, t = kuhn_box(93, 93, 2, 0.26, 0.26, 0.012)
        return p, t, MATERIALS["Iron"], {"num_modes": 200, "num_fem_modes": 215}
    raise KeyError(name)
