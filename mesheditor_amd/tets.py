"""Python binding of the tet-generation front end (modal/tets.hpp: tetra::Tetrahedralize, host C++ in libmodalhost.so)."""
import ctypes as C

import numpy as np


def _lib():
    from . import bank
    L = bank.lib()
    if not getattr(L, "_tets_bound", False):
        vp, u32 = C.c_void_p, C.c_uint32
        L.mhx_tetrahedralize.restype, L.mhx_tetrahedralize.argtypes = vp, [vp, u32, vp, u32, C.c_uint64, C.c_int]
        L.mhx_tetrahedralize2.restype, L.mhx_tetrahedralize2.argtypes = vp, [vp, u32, vp, u32, C.c_uint64, C.c_int, C.c_double]
        L.mhx_tets_error.restype, L.mhx_tets_error.argtypes = C.c_char_p, [vp]
        for name in ("mhx_tets_num_points", "mhx_tets_num_tets", "mhx_tets_boundary_steiner"):
            getattr(L, name).restype, getattr(L, name).argtypes = u32, [vp]
        L.mhx_tets_copy.restype, L.mhx_tets_copy.argtypes = None, [vp, vp, vp]
        L.mhx_tets_free.restype, L.mhx_tets_free.argtypes = None, [vp]
        L.mhx_simplify_surface.restype, L.mhx_simplify_surface.argtypes = None, [vp, C.POINTER(u32), vp, C.POINTER(u32), C.c_float]
        L._tets_bound = True
    return L


def tetrahedralize(points, triangles, max_steiner=0, interior_steiner=True, repair_slivers=True, interior_shell="when_flat", quality=False, max_volume=0.0, break_flat_cells=True):
    """(points float64 [V', 3], tets uint32 [T, 4], boundary_steiner_count): input vertex i keeps index i, added points follow.
    interior_steiner (tetra::Options::InteriorSteiner): the recovery's points are moved off the surface afterwards, so that every
    input triangle is a boundary face (the count returned is what had to stay on it; 0 = the reference's contract holds).
    repair_slivers (tetra::Options::RepairSlivers): connectivity-only sliver repair afterwards (edge removal, 2-3 flips), as the
    reference's tetrahedraliser always does.
    interior_shell (tetra::Options::InteriorShell): "when_flat" (default: a point under every surface vertex only if the fill is left
    with flat cells at the surface), "never", "always".
    quality / max_volume: the reference's tetra::Options::Quality / MaxVolume (src/mesh/Tetrahedralize.h:17-27): interior points until the
    radius-edge ratio is at most 2 where the fixed surface allows / until no tetrahedron is larger than max_volume.
    break_flat_cells (tetra::Options::BreakFlatCells): the always-on last pass that leaves no cell below a shape measure of 1e-2 where an
    interior point beside it helps; False leaves the flat cells of a fine UV sphere's fill in (tests of the solver on such a mesh).
    Raises RuntimeError with the tetrahedraliser's message for open / self-intersecting / unrecoverable surfaces."""
    L = _lib()
    pts = np.ascontiguousarray(points, dtype=np.float64)
    tri = np.ascontiguousarray(triangles, dtype=np.uint32)
    h = L.mhx_tetrahedralize2(pts.ctypes.data_as(C.c_void_p), len(pts), tri.ctypes.data_as(C.c_void_p), len(tri), int(max_steiner),
                              int(bool(interior_steiner)) | (2 if repair_slivers else 0) | {"when_flat": 0, "never": 4, "always": 8}[interior_shell] | (16 if quality else 0) | (0 if break_flat_cells else 32),
                              float(max_volume))
    try:
        err = L.mhx_tets_error(h).decode()
        if err:
            raise RuntimeError("Tetrahedralize: " + err)
        out_p = np.zeros((L.mhx_tets_num_points(h), 3), np.float64)
        out_t = np.zeros((L.mhx_tets_num_tets(h), 4), np.uint32)
        L.mhx_tets_copy(h, out_p.ctypes.data_as(C.c_void_p), out_t.ctypes.data_as(C.c_void_p))
        return out_p, out_t, int(L.mhx_tets_boundary_steiner(h))
    finally:
        L.mhx_tets_free(h)


def simplify_surface(positions, triangles, ratio):
    """SimplifySurface of the reference (src/mesh/Tets.h:8): quadric edge-collapse to `ratio` of the triangles; returns the
    coarsened (positions float32 [V', 3], triangles uint32 [F', 3])."""
    L = _lib()
    pos = np.array(positions, dtype=np.float32, order="C")
    tri = np.array(triangles, dtype=np.uint32, order="C")
    nv, nt = C.c_uint32(len(pos)), C.c_uint32(len(tri))
    L.mhx_simplify_surface(pos.ctypes.data_as(C.c_void_p), C.byref(nv), tri.ctypes.data_as(C.c_void_p), C.byref(nt), float(ratio))
    return pos[: nv.value].copy(), tri[: nt.value].copy()
