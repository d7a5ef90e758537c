"""CPU-side checks of the boundary: libmodalhip.so / libmodalhost.so load and export every symbol the headers declare
(no compute calls without a GPU), the host-side scalar stages agree with the oracle, and the device entry points fail
loudly when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mesheditor_amd import meshes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core():
    from mesheditor_amd import _lib
    _lib.build()
    return _lib


def test_library_exports_every_declared_symbol(core):
    L = core.lib()
    header = open(os.path.join(ROOT, "include", "modalhip.h")).read()
    declared = sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 30
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    # the binding declares a signature for every exported function it uses
    assert set(L._declared) <= set(declared)
    # the boundary holds path entry points only: measurement and experiment entries live in libmodalhip_lab.so
    assert not [n for n in declared if "bench" in n or "tridiagonalize" in n or "elementwise" in n]


def test_the_bindings_struct_images_match_the_library(core):
    """The ctypes images of the boundary's structs have the sizes the library was built with (mh_abi_struct_sizes; no GPU needed): a field
    added on one side only -- round 5 added two to mh_profile -- would otherwise read garbage silently."""
    L = core.lib()
    sizes = (C.c_uint32 * 4)()
    L.mh_abi_struct_sizes(sizes)
    assert list(sizes) == [C.sizeof(core.Profile), C.sizeof(core.SolverConfig), C.sizeof(core.Material), C.sizeof(core.MassProps)], list(sizes)


def test_lab_library_is_separate_from_the_product(core):
    """libmodalhip_lab.so (timing loops, kernel variants called directly, the matrix-free operator) exports what its own header
    declares; the product library exports none of it."""
    from tools import lab
    L = lab.lib()
    header = open(os.path.join(ROOT, "mesheditor_amd", "csrc", "lab", "modalhip_lab.h")).read()
    declared = sorted(set(re.findall(r"\b(mhl_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) == 17 and all(hasattr(L, n) for n in declared)
    P = core.lib()
    assert not [n for n in declared if hasattr(P, n)] and not hasattr(P, "mh_system_bench_spmm")


def test_no_gpu_fails_loudly(core):
    L = core.lib()
    h = C.c_void_p()
    rc = L.mh_context_create(0, C.byref(h))
    if rc == 0:
        L.mh_context_destroy(h)
        pytest.skip("a GPU is present")
    assert rc == core.MH_EHIP and not h.value
    from mesheditor_amd import api
    with pytest.raises(api.ModalHipError):
        api.Context(0)


def test_host_stages_match_oracle(core, oracle):
    from mesheditor_amd import api
    pts, tets = meshes.kuhn_box(5, 3, 2, 0.5, 0.3, 0.2, origin=(-0.25, -0.15, -0.1))
    for scale, si in (((1, 1, 1), 1.0), ((2, 2, 2), 2.0), ((1.5, 1.0, 0.5), 1.0)):
        mg = api.mass_properties(pts, tets, 2700.0, scale, si)
        mo = oracle.mass_properties(pts, tets, 2700.0, scale, si)
        assert mg[0] == mo[0]
        assert np.array_equal(mg[1], mo[1])
        assert np.allclose(mg[2], mo[2], rtol=1e-6)
        # principal frames agree up to axis sign (eigenvector signs are arbitrary in the reference too)
        def rot(q):
            w, x, y, z = q
            return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                             [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                             [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(np.abs(rot(mg[3]).T @ rot(mo[3])), np.eye(3), atol=1e-4)
    mat_g, mat_o = api.material(*meshes.MATERIALS["Glass"]), oracle.material(*meshes.MATERIALS["Glass"])
    rng = np.random.default_rng(2)
    lam = np.sort(np.concatenate([rng.uniform(-1e-7, 1e-6, 6), (2 * np.pi * rng.uniform(5, 30000, 40)) ** 2]))
    shapes = rng.standard_normal((5, len(lam), 3)).astype(np.float32)
    for kw in ({}, {"num_modes": 7}, {"fundamental_freq": 440.0}, {"min_mode_freq": 100.0, "max_mode_freq": 5000.0}):
        a = api.postprocess_modes(lam, shapes, 0.5, mat_g, api.default_config(**kw))
        b = oracle.postprocess_modes(lam, shapes, 0.5, mat_o, oracle.default_config(**kw))
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y)
        assert a[3] == b[3]
    edited = meshes.MATERIALS["Glass"]
    e_g, e_o = api.material(edited[0] * 1.3, edited[1] * 0.7, *edited[2:]), oracle.material(edited[0] * 1.3, edited[1] * 0.7, *edited[2:])
    a = api.rescale_modes(lam, shapes, mat_g, e_g, api.default_config())
    b = oracle.rescale_modes(lam, shapes, mat_o, e_o, oracle.default_config())
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert api.rescale_modes(lam, shapes, mat_g, api.material(2600, 6.2e10, 0.21), api.default_config()) is None
    # nothing in band -> empty result
    assert len(api.postprocess_modes(lam[:6], shapes[:, :6], 1.0, mat_g, api.default_config())[0]) == 0


def test_host_mirror_contact_model(oracle):
    """libmodalhost.so (C++ mirror of src/audio/ContactModel.h) against the oracle and the reference's known answers."""
    from mesheditor_amd import bank as hipbank
    hipbank_path = hipbank.SO_PATH
    if not os.path.exists(hipbank_path):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mesheditor_amd", "cpp")])
    H, O = hipbank.lib(), oracle.lib()
    polymer = np.array([1000.0, 1e9, 0.3, 0.0, 0.0])
    null = np.array([1e6, 1e30, 0.0, 0.0, 0.0])
    inv = np.eye(3, dtype=np.float32).reshape(-1)
    arm, direction = np.zeros(3, np.float32), np.array([0, 0, 1], np.float32)
    smass = H.mhx_striker_mass(1e6, 1e6, 1e6)
    assert smass == O.mo_striker_mass(1e6, 1e6, 1e6)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for curv, area, speed, scale in ((100.0, 0.0, 1.0, 1.0), (0.0, 1e-4, 1.0, 1.0), (10.0, 1e-5, 3.0, 2.0), (100.0, 0.0, 32.0, 1e-6)):
        a = H.mhx_estimate_contact_time(1.0, p(inv), p(arm), p(direction), speed, p(polymer), curv, area, p(null), 1e-6, 1.0 / smass, scale, 0.0)
        b = O.mo_estimate_contact_time(1.0, p(inv), p(arm), p(direction), speed, oracle.material(*polymer), curv, area, oracle.material(*null), 1e-6, 1.0 / smass, scale, 0.0)
        assert abs(a - b) <= 1e-12 * abs(b)  # a scalar model value, not a recurrence: the mirror's own arrangement of the same formulas
    tau = H.mhx_estimate_contact_time(1.0, p(inv), p(arm), p(direction), 1.0, p(polymer), 100.0, 0.0, p(null), 1e-6, 1.0 / smass, 1.0, 0.0)
    assert abs(tau - 1.744e-3) < 2e-2 * 1.744e-3  # tests/ContactModelTest.cpp:55-60
    assert abs(H.mhx_saturation_penetration(10.0, 1e-5) - 3.183e-5) < 1e-3 * 3.183e-5
    q = np.array([0.3, 0.1, -0.5, 0.8], np.float32)
    q /= np.linalg.norm(q)
    diag = np.array([2.0, 5.0, 9.0], np.float32)
    a9, b9 = np.zeros(9, np.float32), np.zeros(9, np.float32)
    H.mhx_inverse_inertia_tensor(p(diag), p(q), p(a9))
    O.mo_inverse_inertia_tensor(p(diag), p(q), p(b9))
    assert np.allclose(a9, b9, rtol=1e-6, atol=1e-7)
    fa, fb = np.zeros(6, np.float32), np.zeros(6, np.float32)
    H.mhx_recoil_object_filter(0.05, 5e-4, 48000.0, p(fa))
    O.mo_recoil_object_filter(0.05, 5e-4, 48000.0, p(fb))
    assert np.array_equal(fa, fb) and fa[0] != 0


def _p1_graph(tets, n_points):
    import scipy.sparse as sp
    t = tets.astype(np.int64)
    i = np.repeat(t, 4, axis=1).ravel()
    j = np.tile(t, (1, 4)).ravel()
    g = sp.coo_matrix((np.ones(len(i)), (i, j)), shape=(n_points, n_points)).tocsr()
    g.sort_indices()
    return g


@pytest.mark.parametrize("name", ["cube_s10k", "scan_s30k"])
def test_graph_aggregates_are_connected_sets(core, name):
    """The rigid-body level's aggregates (host code in libmodalhip, reached through the lab library): every P1 node in exactly one
    aggregate, every aggregate a CONNECTED set of the P1 graph with at least four nodes, the coarse order within its cap, the
    result deterministic -- on a Kuhn grid and on the scan-like mesh whose Morton-run aggregates were the round-2 solver's undoing."""
    import scipy.sparse.csgraph as cg
    from tools import lab
    pts, tets, _, _ = meshes.workload(name)
    g = _p1_graph(tets, len(pts))
    agg, na = lab.graph_aggregates(g.indptr, g.indices)
    agg2, na2 = lab.graph_aggregates(g.indptr, g.indices)
    assert na == na2 and np.array_equal(agg, agg2)
    assert agg.max() == na - 1 and 6 * na <= 6144
    sizes = np.bincount(agg, minlength=na)
    assert sizes.min() >= 4 and 8 <= sizes.mean() <= 64
    same = agg[g.nonzero()[0]] == agg[g.nonzero()[1]]
    inside = g.tocoo()
    import scipy.sparse as sp
    kept = sp.coo_matrix((np.ones(same.sum()), (inside.row[same], inside.col[same])), shape=g.shape)
    ncomp, label = cg.connected_components(kept, directed=False)
    assert ncomp == na, (ncomp, na)  # within-aggregate edges alone connect each aggregate
    # a cap forces merging: still connected, fewer aggregates
    agg3, na3 = lab.graph_aggregates(g.indptr, g.indices, target=16, max_order=6 * (na // 3))
    assert na3 <= na // 3 and len(np.unique(agg3)) == na3
    same3 = agg3[inside.row] == agg3[inside.col]
    ncomp3, _ = cg.connected_components(sp.coo_matrix((np.ones(same3.sum()), (inside.row[same3], inside.col[same3])), shape=g.shape), directed=False)
    assert ncomp3 == na3


def test_graph_aggregates_leave_no_tiny_aggregate_on_chains(core):
    """a one- or two-node aggregate has linearly dependent rigid-body columns (a singular coarse operator).  On graphs
    whose first pass founds only tiny aggregates -- a path (three nodes each), a comb, the sample UV sphere's P1 graph -- the merge
    pass must run until every aggregate that has a neighbour holds at least four nodes, whatever the chain of merges looks like."""
    import scipy.sparse as sp
    from tools import lab

    def graph(edges, n):
        r, c = np.array(edges).T
        g = sp.coo_matrix((np.ones(2 * len(r) + n), (np.r_[r, c, np.arange(n)], np.r_[c, r, np.arange(n)])), shape=(n, n)).tocsr()
        g.sum_duplicates()
        g.sort_indices()
        return g
    path = graph([(i, i + 1) for i in range(200)], 201)
    comb = graph([(i, i + 1) for i in range(0, 120, 2)] + [(i, i + 2) for i in range(0, 119, 2)], 121)  # a spine of even nodes, a tooth on each
    full = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gltf_modal_models_full.npz"))
    tri = full["Pile.gltf|Marble|indices"].astype(np.int64)
    sphere = graph([(t[i], t[(i + 1) % 3]) for t in tri for i in range(3)], int(tri.max()) + 1)
    for name, g in (("path", path), ("comb", comb), ("sphere surface", sphere)):
        agg, na = lab.graph_aggregates(g.indptr.astype(np.uint32), g.indices.astype(np.uint32))
        sizes = np.bincount(agg, minlength=na)
        assert sizes.min() >= 4, (name, sizes.min(), na)
        assert len(np.unique(agg)) == na


def test_tet_front_end_keeps_the_input_triangulation_as_the_boundary():
    """SURVEY 8f row N3 through the Python binding (host C++, no GPU): the scan-like skillet surface filled with the tetrahedraliser's
    default options -- the boundary recovery's points moved inside afterwards -- has exactly the input triangles as boundary faces and
    no added point on one (reference contract, src/mesh/Tetrahedralize.h:59); with interior_steiner=False the same surface comes
    back refined (points left on it), which is what the count reports."""
    from mesheditor_amd import meshes, tets
    v, f = meshes.skillet_scan_surface(0.02, 0.02)
    for interior in (True, False):
        p, t, on_surface = tets.tetrahedralize(v, f, interior_steiner=interior)
        assert np.array_equal(p[: len(v)], v)
        a, b, c, d = (p[t[:, i]] for i in range(4))
        assert (np.einsum("ij,ij->i", np.cross(b - a, c - a), d - a) > 0).all()
        faces = np.sort(np.concatenate([t[:, [1, 2, 3]], t[:, [0, 2, 3]], t[:, [0, 1, 3]], t[:, [0, 1, 2]]]), axis=1)
        uniq, counts = np.unique(faces, axis=0, return_counts=True)
        assert counts.max() <= 2
        boundary = {tuple(r) for r in uniq[counts == 1]}
        given = {tuple(sorted(r)) for r in f.tolist()}
        if interior:
            assert on_surface == 0 and len(p) > len(v)  # points were added, none is left on the surface
            assert boundary == given
        else:
            assert on_surface == len(p) - len(v) > 0
            assert boundary != given and all(max(face) < len(v) for face in boundary & given)


def test_tet_front_end_quality_and_max_volume_options():
    """The reference's tetra::Options::Quality / MaxVolume (src/mesh/Tetrahedralize.h:17-27) through the Python binding: a UV sphere, whose
    default fill keeps flat cells between coplanar ring quads, comes back without any and with the surface untouched; MaxVolume bounds
    every tetrahedron; every added point is strictly inside (the boundary is exactly the input triangulation)."""
    from mesheditor_amd import meshes, tets
    v, f = meshes.uv_sphere_surface(0.15, 32, 16)

    def shapes(p, t):
        q = p[t.astype(np.int64)]
        vol6 = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0])
        e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
        return vol6, vol6 * np.sqrt(2) / e2 ** 1.5

    def boundary_is_the_input(p, t):
        faces = np.sort(np.concatenate([t[:, [1, 2, 3]], t[:, [0, 2, 3]], t[:, [0, 1, 3]], t[:, [0, 1, 2]]]), axis=1)
        uniq, counts = np.unique(faces, axis=0, return_counts=True)
        return counts.max() <= 2 and {tuple(r) for r in uniq[counts == 1]} == {tuple(sorted(r)) for r in f.tolist()}

    p0, t0, _ = tets.tetrahedralize(v, f, interior_shell="never", break_flat_cells=False)
    p1, t1, left1 = tets.tetrahedralize(v, f, quality=True)
    vol0, q0 = shapes(p0, t0)
    vol1, q1 = shapes(p1, t1)
    assert left1 == 0 and np.array_equal(p1[: len(v)], v) and len(p1) > len(v)
    assert vol1.min() > 0 and boundary_is_the_input(p1, t1)
    assert (q0 < 1e-3).sum() > 0 and (q1 < 1e-3).sum() == 0, ((q0 < 1e-3).sum(), (q1 < 1e-3).sum())
    # round 6: the flat-cell pass is always on (tetra::Options::BreakFlatCells) -- the same fill without the quality arm and without
    # the shell has no cell below 1e-3 either, with interior points only
    p2, t2, left2 = tets.tetrahedralize(v, f, interior_shell="never")
    vol2, q2 = shapes(p2, t2)
    assert left2 == 0 and np.array_equal(p2[: len(v)], v) and len(p0) < len(p2) < len(p1)
    assert vol2.min() > 0 and boundary_is_the_input(p2, t2) and (q2 < 1e-3).sum() == 0, q2.min()
    assert abs(vol1.sum() - vol0.sum()) < 1e-12 * vol0.sum()
    bound = vol0.sum() / 6 / 4000
    p2, t2, left2 = tets.tetrahedralize(v, f, max_volume=bound)
    vol2, _ = shapes(p2, t2)
    assert left2 == 0 and vol2.min() > 0 and vol2.max() / 6 <= bound * (1 + 1e-12) and len(t2) >= 4000
    assert boundary_is_the_input(p2, t2)


@pytest.mark.parametrize("h,thickness,ratio", [(0.011, 0.015, 0.25), (0.011, 0.015, 0.1), (0.006, 0.008, 0.25)])
def test_decimated_scan_surfaces_fill_with_their_triangulation_as_the_boundary(h, thickness, ratio):
    """the reference pipeline's actual input is a quadric-decimated scan (src/mesh/Tets.h:8-10 -> GenerateTets).
    Coarse triangles on a thin wall make the conforming Delaunay refinement run away; the constrained recovery (round 4) fills them:
    every input triangle a boundary face (src/mesh/Tetrahedralize.h:59), no added point left on the surface, every tet positive,
    the volume the surface encloses."""
    from mesheditor_amd import tets
    v, f = meshes.skillet_scan_surface(h, thickness)
    v2, f2 = tets.simplify_surface(v, f, ratio)
    assert len(f2) <= ratio * len(f) * 1.05 + 8
    p, t, left_on_surface = tets.tetrahedralize(v2.astype(np.float64), f2)
    assert left_on_surface == 0 and np.array_equal(p[: len(v2)], v2.astype(np.float64))
    assert len(p) - len(v2) <= 64  # a handful of recovery points (<= 16) and, since round 6, the interior points the flat-cell pass puts beside cells below 1e-2 (27 on the 100k one); all inside
    t64 = t.astype(np.int64)
    vol6 = np.einsum("ij,ij->i", np.cross(p[t64[:, 1]] - p[t64[:, 0]], p[t64[:, 2]] - p[t64[:, 0]]), p[t64[:, 3]] - p[t64[:, 0]])
    assert vol6.min() > 0
    count = {}
    for tet in t64:
        for i in range(4):
            key = tuple(sorted(int(tet[j]) for j in range(4) if j != i))
            count[key] = count.get(key, 0) + 1
    assert max(count.values()) == 2
    assert {k for k, c in count.items() if c == 1} == {tuple(sorted(map(int, tri))) for tri in f2}
    # positive tets whose faces pair up and whose free faces are exactly the surface tile the enclosed region; its volume stays within
    # the decimation's few per cent of the undecimated solid's (marching-tetrahedra surface filled in round 3: scan_s30k / scan_s100k)
    full = {0.011: 0.00115437750498, 0.006: 0.000646398052489}[h]
    assert abs(vol6.sum() / 6 - full) < 0.03 * full


def test_tet_front_end_never_hands_on_an_inverted_tetrahedron():
    """Found by the round-6 soak (tools/probe/r06_soak.py): a 13 mm scan wall at a 13 mm lattice makes the conforming recovery add 2 080 points on the
    surface; exact while it runs, ROUNDED on the way out, they left 15 tetrahedra flat or turned over -- and the sliver repair behind them crashed
    (a face looked up that was not there).  Such an attempt is now reported as not converged, the constrained recovery takes over, and no fill is
    handed on with a tetrahedron that is not positively oriented (the reference's validator rejects one: tests/ValidateTetMesh.h:47-140)."""
    from mesheditor_amd import meshes, tets
    v, f = meshes.skillet_scan_surface(0.013, 0.01782900063536733, noise_seed=814)
    p, t, left = tets.tetrahedralize(v, f)
    assert left == 0 and np.array_equal(p[: len(v)], v)
    q = p[t.astype(np.int64)]
    vol6 = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0])
    assert vol6.min() > 0
    faces = np.sort(np.concatenate([t[:, [1, 2, 3]], t[:, [0, 2, 3]], t[:, [0, 1, 3]], t[:, [0, 1, 2]]]), axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    assert counts.max() <= 2 and {tuple(r) for r in uniq[counts == 1]} == {tuple(sorted(r)) for r in f.tolist()}
    e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
    assert (vol6 * np.sqrt(2) / e2 ** 1.5).min() > 1e-3


def test_the_flat_cell_pass_on_a_repaired_scan_fill():
    """tetra::Options::BreakFlatCells (always on; off here for the comparison): the 30k-tet skillet scan through the front end's default options keeps five cells
    below a shape measure of 1e-2 after sliver repair and smoothing (worst 2.9e-3); the pass puts an interior point beside them, or moves the added point that
    makes one flat: nothing below 1e-2 afterwards, the boundary still the scan's own triangulation, every added point inside."""
    from mesheditor_amd import meshes, tets
    v, f = meshes.skillet_scan_surface(0.011, 0.015)

    def worst(p, t):
        q = p[t.astype(np.int64)]
        vol6 = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0])
        e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
        assert vol6.min() > 0
        return float((vol6 * np.sqrt(2) / e2 ** 1.5).min())

    p0, t0, left0 = tets.tetrahedralize(v, f, break_flat_cells=False)
    p1, t1, left1 = tets.tetrahedralize(v, f)
    assert left0 == 0 and left1 == 0 and np.array_equal(p1[: len(v)], v)
    assert worst(p0, t0) < 1e-2 <= worst(p1, t1), (worst(p0, t0), worst(p1, t1))
    assert len(p0) <= len(p1) <= len(p0) + 64
    faces = np.sort(np.concatenate([t1[:, [1, 2, 3]], t1[:, [0, 2, 3]], t1[:, [0, 1, 3]], t1[:, [0, 1, 2]]]), axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    assert counts.max() <= 2 and {tuple(r) for r in uniq[counts == 1]} == {tuple(sorted(r)) for r in f.tolist()}


@pytest.mark.parametrize("name,was", [("torus_sliver", 2e-8), ("torus_edge", 1.4e-5), ("ellipsoid_cap", 3.8e-7)])
def test_the_soak_s_flat_fills_come_back_without_a_flat_cell(name, was):
    """Three surfaces of the round-6 soak (tests/golden/make_flat_fill_surfaces.py replays its random stream) whose default fills kept a cell at `was`
    after the flat-cell pass's insertions: a sliver of four surface vertices across a coarse torus's tube, a recovery point a hair off a surface edge,
    caps on planar quads over a fan of thin cells round a surface vertex (every insertion cavity unravels under the star-shape test).  The pass now
    splits an interior edge of such a cell inside the kernel of its ring, or moves the added point to the Chebyshev centre of its star
    (mesheditor_amd/cpp/src/tetrahedralize.cpp: BreakFlatCells): no cell below 1e-3, the boundary still the input's triangulation, the input
    vertices untouched, every tetrahedron positively oriented, every added point in one."""
    from mesheditor_amd import tets
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flat_fill_surfaces.npz"))
    v, f = data[name + "_P"], data[name + "_F"]
    p, t, left = tets.tetrahedralize(v, f)
    assert left == 0 and np.array_equal(p[: len(v)], v)
    q = p[t.astype(np.int64)]
    vol6 = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0])
    assert vol6.min() > 0
    e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
    assert (vol6 * np.sqrt(2) / e2 ** 1.5).min() > 1e-3
    assert len(np.unique(t)) == len(p)
    faces = np.sort(np.concatenate([t[:, [1, 2, 3]], t[:, [0, 2, 3]], t[:, [0, 1, 3]], t[:, [0, 1, 2]]]), axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    assert counts.max() <= 2 and {tuple(r) for r in uniq[counts == 1]} == {tuple(sorted(r)) for r in f.tolist()}
    # the enclosed volume (divergence theorem over the input triangles) is the fill's
    a, b, c = v[f[:, 0].astype(np.int64)], v[f[:, 1].astype(np.int64)], v[f[:, 2].astype(np.int64)]
    enclosed = abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6
    assert abs(vol6.sum() / 6 - enclosed) < 1e-9 * enclosed


def test_max_volume_has_the_last_word():
    """Found by the round-6 options fuzz (tools/probe/r06_front_end_fuzz.py): the quality arm bounded every volume, and the exchanges, the smoothing
    and the flat-cell pass that FOLLOW it merged and moved cells past the bound again (up to twice MaxVolume on one fill in five).  The bound is
    now enforced once more at the very end -- points only, the best-shaped of circumcentre / centroid / their midpoint, the flat-cell pass (no
    exchanges) in turn with it -- on the fuzz's first case: a 47 x 17 torus, InteriorShell::Always, MaxVolume = enclosed volume / 14 681."""
    import importlib.util
    from mesheditor_amd import tets
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_flat_fill_surfaces.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    v, f, name = mk.soak_surface(11, 1)
    assert name == "torus 47x17"
    bound = 7.35120153672187e-08
    p, t, left = tets.tetrahedralize(v, f, interior_shell="always", max_volume=bound)
    q = p[t.astype(np.int64)]
    vol6 = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0])
    assert left == 0 and np.array_equal(p[: len(v)], v) and vol6.min() > 0
    assert vol6.max() / 6 <= bound * (1 + 1e-12), vol6.max() / 6 / bound
    e2 = sum(((q[:, i] - q[:, j]) ** 2).sum(1) for i in range(4) for j in range(i + 1, 4)) / 6
    assert (vol6 * np.sqrt(2) / e2 ** 1.5).min() > 1e-3
    a, b, c = v[f[:, 0].astype(np.int64)], v[f[:, 1].astype(np.int64)], v[f[:, 2].astype(np.int64)]
    enclosed = abs(np.einsum("ij,ij->i", a, np.cross(b, c)).sum()) / 6
    assert abs(vol6.sum() / 6 - enclosed) < 1e-9 * enclosed
    # without RepairSlivers the quality arm used not to run at all: the bound holds there too
    p, t, _ = tets.tetrahedralize(v, f, repair_slivers=False, max_volume=4 * bound)
    q = p[t.astype(np.int64)]
    vol6 = np.einsum("ij,ij->i", np.cross(q[:, 1] - q[:, 0], q[:, 2] - q[:, 0]), q[:, 3] - q[:, 0])
    assert vol6.max() / 6 <= 4 * bound * (1 + 1e-12)
