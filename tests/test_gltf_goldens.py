"""The reference's own committed solver outputs as pins.

The sample glTFs of the reference embed what its real chain -- tests/ModalSolveTool.cpp:72-123: GenerateTets + modal::mesh2modes,
driven by glTF_PhysicalAudio/samples/generate.py:278-335 -- produced for eight bodies: frequencies, decay rates, every mode
shape at every surface vertex, mass properties, and (as `positions` + `indices`) the INPUT surface itself.
tests/golden/extract_gltf_modal_models.py decodes them into gltf_modal_models.json / gltf_modal_models_full.npz (data only).

Two kinds of pin:

* EXACT (Solved box, Bar, Platform).  These bodies are one grid cell thick, so the reference's tetrahedralisation is one of
  finitely many and tests/golden/find_reference_tets.py identified it from the golden itself (reference_tets.npz).  On that
  mesh the oracle must reproduce the reference's float32 outputs to round-off: every stored frequency within 2e-7 relative
  (measured: bit-equal), decay rates 2e-7, all shapes 1e-6 (measured <= 7e-9), mass 1e-14, inertia and positions to float32.
  "Bar" is an independent confirmation: its mesh was identified from the CERAMIC box, and reproduces the STEEL solve.
  The device path is held to the same goldens directly (-m gpu twin).
* FRONT END (all eight).  The golden surface goes through the path's own tetra::Tetrahedralize (input triangulation kept as
  the boundary, no point left on it) and the solve is compared at the tolerance that the different interior diagonals allow
  -- stated per model below; mass, which depends on the surface only, to 1e-12.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STEEL, CERAMIC, GLASS = (7850.0, 2.0e11, 0.29, 5.0, 3e-8), (2700.0, 7.2e10, 0.19, 6.0, 1e-7), (2500.0, 7.0e10, 0.22, 3.0, 1e-8)
# name -> (key in the .npz, material, MaxModeFreq of the solve [generate.py: 16 kHz default, 60 kHz for the small bodies])
MODELS = {
    "Solved box": ("test/StrikeOne/a_ThreeInstances.gltf|Solved box", CERAMIC, 16000.0),
    "Bar": ("Pile.gltf|Bar", STEEL, 16000.0),
    "Platform": ("Pile.gltf|Platform", CERAMIC, 16000.0),
    "Slab": ("Pile.gltf|Slab", CERAMIC, 60000.0),
    "Cube": ("Pile.gltf|Cube", CERAMIC, 60000.0),
    "Bracket": ("Pile.gltf|Bracket", STEEL, 60000.0),
    "Marble": ("Pile.gltf|Marble", GLASS, 60000.0),
    # (the scene lists this model under its ceramic material entry, but its stored mass is 7850 x the polyhedron's volume)
    "Solved sphere": ("test/AccelerationNoise/a_SteelBead.gltf|Solved sphere", STEEL, 60000.0),
}
REFERENCE_MESH = {"Solved box": "box_12x3x1", "Bar": "box_12x3x1", "Platform": "platform_12x1x12"}
# front end: (relative frequency tolerance over all stored modes, measured worst) -- the interior differs from the reference's
# (Cube: the 8 x 8 x 8 surface grid has an empty interior; our recovery adds 36 points there, which the repair pass then spreads --
# the finer interior LOWERS the P2 frequencies by up to 4.8 % against the reference's fill, which lists surface vertices only;
# without Options::RepairSlivers the same surface gives 0.9 %.  The two balls: the UV sphere's planar quads leave flat cap tetrahedra
# (shape 1e-8) in a fill without interior points, on which no iterative eigensolver converges; the front end therefore puts a shell
# of 266 points under the surface (Options::InteriorShell, WhenFlat) -- 536 points instead of the reference's 266, and frequencies
# 1-14 % BELOW the golden's (a finer interior is softer).  The reference's interior-free mesh gives the golden; ours is another
# discretisation of the same ball.)
FRONT_END_TOL = {"Solved box": 4e-3, "Bar": 4e-3, "Platform": 3e-3, "Slab": 2e-3, "Cube": 6e-2, "Bracket": 1e-2, "Marble": 0.15, "Solved sphere": 0.15}
# ... and the inertia, which the reference sums from vertex-lumped tet volumes (mesh2modes.cpp:61-110): it sees the interior too
FRONT_END_INERTIA_TOL = {"Solved box": 1e-3, "Bar": 1e-3, "Platform": 1e-3, "Slab": 1e-3, "Bracket": 2e-2, "Cube": 0.15, "Marble": 0.35, "Solved sphere": 0.35}


@pytest.fixture(autouse=True)
def small_thread_team(request):
    """These systems have a few thousand unknowns: a full OpenMP team costs more in start-up than it saves (and far more on a busy host)."""
    if "oracle" not in request.fixturenames:
        yield
        return
    po = request.getfixturevalue("oracle")
    before = po.lib().mo_max_threads()
    po.set_threads(2)
    yield
    po.set_threads(before)


@pytest.fixture(scope="module")
def full():
    return np.load(os.path.join(HERE, "golden", "gltf_modal_models_full.npz"))


@pytest.fixture(scope="module")
def reference_tets():
    return np.load(os.path.join(HERE, "golden", "reference_tets.npz"))


def golden_of(full, name):
    key, material, max_freq = MODELS[name]
    return {f: full[key + "|" + f] for f in ("positions", "indices", "shapes", "frequencies", "decayRates")}, material, max_freq


def shape_errors(shapes_position_major, gold_mode_major, gold_freqs):
    """Relative error of every mode's shape over all sample points; modes of nearly equal frequency as subspaces."""
    k = len(gold_freqs)
    got = np.transpose(shapes_position_major[:, :k, :], (1, 0, 2)).astype(np.float64)
    gold = gold_mode_major.astype(np.float64)
    err = np.zeros(k)
    m = 0
    while m < k:
        e = m + 1
        while e < k and gold_freqs[e] - gold_freqs[e - 1] < 3e-3 * gold_freqs[e]:
            e += 1
        a, b = got[m:e].reshape(e - m, -1).T, gold[m:e].reshape(e - m, -1).T
        x, *_ = np.linalg.lstsq(a, b, rcond=None)
        err[m:e] = np.linalg.norm(a @ x - b, axis=0) / np.linalg.norm(b, axis=0)
        m = e
    return err


def check_against_golden(r, g, golden_record, name, shape_tol=1e-6):
    """Everything the solve tool prints (tests/ModalSolveTool.cpp:101-123) against the reference's stored values."""
    gf = g["frequencies"].astype(np.float64)
    assert len(r.freqs) == len(gf), (name, len(r.freqs), len(gf))
    assert np.abs(r.freqs / gf - 1).max() < 2e-7, (name, np.abs(r.freqs / gf - 1).max())
    decay = np.float32(3 * np.log(np.float32(10))) / r.t60s  # the tool's float Ln1000 / T60 (ModalSolveTool.cpp:97-99)
    assert np.abs(decay / g["decayRates"] - 1).max() < 2e-7, (name, np.abs(decay / g["decayRates"] - 1).max())
    assert np.array_equal(r.positions, g["positions"])  # node-local float positions, one per distinct tet point, in request order
    err = shape_errors(r.shapes, g["shapes"], gf)
    assert err.max() < shape_tol, (name, err)
    mp = golden_record["massProperties"]
    assert abs(r.mass - mp["mass"]) <= 1e-14 * mp["mass"]
    assert np.allclose(r.inertia_diagonal, np.array(mp["inertiaDiagonal"], np.float32), rtol=3e-7)
    assert np.abs(np.asarray(r.center_of_mass) - np.array(mp["centerOfMass"], np.float32)).max() < 1e-9 * np.abs(g["positions"]).max()


@pytest.mark.parametrize("name", sorted(REFERENCE_MESH))
def test_oracle_reproduces_the_reference_output_on_the_references_mesh(oracle, golden, full, reference_tets, name):
    g, material, max_freq = golden_of(full, name)
    tets = reference_tets[REFERENCE_MESH[name]]
    # the mesh is a tiling of the golden's own surface: its boundary faces are exactly the input triangles
    assert boundary_faces(tets) == {tuple(sorted(map(int, t))) for t in g["indices"]}
    cfg = oracle.default_config(num_modes=30, num_fem_modes=45, max_mode_freq=max_freq)  # ModalSolveTool.cpp:66-71 with --modes 30
    r = oracle.mesh2modes(g["positions"].astype(np.float64), tets, oracle.material(*material), g["positions"], config=cfg)
    check_against_golden(r, g, golden[name], name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(REFERENCE_MESH))
def test_device_reproduces_the_reference_output_on_the_references_mesh(golden, full, reference_tets, name):
    """The device path against the reference's stored output DIRECTLY (no oracle in between)."""
    from mesheditor_amd import api
    g, material, max_freq = golden_of(full, name)
    ctx = api.Context(0)
    try:
        cfg = api.default_config(num_modes=30, num_fem_modes=45, max_mode_freq=max_freq)
        r = api.mesh2modes(ctx, g["positions"].astype(np.float64), reference_tets[REFERENCE_MESH[name]], api.material(*material), g["positions"], config=cfg)
    finally:
        ctx.close()
    # (the device stops at a relative residual of 1e-4 since round 6 -- 1e-5 before: eigenvalues to ~1e-9, measured worst shape deviation within the 1e-5 held here)
    check_against_golden(r, g, golden[name], name, shape_tol=1e-5)


def boundary_faces(tets):
    count = {}
    for t in np.asarray(tets, np.int64):
        for i in range(4):
            f = tuple(sorted(int(t[j]) for j in range(4) if j != i))
            count[f] = count.get(f, 0) + 1
    return {f for f, c in count.items() if c == 1}


def front_end_mesh(g):
    from mesheditor_amd import tets as front_end
    pts, tets, left_on_surface = front_end.tetrahedralize(g["positions"].astype(np.float64), g["indices"])
    return pts, tets, left_on_surface


@pytest.mark.parametrize("name", sorted(MODELS))
def test_golden_surface_through_the_front_end_and_the_oracle(oracle, golden, full, name):
    g, material, max_freq = golden_of(full, name)
    pts, tets, left_on_surface = front_end_mesh(g)
    # the reference's contract (src/mesh/Tetrahedralize.h:49-61): input vertex i keeps index i, every input triangle is a boundary face
    assert left_on_surface == 0 and np.array_equal(pts[: len(g["positions"])], g["positions"].astype(np.float64))
    assert boundary_faces(tets) == {tuple(sorted(map(int, t))) for t in g["indices"]}
    if name in REFERENCE_MESH:  # one-cell-thick grids: no point needed at all (the degenerate Delaunay cells are re-tiled instead)
        assert len(pts) == len(g["positions"])
    cfg = oracle.default_config(num_modes=30, num_fem_modes=45, max_mode_freq=max_freq)
    r = oracle.mesh2modes(pts, tets, oracle.material(*material), g["positions"], config=cfg)
    gf = g["frequencies"].astype(np.float64)
    k = min(len(gf), len(r.freqs))
    assert k >= len(gf) - 4, (name, k)  # (on the two spheres the last modes of the window fall outside on our coarser interior)
    rel = np.abs(r.freqs[:k] / gf[:k] - 1)
    assert rel.max() < FRONT_END_TOL[name], (name, rel.max())
    assert np.array_equal(r.positions, g["positions"])
    mp = golden[name]["massProperties"]
    assert abs(r.mass - mp["mass"]) <= 1e-12 * mp["mass"]  # the enclosed volume is the surface's: independent of the interior
    assert np.allclose(np.sort(r.inertia_diagonal), np.sort(mp["inertiaDiagonal"]), rtol=FRONT_END_INERTIA_TOL[name])
    if name in ("Marble", "Solved sphere"):  # the five-fold l = 2 multiplet of a ball comes out as a cluster of five
        assert r.freqs[4] / r.freqs[0] < 1.02 and r.freqs[5] / r.freqs[4] > 1.03
        assert gf[4] / gf[0] < 1.02 and gf[5] / gf[4] > 1.03


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_golden_surface_through_the_front_end_and_the_device(oracle, full, name):
    """The same eight unstructured little meshes, device against oracle at the parity bar (eigenvalues 1e-6)."""
    from mesheditor_amd import api
    g, material, max_freq = golden_of(full, name)
    pts, tets, _ = front_end_mesh(g)
    ref = oracle.mesh2modes(pts, tets, oracle.material(*material), g["positions"], config=oracle.default_config(num_modes=30, num_fem_modes=45, max_mode_freq=max_freq))
    ctx = api.Context(0)
    try:
        got = api.mesh2modes(ctx, pts, tets, api.material(*material), g["positions"], config=api.default_config(num_modes=30, num_fem_modes=45, max_mode_freq=max_freq))
    finally:
        ctx.close()
    assert len(got.eigenvalues) == len(ref.eigenvalues) == 45
    elastic = ref.eigenvalues > 1e-6 * ref.eigenvalues[-1]
    assert elastic.sum() == 39
    assert np.abs(got.eigenvalues[elastic] / ref.eigenvalues[elastic] - 1).max() < 1e-6
    assert len(got.freqs) == len(ref.freqs) and np.allclose(got.freqs, ref.freqs, rtol=2e-6) and np.allclose(got.t60s, ref.t60s, rtol=4e-6)
    assert abs(got.mass - ref.mass) <= 1e-12 * ref.mass
    assert shape_errors(got.shapes, np.transpose(ref.shapes, (1, 0, 2)), ref.freqs.astype(np.float64)).max() < 1e-4
