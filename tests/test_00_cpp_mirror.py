"""C++ tests of the host mirror (tests/cpp): the reference's ModalSolverTest / ModalRenderTest / ContactModelTest
properties, written against the mirrored headers under the reference's include paths.  On the CPU: they compile and
link (the API surface is intact) and the scalar contact model runs; on the GPU box: all three run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "mesheditor_amd", "libmodalhost.so")):
        ge.build()
    subprocess.run(["make", "-s", "-C", CPP], check=True)


def _run(name, timeout=600):
    p = subprocess.run([os.path.join(CPP, "bin", name)], capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_cpp_tests_compile_against_the_mirror():
    _build()
    for name in ("modal_solver_test", "modal_render_test", "contact_model_test", "model_io_test", "tets_test", "tet_caller_test", "batch_test"):
        assert os.path.exists(os.path.join(CPP, "bin", name))


def test_contact_model_known_answers():
    _build()
    assert "0 failure(s)" in _run("contact_model_test")


def test_gltf_modal_models_and_modal_store():
    """the KHR_audio_rigid_bodies reader/writer against the reference's committed sample
    scene, and the content-addressed .modal store (host code, no GPU)."""
    _build()
    assert "0 failure(s)" in _run("model_io_test")


def test_star_shaped_tet_fill_and_obj_loader():
    """the tet-generation front end against the validity properties of the reference's
    tests/ValidateTetMesh.h (host code, no GPU)."""
    _build()
    assert "0 failure(s)" in _run("tets_test")


def test_tet_front_end_binds_as_the_reference_declares_it():
    """SURVEY section 8f row N3, the drop-in half: tests/cpp/tet_caller_test.cpp restates the reference's two callers of GenerateTets /
    tetra::Tetrahedralize (tests/ModalSolveTool.cpp:72-77, tests/ModalSolverBench.cpp:285-327) against the mirror's mesh/Tets.h and
    mesh/Tetrahedralize.h -- expected-shaped return, tetra::Profile -- and checks the Profile's arithmetic (host code, no GPU)."""
    _build()
    assert "0 failure(s)" in _run("tet_caller_test")


@pytest.mark.gpu
def test_solve_tool_takes_a_surface_obj(tmp_path):
    """The reference tool's own input: a surface .obj, filled with tets and solved with the .obj's vertices as the
    excitation positions.  A cube's first elastic modes must agree with the same cube meshed as a Kuhn grid (two
    different quadratic-tet discretisations of one solid) and the mass must be exact."""
    import json
    import numpy as np
    _build()
    n, side = 4, 0.1
    ids, pts, tris = {}, [], []
    def vid(c):
        if c not in ids:
            ids[c] = len(pts)
            pts.append([side * v / n for v in c])
        return ids[c]
    for axis in range(3):
        for s in (0, n):
            for u in range(n):
                for v in range(n):
                    def at(uu, vv):
                        c = [0, 0, 0]
                        c[axis], c[(axis + 1) % 3], c[(axis + 2) % 3] = s, uu, vv
                        return vid(tuple(c))
                    a, b, c, d = at(u, v), at(u + 1, v), at(u + 1, v + 1), at(u, v + 1)
                    tris += [(a, b, c), (a, c, d)]
    obj = tmp_path / "cube.obj"
    obj.write_text("".join(f"v {x!r} {y!r} {z!r}\n" for x, y, z in pts) + "".join(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in tris))
    tool = os.path.join(ROOT, "mesheditor_amd", "cpp", "bin", "modal_solve")
    common = ["--young", "7.2e10", "--poisson", "0.19", "--density", "2700", "--modes", "8", "--max-freq", "60000"]
    p = subprocess.run([tool, str(obj), "--layers", "2", *common], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout)
    q = subprocess.run([tool, "--kuhn", repr(side), repr(side), repr(side), "4", "4", "4", *common], capture_output=True, text=True, timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    ref = json.loads(q.stdout)
    assert abs(out["mass"] / (2700 * side ** 3) - 1) < 1e-6
    assert len(out["positions"]) == len(pts) and len(out["indices"]) == 3 * len(tris)
    assert np.allclose(np.array(out["positions"]), np.array(pts, dtype=np.float32), atol=1e-7)
    f, g = np.array(out["frequencies"]), np.array(ref["frequencies"])
    assert len(f) == len(g) == 8 and np.abs(f / g - 1).max() < 0.03, (f, g)
    bad = tmp_path / "open.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 2 4\nf 1 3 4\n")
    r = subprocess.run([tool, str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Tetrahedralization failed" in r.stderr


def _voxel_surface(cells, h):
    """Boundary of a union of grid cubes of edge h (cells: set of integer triples) as (points, triangles): every cube face that is
    not shared by two cells, split along one diagonal."""
    ids, pts, tris = {}, [], []
    def vid(c):
        if c not in ids:
            ids[c] = len(pts)
            pts.append([h * v for v in c])
        return ids[c]
    for (i, j, k) in sorted(cells):
        for axis in range(3):
            for side in (0, 1):
                nb = [i, j, k]
                nb[axis] += 1 if side else -1
                if tuple(nb) in cells:
                    continue
                u, v = (axis + 1) % 3, (axis + 2) % 3
                def corner(du, dv):
                    c = [i, j, k]
                    c[axis] += side
                    c[u] += du
                    c[v] += dv
                    return vid(tuple(c))
                a, b, c, d = corner(0, 0), corner(1, 0), corner(1, 1), corner(0, 1)
                tris += [(a, b, c), (a, c, d)]
    return pts, tris


@pytest.mark.gpu
def test_solve_tool_fills_a_surface_that_is_not_star_shaped(tmp_path):
    """obj -> tets -> modes through the GENERAL tetrahedraliser (an L-bracket is not star-shaped about its
    centroid, which lies outside it) and the device solve: the mass is the bracket's, the input vertices come back as the
    sample points, and a bracket twice the size rings exactly one octave lower (f ~ 1 / L at fixed material) with eight times the mass."""
    import json
    import numpy as np
    _build()
    n = 3  # grid cells per unit length
    cells = {(i, j, k) for i in range(2 * n) for j in range(2 * n) for k in range(n) if i < n or j < n}  # an L of three unit cubes
    tool = os.path.join(ROOT, "mesheditor_amd", "cpp", "bin", "modal_solve")
    common = ["--young", "7.2e10", "--poisson", "0.19", "--density", "2700", "--modes", "6", "--max-freq", "200000"]
    runs = []
    for unit in (0.05, 0.10):
        pts, tris = _voxel_surface(cells, unit / n)
        obj = tmp_path / f"bracket_{unit}.obj"
        obj.write_text("".join(f"v {x!r} {y!r} {z!r}\n" for x, y, z in pts) + "".join(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in tris))
        p = subprocess.run([tool, str(obj), *common], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out = json.loads(p.stdout)
        assert abs(out["mass"] / (2700 * 3 * unit ** 3) - 1) < 1e-6
        assert len(out["positions"]) >= len(pts) and np.allclose(np.array(out["positions"][: len(pts)]), np.array(pts, dtype=np.float32), atol=1e-7)
        # the tetrahedraliser refines this surface (points on it), yet the model's surface is the caller's: the sample points are
        # the input vertices under their own indices and `indices` are the input triangles (reference tests/ModalSolveTool.cpp:84-94)
        assert np.array_equal(np.array(out["indices"]).reshape(-1, 3), np.array(tris))
        runs.append(np.array(out["frequencies"]))
    small, large = runs
    assert len(small) == len(large) == 6 and small[0] > 1000
    assert np.abs(small / (2 * large) - 1).max() < 1e-4, (small, large)


@pytest.mark.gpu
def test_solve_tool_reproduces_the_reference_sample_model(golden):
    """The JSON solve tool (the reference's MeshEditorModalSolve, which its sample generator shells out to) on the
    'Solved box' body of the reference's sample scene: same fields, frequencies within the tetrahedralisation tolerance
    (SURVEY 8c, G3), exact decay law, mass; the --gltf output reads back as the same model."""
    import base64, json, struct, tempfile
    import numpy as np
    _build()
    model = golden["Solved box"]
    lo, hi = np.array(model["positionMin"]), np.array(model["positionMax"])
    ext = hi - lo
    tool = os.path.join(ROOT, "mesheditor_amd", "cpp", "bin", "modal_solve")
    with tempfile.TemporaryDirectory() as tmp:
        gltf = os.path.join(tmp, "box.gltf")
        args = [tool, "--kuhn", *(repr(float(v)) for v in ext), "12", "3", "1", "--origin", *(repr(float(v)) for v in lo), "--young", "7.2e10", "--poisson", "0.19",
                "--density", "2700", "--alpha", "6", "--beta", "1e-7", "--modes", "10", "--gltf", gltf]
        p = subprocess.run(args, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out = json.loads(p.stdout)
        doc = json.load(open(gltf))
    f = np.array(out["frequencies"])
    ref = np.array(model["frequencies"])
    assert len(f) == len(ref) == 10
    assert np.abs(f[:4] / ref[:4] - 1).max() < 3.5e-4 and np.abs(f / ref - 1).max() < 3e-3, f / ref
    # d = (alpha + beta omega^2) / 2 with omega the UNDAMPED rate; the model lists damped frequencies, omega_d^2 = omega^2 - d^2
    wd = 2 * np.pi * f
    d = np.zeros_like(wd)
    for _ in range(8):
        d = (6 + 1e-7 * (wd * wd + d * d)) / 2
    assert np.allclose(out["decayRates"], d, rtol=2e-6)
    assert abs(out["mass"] / model["massProperties"]["mass"] - 1) < 1e-6
    # lumped vertex volumes depend on the tetrahedralisation (Kuhn here, the reference's generator there)
    assert np.allclose(out["inertiaDiagonal"], model["massProperties"]["inertiaDiagonal"], rtol=0.1), out["inertiaDiagonal"]
    npts = len(out["positions"])
    assert npts == model["numPositions"] and len(out["shapes"]) == 10 * npts and len(out["indices"]) == 3 * model["numTriangles"]
    assert max(out["indices"]) < npts
    # the glTF written beside it carries the same numbers
    mm = doc["extensions"]["KHR_audio_rigid_bodies"]["modalModels"][0]
    blob = base64.b64decode(doc["buffers"][0]["uri"].split(",", 1)[1])
    view = doc["bufferViews"][doc["accessors"][mm["frequencies"]]["bufferView"]]
    got = struct.unpack_from("<10f", blob, view["byteOffset"])
    assert np.allclose(got, f, rtol=1e-7)
    assert abs(mm["massProperties"]["mass"] - out["mass"]) < 1e-12


@pytest.mark.gpu
def test_modal_render_properties_cpp():
    _build()
    assert "0 failure(s)" in _run("modal_render_test")


@pytest.mark.gpu
def test_modal_solver_closed_forms_cpp():
    _build()
    assert "0 failure(s)" in _run("modal_solver_test")


@pytest.mark.gpu
def test_solve_batch_cpp_with_rccl_on_one_rank():
    """modal::SolveBatch (the C++ multi-GPU batch driver): one-rank RCCL communicator, one ncclAllGather per batch, records equal
    to direct solves, a failing mesh travels as a failed record (tests/cpp/batch_test.cpp)."""
    _build()
    assert "0 failure(s)" in _run("batch_test")


@pytest.mark.gpu
@pytest.mark.parametrize("name,key,mesh,material", [
    ("Solved box", "test/StrikeOne/a_ThreeInstances.gltf|Solved box", "box_12x3x1", (2700, 7.2e10, 0.19, 6, 1e-7)),
    ("Bar", "Pile.gltf|Bar", "box_12x3x1", (7850, 2.0e11, 0.29, 5, 3e-8)),
    ("Platform", "Pile.gltf|Platform", "platform_12x1x12", (2700, 7.2e10, 0.19, 6, 1e-7)),
])
def test_solve_tool_reproduces_the_reference_output_on_the_references_mesh(golden, name, key, mesh, material):
    """End to end through the C++ mirror: the JSON solve tool (our MeshEditorModalSolve) given the reference's own tetrahedralisation of
    a sample body (tests/golden/reference_tets.npz, identified from the goldens: tests/test_gltf_goldens.py) prints what the reference's
    tool printed into the sample scenes -- frequencies to float32 round-off, decay rates, mass, inertia, every shape, the positions and the
    triangles (as a set: the tool lists the mesh's boundary faces)."""
    import json, tempfile
    import numpy as np
    _build()
    full = np.load(os.path.join(ROOT, "tests", "golden", "gltf_modal_models_full.npz"))
    tets = np.load(os.path.join(ROOT, "tests", "golden", "reference_tets.npz"))[mesh]
    pos = full[key + "|positions"]
    tool = os.path.join(ROOT, "mesheditor_amd", "cpp", "bin", "modal_solve")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "body.tet")
        with open(path, "w") as f:
            f.write("%d %d\n" % (len(pos), len(tets)))
            for p_ in pos.astype(np.float64):
                f.write("%r %r %r\n" % (float(p_[0]), float(p_[1]), float(p_[2])))
            for t in tets:
                f.write("%d %d %d %d\n" % tuple(int(v) for v in t))
        args = [tool, path, "--density", repr(float(material[0])), "--young", repr(float(material[1])), "--poisson", repr(float(material[2])),
                "--alpha", repr(float(material[3])), "--beta", repr(float(material[4])), "--modes", "30"]
        p = subprocess.run(args, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out = json.loads(p.stdout)
    gf = full[key + "|frequencies"].astype(np.float64)
    f = np.array(out["frequencies"])
    assert len(f) == len(gf) and np.abs(f / gf - 1).max() < 2e-7, np.abs(f / gf - 1).max()
    assert np.abs(np.array(out["decayRates"]) / full[key + "|decayRates"] - 1).max() < 1e-6
    # (a tet mesh has no surface vertex order of its own: the tool takes the boundary vertices as it meets them; same points, another order)
    got_pos = np.array(out["positions"], np.float32)
    where = {tuple(p_): i for i, p_ in enumerate(pos.tolist())}
    order = np.array([where[tuple(p_)] for p_ in got_pos.tolist()])
    assert sorted(order.tolist()) == list(range(len(pos)))
    mp = golden[name]["massProperties"]
    assert abs(out["mass"] / mp["mass"] - 1) < 1e-12
    assert np.allclose(out["inertiaDiagonal"], mp["inertiaDiagonal"], rtol=1e-6)
    shapes = np.array(out["shapes"], np.float64).reshape(len(f), len(pos), 3)
    gold = full[key + "|shapes"].astype(np.float64)[:, order, :]
    m = 0
    while m < len(f):  # modes of nearly equal frequency as subspaces (the platform is square)
        e = m + 1
        while e < len(f) and gf[e] - gf[e - 1] < 3e-3 * gf[e]:
            e += 1
        a, b = shapes[m:e].reshape(e - m, -1).T, gold[m:e].reshape(e - m, -1).T
        x, *_ = np.linalg.lstsq(a, b, rcond=None)
        assert (np.linalg.norm(a @ x - b, axis=0) / np.linalg.norm(b, axis=0)).max() < 1e-5, (name, m, e)
        m = e
    got_tris = {tuple(sorted(int(order[v]) for v in out["indices"][3 * k: 3 * k + 3])) for k in range(len(out["indices"]) // 3)}
    assert got_tris == {tuple(sorted(int(v) for v in t)) for t in full[key + "|indices"]}
