"""The metric's own configurations against committed oracle results (tests/golden/oracle_eigs_*.json, made in the build
container by tests/golden/make_oracle_fixtures.py -- minutes of CPU each, the reference algorithm of
src/audio/mesh2modes.cpp:441-512 restated by oracle/analysis.cpp): the device path through the C ABI must reproduce every
eigenvalue to 1e-6 (rigid-body pairs absolutely), the kept frequencies and decay times, the mass properties and the
excitation map.  cube_s100k is bench.py's workload, skillet_s100k / ball_s10k BASELINE configs 3 / 2, cube_s30k the
RealImpact-sized Kuhn grid, scan_s30k / scan_s100k the scan-like unstructured meshes (marching-tetrahedra skillet surface
through the path's own tetrahedraliser: slivers, 2 to 60 tets around a node, no interior points), the "_interior" ones the
same surfaces with the boundary recovery's points moved inside afterwards (the mesh's boundary is then the scan's own
triangulation, as the reference's contract wants; ~15 % more tets), the "_repaired" ones the front end's round-4 default on top of
that (connectivity-only sliver repair + smoothing of the added points, as the reference's tetrahedraliser always runs);
config3_* = the scan meshes with NumModes = 200 (BASELINE configs[2] as written)."""
import json
import os

import numpy as np
import pytest

from mesheditor_amd import meshes

HERE = os.path.dirname(os.path.abspath(__file__))
WORKLOADS = ["ball_s10k", "cube_s30k", "cube_s100k", "skillet_s100k", "scan_s30k", "scan_s100k", "scan_s30k_interior", "scan_s100k_interior",
             # round 4: BASELINE config 3 as written (scan mesh x 200 modes = 215 pairs) at both sizes; the scan surfaces through the
             # front end's default options (points moved inside, sliver repair, smoothing), 65 and 215 pairs
             "config3_s30k", "config3_s100k", "scan_s30k_repaired", "scan_s100k_repaired", "config3_s30k_repaired", "config3_s100k_repaired",
             # BASELINE config 2 as written: the UV-sphere primitive (48 x 24 surface, 9 457 tets after the front end) -- ball_s10k is its Kuhn-mapped stand-in
             "uvsphere_s10k",
             # round 6: a FINE UV sphere (80 x 40) through the front end's default options -- the class that took 57 iterations (96 x 48) or returned
             # nothing (128 x 64) in round 5; the oracle's factorisation of the 96 x 48 and 128 x 64 fills does not fit the build container (> 60 GB)
             "uvsphere_80x40"]


def load_fixture(name):
    with open(os.path.join(HERE, "golden", f"oracle_eigs_{name}.json")) as f:
        return json.load(f)


def mesh_digest(pts, tets):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(pts, np.float64).tobytes() + np.ascontiguousarray(tets, np.uint32).tobytes()).hexdigest()


@pytest.mark.parametrize("name", WORKLOADS)
def test_fixture_describes_the_mesh_the_generator_makes(name):
    """CPU: the committed result belongs to the mesh meshes.workload(name) produces here (the scan meshes pass through numpy's
    generators and the host tetrahedraliser: byte-identical or the comparison below would be meaningless), and its mass is
    the oracle's lumped mass of that mesh."""
    fx = load_fixture(name)
    pts, tets, m, kw = meshes.workload(name)
    assert fx["mesh"]["tets"] == len(tets) and fx["mesh"]["points"] == len(pts)
    assert fx["mesh"]["sha256"] == mesh_digest(pts, tets)
    assert fx["config"]["num_fem_modes"] == kw["num_fem_modes"] == len(fx["eigenvalues"])
    p = pts[tets.astype(np.int64)]
    vol = np.abs(np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])).sum() / 6
    assert abs(fx["mass"] - m[0] * vol) < 1e-6 * fx["mass"]  # (the reference sums the volumes in its own order, with a float 1/6)
    ev = np.array(fx["eigenvalues"])
    assert np.all(np.abs(ev[:6]) < 1e-6 * ev[6]) and np.all(np.diff(ev[6:]) >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", WORKLOADS)
def test_device_result_matches_the_committed_oracle_result(name):
    from mesheditor_amd import api
    fx = load_fixture(name)
    pts, tets, m, kw = meshes.workload(name)
    assert fx["mesh"]["sha256"] == mesh_digest(pts, tets), "the generator produced a different mesh than the fixture was made on"
    ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
    ctx = api.Context(0)
    try:
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(**kw), keep_system=True)
        worst_plain, dropped = r.system.residual_report()
        r.system.close()
    finally:
        ctx.close()
    # what the tolerance meant (VERDICT round 4, item 6 iv): a mesh without sliver patches is accepted in the plain 2-norm (report -1); one
    # with them in the Jacobi-scaled norm, and the worst PLAIN relative residual of the returned elastic pairs is then bounded here
    # (measured 2e-5 ... 3e-3 on the scan fixtures: the slivers' rows carry rounding noise eps ||A|| |x| in the 2-norm)
    assert worst_plain == -1.0 or 0 < worst_plain < 1e-2, (name, worst_plain)
    assert dropped[0] < 50 and dropped[1] < 50, dropped
    # health of the device solve: no Rayleigh-Ritz step redone, every step's self-check at rounding level (mh_profile)
    assert r.profile["sytrd_redos"] == 0 and r.profile["rr_selfcheck"] < 1e-9, r.profile  # (measured 1e-13 ... 3e-12 here, up to 1e-10 on jittered boxes; the solve itself fails at 1e-8)
    ref = np.array(fx["eigenvalues"])
    assert len(r.eigenvalues) == len(ref), r.profile
    elastic = ref > 1e-6 * ref[-1]
    assert elastic.sum() == len(ref) - 6
    rel = np.abs(r.eigenvalues[elastic] - ref[elastic]) / ref[elastic]
    assert rel.max() < 1e-6, (name, rel.max())
    assert np.abs(r.eigenvalues[~elastic]).max() < 1e-6 * ref[elastic][0]
    # what PostprocessModes keeps: frequencies and decay times (float32 in the reference's ModalModes)
    assert len(r.freqs) == len(fx["freqs"])
    assert np.allclose(r.freqs, np.array(fx["freqs"], np.float32), rtol=2e-6)
    assert np.allclose(r.t60s, np.array(fx["t60s"], np.float32), rtol=4e-6)
    assert abs(r.original_fundamental - fx["original_fundamental"]) <= 2e-6 * fx["original_fundamental"]
    # mass properties and the excitation map
    assert abs(r.mass - fx["mass"]) <= 1e-12 * fx["mass"]
    assert np.allclose(r.center_of_mass, fx["center_of_mass"], atol=1e-6 * np.abs(pts).max())
    assert np.allclose(r.inertia_diagonal, fx["inertia_diagonal"], rtol=1e-5)
    assert np.array_equal(r.sample_point_of_excitation, np.array(fx["sample_point_of_excitation"], np.uint32))
    # shapes at the excitation points, ALL pairs: per cluster of (nearly) equal eigenvalues the sum of outer products
    # S = sum_j s_j s_j^T (s_j: the pair's displacement at the points, 3P values) does not depend on the basis chosen inside
    # the eigenspace -- for a simple pair this is the shape up to sign, for the cubes' multiplets the eigenspace itself.
    # The last cluster may be cut by the number of pairs requested and is left out.
    shp = np.array(fx["summary_shapes"], np.float64)  # [position][pair][3]
    got = r.summary_shapes.astype(np.float64)
    first = np.r_[True, np.diff(ref) > 1e-4 * np.maximum(ref[1:], ref[6])]
    starts = np.r_[np.where(first)[0], len(ref)]
    checked = 0
    for a, b in zip(starts[:-2], starts[1:-1]):
        if a < 6:  # the rigid-body six: any basis of translations and rotations
            continue
        A = got[:, a:b, :].transpose(1, 0, 2).reshape(b - a, -1)
        B = shp[:, a:b, :].transpose(1, 0, 2).reshape(b - a, -1)
        SA, SB = A.T @ A, B.T @ B
        assert np.abs(SA - SB).max() <= 5e-3 * np.abs(SB).max() + 1e-12, (name, int(a), int(b))
        checked += b - a
    assert checked >= len(ref) // 2
