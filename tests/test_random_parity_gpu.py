"""Randomised parity sweep: jittered boxes of random shape, size, material and mode count, device eigenvalues against the
oracle's shift-invert result (the 1e-6 bar of BASELINE north_star), all multiplicities included.  Exercises the block
sizes, locking patterns and Rayleigh-Ritz orders the fixed cases do not."""
import numpy as np
import pytest

from mesheditor_amd import meshes

pytestmark = pytest.mark.gpu
SIGMA = -(2 * np.pi * 20.0) ** 2


@pytest.fixture(scope="module")
def api():
    from mesheditor_amd import api as _api
    return _api


@pytest.fixture(scope="module")
def ctx(api):
    c = api.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_boxes_match_oracle(api, ctx, oracle, seed):
    rng = np.random.default_rng(4000 + seed)
    nx, ny, nz = (int(v) for v in rng.integers(3, 9, 3))
    ext = rng.uniform(0.05, 0.6, 3)
    pts, tets = meshes.kuhn_box(nx, ny, nz, *ext)
    h = ext / np.array([nx, ny, nz])
    interior = np.all((pts > 1e-12) & (pts < ext - 1e-12), axis=1)
    pts = pts + interior[:, None] * rng.uniform(-0.15, 0.15, pts.shape) * h  # jitter: breaks the symmetric multiplets
    if seed % 3 == 0:
        pts, tets = meshes.kuhn_box(nx, nx, nx, ext[0], ext[0], ext[0])  # and keep some exactly symmetric cubes (3-fold multiplets)
    name = meshes.MATERIAL_ORDER[seed % 7]
    mat = meshes.MATERIALS[name]
    sysg = api.System(ctx, api.Mesh(ctx, pts, tets), api.material(*mat))
    syso = oracle.System(pts, tets, oracle.material(*mat))
    nev = int(rng.integers(12, min(70, sysg.n // 8)))
    ev, prof = sysg.eigs(nev, SIGMA, 1e-6)
    evo, _, _ = syso.eigs(nev)
    elastic = evo > 1e-6 * evo[-1]
    assert elastic.sum() == nev - 6, (nev, elastic.sum())
    rel = np.abs(ev[elastic] - evo[elastic]) / evo[elastic]
    assert rel.max() < 1e-6, (seed, name, (nx, ny, nz), nev, rel.max())
    assert np.abs(ev[~elastic]).max() < 1e-6 * evo[6]
