"""modal::SolveBatch through its Python binding (mesheditor_amd/batch.py): a one-rank RCCL communicator, one ncclAllGather per
batch; the records must equal direct solves through the C ABI, and a failing mesh must surface after the gather."""
import os
import subprocess
import sys

import numpy as np
import pytest

from mesheditor_amd import meshes

pytestmark = pytest.mark.gpu
INNER = os.environ.get("MH_BATCH_TESTS_INNER") == "1"
inner = pytest.mark.skipif(not INNER, reason="runs in the fresh interpreter started by test_batch_driver_in_a_fresh_interpreter")


@pytest.mark.skipif(INNER, reason="the outer test")
def test_batch_driver_in_a_fresh_interpreter():
    """The tests below run in an interpreter of their own: earlier tests of a whole-suite run import PyTorch, whose wheel bundles
    a second copy of the ROCm runtime, and the system RCCL must not be bound against that copy (mesheditor_amd/batch.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider"], capture_output=True, text=True,
                       timeout=900, cwd=root, env=dict(os.environ, MH_BATCH_TESTS_INNER="1"))
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]


@inner
def test_record_layouts_agree():
    from mesheditor_amd import batch, sharding
    L = batch._lib()
    for nev, pos in ((256, 16), (45, 10), (65, 4)):
        assert int(L.mhx_batch_record_length(nev, pos)) == sharding.record_length(nev, pos)


@inner
def test_solve_batch_world_one_matches_direct_solves():
    from mesheditor_amd import api, batch
    items = []
    for i, n in enumerate((5, 4, 6, 5)):
        p, t = meshes.jittered_box(n, 1000 + i)
        items.append((p, t, meshes.MATERIALS[meshes.MATERIAL_ORDER[i]], {"num_modes": 12, "num_fem_modes": 24}))
    comm = batch.BatchComm(1, 0, 0, batch.make_id())
    try:
        records = batch.solve_batch(comm, items, nev_max=32, pos_max=16, threads=2)
        again = batch.solve_batch(comm, items, nev_max=32, pos_max=16, threads=3)  # a second batch on the same communicator
    finally:
        comm.close()
    assert [r["index"] for r in records] == [0, 1, 2, 3]
    ctx = api.Context(0)
    for i, (p, t, m, kw) in enumerate(items):
        ex = p[(np.arange(10) * len(p)) // 10].astype(np.float32)
        d = api.mesh2modes(ctx, p, t, api.material(*m), ex, config=api.default_config(**kw))
        r = records[i]
        assert np.array_equal(r["eigenvalues"], d.eigenvalues)  # the same deterministic solve, whichever thread and context ran it
        assert np.array_equal(r["freqs"], d.freqs) and np.array_equal(r["t60s"], d.t60s)
        assert r["mass"] == d.mass and np.array_equal(r["positions"], d.positions)
        assert np.array_equal(r["summary_shapes"], d.summary_shapes)
        assert np.array_equal(again[i]["eigenvalues"], d.eigenvalues)
    ctx.close()


@inner
def test_a_failing_mesh_is_reported_after_the_gather():
    from mesheditor_amd import batch
    p, t = meshes.kuhn_box(4, 4, 4, 0.1, 0.1, 0.1)
    flat_p = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]])
    flat_t = np.array([[0, 1, 2, 3]], np.uint32)
    items = [(p, t, meshes.MATERIALS["Glass"], {"num_modes": 10, "num_fem_modes": 20}), (flat_p, flat_t, meshes.MATERIALS["Glass"], {"num_modes": 10, "num_fem_modes": 20})]
    comm = batch.BatchComm(1, 0, 0, batch.make_id())
    try:
        with pytest.raises(RuntimeError, match=r"mesh\(es\) \[1\] failed"):
            batch.solve_batch(comm, items, nev_max=32, pos_max=16, threads=2)
    finally:
        comm.close()
