"""The N > 1 path on CPU: two gloo ranks deal a batch of meshes by cost, 'solve' their shares (a deterministic stand-in
for the device solve -- the GPU is absent here), and gather fixed-size records with the same collective the GPU box
runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from mesheditor_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_deal_matches_reference_rule():
    costs = [5, 9, 5, 1, 7, 7, 3]
    deal = sharding.lpt_deal(costs, 3)
    # heaviest first (9 | 7 | 7), then 5 -> least loaded lowest index ...
    assert sorted(sum(deal, [])) == list(range(len(costs)))
    loads = [sum(costs[i] for i in d) for d in deal]
    assert max(loads) - min(loads) <= max(costs)
    assert deal == [sorted(d) for d in deal]
    assert sharding.lpt_deal(costs, 1) == [list(range(len(costs)))]
    assert sharding.lpt_deal([4, 4, 4, 4], 2) == [[0, 2], [1, 3]]  # ties: lowest index first, lowest rank first


class _FakeResult:
    def __init__(self, i, nev):
        rng = np.random.default_rng(100 + i)
        self.eigenvalues = np.sort(rng.uniform(0, 1e9, nev))
        self.freqs = np.sqrt(self.eigenvalues[6:36]).astype(np.float32)
        self.t60s = (1.0 / (1 + self.freqs)).astype(np.float32)
        self.mass = 1.0 + i
        self.profile = {"dofs": 3000 + i, "restarts": 20 + i}


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 9
        meshes = [(None, np.zeros((1000 + 137 * (i % 4), 4)), None, {"num_fem_modes": 45 if i % 2 else 65}) for i in range(n)]
        solved = []

        def solve(i, m):
            solved.append(i)
            return _FakeResult(i, m[3]["num_fem_modes"])
        recs = sharding.solve_batch(meshes, solve, 65, dist)
        q.put((rank, solved, [(r["index"], r["dofs"], float(r["eigenvalues"].sum()), len(r["freqs"])) for r in recs]))
    finally:
        dist.destroy_process_group()


def test_two_rank_batch_gather():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    out.sort()
    (r0, solved0, recs0), (r1, solved1, recs1) = out
    assert sorted(solved0 + solved1) == list(range(9)) and not set(solved0) & set(solved1)
    assert recs0 == recs1 and [r[0] for r in recs0] == list(range(9))
    for i, dofs, evsum, k in recs0:
        ref = _FakeResult(i, 45 if i % 2 else 65)
        assert dofs == 3000 + i and abs(evsum - ref.eigenvalues.sum()) < 1e-6 * evsum and k == 30


def test_threaded_share_solves_every_mesh_once():
    """Several host threads per rank (concurrent solves on one GPU): the rank's share is dealt over the workers by the same
    rule, every mesh is solved exactly once, records come back in index order, a worker's failure is raised."""
    n = 11
    meshes = [(None, np.zeros((900 + 211 * (i % 5), 4)), None, {"num_fem_modes": 45}) for i in range(n)]
    seen = []

    def solve(i, m, worker):
        seen.append((worker, i))
        return _FakeResult(i, 45)
    recs = sharding.solve_batch(meshes, solve, 45, None, threads=3)
    assert [r["index"] for r in recs] == list(range(n))
    assert sorted(i for _, i in seen) == list(range(n)) and {w for w, _ in seen} == {0, 1, 2}
    for r in recs:
        assert np.array_equal(r["eigenvalues"], _FakeResult(r["index"], 45).eigenvalues)

    def failing(i, m, worker):
        if i == 4:
            raise RuntimeError("solve 4 failed")
        return _FakeResult(i, 45)
    import pytest
    with pytest.raises(RuntimeError, match="solve 4 failed"):
        sharding.solve_batch(meshes, failing, 45, None, threads=2)


def test_record_round_trip_and_rank_order_mix():
    r = _FakeResult(3, 45)
    rec = sharding.pack_record(3, r, 65, 0.25)
    back = sharding.unpack_record(rec, 65)
    assert back["index"] == 3 and np.array_equal(back["eigenvalues"], r.eigenvalues) and np.allclose(back["freqs"], r.freqs)
    parts = [np.float32(x) * np.ones(8, np.float32) for x in (1e8, 1.0, -1e8)]
    assert np.array_equal(sharding.mix_partial_signals(parts), (parts[0] + parts[1]) + parts[2])
