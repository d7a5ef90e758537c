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
    """Deterministic stand-in for a device solve (the GPU is absent here): every field of the section-8e record."""

    def __init__(self, i, nev, n_pos=3):
        rng = np.random.default_rng(100 + i)
        self.eigenvalues = np.sort(rng.uniform(0, 1e9, nev))
        self.freqs = np.sqrt(self.eigenvalues[6:36]).astype(np.float32)
        self.t60s = (1.0 / (1 + self.freqs)).astype(np.float32)
        self.original_fundamental = float(self.freqs[0])
        self.positions = rng.uniform(-1, 1, (n_pos, 3)).astype(np.float32)
        self.summary_shapes = rng.normal(size=(n_pos, nev, 3)).astype(np.float32)
        self.mass = 1.0 + i
        self.center_of_mass = rng.normal(size=3).astype(np.float32)
        self.inertia_diagonal = rng.uniform(1, 2, 3).astype(np.float32)
        self.inertia_orientation_wxyz = np.array([1, 0, 0, 0], np.float32)
        self.profile = {"dofs": 3000 + i, "restarts": 20 + i, "assemble": 0.001 * i, "iterate": 0.5 + i, "op_solve": 0.25, "factorize": 0.1}


def _nev(i):
    return 45 if i % 2 else 65


def _batch(n):
    # uneven costs: tet counts between 1000 and 4000, two eigenpair counts
    return [(None, np.zeros((1000 + 997 * (i * i % 4), 4)), None, {"num_fem_modes": _nev(i)}) for i in range(n)]


def _worker(rank, world, port, q, mode):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if mode == "bank":
            q.put((rank, _bank_rank(dist)))
            return
        meshes = _batch(9 if world == 2 else 14)
        solved = []

        def solve(i, m):
            solved.append(i)
            if mode == "fail" and i == 5:
                raise RuntimeError("solve 5 failed")
            return _FakeResult(i, m[3]["num_fem_modes"])
        try:
            recs = sharding.solve_batch(meshes, solve, 65, dist)
            q.put((rank, solved, [(r["index"], r["dofs"], float(r["eigenvalues"].sum()), len(r["freqs"]), float(np.abs(r["summary_shapes"]).sum()), r["mass"],
                                   float(r["inertia_diagonal"].sum()), r["profile"]["iterate"]) for r in recs]))
        except RuntimeError as e:
            q.put((rank, solved, "raised: %s" % e))
    finally:
        dist.destroy_process_group()


def _spawn(world, mode):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


@pytest.mark.parametrize("world,n", [(2, 9), (4, 14)])
def test_batch_gather_over_ranks(world, n):
    out = _spawn(world, "ok")
    solved = [o[1] for o in out]
    assert sorted(sum(solved, [])) == list(range(n)) and len(set(sum(solved, []))) == n  # every mesh exactly once
    assert all(len(s) > 0 for s in solved)
    costs = [sharding.mesh_cost(len(m[1]), m[3]["num_fem_modes"]) for m in _batch(n)]
    assert [sorted(s) for s in solved] == sharding.lpt_deal(costs, world)  # the deal is the reference's LPT rule
    recs = out[0][2]
    assert all(o[2] == recs for o in out) and [r[0] for r in recs] == list(range(n))  # every rank holds every record
    for i, dofs, evsum, k, shape_sum, mass, inertia, iterate in recs:
        ref = _FakeResult(i, _nev(i))
        assert dofs == 3000 + i and abs(evsum - ref.eigenvalues.sum()) < 1e-6 * evsum and k == 30
        assert abs(shape_sum - float(np.abs(ref.summary_shapes).sum())) < 1e-3 and mass == 1.0 + i
        assert abs(inertia - float(ref.inertia_diagonal.sum())) < 1e-5 and iterate == 0.5 + i


def test_a_failed_solve_does_not_hang_the_other_ranks():
    """One rank's solve raises: every rank still completes the gather and then raises (no rank is left blocked in the
    collective)."""
    out = _spawn(2, "fail")
    assert all(isinstance(o[2], str) and "raised" in o[2] for o in out), out
    assert any("solve 5 failed" in o[2] for o in out) and any("another rank" in o[2] for o in out)


def test_oversized_requests_are_rejected_up_front():
    with pytest.raises(ValueError, match="eigenpairs"):
        sharding.solve_batch([(None, np.zeros((10, 4)), None, {"num_fem_modes": 80})], lambda i, m: None, 65)
    with pytest.raises(ValueError, match="do not fit"):
        sharding.pack_record(0, _FakeResult(0, 45, n_pos=20), 65, 16)


def _bank_objects(n):
    from tests import bank_harness as bh
    return [dict(bh.make_modes(8 + 5 * (o % 4), 0.05 + 0.01 * o, freq_scale=1.0 + 0.01 * o)) for o in range(n)]


def _bank_rank(dist):
    """One rank of the sharded bank with the oracle's CPU bank standing in for the device scene."""
    from oracle import pyoracle as po
    from tests import bank_harness as bh
    objects = _bank_objects(7)

    class OracleSceneAdapter:
        dtype = np.float32

        def __init__(self):
            self.b = po.Bank(bh.SAMPLE_RATE)
            self.b.set_renderers(2)
        add_object = lambda self, *a: self.b.add_object(*a)
        tune_object = lambda self, *a: self.b.tune_object(*a)
        set_gains = lambda self, *a: self.b.set_gains(*a)
        enqueue = lambda self, e: self.b.enqueue(e)
        render = lambda self, out: self.b.render(out)

        def install(self):
            self.b.install()
            self.b.render(np.zeros(bh.BLOCK, np.float32))
    bank = sharding.ShardedBank(objects, OracleSceneAdapter, dist)
    for o in range(len(objects)):
        bank.enqueue(o, lambda slot, o=o: bh.impact_event(po, slot, 1.0 - 0.1 * o, o % 4))
    return np.concatenate([bank.render(bh.BLOCK) for _ in range(3)])


def test_bank_sharded_by_object_mixes_in_rank_order():
    """Two gloo ranks each render their share of the objects (the oracle stands in for the device bank); the gathered,
    rank-ordered mix is the same on both ranks and equals the partials of the same deal added in rank order."""
    from oracle import pyoracle as po
    from tests import bank_harness as bh
    out = _spawn(2, "bank")
    assert np.array_equal(out[0][1], out[1][1]) and np.abs(out[0][1]).max() > 0
    objects = _bank_objects(7)
    deal = sharding.lpt_deal([len(o["freqs"]) for o in objects], 2)
    partials = []
    for share in deal:
        b = po.Bank(bh.SAMPLE_RATE)
        b.set_renderers(2)
        slots = {}
        for o in share:
            slots[o] = b.add_object(o, objects[o]["shapes"], objects[o]["positions"], objects[o]["indices"])
            b.tune_object(slots[o], objects[o]["freqs"], objects[o]["t60s"])
            b.set_gains(slots[o], 1.0, 1.0)
        b.install()
        b.render(np.zeros(bh.BLOCK, np.float32))
        for o in share:
            b.enqueue(bh.impact_event(po, slots[o], 1.0 - 0.1 * o, o % 4))
        sig = np.zeros(3 * bh.BLOCK, np.float32)
        for k in range(3):
            b.render(sig[k * bh.BLOCK:(k + 1) * bh.BLOCK])
        partials.append(sig)
    assert np.array_equal(out[0][1], sharding.mix_partial_signals(partials))


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
    with pytest.raises(RuntimeError, match="solve 4 failed"):
        sharding.solve_batch(meshes, failing, 45, None, threads=2)


def test_record_round_trip_and_rank_order_mix():
    r = _FakeResult(3, 45)
    rec = sharding.pack_record(3, r, 65, 16, 0.25)
    assert len(rec) == sharding.record_length(65, 16)
    back = sharding.unpack_record(rec, 65, 16)
    assert back["index"] == 3 and back["ok"] and np.array_equal(back["eigenvalues"], r.eigenvalues) and np.array_equal(back["freqs"], r.freqs)
    assert np.array_equal(back["summary_shapes"], r.summary_shapes) and np.array_equal(back["positions"], r.positions)
    assert np.array_equal(back["center_of_mass"].astype(np.float32), r.center_of_mass) and back["profile"]["op_solve"] == 0.25
    assert not sharding.unpack_record(sharding.failed_record(3, 65, 16), 65, 16)["ok"]
    parts = [np.float32(x) * np.ones(8, np.float32) for x in (1e8, 1.0, -1e8)]
    assert np.array_equal(sharding.mix_partial_signals(parts), (parts[0] + parts[1]) + parts[2])
