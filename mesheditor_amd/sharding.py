"""Multi-GPU sharding of the path: independent meshes (and independent bank objects) are dealt to ranks and solved
with no data-path collective; one gather of fixed-size result records at the end (RCCL over xGMI on the GPU box,
gloo in the CPU tests).  One process per GPU, torch.distributed for the collective only.

The deal is the reference's own longest-processing-time rule for objects -> renderers
(src/audio/ModalAudio.cpp:450-458): heaviest first, ties by index, onto the least-loaded rank (lowest rank on ties);
each rank then processes its items in ascending order.
"""
import numpy as np


def lpt_deal(costs, n_ranks):
    """-> list (per rank) of item indices, ascending within a rank."""
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * n_ranks
    out = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        load[r] += int(costs[i])
        out[r].append(i)
    return [sorted(x) for x in out]


def mesh_cost(n_tets, nev):
    return int(n_tets) * int(nev)


RECORD_HEADER = 8  # index, nev, n_modes, dofs, iterations, seconds, mass, reserved


def pack_record(index, result, nev_max, seconds=0.0):
    """Fixed-size float64 record of one solved mesh: header, eigenvalues[nev_max], freqs[nev_max], t60s[nev_max]."""
    rec = np.zeros(RECORD_HEADER + 3 * nev_max)
    nev, k = len(result.eigenvalues), len(result.freqs)
    rec[:RECORD_HEADER] = [index, nev, k, result.profile.get("dofs", 0), result.profile.get("restarts", 0), seconds, result.mass, 0.0]
    rec[RECORD_HEADER:RECORD_HEADER + nev] = result.eigenvalues
    rec[RECORD_HEADER + nev_max:RECORD_HEADER + nev_max + k] = result.freqs
    rec[RECORD_HEADER + 2 * nev_max:RECORD_HEADER + 2 * nev_max + k] = result.t60s
    return rec


def unpack_record(rec, nev_max):
    nev, k = int(rec[1]), int(rec[2])
    return {"index": int(rec[0]), "dofs": int(rec[3]), "iterations": int(rec[4]), "seconds": float(rec[5]), "mass": float(rec[6]),
            "eigenvalues": rec[RECORD_HEADER:RECORD_HEADER + nev].copy(),
            "freqs": rec[RECORD_HEADER + nev_max:RECORD_HEADER + nev_max + k].copy(),
            "t60s": rec[RECORD_HEADER + 2 * nev_max:RECORD_HEADER + 2 * nev_max + k].copy()}


def gather_records(local_records, n_items, dist=None, device="cpu"):
    """All ranks end with every item's record, ordered by item index.  local_records: {item index: record}."""
    import torch
    reclen = len(next(iter(local_records.values()))) if local_records else 0
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_records[i] for i in sorted(local_records)]
    world = dist.get_world_size()
    meta = torch.tensor([len(local_records), reclen], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    max_items = max(int(m[0]) for m in metas)
    reclen = max(int(m[1]) for m in metas)
    buf = torch.zeros((max_items, reclen + 1), dtype=torch.float64, device=device)
    for k, i in enumerate(sorted(local_records)):
        buf[k, 0] = 1.0
        buf[k, 1:] = torch.from_numpy(np.asarray(local_records[i], dtype=np.float64)).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)  # the one collective of the path
    records = {}
    for b in bufs:
        b = b.cpu().numpy()
        for row in b:
            if row[0] == 1.0:
                records[int(row[1])] = row[1:].copy()
    assert len(records) == n_items, (len(records), n_items)
    return [records[i] for i in sorted(records)]


def solve_batch(meshes, solve_fn, nev_max, dist=None, device="cpu", threads=1):
    """meshes: list of (points, tets, material tuple, config kwargs).  solve_fn(index, mesh tuple) -> ModalResult-like
    (with threads > 1: solve_fn(index, mesh tuple, worker) and `threads` host threads per rank, worker = 0 .. threads - 1,
    each meant to use its own context: solves of different contexts iterate side by side on one GPU, the reference's
    "one job per entity, several at a time").  Deals the batch over the ranks by cost, solves the local share, gathers
    every record to every rank."""
    import time
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    costs = [mesh_cost(len(m[1]), m[3].get("num_fem_modes", 45)) for m in meshes]
    mine = lpt_deal(costs, world)[rank]
    local = {}
    if threads <= 1:
        for i in mine:
            t0 = time.perf_counter()
            res = solve_fn(i, meshes[i])
            local[i] = pack_record(i, res, nev_max, time.perf_counter() - t0)
    else:
        import threading
        # the rank's share is dealt again over its workers by the same rule (largest first)
        shares = lpt_deal([costs[i] for i in mine], threads)
        errors = []

        def work(worker):
            try:
                for j in shares[worker]:
                    i = mine[j]
                    t0 = time.perf_counter()
                    res = solve_fn(i, meshes[i], worker)
                    local[i] = pack_record(i, res, nev_max, time.perf_counter() - t0)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
        pool = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        [t.start() for t in pool]
        [t.join() for t in pool]
        if errors:
            raise errors[0]
    return [unpack_record(r, nev_max) for r in gather_records(local, len(meshes), dist, device)]


def mix_partial_signals(partials):
    """Bank sharded by object: each rank renders a partial mix; the final signal adds them in rank order, so the sum
    does not depend on arrival order (an all-reduce would)."""
    out = np.zeros_like(partials[0])
    for p in partials:
        out += p
    return out
