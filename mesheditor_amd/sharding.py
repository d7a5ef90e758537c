"""Multi-GPU sharding of the path: independent meshes (and independent bank objects) are dealt to ranks and processed
with no data-path collective; one gather of fixed-size records at the end of a batch (RCCL over xGMI on the GPU box,
gloo in the CPU tests), one gather of per-rank partial signals per audio block.  One process per GPU,
torch.distributed for the collectives only.  A single solve is never split across GPUs.

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


# ---- the per-mesh record: everything a ModalResult holds, at a fixed size ----------------------
# float64 words: header | MassProperties | SolveProfile | eigenvalues[nev_max] | freqs[nev_max] | t60s[nev_max] |
#                positions[pos_max x 3] | summary shapes[pos_max x nev_max x 3] (eigenvector rows at the sample points)
HEADER = ("index", "status", "nev", "n_modes", "n_pos", "seconds", "original_fundamental", "reserved")
MASS = 11  # mass, centre of mass (3), inertia diagonal (3), inertia orientation w x y z (4)
PROFILE = ("mass_props", "quad_mesh", "assemble", "sample_excite", "factorize", "iterate", "op_solve", "extract", "dofs", "stiffness_nonzeros",
           "op_applications", "restarts")
STATUS_OK, STATUS_FAILED = 0.0, 1.0


def record_length(nev_max, pos_max):
    return len(HEADER) + MASS + len(PROFILE) + 3 * nev_max + 3 * pos_max + 3 * pos_max * nev_max


def pack_record(index, result, nev_max, pos_max, seconds=0.0):
    """One solved mesh as a fixed-size float64 record.  `result` is an api.ModalResult (or anything with its fields)."""
    nev, k = len(result.eigenvalues), len(result.freqs)
    shapes = np.asarray(result.summary_shapes, np.float64).reshape(-1, nev, 3) if nev else np.zeros((0, 0, 3))
    positions = np.asarray(result.positions, np.float64).reshape(-1, 3)
    n_pos = len(positions)
    if nev > nev_max or n_pos > pos_max:
        raise ValueError(f"mesh {index}: {nev} eigenpairs / {n_pos} sample points do not fit a record of {nev_max} / {pos_max}")
    rec = np.zeros(record_length(nev_max, pos_max))
    rec[:len(HEADER)] = [index, STATUS_OK, nev, k, n_pos, seconds, float(getattr(result, "original_fundamental", 0.0)), 0.0]
    o = len(HEADER)
    rec[o] = result.mass
    rec[o + 1:o + 4] = result.center_of_mass
    rec[o + 4:o + 7] = result.inertia_diagonal
    rec[o + 7:o + 11] = result.inertia_orientation_wxyz
    o += MASS
    rec[o:o + len(PROFILE)] = [float(result.profile.get(name, 0.0)) for name in PROFILE]
    o += len(PROFILE)
    rec[o:o + nev] = result.eigenvalues
    rec[o + nev_max:o + nev_max + k] = result.freqs
    rec[o + 2 * nev_max:o + 2 * nev_max + k] = result.t60s
    o += 3 * nev_max
    rec[o:o + 3 * n_pos] = positions.reshape(-1)
    o += 3 * pos_max
    block = rec[o:o + 3 * pos_max * nev_max].reshape(pos_max, nev_max, 3)
    block[:n_pos, :nev] = shapes
    return rec


def failed_record(index, nev_max, pos_max, seconds=0.0):
    """Placeholder of a mesh whose solve raised: the rank still takes part in the gather; the failure is raised after it."""
    rec = np.zeros(record_length(nev_max, pos_max))
    rec[0], rec[1], rec[5] = index, STATUS_FAILED, seconds
    return rec


def unpack_record(rec, nev_max, pos_max):
    head = dict(zip(HEADER, rec[:len(HEADER)]))
    nev, k, n_pos = int(head["nev"]), int(head["n_modes"]), int(head["n_pos"])
    o = len(HEADER)
    out = {"index": int(head["index"]), "ok": head["status"] == STATUS_OK, "seconds": float(head["seconds"]), "original_fundamental": float(head["original_fundamental"]),
           "mass": float(rec[o]), "center_of_mass": rec[o + 1:o + 4].copy(), "inertia_diagonal": rec[o + 4:o + 7].copy(), "inertia_orientation_wxyz": rec[o + 7:o + 11].copy()}
    o += MASS
    out["profile"] = dict(zip(PROFILE, rec[o:o + len(PROFILE)].tolist()))
    out["dofs"], out["iterations"] = int(out["profile"]["dofs"]), int(out["profile"]["restarts"])
    o += len(PROFILE)
    out["eigenvalues"] = rec[o:o + nev].copy()
    out["freqs"] = rec[o + nev_max:o + nev_max + k].astype(np.float32)
    out["t60s"] = rec[o + 2 * nev_max:o + 2 * nev_max + k].astype(np.float32)
    o += 3 * nev_max
    out["positions"] = rec[o:o + 3 * n_pos].reshape(n_pos, 3).astype(np.float32)
    o += 3 * pos_max
    out["summary_shapes"] = rec[o:o + 3 * pos_max * nev_max].reshape(pos_max, nev_max, 3)[:n_pos, :nev].astype(np.float32)
    return out


def gather_records(local_records, n_items, dist=None, device="cpu"):
    """All ranks end with every item's record, ordered by item index.  local_records: {item index: record}.  Every
    rank must call this exactly once per batch, whatever happened to its solves."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_records[i] for i in sorted(local_records)]
    world = dist.get_world_size()
    reclen = len(next(iter(local_records.values()))) if local_records else 0
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
    dist.all_gather(bufs, buf)  # the one collective of the analysis path
    records = {}
    for b in bufs:
        for row in b.cpu().numpy():
            if row[0] == 1.0:
                records[int(row[1])] = row[1:].copy()
    if len(records) != n_items:
        raise RuntimeError(f"gathered {len(records)} records for {n_items} meshes")
    return [records[i] for i in sorted(records)]


def solve_batch(meshes, solve_fn, nev_max, dist=None, device="cpu", threads=1, pos_max=16):
    """meshes: list of (points, tets, material tuple, config kwargs).  solve_fn(index, mesh tuple) -> ModalResult-like
    (with threads > 1: solve_fn(index, mesh tuple, worker) and `threads` host threads per rank, worker = 0 .. threads - 1,
    each meant to use its own context: solves of different contexts iterate side by side on one GPU, the reference's
    "one job per entity, several at a time").  Deals the batch over the ranks by cost, solves the local share, gathers
    every record to every rank.  A solve that raises does not keep its rank out of the gather: it contributes a failed
    record and the first failure (lowest mesh index, on every rank alike) is raised once the collective has completed."""
    import time
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    for i, m in enumerate(meshes):
        if m[3].get("num_fem_modes", 45) > nev_max:
            raise ValueError(f"mesh {i} asks for {m[3]['num_fem_modes']} eigenpairs but records hold {nev_max}")
    costs = [mesh_cost(len(m[1]), m[3].get("num_fem_modes", 45)) for m in meshes]
    mine = lpt_deal(costs, world)[rank]
    local, errors = {}, {}

    def run(i, *extra):
        t0 = time.perf_counter()
        try:
            local[i] = pack_record(i, solve_fn(i, meshes[i], *extra), nev_max, pos_max, time.perf_counter() - t0)
        except Exception as e:  # noqa: BLE001 -- reported after the gather
            errors[i] = e
            local[i] = failed_record(i, nev_max, pos_max, time.perf_counter() - t0)
    if threads <= 1:
        for i in mine:
            run(i)
    else:
        import threading
        shares = lpt_deal([costs[i] for i in mine], threads)  # the rank's share, dealt again over its workers

        def work(worker):
            for j in shares[worker]:
                run(mine[j], worker)
        pool = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        [t.start() for t in pool]
        [t.join() for t in pool]
    records = [unpack_record(r, nev_max, pos_max) for r in gather_records(local, len(meshes), dist, device)]
    failed = [r["index"] for r in records if not r["ok"]]
    if failed:
        first = failed[0]
        if first in errors:
            raise errors[first]
        raise RuntimeError(f"solve of mesh {first} failed on another rank ({len(failed)} of {len(meshes)} meshes failed)")
    return records


# ---- the resonator bank, sharded by object ---------------------------------------------------------------------------
def mix_partial_signals(partials):
    """Rank-ordered mix: the final block adds the ranks' partial blocks in rank order, so the sum does not depend on
    arrival order (an all-reduce would reorder the floating-point adds)."""
    out = np.zeros_like(partials[0])
    for p in partials:
        out += p
    return out


class ShardedBank:
    """A bank whose objects are dealt over the ranks by mode count with the reference's LPT rule (ModalAudio.cpp:450-458
    applied across devices instead of render threads).  Every rank builds a scene holding only its own objects
    (ascending), renders its partial block, and the block is completed by ONE all_gather of the partial signals followed
    by the rank-ordered mix -- on the device when the scenes render there.  Events are routed to the owning rank.

    scene_factory() -> an object with add_object / tune_object / set_gains / install / enqueue / render and a numpy
    `dtype` (mesheditor_amd.bank.Scene on the GPU box; the CPU tests pass the oracle's bank)."""

    def __init__(self, objects, scene_factory, dist=None, device="cpu"):
        self.dist = dist if dist is not None and dist.is_initialized() and dist.get_world_size() > 1 else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.device = device
        self.owner, self.local_slot = {}, {}
        deal = lpt_deal([len(o["freqs"]) for o in objects], self.world)
        for r, share in enumerate(deal):
            for o in share:
                self.owner[o] = r
        self.scene = scene_factory()
        for o in deal[self.rank]:
            obj = objects[o]
            slot = self.scene.add_object(o, obj["shapes"], obj["positions"], obj["indices"])
            self.scene.tune_object(slot, obj["freqs"], obj["t60s"])
            self.scene.set_gains(slot, obj.get("out_gain", 1.0), obj.get("listener_gain", 1.0))
            self.local_slot[o] = slot
        self.scene.install()
        self.dtype = getattr(self.scene, "dtype", np.float32)

    def enqueue(self, obj, make_event):
        """make_event(local slot) -> event; only the owner queues it.  True everywhere so that callers stay in lockstep."""
        if self.owner[obj] == self.rank:
            return self.scene.enqueue(make_event(self.local_slot[obj]))
        return True

    def render(self, frames):
        """One block: the local partial, then the gather + rank-ordered mix.  Returns the complete block on every rank."""
        partial = np.zeros(frames, self.dtype)
        self.scene.render(partial)
        if not self.dist:
            return partial
        import torch
        mine = torch.from_numpy(partial).to(self.device)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)  # the one collective of a block: world x frames samples
        out = torch.zeros_like(mine)
        for p in parts:  # rank order, on the device the partials arrived on
            out += p
        return out.cpu().numpy()
