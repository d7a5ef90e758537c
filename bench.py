#!/usr/bin/env python3
"""Headline benchmark: eigenpairs/sec of the modal solve (K/M assembly + 50-mode solve) on a 100k-tet mesh.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one mesh already resident in HBM: mh_assemble (FilterDegenerate, BuildQuadMesh,
element bases, K/M assembly) + excitation sampling + mh_eigs (65 eigenpairs requested for 50 kept modes, as the
reference's NumFemModes = NumModes + 15) + shape gather + PostprocessModes + mass properties.  With N ranks every rank
solves its own mesh of the same size (independent objects: weak scaling) and the per-mesh result records are gathered
with one RCCL all_gather per step.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable with a streaming copy


def pmc_traffic():
    """HBM bytes per SpMM launch from the committed rocprofv3 --pmc passes of this command (profiles/, made by
    tools/pmc_traffic.py: FETCH_SIZE doubled per the gfx950 rule and checked on a kernel of known byte count) -- a
    separate profiled run, never this one; None when the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["spmm_family"]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def bank_metric(blocks=72):
    """Secondary figure: the 1024 x 256 resonator bank at 48 kHz through the C++
    mirror's RenderModal, as x real time."""
    try:
        from tools import bank_bench
        return bank_bench.run(blocks=blocks)
    except Exception as e:  # the headline line must not depend on it
        return {"error": str(e)[:200]}


def concurrent_throughput(api, device, pts, tets, mat, ex, cfg, threads=3, per_thread=2):
    """Secondary figure (not `value`): the same mesh solved by several host threads at once, one context each -- the
    reference's "one job per entity, several at a time".  Whole-GPU eigenpairs per second."""
    import threading
    ctxs = [api.Context(device) for _ in range(threads)]
    ms = [api.Mesh(c, pts, tets) for c in ctxs]
    pairs, errs = [0] * threads, []

    def work(k):
        try:
            for _ in range(per_thread):
                pairs[k] += len(api.mesh2modes(ctxs[k], pts, tets, mat, ex, config=cfg, mesh=ms[k]).eigenvalues)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e)[:200])
    t0 = time.perf_counter()
    pool = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    [t.start() for t in pool]
    [t.join() for t in pool]
    [c.synchronize() for c in ctxs]
    dt = time.perf_counter() - t0
    out = {"threads": threads, "meshes": threads * per_thread, "seconds": dt, "eigenpairs_per_s": sum(pairs) / dt, "errors": errs}
    for m_, c in zip(ms, ctxs):
        m_.close()
        c.close()
    return out


def cpu_baseline(seconds_budget=60.0):
    """The CPU oracle (restated reference algorithm: multifrontal Cholesky shift-invert + Lanczos, one thread) on a
    bounded sample of the same workload: the 10k-tet cube with the same 65 requested eigenpairs."""
    from oracle import pyoracle as po
    from mesheditor_amd import meshes
    pts, tets, m, kw = meshes.workload("cube_s10k")
    cfg = po.default_config(num_modes=kw["num_modes"], num_fem_modes=kw["num_fem_modes"])
    ex = pts[:: len(pts) // 10][:10].astype(np.float32)
    t0 = time.perf_counter()
    r = po.mesh2modes(pts, tets, po.material(*m), ex, config=cfg)
    dt = time.perf_counter() - t0
    nev = len(r.eigenvalues)
    return {"value": nev / dt if dt > 0 and nev else 0.0, "unit": "eigenpairs/s", "cores": 1, "kind": "port", "seconds": dt,
            "sample": "cube_s10k: 10,368 tets / 46,875 DOF, 65 eigenpairs, whole mesh2modes path on 1 host thread "
                      "(1/10 of the metric's mesh; the direct solve grows ~quadratically with size)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cube_s100k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if torch.cuda.is_available() else 0

    from mesheditor_amd import api, meshes
    ctx = api.Context(device)
    pts, tets, m, kw = meshes.workload(args.workload)
    if world > 1:  # every rank its own object of the same size: jitter the extents deterministically per rank
        rng = np.random.Generator(np.random.MT19937(1000 + rank))
        pts = pts * rng.uniform(0.9, 1.1, 3)[None, :]
    mat = api.material(*m)
    cfg = api.default_config(num_modes=kw["num_modes"], num_fem_modes=kw["num_fem_modes"])
    ex = pts[:: len(pts) // 10][:10].astype(np.float32)  # P = 10 excitation positions, as the app and bench use
    mesh = api.Mesh(ctx, pts, tets)  # inputs resident in HBM before the timed region

    def sync():
        ctx.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    record = None

    def step():
        nonlocal record
        r = api.mesh2modes(ctx, pts, tets, mat, ex, config=cfg, mesh=mesh)
        if len(r.eigenvalues) == 0:
            raise RuntimeError("solve failed: %s" % r.profile)
        if dist is not None:  # the final gather: fixed-size record per mesh over RCCL
            rec = torch.zeros(256, dtype=torch.float64, device="cuda")
            rec[: len(r.eigenvalues)] = torch.from_numpy(r.eigenvalues).to("cuda")
            out = [torch.empty_like(rec) for _ in range(world)]
            dist.all_gather(out, rec)
            record = out
        return r

    for _ in range(args.warmup):
        step()
    ctx.time_kernels(True)
    sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    sync()
    dt = time.perf_counter() - t0
    stats = ctx.kernel_stats()
    ctx.time_kernels(False)

    nev = len(last.eigenvalues)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_pairs = nev * args.steps * world
    line = {
        "metric": "eigenpairs/sec (K/M assembly + 50-mode solve, 100k-tet mesh)",
        "value": total_pairs / dt,
        "unit": "eigenpairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: Kuhn cube %d tets / %d DOF, Iron, NumModes=%d NumFemModes=%d, P=10 excitation points, one mesh per GPU"
                               % (args.workload, len(tets), last.profile.get("dofs", 0), cfg.num_modes, cfg.num_fem_modes),
                   "eigenpairs_per_mesh": nev, "lobpcg_iterations": last.profile.get("restarts"), "parallelism": "mesh-per-gpu x%d" % world},
    }
    if stats["launches"]:
        achieved = stats["total_bytes"] / (stats["total_ms"] * 1e-3) / 1e9
        line["roofline"] = {"bound": "hbm",
                            "kernel": "k_spmm_wide / k_spmm: BSR 3x3 SpMM of the P2 and P1 operators over n-by-w panels "
                                      "(every launch of the solve: fp32 smoother products, mixed fp64-A x fp32-panel residuals, fp64 operator products)",
                            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
                            "launches": stats["launches"], "avg_launch_us": 1e3 * stats["total_ms"] / stats["launches"],
                            "algorithmic_bytes_per_launch": stats["total_bytes"] / stats["launches"]}
    line["profile"] = {k: last.profile.get(k) for k in ("assemble", "factorize", "iterate", "op_solve", "restarts", "op_applications")}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["concurrent_solves"] = concurrent_throughput(api, device, pts, tets, mat, ex, cfg)
        line["cpu_baseline"] = cpu_baseline()
        line["resonator_bank"] = bank_metric()
    if rank == 0:
        print(json.dumps(line), flush=True)
    mesh.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
