#!/usr/bin/env python3
"""Headline benchmark: eigenpairs/sec of the modal solve (K/M assembly + 50-mode solve) on a 100k-tet mesh.

    python bench.py --gpus N --steps K --warmup W                (N > 1 without a launcher: starts N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one mesh already resident in HBM: mh_assemble (FilterDegenerate, BuildQuadMesh,
element bases, K/M assembly) + excitation sampling + mh_eigs (65 eigenpairs requested for 50 kept modes, as the
reference's NumFemModes = NumModes + 15) + shape gather + PostprocessModes + mass properties.  With N ranks every rank
solves its own mesh of the same size (independent objects: weak scaling) and the per-mesh result records -- the whole
SURVEY 8e record: eigenvalues, frequencies, decay times, shapes, mass properties, solve profile -- are gathered with one
RCCL all_gather per step.  `--workload batch64` runs BASELINE config 4 instead: 64 jittered 29k-tet boxes dealt over the
ranks by cost (8 per GPU at N = 8), three solves in flight per GPU, one gather per pass.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable with a streaming copy
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X data sheet, fp64 matrix (the guide has no fp64 row)
NEV_MAX, POS_MAX = 256, 16  # record capacity of the gather


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never will) and leave with its exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


PMC_WORKLOAD = "cube_s100k"  # the workload the committed --pmc passes ran (the default command line of this file)


def pmc_traffic(family="spmm_family", workload=PMC_WORKLOAD):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc passes of this command (profiles/, made
    by tools/pmc_traffic.py: FETCH_SIZE doubled per the gfx950 rule and checked on a kernel of known byte count) -- a
    separate profiled run, never this one; None when the summary is absent, and None for any workload other than the one
    those passes measured (a per-launch figure of the cube says nothing about a batch of 30k-tet meshes)."""
    if workload != PMC_WORKLOAD:
        return None
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return float(json.load(f)[family]["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            continue
    return None


def bank_metric(blocks=72):
    """Secondary figure: the 1024 x 256 resonator bank at 48 kHz through the C++
    mirror's RenderModal -- all-live and steady-state phases, the resonator kernel's share of the fp32 vector peak."""
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


def operator_forms(api, ctx, mesh, mat, widths=(1, 16, 64, 80)):
    """One fp64 product y = (K - sigma M) x over an n x w panel of the metric's mesh (w = 1 is the north star's literal SpMV), in
    both forms that exist: the BSR SpMM the solver uses and the matrix-free element-by-element product (csrc/mh_elem.hip,
    atomic scatter).  Fractions of the 8 TB/s roofline on two byte counts, both from SURVEY 8(d): the BSR bytes the kernel
    moves (76 B per node block + 4 B per row pointer + 16 B per panel entry; `roofline.frac` uses this one) and the canonical
    scalar-CSR count the 40 % target is quoted on (12 B per non-zero + 4 B per row pointer + 16 B per panel entry)."""
    from tools import lab  # libmodalhip_lab.so: timing loops outside the path's ABI
    system = api.System(ctx, mesh, mat)
    n_nodes, n_blocks = system.node_count, system.node_blocks
    out = []
    for w in widths:
        ms, by = lab.bench_spmm(system, w, 10)
        em = lab.bench_elementwise(system, w, 10)
        csr = 12.0 * 9 * n_blocks + 4.0 * (3 * n_nodes + 1) + 16.0 * 3 * n_nodes * w
        out.append({"w": w, "algorithmic_bytes": by, "canonical_csr_bytes": csr, "bsr_us": 1e3 * ms, "bsr_frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "bsr_frac_on_csr_bytes": csr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "elementwise_us": 1e3 * em, "elementwise_frac": by / (em * 1e-3) / 1e9 / HBM_PEAK_GBS})
    system.close()
    return out


def cpu_baseline(workload="cube_s10k"):
    """The CPU oracle (restated reference algorithm: nested-dissection multifrontal Cholesky shift-invert + restarted
    Lanczos) on a bounded sample of the metric's workload, timed on this box's host cores: once on one thread (the
    reference runs one solve per job thread) and once with an OpenMP team on every core, stage by stage as the
    reference's SolveProfile."""
    from oracle import pyoracle as po
    from mesheditor_amd import meshes
    pts, tets, m, kw = meshes.workload(workload)
    cfg = po.default_config(num_modes=kw["num_modes"], num_fem_modes=kw["num_fem_modes"])
    ex = pts[:: len(pts) // 10][:10].astype(np.float32)
    cores = po.available_cores()  # affinity mask cut to the cgroup CPU quota (the GPU box: 256 logical CPUs, quota 16)
    team = min(cores, 16)  # the oracle's loops are front- and panel-sized: it stops scaling at 8-16 threads
    runs = {}
    for threads in (1, team) if team > 1 else (1,):
        po.set_threads(threads)
        t0 = time.perf_counter()
        r = po.mesh2modes(pts, tets, po.material(*m), ex, config=cfg)
        dt = time.perf_counter() - t0
        nev = len(r.eigenvalues)
        runs[threads] = {"threads": threads, "seconds": dt, "eigenpairs_per_s": nev / dt if dt > 0 and nev else 0.0, "eigenpairs": nev,
                         "stages_s": {k: r.profile[k] for k in ("mass_props", "quad_mesh", "assemble", "sample_excite", "factorize", "iterate", "op_solve", "extract")}}
    po.set_threads(1)
    best = runs[max(runs)]
    return {"value": best["eigenpairs_per_s"], "unit": "eigenpairs/s", "cores": best["threads"], "kind": "port", "host_cores": cores,
            "sample": "%s: %d tets / %d DOF, %d eigenpairs, whole mesh2modes path (1/10 of the metric's mesh, timed live on this box; the metric's own mesh: "
                      "cpu_baseline_metric_mesh, a recorded run)" % (workload, len(tets), int(r.profile.get("dofs", 0)), best["eigenpairs"]),
            "single_thread": runs[1], "threaded": runs[max(runs)],
            "logical_cpus": os.cpu_count(),
            "note": "host_cores = CPUs this container may use (affinity and cgroup quota); threaded row: OpenMP team of min(host_cores, 16)"}


def cpu_baseline_metric_mesh_live(workload="cube_s100k"):
    """`--cpu-baseline-metric-mesh`: the oracle on the METRIC'S OWN mesh, timed on THIS box's host cores (about ten minutes on
    the 16 cores the GPU box grants).  Opt-in; the record it prints is committed as profiles/r04_cpu_baseline_metric_mesh.json
    and the default line cites it."""
    import datetime
    import platform
    from oracle import pyoracle as po
    from mesheditor_amd import meshes
    pts, tets, m, kw = meshes.workload(workload)
    cores = po.available_cores()
    team = min(cores, 16)
    po.set_threads(team)
    cfg = po.default_config(num_modes=kw["num_modes"], num_fem_modes=kw["num_fem_modes"])
    ex = pts[:: len(pts) // 10][:10].astype(np.float32)
    t0 = time.perf_counter()
    r = po.mesh2modes(pts, tets, po.material(*m), ex, config=cfg)
    dt = time.perf_counter() - t0
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": len(r.eigenvalues) / dt, "unit": "eigenpairs/s", "cores": team, "kind": "port", "workload": workload, "seconds": dt,
            "eigenpairs": int(len(r.eigenvalues)), "mesh": {"tets": int(len(tets)), "dof": int(r.profile["dofs"])},
            "host": {"cpu": cpu or platform.processor(), "cores_available": cores, "logical_cpus": os.cpu_count(), "threads": team, "node": platform.node()},
            "date": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d"), "command": "python bench.py --cpu-baseline-metric-mesh",
            "stages_s": {k: r.profile[k] for k in ("mass_props", "quad_mesh", "assemble", "sample_excite", "factorize", "iterate", "op_solve", "extract")},
            "first_eigenvalues": [float(v) for v in r.eigenvalues[6:12]],
            "measured": "live on the GPU box's host cores (oracle = restated reference algorithm, OpenMP team of min(host cores, 16))"}


def cpu_baseline_metric_mesh(workload="cube_s100k"):
    """The oracle on the METRIC'S OWN mesh -- the same-input partner of `value`.  The run takes ten minutes, the default bench
    may not, so this reads a committed record: profiles/r04_cpu_baseline_metric_mesh.json (timed on a GPU box's host cores by
    `bench.py --cpu-baseline-metric-mesh` through gpurun) when it describes this workload, else the build container's record in
    tests/golden/oracle_eigs_<workload>.json.  The live `cpu_baseline` stays the on-this-box figure on the 10k-tet sample."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_cpu_baseline_metric_mesh.json")) as f:
            d = json.load(f)
        if d.get("workload") == workload:
            d["measured"] = "recorded run on a GPU box of this pool (same host type as this run), not timed in this invocation"
            d["record"] = "profiles/r04_cpu_baseline_metric_mesh.json"
            return d
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(ROOT, "tests", "golden", "oracle_eigs_%s.json" % workload)) as f:
            d = json.load(f)
        return {"value": d["eigenpairs_per_second"], "unit": "eigenpairs/s", "cores": d["host"]["threads"], "kind": "port", "workload": workload,
                "seconds": d["seconds"], "eigenpairs": len(d["eigenvalues"]), "mesh": d["mesh"], "host": d["host"], "date": d["date"], "command": d["generator"],
                "stages_s": {k: d["profile"][k] for k in ("mass_props", "quad_mesh", "assemble", "sample_excite", "factorize", "iterate", "op_solve", "extract")},
                "measured": "recorded run (build container), not this box", "record": "tests/golden/oracle_eigs_%s.json" % workload}
    except (OSError, KeyError, ValueError) as e:
        return {"error": str(e)[:200]}


def edit_loop(api, ctx, pts, tets, m, ex, cfg, mesh):
    """The reference bench's synthetic interactive-edit loop (tests/ModalSolverBench.cpp:346-411) on the metric's mesh: one cold
    solve keeping the basis; a Poisson-ratio edit solved cold and warm-started from that basis (same kept-mode count and
    fundamental within 0.05 Hz, the reference's criterion); a Young's-modulus-and-density edit solved cold against
    RescaleModes of the first solve's summary (no eigensolve)."""
    def timed(mat, **kw):
        t0 = time.perf_counter()
        r = api.mesh2modes(ctx, pts, tets, mat, ex, config=cfg, mesh=mesh, **kw)
        ctx.synchronize()
        return r, time.perf_counter() - t0
    try:
        initial, t_initial = timed(api.material(*m), keep_basis=True)
        nu = (m[0], m[1], min(m[2] + 0.02, 0.49), m[3], m[4])
        cold, t_cold = timed(api.material(*nu))
        warm, t_warm = timed(api.material(*nu), seed_basis=initial.basis)
        scaled = (m[0] * 0.8, m[1] * 1.5, m[2], m[3], m[4])
        cold2, t_cold2 = timed(api.material(*scaled))
        t0 = time.perf_counter()
        resc = api.rescale_modes(initial.eigenvalues, initial.summary_shapes, api.material(*m), api.material(*scaled), cfg)
        t_resc = time.perf_counter() - t0
        f1 = lambda fr: float(fr[0]) if len(fr) else 0.0  # noqa: E731
        return {"initial_cold_ms": 1e3 * t_initial,
                "nu_edit": {"cold_ms": 1e3 * t_cold, "warm_ms": 1e3 * t_warm, "speedup": t_cold / t_warm, "cold_iterations": cold.profile.get("restarts"),
                            "warm_iterations": warm.profile.get("restarts"), "modes": [len(cold.freqs), len(warm.freqs)], "f1_hz": [f1(cold.freqs), f1(warm.freqs)],
                            "match": len(cold.freqs) == len(warm.freqs) and abs(f1(cold.freqs) - f1(warm.freqs)) < 0.05},
                "e_rho_edit": {"cold_ms": 1e3 * t_cold2, "rescale_ms": 1e3 * t_resc, "speedup": t_cold2 / max(t_resc, 1e-9), "modes": [len(cold2.freqs), len(resc[0]) if resc else 0],
                               "f1_hz": [f1(cold2.freqs), f1(resc[0]) if resc else 0.0],
                               "match": resc is not None and len(cold2.freqs) == len(resc[0]) and abs(f1(cold2.freqs) - f1(resc[0])) < 0.05}}
    except Exception as e:  # noqa: BLE001 -- the headline line must not depend on it
        return {"error": repr(e)[:200]}


def scan_like_rows(api, ctx, names=("ball_s10k", "uvsphere_s10k", "cube_s30k", "scan_s30k", "scan_s100k", "scan_s30k_interior", "scan_s100k_interior", "scan_s30k_repaired", "scan_s100k_repaired"), reps=2):
    """Secondary rows: BASELINE config 2 (the Kuhn-mapped ball of rounds 1-3 and the UV-sphere primitive itself through the front end), then
    the scan-like unstructured meshes (marching-tetrahedra skillet surface through the path's own
    tetrahedraliser: slivers, 2-60 tets per node, no interior points) beside the Kuhn grid of the same size -- iterations,
    milliseconds and eigenpairs per second of the whole mesh2modes path, 65 pairs each.  "_interior": recovery points moved off
    the surface; "_repaired": the front end's default since round 4 (that plus sliver repair and smoothing of the added points)."""
    from mesheditor_amd import meshes
    out = []
    for name in names:
        try:
            pts, tets, m, kw = meshes.workload(name)
            ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
            cfg = api.default_config(**kw)
            mesh = api.Mesh(ctx, pts, tets)
            ts, r = [], None
            for _ in range(reps + 1):
                t0 = time.perf_counter()
                r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=cfg, mesh=mesh)
                ctx.synchronize()
                ts.append(time.perf_counter() - t0)
            mesh.close()
            dt = float(np.median(ts[1:]))
            out.append({"workload": name, "tets": int(len(tets)), "dof": int(r.profile.get("dofs", 0)), "eigenpairs": int(len(r.eigenvalues)),
                        "lobpcg_iterations": int(r.profile.get("restarts", 0)), "ms": 1e3 * dt, "eigenpairs_per_s": len(r.eigenvalues) / dt if dt > 0 else 0.0})
        except Exception as e:  # noqa: BLE001
            out.append({"workload": name, "error": repr(e)[:200]})
    return out


def config3_rows(api, ctx):
    """BASELINE config 3 as written -- a scanned ~100k-tet mesh, 200 modes (215 pairs) -- at the metric's size and at the
    RealImpact-true size, beside the Kuhn plate that stood in for it in rounds 1-3: iterations, ms, eigenpairs per second."""
    return scan_like_rows(api, ctx, names=("config3_s30k", "config3_s100k", "config3_s30k_repaired", "config3_s100k_repaired", "skillet_s100k"), reps=1)


def batch64_pass(api, device, threads=3, scans=False):
    """BASELINE config 4 on the GPU(s) of this run's first rank: the 64 jittered 29k-tet boxes solved by `threads` host threads
    (one context each), one pass after a warm-up solve; N = 1 here -- the multi-GPU form is `--workload batch64 --gpus N`.
    scans: the same with 64 jittered scan-like thin-walled fills (RealImpact's shape) instead of Kuhn boxes."""
    from mesheditor_amd import sharding
    try:
        items = batch_scan_meshes() if scans else batch_meshes()
        ctxs = [api.Context(device) for _ in range(threads)]
        ex_of = [m[0][:: len(m[0]) // 10][:10].astype(np.float32) for m in items]

        def solve(i, m, worker=0):
            return api.mesh2modes(ctxs[worker], m[0], m[1], api.material(*m[2]), ex_of[i], config=api.default_config(**m[3]))
        solve(0, items[0])
        [c.synchronize() for c in ctxs]
        t0 = time.perf_counter()
        recs = sharding.solve_batch(items, solve, 64, None, "cpu", threads=threads, pos_max=POS_MAX)  # (45 pairs per mesh: records of 64, not of NEV_MAX)
        [c.synchronize() for c in ctxs]
        dt = time.perf_counter() - t0
        [c.close() for c in ctxs]
        pairs = sum(len(r["eigenvalues"]) for r in recs)
        what = ("64 jittered scan-like thin-walled fills (8 skillet scan surfaces through the front end's default options, stretched by U(0.8, 1.25) per axis) of %d-%d tets"
                % (min(len(m[1]) for m in items), max(len(m[1]) for m in items))) if scans else "64 jittered Kuhn boxes of %d tets" % len(items[0][1])
        return {"workload": what + ", 7 materials cycled, 45 eigenpairs each, mesh upload included", "n_gpus": 1,
                "threads_per_gpu": threads, "seconds": dt, "eigenpairs_per_s": pairs / dt, "meshes_per_s": len(items) / dt,
                "iterations_mean": float(np.mean([r["iterations"] for r in recs]))}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}


def cpu_bank_baseline(blocks=2):
    """The oracle's bank (the reference's RenderObjectFast loop, fp32) on config 5 with every mode live: seconds per
    512-frame block and RenderShare (= seconds x SR / frames; > 1 underruns) at 1 and 4 renderers (the reference's
    default pool, AudioTypes.h:26)."""
    try:
        from oracle import pyoracle as po
        from tools import bank_bench as bb
        out = {}
        for renderers in (1, 4):
            po.set_threads(renderers)  # one OpenMP thread per renderer, as the reference's render pool
            b = po.Bank(bb.SR)
            b.set_renderers(renderers)
            pos = np.array([[p * 0.01, 0.0, 0.02 if p % 2 else 0.0] for p in range(bb.POINTS)], np.float32)
            idx = np.array([[p, p + 1, p + 2] for p in range(bb.POINTS - 2)], np.uint32).reshape(-1)
            for o in range(1024):
                f, t, sh = bb.modes_for(o, 256)
                s = b.add_object(o, sh, pos, idx)
                b.tune_object(s, f, t)
                b.set_gains(s, 1.0, 1.0)
            b.install()
            b.set_max_impacts(4096)
            buf = np.zeros(bb.BLOCK, np.float32)
            b.render(buf)
            step = np.float32(1.0 / (4 * bb.BLOCK))
            times = []
            for blk in range(4 + blocks):  # four blocks to strike everything (256 events per block), then timed all-live blocks
                for o in range((blk % 4) * 256, (blk % 4 + 1) * 256):
                    b.enqueue(po.Event(0, o, 0, 1.0, 0.5, 0.0, step, 2 * step, 0.0, 0.0, 0.0, 0.0))
                t0 = time.perf_counter()
                b.render(buf)
                times.append(time.perf_counter() - t0)
            sec = float(np.mean(times[4:]))
            out["renderers_%d" % renderers] = {"seconds_per_block": sec, "render_share": sec * bb.SR / bb.BLOCK, "x_real_time": bb.BLOCK / bb.SR / sec}
        out["sample"] = "oracle bank 1024x256 @48k, all 262,144 modes live, %d timed 512-frame blocks" % blocks
        po.set_threads(min(po.available_cores(), 16))
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}


def batch_meshes(count=64, n=17):
    """BASELINE config 4: `count` jittered boxes of the RealImpact size (29,478 tets), materials cycled, 45 eigenpairs."""
    from mesheditor_amd import meshes
    out = []
    for i in range(count):
        p, t = meshes.jittered_box(n, 1000 + i)
        out.append((p, t, meshes.MATERIALS[meshes.MATERIAL_ORDER[i % len(meshes.MATERIAL_ORDER)]], {"num_modes": 30, "num_fem_modes": 45}))
    return out


def batch_scan_meshes(count=64):
    """config 4's batch on RealImpact's SHAPE -- `count` jittered scan-like thin-walled fills of ~30k tets
    (meshes.jittered_scan), materials cycled, 45 eigenpairs."""
    from mesheditor_amd import meshes
    out = []
    for i in range(count):
        p, t = meshes.jittered_scan(i)
        out.append((p, t, meshes.MATERIALS[meshes.MATERIAL_ORDER[i % len(meshes.MATERIAL_ORDER)]], {"num_modes": 30, "num_fem_modes": 45}))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cube_s100k", help="a meshes.workload name (one mesh per rank per step) or batch64 (BASELINE config 4)")
    ap.add_argument("--threads", type=int, default=3, help="batch64: solves in flight per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-metric-mesh", action="store_true", help="only time the CPU oracle on the metric's own mesh on this box (~10 min) and print that record")
    args = ap.parse_args()
    if args.cpu_baseline_metric_mesh:  # no GPU work at all
        print(json.dumps(cpu_baseline_metric_mesh_live(args.workload)), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))  # before anything here touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    share_gpu = bool(os.environ.get("BENCH_SHARE_GPU"))  # test mode: every rank on device 0, gloo instead of RCCL
    import torch
    dist = None
    # The data path's one collective -- the gather of the per-mesh records -- is ncclAllGather called from C++
    # (modal::SolveBatch, mesheditor_amd/cpp/src/batch.cpp) whenever the ranks sit on different GPUs; torch.distributed (gloo)
    # only launches the ranks' rendezvous: it ships the 128-byte communicator id and carries the timing barriers.
    # BENCH_GATHER=torch keeps the records' gather in torch.distributed (RCCL through torch); BENCH_SHARE_GPU (all ranks on
    # device 0, a test mode) always does, over gloo.
    cpp_driver = (world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))) and not share_gpu and os.environ.get("BENCH_GATHER", "rccl") == "rccl"
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # BENCH_FORCE_DIST: exercise the distributed path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share_gpu or cpp_driver:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = 0 if share_gpu or not torch.cuda.is_available() else local_rank
    gather_device = "cpu" if share_gpu or dist is None or cpp_driver else "cuda"

    from mesheditor_amd import api, meshes, sharding
    comm = None
    if cpp_driver:
        from mesheditor_amd import batch as batch_driver
        box = [batch_driver.make_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        try:
            comm = batch_driver.BatchComm(world, rank, device, box[0])
        except RuntimeError as e:  # no RCCL communicator: say so and gather through torch.distributed (gloo) instead
            print("bench: %s -- falling back to the torch.distributed gather" % e, file=sys.stderr)
            cpp_driver = False

    def sync(ctxs):
        for c in ctxs:
            c.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    batch = args.workload == "batch64"
    if batch:
        items = batch_meshes()
        ctxs = [api.Context(device) for _ in range(max(1, args.threads))]
        ex_of = [m[0][:: len(m[0]) // 10][:10].astype(np.float32) for m in items]
        resident = {}  # meshes of this rank's share, uploaded before the timed region

        def solve(i, m, worker=0):
            c = ctxs[worker]
            key = (i, worker)
            if key not in resident:
                resident[key] = api.Mesh(c, m[0], m[1])
            return api.mesh2modes(c, m[0], m[1], api.material(*m[2]), ex_of[i], config=api.default_config(**m[3]), mesh=resident[key])
        records = None

        def step():
            nonlocal records
            if comm is not None:  # C++ driver: deal, solve (host threads per GPU), ncclAllGather of the records
                records = batch_driver.solve_batch(comm, items, NEV_MAX, POS_MAX, threads=max(1, args.threads), excite=ex_of)
            else:
                records = sharding.solve_batch(items, solve, NEV_MAX, dist, gather_device, threads=len(ctxs), pos_max=POS_MAX)
            return records
    else:
        ctx = api.Context(device)
        ctxs = [ctx]
        pts, tets, m, kw = meshes.workload(args.workload)
        if world > 1:  # every rank its own object of the same size: jitter the extents deterministically per rank
            rng = np.random.Generator(np.random.MT19937(1000 + rank))
            pts = pts * rng.uniform(0.9, 1.1, 3)[None, :]
        mat = api.material(*m)
        cfg = api.default_config(num_modes=kw["num_modes"], num_fem_modes=kw["num_fem_modes"])
        ex = pts[:: len(pts) // 10][:10].astype(np.float32)  # P = 10 excitation positions, as the app and bench use
        mesh = api.Mesh(ctx, pts, tets)  # inputs resident in HBM before the timed region
        records = None
        if comm is not None:  # every rank describes every rank's mesh (the same jitter rule): equal costs deal mesh r to rank r
            rank_items, rank_ex = [], []
            base = meshes.workload(args.workload)[0]
            for r_ in range(world):
                g = np.random.Generator(np.random.MT19937(1000 + r_))
                p_ = base * g.uniform(0.9, 1.1, 3)[None, :] if world > 1 else base
                rank_items.append((p_, tets, m, kw))
                rank_ex.append(p_[:: len(p_) // 10][:10].astype(np.float32))

        def step():
            nonlocal records
            if comm is not None:  # the C++ batch driver: this rank's mesh goes host -> HBM inside the step (+ ~1 ms at this size)
                records = batch_driver.solve_batch(comm, rank_items, NEV_MAX, POS_MAX, threads=1, excite=rank_ex)
                mine = records[rank]
                return api.ModalResult(mine["freqs"], mine["t60s"], None, mine["positions"], mine["original_fundamental"], mine["eigenvalues"], mine["summary_shapes"],
                                       mine["mass"], mine["center_of_mass"], mine["inertia_diagonal"], mine["inertia_orientation_wxyz"], mine["profile"], None)
            t0 = time.perf_counter()
            r = api.mesh2modes(ctx, pts, tets, mat, ex, config=cfg, mesh=mesh)
            failed = len(r.eigenvalues) == 0
            if dist is not None:  # the final gather: the whole fixed-size record of every rank's mesh, one collective
                # a rank whose solve failed still joins the collective (with a failed record): the others must not hang in it
                rec = sharding.failed_record(rank, NEV_MAX, POS_MAX, time.perf_counter() - t0) if failed else sharding.pack_record(rank, r, NEV_MAX, POS_MAX, time.perf_counter() - t0)
                records = [sharding.unpack_record(x, NEV_MAX, POS_MAX) for x in sharding.gather_records({rank: rec}, world, dist, gather_device)]
                bad = [x["index"] for x in records if not x["ok"]]
                if bad:
                    raise RuntimeError("solve failed on rank(s) %s" % bad)
            if failed:
                raise RuntimeError("solve failed: %s" % r.profile)
            return r

    for _ in range(args.warmup):
        step()
    # the timed region: EXACTLY `steps` steps between two barriers, no instrumentation running
    sync(ctxs)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    sync(ctxs)
    dt = time.perf_counter() - t0
    # the roofline objects: the same steps once more with HIP events around every launch of the named kernel classes (on the
    # solver's own stream, inside the library).  Kept out of the timed region: ~700 event pairs per solve cost ~5 ms per step
    # (measured: 158.6 / 156.9 ms with them, 151.3 / 153.6 ms without, same box), which is the instrument's time, not the path's.
    for c in ctxs:
        c.time_kernels(True)
    sync(ctxs)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync(ctxs)
    dt_instrumented = time.perf_counter() - t1
    stats = [{k: sum(c.kernel_stats(cls)[k] for c in ctxs) for k in ("launches", "total_ms", "total_bytes")} for cls in (0, 1, 3, 4, 5)]
    for c in ctxs:
        c.time_kernels(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=gather_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if batch:
        pairs_per_step = sum(len(r["eigenvalues"]) for r in last)  # every rank holds every record after the gather
        modes_per_step = sum(len(r["freqs"]) for r in last)
        total_pairs, total_modes = pairs_per_step * args.steps, modes_per_step * args.steps
        workload = ("batch64: 64 jittered Kuhn boxes of %d tets (RealImpact size), 7 materials cycled, NumModes=30 NumFemModes=45, P=10, LPT deal over %d GPU(s), "
                    "%d solves in flight per GPU" % (len(items[0][1]), world, len(ctxs)))
        config = {"workload": workload, "meshes": len(items), "eigenpairs_per_mesh": 45, "parallelism": "lpt-deal x%d" % world}
        scaling = "strong"
    else:
        nev = len(last.eigenvalues)
        total_pairs, total_modes = nev * args.steps * world, len(last.freqs) * args.steps * world
        kind = "scan-like unstructured mesh (marching-tetrahedra skillet surface through the path's tetrahedraliser)" if args.workload.startswith("scan_") else "Kuhn mesh"
        config = {"workload": "%s: %s %d tets / %d DOF, NumModes=%d NumFemModes=%d, P=10 excitation points, one mesh per GPU"
                              % (args.workload, kind, len(tets), last.profile.get("dofs", 0), cfg.num_modes, cfg.num_fem_modes),
                  "eigenpairs_per_mesh": nev, "kept_modes_per_mesh": len(last.freqs), "lobpcg_iterations": last.profile.get("restarts"), "parallelism": "mesh-per-gpu x%d" % world}
        scaling = "weak"
    line = {
        "metric": "eigenpairs/sec (K/M assembly + 50-mode solve, 100k-tet mesh)",
        "value": total_pairs / dt,
        "unit": "eigenpairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": config,
        "modes_per_s": total_modes / dt,  # kept modes only (`value` counts every converged pair: 6 rigid-body + guard pairs included)
    }
    if dist is not None and records:
        line["gathered_records"] = {"count": len(records), "words_per_record": sharding.record_length(NEV_MAX, POS_MAX),
                                    "fields": "eigenvalues, freqs, t60s, positions, shapes, mass properties, solve profile",
                                    "collective": "ncclAllGather from C++ (modal::SolveBatch)" if comm is not None else "torch.distributed all_gather (%s)" % ("gloo" if share_gpu else "nccl")}
    spmm, asm, comb, comb_bytes, comb_full = stats
    if spmm["launches"]:
        achieved = spmm["total_bytes"] / (spmm["total_ms"] * 1e-3) / 1e9
        line["roofline"] = {"bound": "hbm",
                            "kernel": "k_spmm_wide / k_spmm: BSR 3x3 SpMM of the P2 and P1 operators over n-by-w panels "
                                      "(every launch of the solve: fp32 smoother products, mixed fp64-A x fp32-panel residuals, fp64 operator products)",
                            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("spmm_family", args.workload), "traffic_source": "committed profile (profiles/r0N_pmc_traffic.json: separate rocprofv3 --pmc passes of this command), not measured in this run",
                            "launches": spmm["launches"], "avg_launch_us": 1e3 * spmm["total_ms"] / spmm["launches"],
                            "algorithmic_bytes_per_launch": spmm["total_bytes"] / spmm["launches"],
                            "measured_in": "%d further steps after the timed region, HIP events around every launch (%.1f ms per step with them)" % (args.steps, 1e3 * dt_instrumented / args.steps)}
    if asm["launches"]:
        achieved = asm["total_bytes"] / (asm["total_ms"] * 1e-3) / 1e9
        line["roofline_assembly"] = {"bound": "hbm", "kernel": "K/M assembly of the quadratic level (SURVEY 8d bytes: 152 B read per tet, 80 B written per node block)",
                                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("assembly", args.workload),
                                     "launches": asm["launches"], "avg_launch_us": 1e3 * asm["total_ms"] / asm["launches"],
                                     "algorithmic_bytes_per_launch": asm["total_bytes"] / asm["launches"]}
    if comb["launches"]:
        tflops = comb["total_bytes"] / (comb["total_ms"] * 1e-3) / 1e12  # (the class's work is flops)
        line["roofline_combine"] = {"bound": "mfma", "kernel": "k_combine: basis updates out = [X | W | P] C of the eigensolver, fp64 MFMA (v_mfma_f64_16x16x4_f64); every call of a solve",
                                    "achieved": tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                                    "launches": comb["launches"], "avg_launch_us": 1e3 * comb["total_ms"] / comb["launches"],
                                    "algorithmic_flops_per_launch": comb["total_bytes"] / comb["launches"],
                                    "algorithmic_bytes_per_launch": comb_bytes["total_bytes"] / max(1, comb_bytes["launches"]),
                                    "hbm_GBps_at_that_time": comb_bytes["total_bytes"] / (comb["total_ms"] * 1e-3) / 1e9,
                                    "ms_per_step": comb["total_ms"] / args.steps / max(1, len(ctxs)),
                                    # the iteration's full-size update alone (>= 200 basis columns into >= 128 output columns: X and P together,
                                    # 240 -> 160 on the 65-pair solve); the class average above also holds the HBM-bound projections (80 -> 80,
                                    # accumulate: 13 flop per byte) and the coarse solve's short products
                                    "full_size_update": ({"launches": comb_full["launches"], "avg_launch_us": 1e3 * comb_full["total_ms"] / comb_full["launches"],
                                                          "algorithmic_flops_per_launch": comb_full["total_bytes"] / comb_full["launches"],
                                                          "achieved": comb_full["total_bytes"] / (comb_full["total_ms"] * 1e-3) / 1e12,
                                                          "frac": comb_full["total_bytes"] / (comb_full["total_ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS} if comb_full["launches"] else None),
                                    "peak_note": "78.6 TFLOP/s: the MI355X data sheet's fp64 matrix figure (the microarchitecture guide lists none); the device holds 2.0-2.1 GHz under these kernels, 65 TFLOP/s there"}
    if not batch:
        line["profile"] = {k: last.profile.get(k) for k in ("assemble", "sample_excite", "factorize", "iterate", "op_solve", "extract", "restarts", "op_applications", "sytrd_redos", "rr_selfcheck")}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if not batch and "roofline" in line:
            line["roofline"]["single_launch"] = operator_forms(api, ctxs[0], mesh, mat)
            from tools import lab
            copy_gbs, read_gbs = lab.bench_stream(ctxs[0], 4 << 30, 10)  # a measured copy-kernel ceiling beside the nominal peak
            # (no fraction against these: the best of the lab's copy forms reaches 5.4-5.5 TB/s, under the guide's 6.29 for a float4 copy --
            # the denominator of `frac` is the nominal 8 TB/s only)
            line["roofline"]["measured_ceiling"] = {"copy_GBps": copy_gbs, "read_GBps": read_gbs,
                                                    "note": "best of the lab library's streaming kernels over 4 GiB (4-16 loads of 16 bytes in flight per lane, temporal and not, 4-32 workgroups per CU, the runtime's own copy); copy counts bytes read + written; the guide's float4 copy: 6 290"}
        if not batch:
            line["concurrent_solves"] = concurrent_throughput(api, device, pts, tets, mat, ex, cfg)
        if not batch:
            line["edit_loop"] = edit_loop(api, ctxs[0], pts, tets, m, ex, cfg, mesh)
            line["scan_like"] = scan_like_rows(api, ctxs[0])
            line["config3"] = config3_rows(api, ctxs[0])
            line["batch64"] = batch64_pass(api, device)
            line["batch64_scan"] = batch64_pass(api, device, scans=True)
        # The CPU partner of `value`, timed LIVE on this box's host cores on the metric's own mesh (round 6; ~150 s on the 16 cores the GPU
        # box grants: the driver gives the bench 1 800 s).  The 10k-tet sample of rounds 1-5 (one thread and the team) stays as
        # cpu_baseline_small; MH_BENCH_CPU_SAMPLE=1 puts it back as the main row for a quick run.
        small = cpu_baseline()
        if os.environ.get("MH_BENCH_CPU_SAMPLE") == "1":
            line["cpu_baseline"] = small
            line["cpu_baseline_metric_mesh"] = cpu_baseline_metric_mesh("cube_s100k" if batch else args.workload)
        else:
            live = cpu_baseline_metric_mesh_live("cube_s100k" if batch else args.workload)
            live["sample"] = "%s: %d tets / %d DOF, %d eigenpairs, whole mesh2modes path -- the metric's own mesh, timed live on this box in this run (%.0f s)" % (
                live["workload"], live["mesh"]["tets"], live["mesh"]["dof"], live["eigenpairs"], live["seconds"])
            live["host_cores"] = live["host"]["cores_available"]
            line["cpu_baseline"] = live
            line["cpu_baseline_small"] = small
            line["cpu_baseline_metric_mesh"] = live  # (the key of rounds 4-5: the same record)
        bank = bank_metric()
        line["resonator_bank"] = bank
        if isinstance(bank.get("all_live"), dict):
            line["roofline_bank"] = bank["all_live"]["roofline_bank"]
        line["cpu_bank_baseline"] = cpu_bank_baseline()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which sits in that buffer until the process exits when stdout is a
        # file or a pipe: push it out first, so that the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    if not batch:
        mesh.close()
    for c in ctxs:
        c.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
