#!/usr/bin/env python3
"""Solve named workloads on the GPU and print size, iterations and time per solve (Kuhn grids beside the scan-like meshes).
    python tools/scan_probe.py cube_s30k scan_s30k cube_s100k scan_s100k [--reps 3] [--json out.json]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api, meshes

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="+")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--json", default=None)
a = ap.parse_args()
ctx = api.Context(0)
rows = []
for name in a.workloads:
    pts, tets, m, kw = meshes.workload(name)
    ex = pts[(np.arange(10) * len(pts)) // 10].astype(np.float32)
    cfg = api.default_config(**kw)
    mesh = api.Mesh(ctx, pts, tets)
    times, r = [], None
    for rep in range(a.reps + 1):
        t0 = time.perf_counter()
        r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=cfg, mesh=mesh)
        ctx.synchronize()
        times.append(time.perf_counter() - t0)
    ok = len(r.eigenvalues)
    row = {"workload": name, "tets": int(len(tets)), "points": int(len(pts)), "dof": int(r.profile.get("dofs", 0)), "eigenpairs": int(ok),
           "iterations": int(r.profile.get("restarts", 0)), "ms": 1e3 * float(np.median(times[1:])), "all_ms": [round(1e3 * t, 1) for t in times], "factorize_ms": 1e3 * r.profile.get("factorize", 0),
           "iterate_ms": 1e3 * r.profile.get("iterate", 0), "assemble_ms": 1e3 * r.profile.get("assemble", 0),
           "first_elastic_hz": float(np.sqrt(max(r.eigenvalues[6], 0)) / (2 * np.pi)) if ok > 6 else None}
    fx = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oracle_eigs_%s.json" % name)
    if ok and os.path.exists(fx):  # the oracle's eigenvalues for this workload (tests/golden/make_oracle_fixtures.py)
        with open(fx) as f:
            ref = np.array(json.load(f)["eigenvalues"])
        k = min(len(ref), ok)
        el = ref[:k] > 1e-6 * ref[k - 1]
        row["max_rel_err_vs_oracle"] = float((np.abs(r.eigenvalues[:k][el] - ref[:k][el]) / ref[:k][el]).max())
        row["rigid_abs_over_lambda7"] = float(np.abs(r.eigenvalues[:k][~el]).max() / ref[:k][el][0]) if (~el).any() else 0.0
    rows.append(row)
    print(json.dumps(row), flush=True)
    mesh.close()
if a.json:
    with open(a.json, "w") as f:
        json.dump(rows, f, indent=1)
ctx.close()
