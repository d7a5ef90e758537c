"""What the residual tolerance buys: iterations, time and eigenvalue error against the oracle fixtures at 1e-5 (the default mapping 0.1 sqrt(Tolerance)), 3e-5 and 1e-4.
    python tools/probe/tolerance_probe.py cube_s100k ball_s10k scan_s100k_repaired"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mesheditor_amd import api, meshes
SIGMA = -(2 * np.pi * 20.0) ** 2
ctx = api.Context(0)
for name in sys.argv[1:]:
    pts, tets, m, kw = meshes.workload(name)
    ref = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_eigs_%s.json" % name)))["eigenvalues"])
    el = ref > 1e-6 * ref[-1]
    mesh = api.Mesh(ctx, pts, tets)
    for tol in [float(t) for t in os.environ.get("TOLS", "1e-5,3e-5,1e-4").split(",")]:
        best = None
        for rep in range(3):
            s = api.System(ctx, mesh, api.material(*m))
            t0 = time.perf_counter()
            ev, prof = s.eigs(kw["num_fem_modes"], SIGMA, tol)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            s.close()
            best = dt if best is None else min(best, dt)
        err = np.abs(ev[el] - ref[el]) / ref[el]
        print("%-22s tol %.0e: %2d iterations, %7.1f ms (solve only), eigenvalue error max %.1e, median %.1e" % (name, tol, prof["restarts"], 1e3 * best, err.max(), np.median(err)), flush=True)
