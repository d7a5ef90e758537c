"""Different workloads and pair counts at once from several host threads (one context each), each repeated; every result against the serial
solve of the same request, bit for bit.   python tools/probe/mixed_soak.py"""
import sys, threading, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
jobs = [("scan_s30k_repaired", 65), ("cube_s30k", 215), ("ball_s10k", 140), ("uvsphere_s10k", 65), ("scan_s30k", 129), ("cube_s10k", 280)]
data = {n: meshes.workload(n) for n, _ in jobs}
def solve(ctx, name, pairs):
    pts, tets, m, _ = data[name]
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    ev, prof = s.eigs(pairs, -(2 * np.pi * 20.0) ** 2, 1e-6)
    s.close(); mesh.close()
    return ev
c0 = api.Context(0)
t0 = time.perf_counter()
ref = {j: solve(c0, *j) for j in jobs}
serial = time.perf_counter() - t0
out, errs = {}, []
def work(k, ctx):
    try:
        for rep in range(3):
            for i, j in enumerate(jobs):
                if (i + rep) % 3 == k:
                    out[(j, rep)] = solve(ctx, *j)
    except Exception as e:  # noqa: BLE001
        errs.append(repr(e)[:300])
ctxs = [api.Context(0) for _ in range(3)]
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k, ctxs[k])) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
par = time.perf_counter() - t0
bad = [(j, rep) for (j, rep), ev in out.items() if not np.array_equal(ev, ref[j])]
worst = max((np.abs(ev[6:] / ref[j][6:] - 1).max() for (j, rep), ev in out.items()), default=0.0)
print(f"{len(out)} of {3 * len(jobs)} concurrent solves done in {par:.1f} s (serial pass of {len(jobs)}: {serial:.1f} s); errors {errs}; not bit-identical {bad}; worst relative difference {worst:.1e}")
