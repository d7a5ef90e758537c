"""A long-lived process: 150 solves of alternating workloads and pair counts on one context, bank scenes created and dropped in between;
device bytes held by the pool and host resident memory at intervals.   python tools/probe/leak_probe.py"""
import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from mesheditor_amd import api, meshes
from tests import bank_harness as bh
import lab
def rss_mb():
    with open("/proc/self/status") as f:
        for l in f:
            if l.startswith("VmRSS"): return int(l.split()[1]) / 1024
ctx = api.Context(0)
names = ["cube_s10k", "ball_s10k", "uvsphere_s10k", "cube_s30k"]
data = {n: meshes.workload(n) for n in names}
for it in range(150):
    n = names[it % 4]
    pts, tets, m, _ = data[n]
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    s.eigs([30, 65, 140][it % 3], -(2 * np.pi * 20.0) ** 2, 1e-5)
    s.close(); mesh.close()
    if it % 10 == 0:
        sc = bh.DeviceScene(8, 64, 0.5, 2)
        sc.render(2, bh.BLOCK)
        del sc
    if it % 25 == 0 or it == 149:
        r, i, c = lab.pool_stats(ctx)
        print(f"after {it + 1:3d} solves: pool holds {r / 2**20:8.1f} MB ({i / 2**20:8.1f} idle), host RSS {rss_mb():8.1f} MB", flush=True)
