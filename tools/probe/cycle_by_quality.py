"""Iterations / ms / worst element shape / sliver patches of a few bodies under the current MH_CYCLE (run once per setting)."""
import sys, os, time, re, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes, tets as front_end
ctx = api.Context(0)
def bodies():
    for name in ("cube_s30k", "uvsphere_s10k", "ball_s10k", "bar_thin"):
        p, t, m, kw = meshes.workload(name)
        yield name, p, t, m
    p, t = meshes.jittered_box(17, 1003)
    yield "jittered_box_17", p, t, meshes.MATERIALS["Ceramic"]
    P, F = meshes.uv_sphere_surface(0.15, 96, 48)
    pts, cells, left = front_end.tetrahedralize(P, F, quality=True)
    yield "uv96x48_quality", pts, cells, meshes.MATERIALS["Ceramic"]
    P, F = meshes.uv_sphere_surface(0.15, 64, 32)
    pts, cells, left = front_end.tetrahedralize(P, F)
    yield "uv64x32_default_fill", pts, cells, meshes.MATERIALS["Ceramic"]
for name, p, t, m in bodies():
    mesh = api.Mesh(ctx, p, t); s = api.System(ctx, mesh, api.material(*m))
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); ev, prof = s.eigs(65, residual_tol=1e-5); best = min(best, time.perf_counter() - t0)
    print(os.environ.get("MH_CYCLE", "default"), name, "tets", len(t), "iterations", prof["restarts"], "ms %.1f" % (best * 1e3), flush=True)
    s.close(); mesh.close()
