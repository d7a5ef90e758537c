import sys, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
ctx = api.Context(0)
for pairs in (65, 120, 215):
    p, t = meshes.jittered_box(12, 1000)
    mesh = api.Mesh(ctx, p, t)
    s = api.System(ctx, mesh, api.material(*meshes.MATERIALS["Steel"]))
    ev, prof = s.eigs(pairs, residual_tol=1e-5)
    print("pairs", pairs, "iterations", prof["restarts"], flush=True)
    s.close(); mesh.close()
ctx.close()
print("done", flush=True)
