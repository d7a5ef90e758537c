import sys, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
ctx = api.Context(0)
pts, tets, m, kw = meshes.workload(sys.argv[1])
mesh = api.Mesh(ctx, pts, tets)
s = api.System(ctx, mesh, api.material(*m))
try:
    ev, prof = s.eigs(65, -(2 * np.pi * 20.0) ** 2, float(sys.argv[2]))
    print("ok", prof["restarts"])
except Exception as e:
    print("FAILED", e)
