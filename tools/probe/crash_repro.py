import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mesheditor_amd import api, meshes, tets as front_end
P, F = meshes.uv_sphere_surface(0.045, 24, 12)
pts, tets, _ = front_end.tetrahedralize(P, F, repair_slivers=False)
print("mesh", len(pts), len(tets), flush=True)
ctx = api.Context(0)
system = api.System(ctx, api.Mesh(ctx, pts, tets), api.material(*meshes.MATERIALS["Glass"]))
print("assembled", system.n, flush=True)
ev, prof = system.eigs(45, -(2 * np.pi * 20.0) ** 2, 1e-5, max_iters=300)
print(prof["restarts"], ev[6:10])
