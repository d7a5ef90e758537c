"""Iterations of the 96 x 48 UV sphere (quality fill) and of the bench mesh under the current library: run with MH_TEST=... for A/B."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
from mesheditor_amd import api, meshes, tets as front_end
P, F = meshes.uv_sphere_surface(0.15, 96, 48)
pts, tets, left = front_end.tetrahedralize(P, F, quality=True)
c = api.Context(0)
m = meshes.MATERIALS["Ceramic"]
ex = pts[(np.arange(10) * len(P)) // 10].astype(np.float32)
for rep in range(2):
    r = api.mesh2modes(c, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=50, num_fem_modes=65))
    print({k: v for k, v in r.profile.items() if not isinstance(v, (list, dict))}, flush=True)
    print(os.environ.get("MH_TEST", "-"), "sphere: restarts", r.profile.get("restarts"), "selfcheck", r.profile.get("rr_selfcheck"), "redos", r.profile.get("sytrd_redos"), "ev[6:9]", r.eigenvalues[6:9], flush=True)
