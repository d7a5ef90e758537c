import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from mesheditor_amd import api, meshes
ctx = api.Context(0)
base_p, base_t = meshes.kuhn_box(7, 6, 5, 0.14, 0.12, 0.1)
rng = np.random.default_rng(9)
base_p = base_p + rng.uniform(-1, 1, base_p.shape) * 0.003
m = meshes.MATERIALS["Ceramic"]
d = float(sys.argv[1])
pts = base_p + np.array([d, -0.5 * d, 0.25 * d])
sysg = api.System(ctx, api.Mesh(ctx, pts, base_t), api.material(*m))
try:
    ev, prof = sysg.eigs(45, -(2 * np.pi * 20) ** 2, 1e-4)
    print(len(ev), prof)
except Exception as e:
    print("EXC", repr(e))
