import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api, meshes
ctx = api.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "skillet_s100k"
nev = int(sys.argv[2]) if len(sys.argv) > 2 else 215
p, t, m, kw = meshes.workload(name)
mesh = api.Mesh(ctx, p, t)
api.System(ctx, mesh, api.material(*m)).eigs(min(nev, 20), residual_tol=1e-3, max_iters=60)  # warm-up (code objects)
t0 = time.perf_counter()
s = api.System(ctx, mesh, api.material(*m))
try:
    ev, prof = s.eigs(nev, residual_tol=1e-6, max_iters=int(os.environ.get("ITERS", 60)))
    ctx.synchronize()
    print("%s %d pairs: %.3f s, iterations %d" % (name, nev, time.perf_counter() - t0, prof["restarts"]))
except Exception as e:
    print("FAILED", e)
