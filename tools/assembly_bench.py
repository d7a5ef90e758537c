"""Assembly kernel timing on the GPU box: builds the S100k system a few times and prints the timed class-1 kernel's average."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api, meshes
ctx = api.Context(0)
ctx.time_kernels(True)
for name in sys.argv[1:] or ["cube_s100k"]:
    p, t, m, kw = meshes.workload(name)
    mesh = api.Mesh(ctx, p, t)
    for rep in range(4):
        s = api.System(ctx, mesh, api.material(*m))
        ctx.synchronize()
        s.close()
    st = ctx.kernel_stats(1)
    print(name, os.environ.get("MH_ASM_XP", "-"), os.environ.get("MH_ASSEMBLE_BY_BLOCK", "-"), st, flush=True)
