"""Does the fp32 smoother step cost more per column on a wide panel than on narrow ones?  (round 5: the P1-level products of the 215-pair
solves run at a third of the per-column rate of the 65-pair ones.)   python tools/probe/cheb_slab_probe.py [workload]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mesheditor_amd import api, meshes
from tools import lab
name = sys.argv[1] if len(sys.argv) > 1 else "scan_s100k_repaired"
pts, tets, m, kw = meshes.workload(name)
ctx = api.Context(0)
mesh = api.Mesh(ctx, pts, tets)
s = api.System(ctx, mesh, api.material(*m))
s.eigs(20, residual_tol=1e-3)  # builds the hierarchy and its fp32 copies
L = lab.lib()
L.mhl_system_bench_cheb_step.restype = C.c_int
L.mhl_system_bench_cheb_step.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
for level in (1, 2):
    for width, slab_list in ((240, (1, 2, 3, 4, 6)), (80, (1, 2)), (256, (1, 2, 4))):
        row = []
        for slabs in slab_list:
            ms = C.c_double(0)
            ctx.check(L.mhl_system_bench_cheb_step(s.h, level, width, slabs, 20, C.byref(ms)))
            row.append("%d x %d cols: %.1f us" % (slabs, width // slabs, 1e3 * ms.value))
        print("%s level %d width %d: %s" % (name, level, width, "; ".join(row)), flush=True)
