"""One width of the SpMM microbenchmark, for rocprofv3 --pmc passes (HBM traffic of k_spmm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api, meshes
import lab  # tools/lab.py: libmodalhip_lab.so
w = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = api.Context(0)
p, t, m, kw = meshes.workload("cube_s100k")
s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*m))
ms, by = lab.bench_spmm(s, w, 5)
print(f"w={w} {ms*1e3:.1f} us {by/1e6:.1f} MB algorithmic")
