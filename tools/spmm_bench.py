"""SpMM roofline microbenchmark on the GPU box: K x over n x w panels at S100k."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api, meshes
import lab  # tools/lab.py: libmodalhip_lab.so
ctx = api.Context(0)
for name in sys.argv[1:] or ["cube_s100k"]:
    p, t, m, kw = meshes.workload(name)
    s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*m))
    for w in [int(v) for v in os.environ.get('WIDTHS', '1,8,16,32,64,75,128,230').split(',')]:
        ms, by = lab.bench_spmm(s, w, 20)
        em = lab.bench_elementwise(s, w, 20)  # the same product element by element (no matrix, atomic scatter); same BSR-equivalent byte count
        print(f"{name} w={w:4d}  BSR {ms*1e3:8.1f} us  {by/1e6:8.1f} MB  {by/ms/1e6:8.1f} GB/s  {100*by/ms/1e6/8000:5.1f}% of 8 TB/s   |  element-wise {em*1e3:8.1f} us"
              f"  {100*by/em/1e6/8000:5.1f}% (BSR-equivalent bytes)", flush=True)
