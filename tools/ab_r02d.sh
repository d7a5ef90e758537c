mkdir -p gpurun_out/r02d
cat /sys/fs/cgroup/cpu.max > gpurun_out/r02d/cpu_max.txt 2>&1; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())" >> gpurun_out/r02d/cpu_max.txt
for rep in 1 2; do
for small in 1 0; do for blk in 0 1; do
  MH_SPMM_SMALL=$small MH_ASSEMBLE_BY_BLOCK=$blk python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep small $small byblock $blk ms %.2f frac %.4f avg_us %.1f asm_us %.1f asm_frac %.4f'%(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline_assembly']['avg_launch_us'], d['roofline_assembly']['frac']))" >> gpurun_out/r02d/ab.txt
done; done; done
python - <<'PY' >> gpurun_out/r02d/omp.txt 2>&1
import time, os, numpy as np
from oracle import pyoracle as po
from mesheditor_amd import meshes
pts, tets, m, kw = meshes.workload("cube_s10k")
cfg = po.default_config(num_modes=50, num_fem_modes=65)
ex = pts[:: len(pts) // 10][:10].astype(np.float32)
for th in (1, 4, 8, 16):
    po.set_threads(th)
    t0=time.perf_counter(); r = po.mesh2modes(pts, tets, po.material(*m), ex, config=cfg); dt=time.perf_counter()-t0
    print(th, round(dt,2), {k:round(v,2) for k,v in r.profile.items() if isinstance(v,float) and v>0.01}, flush=True)
PY
cat gpurun_out/r02d/ab.txt
