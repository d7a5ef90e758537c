# A/B of one environment switch on the GPU box: bash tools/ab_env.sh NAME [steps]
# prints ms per step for NAME=0 / NAME=1 alternating, and whether the eigenvalues of the metric solve are bit-identical
NAME=$1; STEPS=${2:-5}
mkdir -p gpurun_out/ab
for v in 0 1; do
  env $NAME=$v timeout 300 python - <<PY
import numpy as np, sys
sys.path.insert(0, '.')
from mesheditor_amd import api, meshes
ctx = api.Context(0)
p, t, m, kw = meshes.workload("cube_s30k")
s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*m))
ev, _ = s.eigs(65, -(2 * np.pi * 20.0) ** 2, 1e-5)
np.save("gpurun_out/ab/ev_$v.npy", np.asarray(ev))
PY
done
python - <<PY
import numpy as np
a, b = np.load("gpurun_out/ab/ev_0.npy"), np.load("gpurun_out/ab/ev_1.npy")
print("$NAME: eigenvalues bit-identical:", bool(np.array_equal(a, b)), " max rel diff %.2e" % float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-300))))
PY
for rep in 1 2; do for v in 0 1; do
  env $NAME=$v timeout 600 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep $NAME=$v ms %.2f value %.1f iters %d'%(d['ms_per_step'], d['value'], d['config']['lobpcg_iterations']))"
done; done
