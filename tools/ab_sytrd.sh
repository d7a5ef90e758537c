# Rayleigh-Ritz tridiagonalisation A/B on the GPU box: one workgroup (MH_SYTRD_MULTI=0) against several (=1), through bench.py
mkdir -p gpurun_out/r02g
rm -f gpurun_out/r02g/ab.txt
for rep in 1 2; do for multi in 0 1; do
  MH_SYTRD_MULTI=$multi timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep multi $multi ms %.2f value %.1f iters %d spmm_frac %.4f'%(d['ms_per_step'], d['value'], d['config']['lobpcg_iterations'], d['roofline']['frac']))" >> gpurun_out/r02g/ab.txt
done; done
cat gpurun_out/r02g/ab.txt
