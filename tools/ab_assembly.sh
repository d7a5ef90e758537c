# assembly kernel A/B on the GPU box: one lane per contribution (default) against one thread per node block (MH_ASSEMBLE_BY_BLOCK=1)
mkdir -p gpurun_out/r02f
rm -f gpurun_out/r02f/ab.txt
for rep in 1 2; do for blk in 0 1; do
  MH_ASSEMBLE_BY_BLOCK=$blk python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rep $rep by_block $blk ms %.2f spmm_frac %.4f asm_us %.1f asm_frac %.4f'%(d['ms_per_step'], d['roofline']['frac'], d['roofline_assembly']['avg_launch_us'], d['roofline_assembly']['frac']))" >> gpurun_out/r02f/ab.txt
done; done
python -m pytest tests/test_analysis_gpu.py -q -m gpu -k "assembl or degenerate or bit_reproducible or eigenvalues_match or elementwise or matvec" > gpurun_out/r02f/tests.log 2>&1
cat gpurun_out/r02f/ab.txt; tail -3 gpurun_out/r02f/tests.log
