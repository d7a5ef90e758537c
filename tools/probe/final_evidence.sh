python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
tail -c 400 gpurun_out/r05_bench_line.err
BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_gpus2_shared_gpu.json 2> gpurun_out/r05_bench_gpus2.err
tail -c 300 gpurun_out/r05_bench_gpus2_shared_gpu.json
BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_forced_dist.json 2> gpurun_out/r05_bench_forced_dist.err
tail -c 300 gpurun_out/r05_bench_forced_dist.json
python bench.py --workload batch64 --no-cpu-baseline > gpurun_out/r05_bench_batch64.json 2>/dev/null
tail -c 400 gpurun_out/r05_bench_batch64.json
