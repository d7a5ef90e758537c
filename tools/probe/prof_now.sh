ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_bench.log 2>&1
cd $ROOT
python3 tools/trace_by_grid.py /tmp/prof_bench 0.5 > gpurun_out/bench_by_grid_now.txt 2>&1
cp $(ls -t /tmp/prof_bench/*/*kernel_stats.csv | head -1) gpurun_out/bench_kernel_stats_now.csv
