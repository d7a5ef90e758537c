ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_batch -- python3 $ROOT/bench.py --workload batch64 --steps 1 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_batch.log 2>&1
cd $ROOT
python3 tools/trace_by_grid.py /tmp/prof_batch 0.5 > gpurun_out/batch_by_grid_now.txt 2>&1
