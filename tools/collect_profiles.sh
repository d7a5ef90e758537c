#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh r03
# Writes the rocprofv3 summaries judged under profiles/ into gpurun_out/final/ (copied into profiles/ afterwards).
# Counter passes (--pmc) run on their own, without any trace option beside them.
R=${1:-r05}
ROOT=$PWD
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
cp $(ls -t /tmp/prof_bench/*/*kernel_stats.csv | head -1) $OUT/${R}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bank -- python3 $ROOT/tools/bank_bench.py --blocks 72 > $OUT/prof_bank.log 2>&1
cp $(ls -t /tmp/prof_bank/*/*kernel_stats.csv | head -1) $OUT/${R}_bank_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
cd $ROOT
python3 tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $OUT/${R}_pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
python3 tools/prof_summary.py /tmp/prof_bench 30 > $OUT/summary_bench.txt 2>&1
python3 tools/prof_summary.py /tmp/prof_bank 10 > $OUT/summary_bank.txt 2>&1
python3 tools/trace_by_grid.py /tmp/prof_bench 0.5 > $OUT/${R}_bench_by_grid.txt 2>&1
