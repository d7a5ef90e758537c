#!/bin/bash
# Preconditioner variations on the scan-like 30k-tet mesh (iterations and ms per solve); output: gpurun_out/r03a/scan_sweep.txt
out=${1:-gpurun_out/r03a/scan_sweep.txt}
wl=${2:-scan_s30k}
mkdir -p $(dirname $out); : > $out
run() { echo "== $*" >> $out; env "$@" python tools/scan_probe.py $wl --reps 1 2>&1 | grep workload | python -c "import sys,json; [print({k:r[k] for k in ('iterations','ms','eigenpairs')}) for r in map(json.loads, sys.stdin)]" >> $out; }
run X=0
run MH_PRECOND_FP64=1
run MH_DEG2=4
run MH_DEG2=6
run MH_DEG1=8
run MH_DEG2=4 MH_DEG1=8
run MH_CHEB_RATIO=30 MH_DEG2=6 MH_DEG1=8
run MH_CHEB_RATIO=30 MH_DEG2=6 MH_DEG1=8 MH_PRECOND_FP64=1
run MH_AGG=8
run MH_AGG=16
run MH_AGG=64
run MH_GAMMA=1
run MH_GAMMA=6
run MH_GUARD_ABS=40
cat $out
