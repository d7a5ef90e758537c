#!/bin/bash
out=${1:-gpurun_out/r03a/scan_sweep2.txt}
mkdir -p $(dirname $out); : > $out
run() { wl=$1; shift; echo "== $wl $*" >> $out; env "$@" timeout 300 python tools/scan_probe.py $wl --reps 1 2>&1 | grep -E "workload|rror" | python -c "import sys,json
for l in sys.stdin:
    try:
        r=json.loads(l); print({k:r[k] for k in ('iterations','ms','eigenpairs','factorize_ms')})
    except Exception: print(l.strip()[:300])" >> $out; }
run cube_s30k X=0
run cube_s100k X=0
run cube_s100k MH_COARSE_CAP=4096
run scan_s30k X=0
run scan_s30k MH_AGG=32
run scan_s30k MH_DEG2=4 MH_CHEB_RATIO=30
run scan_s30k MH_DEG2=6 MH_CHEB_RATIO=30
run scan_s30k MH_DEG2=6 MH_CHEB_RATIO=30 MH_DEG1=8
run scan_s30k MH_DEG2=8 MH_CHEB_RATIO=60
run scan_s100k MH_DEG2=4 MH_CHEB_RATIO=30
run scan_s100k MH_DEG2=6 MH_CHEB_RATIO=30
run scan_s100k MH_DEG2=8 MH_CHEB_RATIO=60
run skillet_s100k X=0
cat $out
