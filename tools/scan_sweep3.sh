#!/bin/bash
out=${1:-gpurun_out/r03a/scan_sweep3.txt}
mkdir -p $(dirname $out); : > $out
run() { wl=$1; shift; echo "== $wl $*" >> $out; env "$@" timeout 300 python tools/scan_probe.py $wl --reps 1 2>&1 | grep -E "workload|rror" | python -c "import sys,json
for l in sys.stdin:
    try:
        r=json.loads(l); print({k:r[k] for k in ('iterations','ms','eigenpairs','factorize_ms')})
    except Exception: print(l.strip()[:300])" >> $out; }
run cube_s30k X=0
run cube_s100k X=0
run ball_s10k X=0
run scan_s30k MH_PATCH_Q=0
run scan_s30k X=0
run scan_s30k MH_PATCH_Q=0.03
run scan_s30k MH_PATCH_Q=0.1
run scan_s30k MH_DEG2=3 MH_CHEB_RATIO=16
run scan_s30k MH_DEG2=4 MH_CHEB_RATIO=30
run scan_s30k MH_PRECOND_FP64=1
run scan_s100k X=0
run scan_s100k MH_DEG2=4 MH_CHEB_RATIO=30
cat $out
MH_VERBOSE=1 timeout 200 python tools/scan_probe.py scan_s100k --reps 0 2>&1 | grep -E "workload|\|\|A\|\||lobpcg\] it +(0|1|2|5|10|20|30|40|60|80|100|150|200) .*conv" | cut -c1-170
