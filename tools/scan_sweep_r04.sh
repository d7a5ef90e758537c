#!/bin/bash
# Round 4: preconditioner-cycle variations on the sliver-repaired scan meshes (front-end default), iterations and ms per solve.
#   bash tools/scan_sweep_r04.sh gpurun_out/r04_scan_sweep.txt
out=${1:-gpurun_out/r04_scan_sweep.txt}; shift
wls=${@:-"scan_s30k_repaired scan_s100k_repaired"}
mkdir -p $(dirname $out); : > $out
for v in "X=0" "MH_CYCLE=2,0,0,8" "MH_CYCLE=3,0,0,16" "MH_CYCLE=4,0,0,30" "MH_CYCLE=4,0,0,60" "MH_CYCLE=6,0,0,60" "MH_CYCLE=5,4,2,60" "MH_CYCLE=5,5,2,60" "MH_PATCH_Q=0.05" "MH_PATCH_Q=0.01" "MH_PATCH_Q=0"; do
  echo "== $v" >> $out
  env $v timeout 900 python tools/scan_probe.py $wls --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], {k:(round(r[k],2) if isinstance(r.get(k),float) else r.get(k)) for k in ('iterations','ms','factorize_ms')})" >> $out
done
cat $out
