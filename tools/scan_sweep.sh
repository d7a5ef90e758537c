#!/bin/bash
# Preconditioner-cycle and patch variations on named workloads (iterations and ms per solve), on the GPU box:
#   bash tools/scan_sweep.sh gpurun_out/sweep.txt scan_s30k scan_s100k cube_s100k
# MH_CYCLE = degree of the P2 smoother, degree of the P1 smoother, P1 cycles per application, spectrum ratio (0 = built-in).
out=${1:-gpurun_out/scan_sweep.txt}; shift
wls=${@:-"cube_s30k scan_s30k scan_s100k"}
mkdir -p $(dirname $out); : > $out
for v in "X=0" "MH_CYCLE=2,0,0,8" "MH_CYCLE=3,0,0,16" "MH_CYCLE=6,0,0,30" "MH_CYCLE=0,6,0,0" "MH_CYCLE=0,0,2,0" "MH_PATCH_Q=0" "MH_PATCH_Q=0.05" "MH_AGG=32" "MH_PRECOND_FP64=1"; do
  echo "== $v" >> $out
  env $v timeout 900 python tools/scan_probe.py $wls --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], {k:(round(r[k],2) if isinstance(r.get(k),float) else r.get(k)) for k in ('iterations','ms','factorize_ms','max_rel_err_vs_oracle')})" >> $out
done
cat $out
