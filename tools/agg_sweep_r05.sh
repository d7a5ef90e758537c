#!/bin/bash
# Round 5: aggregate-size target of the rigid-body level (MH_AGG) against set-up time and iteration counts, every bench workload.
out=${1:-gpurun_out/r05_agg_sweep.txt}
mkdir -p $(dirname $out); : > $out
for v in "X=0" "MH_AGG=24" "MH_AGG=32" "MH_AGG=48" "MH_AGG=64"; do
  echo "== $v" >> $out
  env $v timeout 1500 python tools/scan_probe.py cube_s100k cube_s30k ball_s10k uvsphere_s10k scan_s30k scan_s100k scan_s100k_repaired skillet_s100k --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], {k:(round(r[k],2) if isinstance(r.get(k),float) else r.get(k)) for k in ('iterations','ms','factorize_ms','max_rel_err_vs_oracle')})" >> $out
done
cat $out
