#!/bin/bash
# Round 5: the cold start block from low-degree polynomial fields (default) against the random block of rounds 1-4 (MH_TEST=no_poly_start)
out=${1:-gpurun_out/r05_poly_start_ab.txt}
mkdir -p $(dirname $out); : > $out
for v in "MH_TEST=no_poly_start" "X=0"; do
  echo "== $v" >> $out
  env $v timeout 2400 python tools/scan_probe.py cube_s100k cube_s30k ball_s10k uvsphere_s10k scan_s30k scan_s100k scan_s30k_repaired scan_s100k_repaired skillet_s100k config3_s30k config3_s100k_repaired --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], {k:(round(r[k],2) if isinstance(r.get(k),float) and k!='max_rel_err_vs_oracle' else r.get(k)) for k in ('iterations','ms','factorize_ms','max_rel_err_vs_oracle')})" >> $out
done
cat $out
