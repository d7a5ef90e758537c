#!/bin/bash
out=${1:-gpurun_out/r03b/sweep5.txt}
mkdir -p $(dirname $out); : > $out
for v in "X=0" "MH_AGG=40" "MH_AGG=40 MH_GAMMA=2" "MH_AGG=40 MH_DEG1=3" "MH_AGG=64"; do
  echo "== $v" >> $out
  env $v timeout 600 python tools/scan_probe.py cube_s100k cube_s30k ball_s10k scan_s30k scan_s100k skillet_s100k --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], {k:(round(r[k],2) if isinstance(r.get(k),float) else r.get(k)) for k in ('iterations','ms','factorize_ms')})" >> $out
done
cat $out
