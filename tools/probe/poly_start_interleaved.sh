for i in 1 2 3; do
for v in "MH_TEST=no_poly_start" "X=0"; do
  echo "== $v" 
  env $v python tools/scan_probe.py cube_s100k cube_s30k --reps 3 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], r['all_ms'])"
done
done
