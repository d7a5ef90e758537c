for c in "5,5,3,60" "6,5,3,80" "7,5,3,100" "5,7,3,60" "5,5,4,60" "5,5,2,60" "4,5,3,60" "5,5,3,100" "5,4,3,60" "8,5,3,150"; do
  echo "== MH_CYCLE=$c"
  MH_CYCLE=$c python tools/scan_probe.py skillet_s100k uvsphere_s10k bar_thin --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x,1) for x in r['all_ms'][1:]])"
done
