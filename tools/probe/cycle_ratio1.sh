for c in "5,5,3,60" "5,5,3,60,8" "5,5,3,60,20" "5,5,3,60,30" "5,5,3,60,120" "5,3,3,60,8" "5,4,3,60,12"; do
  echo "== MH_CYCLE=$c"
  MH_CYCLE=$c python tools/scan_probe.py skillet_s100k uvsphere_s10k scan_s100k scan_s30k_repaired ball_s10k --reps 1 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x,1) for x in r['all_ms'][1:]])"
done
