for c in "5,5,3,60" "4,5,3,60" "4,5,3,40" "5,5,2,60" "5,4,3,60" "4,4,3,60" "5,3,3,60" "5,5,3,40"; do
  echo "== MH_CYCLE=$c"
  MH_CYCLE=$c python tools/scan_probe.py config3_s100k_repaired config3_s30k_repaired skillet_s100k --reps 1 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x,1) for x in r['all_ms'][1:]])"
done
