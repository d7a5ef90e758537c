for c in "5,5,3,60" "4,5,3,40" "3,5,3,20" "4,5,3,60" "3,5,3,30" "2,5,3,8" "4,4,3,40" "6,5,3,80" "4,5,2,40"; do
  echo "== MH_CYCLE=$c"
  MH_CYCLE=$c python tools/scan_probe.py scan_s100k scan_s30k ball_s10k uvsphere_s10k --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x,1) for x in r['all_ms'][1:]], r.get('max_rel_err_vs_oracle'))"
done
