for c in "2,5,3,8" "2,7,2,8" "2,6,2,8" "2,8,2,8" "2,5,2,8" "2,4,3,8" "2,10,1,8" "2,3,4,8" "3,5,3,8" "2,5,3,6" "2,5,3,10"; do
  echo "== MH_CYCLE=$c"
  MH_CYCLE=$c python tools/scan_probe.py cube_s100k cube_s30k ball_s10k --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x,1) for x in r['all_ms'][1:]], r.get('max_rel_err_vs_oracle'))"
done
