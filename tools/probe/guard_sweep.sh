for g in 15 31 47 63; do
  echo "== MH_GUARDS=$g"
  MH_GUARDS=$g python tools/scan_probe.py scan_s100k scan_s30k cube_s100k ball_s10k --reps 1 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x) for x in r['all_ms']], round(r['factorize_ms'],1))"
done
