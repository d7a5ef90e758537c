for q in 0.02 0.05 0.1 0.2; do
  echo "== MH_PATCH_Q=$q"
  MH_PATCH_Q=$q python tools/scan_probe.py scan_s100k scan_s30k --reps 1 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x) for x in r['all_ms']], round(r['factorize_ms'],1))"
done
