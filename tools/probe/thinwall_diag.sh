for v in "X=0" "MH_CYCLE=5,10,6,60" "MH_CYCLE=8,5,3,100" "MH_CYCLE=10,10,6,200" "MH_CYCLE=5,5,3,30" "MH_PRECOND_FP64=1" "MH_PATCH_Q=0.03" "MH_AGG=8"; do
  echo "== $v"
  env $v python tools/scan_probe.py scan_s100k scan_s30k --reps 1 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], [round(x) for x in r['all_ms']], round(r['factorize_ms'],1))"
done
