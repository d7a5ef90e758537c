for c in "default" "5,5,3,60" "4,5,3,40" "3,5,3,20" "3,5,3,12"; do
  echo "== MH_CYCLE=$c"
  if [ "$c" = "default" ]; then unset MH_CYCLE; else export MH_CYCLE=$c; fi
  MH_VERBOSE=1 python tools/scan_probe.py cube_s100k cube_s30k uvsphere_s10k scan_s30k_repaired scan_s100k_repaired --reps 2 2>&1 | grep -E "workload|sliver patches" | python -c "import sys,json,re
p=None
for l in sys.stdin:
    m=re.search(r'sliver patches (\d+)', l)
    if m: p=m.group(1); continue
    if l.startswith('{'):
        r=json.loads(l); print(r['workload'], 'patches', p, 'iterations', r['iterations'], [round(x,1) for x in r['all_ms'][1:]])"
done
