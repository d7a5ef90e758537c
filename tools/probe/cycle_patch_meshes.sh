for c in "5,5,3,60" "5,16,1,60,250" "5,12,2,60,250"; do
  echo "== MH_CYCLE=$c"
  MH_CYCLE=$c MH_VERBOSE=1 python tools/scan_probe.py ball_s10k scan_s30k_repaired scan_s100k_repaired config3_s30k_repaired config3_s100k_repaired scan_s30k scan_s100k config3_s30k --reps 2 2>&1 | grep -E "workload|sliver patches" | python -c "import sys,json,re
p=None
for l in sys.stdin:
    m=re.search(r'sliver patches (\d+)', l)
    if m: p=m.group(1); continue
    if l.startswith('{'):
        r=json.loads(l); print(r['workload'], 'patches', p, 'pairs', r['eigenpairs'], 'iterations', r['iterations'], [round(x,1) for x in r['all_ms'][1:]])"
done
