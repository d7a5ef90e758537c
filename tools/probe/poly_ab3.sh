python tools/scan_probe.py skillet_s100k cube_s100k cube_s30k --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], r['all_ms'], r.get('max_rel_err_vs_oracle'))"
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
