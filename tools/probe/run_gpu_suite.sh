python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/suite_tests.txt; cat gpurun_out/suite_tests.txt
python bench.py > gpurun_out/suite_bench.json 2> gpurun_out/suite_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/suite_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["profile"])
print([ (r["workload"], r["lobpcg_iterations"], round(r["ms"])) for r in d.get("scan_like",[])])
print([ (r["workload"], r["lobpcg_iterations"], round(r["ms"])) for r in d.get("config3",[])])
print(d.get("concurrent_solves"), d.get("batch64"))
PY
