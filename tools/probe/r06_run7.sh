python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_suite_tests.txt; cat gpurun_out/r06_suite_tests.txt
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err; tail -3 gpurun_out/r06_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_line.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["profile"])
print("combine", {k:d["roofline_combine"][k] for k in ("achieved","frac","avg_launch_us")}, d["roofline_combine"].get("full_size_update"))
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","cores","sample","seconds")})
print("cpu_small", d.get("cpu_baseline_small",{}).get("value"))
print([ (r["workload"], r["lobpcg_iterations"], round(r["ms"])) for r in d.get("scan_like",[])])
print([ (r["workload"], r["lobpcg_iterations"], round(r["ms"])) for r in d.get("config3",[])])
print(d.get("concurrent_solves"), d.get("batch64"), d.get("batch64_scan"))
PY
