python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r05.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, "frac", d["roofline"]["frac"], d["roofline"].get("measured_ceiling"))
print("combine", {k:d["roofline_combine"][k] for k in ("achieved","frac","launches","avg_launch_us","ms_per_step","hbm_GBps_at_that_time")})
print(d["profile"])
print([ (r["workload"], r["lobpcg_iterations"], round(r["ms"])) for r in d.get("scan_like",[])])
print([ (r["workload"], r["lobpcg_iterations"], round(r["ms"])) for r in d.get("config3",[])])
print(d.get("concurrent_solves"), d.get("batch64"))
print(d["resonator_bank"]["all_live"]["ms_per_block"], d["cpu_baseline"]["value"])
PY
