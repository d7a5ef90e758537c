#!/bin/bash
# Scan of the preconditioner's knobs on the bench workload (GPU box): iterations, ms per solve, ms in the preconditioner.
run() { echo -n "$* : "; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['lobpcg_iterations'], round(d['ms_per_step'],1), round(d['profile']['op_solve']*1e3,1))"; }
for cfg in "$@"; do run $cfg; done
