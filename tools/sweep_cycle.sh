# preconditioner cycle shape sweep through bench.py (ms per step, iterations)
run() { env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],1), d['config']['lobpcg_iterations'])"; }
run MH_NOP=1
run MH_DEG2=1
run MH_DEG2=3
run MH_DEG1=3
run MH_DEG1=6
run MH_GAMMA=2
run MH_GAMMA=4
run MH_DEG2=3 MH_GAMMA=2
run MH_AGG=16
run MH_AGG=64
run MH_GUARD_ABS=7
run MH_GUARD_ABS=31
