"""Does a one-CU serial kernel (k_sytrd_regs at order 222: LDS round trips and barriers, nothing in memory) slow down while the
library's fp64 GEMM runs on the rest of the device from another context?  (The pivot inverse of the coarse set-up runs 199 us per
call beside the rank-128 updates against 68-75 alone; reserving the CU's LDS changed nothing.)"""
import sys, os, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from mesheditor_amd import api
import lab
a_ctx, b_ctx = api.Context(0), api.Context(0)
rng = np.random.default_rng(3)
m = 222
b = rng.standard_normal((m, m)); a = b + b.T
lab.tridiagonalize(a_ctx, a, variant=3, reps=5)
alone = [lab.tridiagonalize(a_ctx, a, variant=3, reps=200)[2] * 1e3 for _ in range(3)]
print("alone: us per call", [round(x, 1) for x in alone], flush=True)
for kind, label, args in ((0, "Gram 256 x 256 over 500k rows (library dgemm)", (500000, 256, 256)), (1, "combine 240 cols over 542k rows (HBM-bound)", (542000, 240, 80))):
    stop = False
    def load():
        while not stop:
            lab.bench_dense(b_ctx, kind, *args, reps=20)
    t = threading.Thread(target=load); t.start()
    time.sleep(0.5)
    beside = [lab.tridiagonalize(a_ctx, a, variant=3, reps=200)[2] * 1e3 for _ in range(3)]
    stop = True; t.join()
    print(f"beside {label}: us per call", [round(x, 1) for x in beside], flush=True)
