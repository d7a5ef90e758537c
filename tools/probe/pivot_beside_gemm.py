"""The coarse set-up's pivot inverse (one workgroup, 96 registers x 1 024 threads, 2 KB of LDS) alone and beside the library's fp64 GEMM
or an HBM-bound kernel on another context.  (Round 5: 123.3 us in all cases before the kernel's rewrite, 92.4 after; a temporary switch that
reserved 144 KB of LDS for the workgroup changed nothing and is gone: docs/LAB_NOTEBOOK.md section 12.)"""
import sys, os, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from mesheditor_amd import api
import lab
a_ctx, b_ctx = api.Context(0), api.Context(0)
rng = np.random.default_rng(3)
w = 128
b = rng.standard_normal((w, w)); a = b @ b.T + w * np.eye(w)
inv, _ = lab.spd_inverse(a_ctx, a, reps=3)
print("inverse error", np.abs(inv @ a - np.eye(w)).max())
alone = [lab.spd_inverse(a_ctx, a, reps=200)[1] * 1e3 for _ in range(3)]
print("alone: us per call", [round(x, 1) for x in alone], flush=True)
for kind, label, args in ((0, "Gram 256 x 256 over 500k rows (library dgemm)", (500000, 256, 256)), (1, "combine 240 cols over 542k rows (HBM-bound)", (542000, 240, 80))):
    stop = False
    def load():
        while not stop:
            lab.bench_dense(b_ctx, kind, *args, reps=20)
    t = threading.Thread(target=load); t.start()
    time.sleep(0.5)
    beside = [lab.spd_inverse(a_ctx, a, reps=200)[1] * 1e3 for _ in range(3)]
    stop = True; t.join()
    print(f"beside {label}: us per call", [round(x, 1) for x in beside], flush=True)
