import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesheditor_amd import api, meshes
pts, tets, m, kw = meshes.workload("cube_s30k")
mat = api.material(*m)
mode = sys.argv[1]
ctxA, ctxB = api.Context(0), api.Context(0)
meshA, meshB = api.Mesh(ctxA, pts, tets), api.Mesh(ctxB, pts, tets)
sB = api.System(ctxB, meshB, mat)
errs, stop = [], False
bank_scene = None
if mode == "bank":
    from tools import bank_bench
    from mesheditor_amd import bank as hipbank
    bank_scene = hipbank.Scene(48000.0, 0)
    bank_scene.set_renderers(4)
    pos = np.array([[p * 0.01, 0.0, 0.02 if p % 2 else 0.0] for p in range(4)], np.float32)
    idx = np.array([[p, p + 1, p + 2] for p in range(2)], np.uint32).reshape(-1)
    for o in range(256):
        f, t, sh = bank_bench.modes_for(o, 256)
        slot = bank_scene.add_object(o, sh, pos, idx)
        bank_scene.tune_object(slot, f, t)
        bank_scene.set_gains(slot, 1.0, 1.0)
    bank_scene.install()
    out_block = np.zeros(512, np.float32)
    bank_scene.render(out_block)
def a():
    global stop
    for _ in range(10):
        s = api.System(ctxA, meshA, mat)
        try:
            s.eigs(45, max_iters=0)
        except Exception as e:
            if "EFACTOR" in str(e): errs.append(str(e)[:90])
        s.close()
    stop = True
def b():
    while not stop:
        if mode == "spmm": sB.bench_spmm(64, 50)
        elif mode == "gram": ctxB.bench_dense(0, 128625, 64, 64, 50)
        elif mode == "combine": ctxB.bench_dense(1, 128625, 64, 64, 20)
        elif mode == "assemble": api.System(ctxB, meshB, mat).close()
        elif mode == "bank":
            for o in range(0, 256, 4):
                bank_scene.L.mhx_enqueue(bank_scene.h, hipbank.Event(0, o, 0, 1.0, 0.5, 0.0, 1.0 / 300.0, 20.0, 0.0, 0.0, 0.0, 0.0))
            for _ in range(20): bank_scene.render(out_block)
        else: time.sleep(0.01)
ta, tb = threading.Thread(target=a), threading.Thread(target=b)
ta.start(); tb.start(); ta.join(); tb.join()
print(mode, "errors", len(errs), errs[:1])
