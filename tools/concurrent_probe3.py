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
        else: time.sleep(0.01)
ta, tb = threading.Thread(target=a), threading.Thread(target=b)
ta.start(); tb.start(); ta.join(); tb.join()
print(mode, "errors", len(errs), errs[:1])
