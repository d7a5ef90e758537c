"""Tiny systems (solved densely under the exclusive device phase) beside iterating solves, from several threads: no deadlock, every result
equal to its serial one.   python tools/probe/tiny_soak.py"""
import sys, threading, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
big = meshes.workload("cube_s10k")
tiny = [meshes.jittered_box(2, 50 + i) for i in range(4)]  # a few dozen tets each: n < 768 -> one dense eigensolve
mat = meshes.MATERIALS["Steel"]
def solve(ctx, pts, tets, m, pairs):
    mesh = api.Mesh(ctx, pts, tets)
    s = api.System(ctx, mesh, api.material(*m))
    ev, prof = s.eigs(pairs, -(2 * np.pi * 20.0) ** 2, 1e-6)
    s.close(); mesh.close()
    return ev
c0 = api.Context(0)
ref_big = solve(c0, big[0], big[1], big[2], 65)
ref_tiny = [solve(c0, p, t, mat, 12) for p, t in tiny]
print("tiny systems:", [3 * (len(np.unique(t)) ) for p, t in tiny][:1], "P1 dofs (P2 more); big", len(big[1]), "tets", flush=True)
out, errs = [], []
def work(k, ctx):
    try:
        for rep in range(12):
            if k == 0:
                out.append(("big", np.array_equal(solve(ctx, big[0], big[1], big[2], 65), ref_big)))
            else:
                i = (rep + k) % 4
                out.append(("tiny", np.array_equal(solve(ctx, tiny[i][0], tiny[i][1], mat, 12), ref_tiny[i])))
    except Exception as e:  # noqa: BLE001
        errs.append(repr(e)[:300])
ctxs = [api.Context(0) for _ in range(4)]
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k, ctxs[k])) for k in range(4)]
[t.start() for t in th]; [t.join(timeout=300) for t in th]
alive = [t.is_alive() for t in th]
print(f"{len(out)} solves in {time.perf_counter() - t0:.1f} s; threads still running {alive}; errors {errs}; results differing from serial: {[k for k, ok in out if not ok]}", flush=True)
import os; os._exit(0 if not any(alive) else 3)
