"""Concurrent solves (MH_CONCURRENT_SOLVES=1): T host threads, each with its own context, solve a share of a batch of
jittered boxes; every result is compared with the serial solve of the same mesh, and the throughput with the serial loop.

    MH_CONCURRENT_SOLVES=1 python tools/concurrent_solves.py [threads] [meshes] [n] [pairs]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesheditor_amd import api, meshes
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = int(sys.argv[3]) if len(sys.argv) > 3 else 17
nev = int(sys.argv[4]) if len(sys.argv) > 4 else 45  # (215: the wide Rayleigh-Ritz kernels and the library pieces beside them run concurrently)
batch = [meshes.jittered_box(n, 1000 + i) + (meshes.MATERIALS[meshes.MATERIAL_ORDER[i % 7]],) for i in range(N)]
ctxs = [api.Context(0) for _ in range(T)]


def solve(ctx, i):
    p, t, mat = batch[i]
    s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*mat))
    ev, prof = s.eigs(nev, residual_tol=1e-5)
    s.close()
    return ev


solve(ctxs[0], 0)
t0 = time.perf_counter()
ref = [solve(ctxs[0], i) for i in range(N)]
serial = time.perf_counter() - t0
out, errs = {}, []


def work(k):
    try:
        for i in range(k, N, T):
            out[i] = solve(ctxs[k], i)
    except Exception as e:  # noqa: BLE001
        errs.append(repr(e)[:200])


t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
[t.start() for t in th]
[t.join() for t in th]
par = time.perf_counter() - t0
bad = [i for i in range(N) if i not in out or not np.array_equal(out[i], ref[i])]
worst = max((np.abs(out[i][6:] / ref[i][6:] - 1).max() for i in out), default=0.0)
print(f"threads {T} meshes {N} ({len(batch[0][1])} tets): serial {serial:.2f} s, concurrent {par:.2f} s, speed-up {serial / par:.2f}; errors {errs}; "
      f"not bit-identical {len(bad)}; worst relative eigenvalue difference {worst:.2e}")
