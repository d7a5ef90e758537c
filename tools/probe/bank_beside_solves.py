"""The bank rendering blocks on one thread while two others solve: the signal and the eigenvalues must equal their solo runs, bit for bit.
    python tools/probe/bank_beside_solves.py"""
import sys, threading, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes, bank as hipbank
from tests import bank_harness as bh
class O:
    Event = lambda *a: hipbank.Event(*a)
def bank_run(blocks=60):
    sc = bh.DeviceScene(64, 128, 0.5, 4)
    for o in sc.objects: sc.enqueue(bh.impact_event(O, o, 1.0, click=False))
    a = sc.render(blocks // 2, bh.BLOCK)
    for o in sc.objects[::3]: sc.enqueue(bh.impact_event(O, o, -0.3, 1, 1.0 / 120.0, click=False))
    b = sc.render(blocks - blocks // 2, bh.BLOCK)
    return np.concatenate([a, b])
pts, tets, m, _ = meshes.workload("cube_s30k")
def solve(ctx, pairs):
    mesh = api.Mesh(ctx, pts, tets); s = api.System(ctx, mesh, api.material(*m))
    ev, _ = s.eigs(pairs, -(2 * np.pi * 20.0) ** 2, 1e-6); s.close(); mesh.close(); return ev
c0 = api.Context(0)
ref_sig = bank_run(); ref65 = solve(c0, 65); ref140 = solve(c0, 140)
res, errs = {}, []
def t_bank():
    try: res["sig"] = [np.array_equal(bank_run(), ref_sig) for _ in range(3)]
    except Exception as e: errs.append(repr(e)[:300])
def t_solve(k, pairs, ref):
    try:
        ctx = api.Context(0)
        res[k] = [np.array_equal(solve(ctx, pairs), ref) for _ in range(6)]
    except Exception as e: errs.append(repr(e)[:300])
th = [threading.Thread(target=t_bank), threading.Thread(target=t_solve, args=("s65", 65, ref65)), threading.Thread(target=t_solve, args=("s140", 140, ref140))]
t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
print(f"{time.perf_counter() - t0:.1f} s; errors {errs}; bank signal equal {res.get('sig')}; 65 pairs equal {res.get('s65')}; 140 pairs equal {res.get('s140')}; max |signal| {np.abs(ref_sig).max():.3e}")
