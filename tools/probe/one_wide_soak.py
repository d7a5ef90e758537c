"""One thread solving wide blocks (120 pairs) beside two threads solving narrow ones (65 pairs): does a wide solve need ANOTHER wide solve
beside it to be disturbed?   python tools/probe/one_wide_soak.py [wide threads]"""
import sys, threading, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
nwide = int(sys.argv[1]) if len(sys.argv) > 1 else 1
boxes = [meshes.jittered_box(12, 1000 + i) + (meshes.MATERIALS[meshes.MATERIAL_ORDER[i % 7]],) for i in range(30)]
def solve(ctx, i, pairs):
    p, t, mat = boxes[i]
    s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*mat))
    ev, _ = s.eigs(pairs, residual_tol=1e-5)
    s.close()
    return ev
c0 = api.Context(0)
ref = {pairs: [solve(c0, i, pairs) for i in range(30)] for pairs in (120, 65)}
bad, errs = [], []
def work(k, ctx):
    pairs = 120 if k < nwide else 65
    try:
        for i in range(30):
            if not np.array_equal(solve(ctx, i, pairs), ref[pairs][i]): bad.append((k, pairs, i))
    except Exception as e:  # noqa: BLE001
        errs.append((k, pairs, repr(e)[:160]))
ctxs = [api.Context(0) for _ in range(3)]
th = [threading.Thread(target=work, args=(k, ctxs[k])) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
print(f"{nwide} wide thread(s) of 3: errors {errs}; differing {bad}")
