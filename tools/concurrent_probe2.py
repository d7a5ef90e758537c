import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesheditor_amd import api, meshes
pts, tets, m, kw = meshes.workload(os.environ.get("WL", "cube_s30k"))
mat = api.material(*m)
mode = sys.argv[1]
S = 3
ctxs = [api.Context(0) for _ in range(S)]
mesh = [api.Mesh(c, pts, tets) for c in ctxs]
lock = threading.Lock()
errs = []
def work(i):
    try:
        for _ in range(4):
            if mode == "lock_assemble":
                with lock:
                    s = api.System(ctxs[i], mesh[i], mat); ctxs[i].synchronize()
            else:
                s = api.System(ctxs[i], mesh[i], mat)
            if mode == "lock_eigs":
                with lock:
                    ev, prof = s.eigs(45); ctxs[i].synchronize()
            elif mode == "assemble_only":
                ctxs[i].synchronize()
            elif mode == "setup_only":
                try:
                    s.eigs(45, max_iters=0)
                except Exception as e:
                    if "EFACTOR" in str(e):
                        raise
            else:
                ev, prof = s.eigs(45)
            s.close()
    except Exception as e:
        errs.append(str(e)[:120])
s0 = api.System(ctxs[0], mesh[0], mat); s0.eigs(45); s0.close()
th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
[t.start() for t in th]; [t.join() for t in th]
print(mode, "errors:", len(errs), errs[:2])
