import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mesheditor_amd import api, meshes
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()
ctx = api.Context(0)
m = meshes.MATERIALS["Ceramic"]
def weld(parts):
    pts = np.concatenate([p for p, _ in parts]); off = np.cumsum([0] + [len(p) for p, _ in parts[:-1]])
    tets = np.concatenate([t + o for (_, t), o in zip(parts, off)]).astype(np.uint32)
    key = np.round(pts * 1e9).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    return pts[first], inv.reshape(-1)[tets].astype(np.uint32)
a = meshes.kuhn_box(4, 4, 4, 0.08, 0.08, 0.08)
for name, org in (("vertex", (0.08, 0.08, 0.08)), ("edge", (0.08, 0.08, 0.0))):
    pts, tets = weld([a, meshes.kuhn_box(4, 4, 4, 0.08, 0.08, 0.08, origin=org)])
    sysg = api.System(ctx, api.Mesh(ctx, pts, tets), api.material(*m))
    syso = pyoracle.System(pts, tets, pyoracle.material(*m))
    K, M = sysg.to_scipy()
    Ko = syso.full(0)
    print(name, "n", sysg.n, "K vs oracle", abs(K - Ko).max() / abs(Ko).max(), "node order equal", np.array_equal(sysg.element_nodes(), syso.element_nodes()))
    n = K.shape[0]
    for d in range(3):
        t = np.zeros(n); t[d::3] = 1.0
        print("   translation", d, "|K t| / (|K| |t|)", np.abs(K @ t).max() / (abs(K).max() * 1.0))
    import scipy.sparse.linalg as sla
    w = np.linalg.eigvalsh(K.toarray())[:12] if n < 6000 else None
    print("   lowest eigenvalues of K / max:", None if w is None else w / abs(K).max())
