"""Bad and odd surfaces through the tetrahedraliser (host code): each must come back as a valid fill or as an error message, never
hang or crash.   python tools/probe/tet_fuzz.py"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import tets as T, meshes

def ico(sub=2, r=1.0, c=(0, 0, 0)):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    for _ in range(sub):
        cache, nf = {}, []
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                p = v[a] + v[b]; v.append(p / np.linalg.norm(p)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c_ in f:
            ab, bc, ca = mid(a, b), mid(b, c_), mid(c_, a)
            nf += [(a, ab, ca), (b, bc, ab), (c_, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v) * r + np.array(c, float), np.array(f, np.uint32)

def run(name, p, t):
    t0 = time.perf_counter()
    try:
        r = T.tetrahedralize(p, t)
        pts, tt = r[0], r[1]
        vol = np.abs(np.einsum("ij,ij->i", np.cross(pts[tt[:, 1]] - pts[tt[:, 0]], pts[tt[:, 2]] - pts[tt[:, 0]]), pts[tt[:, 3]] - pts[tt[:, 0]])).sum() / 6
        print(f"{name:34s} ok: {len(pts)} points, {len(tt)} tets, volume {vol:.4f}  ({time.perf_counter() - t0:.2f} s)", flush=True)
    except Exception as e:
        print(f"{name:34s} error: {str(e)[:120]}  ({time.perf_counter() - t0:.2f} s)", flush=True)

p, t = ico(2)
run("icosphere", p, t)
run("one triangle missing (open)", p, t[1:])
run("one triangle flipped", p, np.vstack([t[:1, ::-1], t[1:]]))
run("all triangles flipped (inside out)", p, t[:, ::-1].copy())
run("duplicate triangle", p, np.vstack([t, t[:1]]))
run("unused extra point", np.vstack([p, [[5, 5, 5]]]), t)
q = p.copy(); q[3] = q[7]
run("two points coincide", q, t)
run("zero-area triangle added", p, np.vstack([t, [[0, 0, 1]]]).astype(np.uint32))
p2, t2 = ico(2, 0.4, (3, 0, 0))
run("two disjoint spheres", np.vstack([p, p2]), np.vstack([t, t2 + len(p)]).astype(np.uint32))
p3, t3 = ico(2, 0.4)
run("nested sphere (cavity, inward)", np.vstack([p, p3]), np.vstack([t, t3[:, ::-1] + len(p)]).astype(np.uint32))
run("nested sphere (same orientation)", np.vstack([p, p3]), np.vstack([t, t3 + len(p)]).astype(np.uint32))
run("flat (all points coplanar)", p * [1, 1, 0], t)
run("needle (scaled 1e-4 in z)", p * [1, 1, 1e-4], t)
run("huge coordinates (x 1e6)", p * 1e6, t)
run("tiny coordinates (x 1e-6)", p * 1e-6, t)
run("index out of range", p, np.vstack([t, [[0, 1, 9999]]]).astype(np.uint32))
run("empty", np.zeros((0, 3)), np.zeros((0, 3), np.uint32))
run("single tetrahedron surface", np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]), np.array([[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]], np.uint32))
rng = np.random.default_rng(1)
run("NaN coordinate", np.where(rng.random(p.shape) < 0.01, np.nan, p), t)
