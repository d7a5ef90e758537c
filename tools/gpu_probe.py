"""Ad-hoc GPU probe: assembly parity, eigensolve parity and timings across sizes. Run on the GPU box."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesheditor_amd import api, meshes
from oracle import pyoracle as po

def main():
    ctx = api.Context(0)
    names = sys.argv[1:] or ["cube_small", "bar_square", "cube_s10k", "cube_s30k", "cube_s100k"]
    for name in names:
        p, t, m, kw = meshes.workload(name)
        mat = api.material(*m)
        t0 = time.time()
        mesh = api.Mesh(ctx, p, t)
        t1 = time.time()
        sysd = api.System(ctx, mesh, mat)
        t2 = time.time()
        print(f"[{name}] tets {len(t)} dofs {sysd.n} blocks {sysd.node_blocks} upload {t1-t0:.3f}s assemble {t2-t1:.3f}s", flush=True)
        if len(t) <= 4000:
            so = po.System(p, t, po.material(*m))
            assert (so.element_nodes() == sysd.element_nodes()).all(), "element nodes differ"
            K, M = sysd.to_scipy()
            Ko, Mo = so.full(0), so.full(1)
            dk = abs(K - Ko).max() / abs(Ko).max(); dm = abs(M - Mo).max() / abs(Mo).max()
            print(f"   assembly parity: K {dk:.2e} M {dm:.2e}", flush=True)
            x = np.random.default_rng(0).standard_normal((sysd.n, 3))
            y = sysd.matvec(0, x)
            print("   spmm parity", abs(y - Ko @ x).max() / abs(y).max(), flush=True)
        nev = kw.get("num_fem_modes", 45)
        for rep in range(2):
            t3 = time.time()
            try:
                ev, prof = sysd.eigs(nev, residual_tol=1e-6)
            except Exception as e:
                print("   eigs failed:", e, flush=True); break
            t4 = time.time()
            print(f"   eigs nev {nev}: {t4-t3:.3f}s iters {prof['restarts']} factorize {prof['factorize']:.3f} iterate {prof['iterate']:.3f} prec {prof['op_solve']:.3f}", flush=True)
        f = np.sqrt(np.maximum(ev, 0)) / (2 * np.pi)
        print("   freqs", np.round(f[:12], 2), flush=True)
        if len(t) <= 4000:
            evo, _, _ = so.eigs(nev, vectors=False)
            rel = np.abs(ev - evo) / np.maximum(np.abs(evo), (2*np.pi*20)**2)
            print(f"   eigenvalue parity vs oracle: {rel.max():.2e}", flush=True)
        sysd.close(); mesh.close()

if __name__ == "__main__":
    main()
