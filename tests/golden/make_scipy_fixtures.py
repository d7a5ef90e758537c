"""Generates tests/golden/scipy_eigs.json (BUILD CONTAINER ONLY): SciPy's ARPACK shift-invert eigenvalues -- an independent
third-party solver (ARPACK + SuperLU), not the reference -- of the pencils the oracle's CPU assembly exports, so that the 1e-9
anchor of the oracle's own Cholesky + Lanczos restatement (pin G8) is committed data rather than a live computation.

    python tests/golden/make_scipy_fixtures.py
"""
import json
import os
import sys

import numpy as np
import scipy
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SIGMA = -(2 * np.pi * 20.0) ** 2
CASES = (("cube_small", 30), ("bar_thin", 30), ("bar_square", 40), ("cube_s10k", 65))

if __name__ == "__main__":
    from mesheditor_amd import meshes
    from oracle import pyoracle as po
    out = {"generator": "tests/golden/make_scipy_fixtures.py", "solver": "scipy %s eigsh(K, k, M=M, sigma=-(2 pi 20)^2, which='LM', tol=1e-12)" % scipy.__version__,
           "sigma": SIGMA, "cases": {}}
    for name, nev in CASES:
        pts, tets, m, _ = meshes.workload(name)
        s = po.System(pts, tets, po.material(*m))
        K, M = s.full(0).tocsc(), s.full(1).tocsc()
        ev = np.sort(spla.eigsh(K, k=nev, M=M, sigma=SIGMA, which="LM", tol=1e-12)[0])
        out["cases"][name] = {"nev": nev, "dofs": int(s.n), "eigenvalues": [float(v) for v in ev]}
        print(name, nev, s.n, ev[6:9], flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scipy_eigs.json"), "w") as f:
        json.dump(out, f)
        f.write("\n")
