"""A caller's own mesh with flat cells through the solver under different settings, one subprocess each (the switches are read once): a fine UV sphere's
default fill with 60 interior points moved almost into a face of one of their tetrahedra (meshes.with_flat_cells: 60+ cells flat to 1e-9), or -- with `raw` --
its RAW constrained-Delaunay fill without any repair pass (every seventh cell flat: minutes on the device, kept for the record).
    python tools/probe/flat_sphere_probe.py [seg rings] [raw] [ENV=VALUE,ENV=VALUE ... | default]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, time, numpy as np
sys.path.insert(0, %r)
from mesheditor_amd import meshes, tets as T, api
seg, rings = %d, %d
P, F = meshes.uv_sphere_surface(0.15, seg, rings)
if %d:
    pts, tets, left = T.tetrahedralize(P, F, repair_slivers=False)  # the raw constrained-Delaunay fill: no repair pass at all (every seventh cell flat on a fine UV sphere)
else:
    pts, tets, left = T.tetrahedralize(P, F)
    pts, made = meshes.with_flat_cells(pts, tets, len(P), count=60, eps=float(os.environ.get("FLAT_EPS", "1e-6")), seed=seg)
ctx = api.Context(0)
m = meshes.MATERIALS["Ceramic"]
ex = pts[(np.arange(10) * len(P)) // 10].astype(np.float32)
t0 = time.time()
r = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(num_modes=50, num_fem_modes=65))
ctx.synchronize()
print("RESULT %%d tets: %%d pairs, %%s iterations, %%.0f ms, f7 %%.2f Hz" %% (len(tets), len(r.eigenvalues), r.profile.get("restarts"), 1e3 * (time.time() - t0), np.sqrt(max(r.eigenvalues[6], 0)) / 2 / np.pi if len(r.eigenvalues) > 6 else 0), flush=True)
'''

if __name__ == "__main__":
    raw = "raw" in sys.argv[1:]
    args = [a for a in sys.argv[1:] if "=" not in a and a not in ("default", "raw")]
    seg, rings = (int(args[0]), int(args[1])) if len(args) >= 2 else (128, 64)
    settings = [("" if a == "default" else a) for a in sys.argv[1:] if "=" in a or a == "default"] or [""]
    for s in settings:
        env = dict(os.environ, MH_VERBOSE="1")
        for kv in filter(None, s.split(",")):
            k, v = kv.split("=", 1)
            env[k] = v
        p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, seg, rings, int(raw))], env=env, capture_output=True, text=True, timeout=900)
        lines = [l for l in p.stderr.splitlines() if ("lobpcg]" in l and "Cholesky-QR" not in l) or "Error" in l or "error" in l]
        print("== %dx%d [%s]" % (seg, rings, s or "default"))
        for l in p.stdout.splitlines():
            if l.startswith("RESULT"):
                print("   ", l)
        keep = lines[:3] + [l for l in lines[3:-6] if " conv " not in l or int(l.split(" it ")[1].split()[0]) % 10 == 0] + lines[-6:]
        for l in keep:
            print("   ", l[:240])
        sys.stdout.flush()
