"""Three PROCESSES (own HIP runtime each, so the library's in-process phase lock plays no part) solving the same 15 boxes at once with 120
pairs each; the three result sets must agree bit for bit.  Tells a device-level disturbance (fails here too) from a host-side one in this
library or its runtime (would pass here).   python tools/probe/multiproc_soak.py [pairs] [boxes]"""
import os, subprocess, sys, json
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    from mesheditor_amd import api, meshes
    pairs, boxes, tag = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    ctx = api.Context(0)
    out, errs = {}, []
    for i in range(boxes):
        p, t = meshes.jittered_box(12, 1000 + i)
        mat = meshes.MATERIALS[meshes.MATERIAL_ORDER[i % 7]]
        try:
            s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*mat))
            ev, _ = s.eigs(pairs, residual_tol=1e-5)
            s.close()
            out[str(i)] = [float(v).hex() for v in ev]
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)[:200]))
    json.dump({"out": out, "errs": errs}, open("/tmp/multiproc_%s.json" % tag, "w"))
    sys.exit(0)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
boxes = int(sys.argv[2]) if len(sys.argv) > 2 else 15
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(pairs), str(boxes), str(k)]) for k in range(3)]
[p.wait() for p in procs]
res = [json.load(open("/tmp/multiproc_%d.json" % k)) for k in range(3)]
errs = [r["errs"] for r in res]
diff = [i for i in res[0]["out"] if any(res[k]["out"].get(i) != res[0]["out"][i] for k in (1, 2))]
print(f"{pairs} pairs, {boxes} boxes, 3 processes: errors {errs}; boxes whose results differ between processes: {diff}")
