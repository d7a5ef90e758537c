"""Generates tests/golden/oracle_eigs_<workload>.json: the CPU oracle's result for the named workloads of
mesheditor_amd/meshes.py (BUILD CONTAINER ONLY -- minutes of CPU per workload; the GPU tests read the committed JSON).

The oracle (oracle/analysis.cpp) restates the reference's mesh2modes path (src/audio/mesh2modes.cpp:441-512: shifted
operator, sparse Cholesky, shift-invert Lanczos at Tolerance 1e-8), so these are the eigenvalues the reference algorithm
gives on these meshes.  Each file also records the wall time per SolveProfile stage, the thread team, the host and the
command -- bench.py's `cpu_baseline_metric_mesh` object is read from the cube_s100k file.

    python tests/golden/make_oracle_fixtures.py cube_s100k skillet_s100k ball_s10k cube_s30k scan_s30k [--threads 8]
"""
import argparse
import datetime
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def excite_positions(pts, count=10):
    """P = 10 excitation positions = points i*V/10."""
    return pts[(np.arange(count) * len(pts)) // count].astype(np.float32)


def mesh_digest(pts, tets):
    """Identity of the generated mesh (the scan workloads go through numpy's generators and the tetrahedraliser: a test that
    compares with a fixture first checks that it solved the same mesh)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(pts, np.float64).tobytes() + np.ascontiguousarray(tets, np.uint32).tobytes()).hexdigest()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def run(name, threads):
    from mesheditor_amd import meshes
    from oracle import pyoracle as po
    pts, tets, m, kw = meshes.workload(name)
    po.set_threads(threads)
    cfg = po.default_config(**kw)
    ex = excite_positions(pts)
    t0 = time.perf_counter()
    r = po.mesh2modes(pts, tets, po.material(*m), ex, config=cfg)
    secs = time.perf_counter() - t0
    nev = len(r.eigenvalues)
    if nev == 0:
        raise RuntimeError(f"{name}: the oracle returned an empty result")
    rec = {
        "workload": name,
        "generator": "tests/golden/make_oracle_fixtures.py " + name + f" --threads {threads}",
        "algorithm": "oracle/analysis.cpp mo_mesh2modes (restates /root/reference/src/audio/mesh2modes.cpp:441-512, cold branch)",
        "date": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d"),
        "host": {"cpu": cpu_model(), "cores_available": po.available_cores(), "threads": threads},
        "mesh": {"tets": int(len(tets)), "points": int(len(pts)), "dof": int(r.profile["dofs"]), "sha256": mesh_digest(pts, tets)},
        "material": list(m),
        "config": {"num_modes": int(cfg.num_modes), "num_fem_modes": int(cfg.num_fem_modes), "tolerance": cfg.tolerance,
                   "min_mode_freq": cfg.min_mode_freq, "max_mode_freq": cfg.max_mode_freq},
        "seconds": secs,
        "eigenpairs_per_second": nev / secs,
        "profile": r.profile,
        "eigenvalues": [float(v) for v in r.eigenvalues],
        "freqs": [float(v) for v in r.freqs],
        "t60s": [float(v) for v in r.t60s],
        "original_fundamental": float(r.original_fundamental),
        "mass": r.mass,
        "center_of_mass": [float(v) for v in r.center_of_mass],
        "inertia_diagonal": [float(v) for v in r.inertia_diagonal],
        "inertia_orientation_wxyz": [float(v) for v in r.inertia_orientation_wxyz],
        "sample_point_of_excitation": [int(v) for v in r.sample_point_of_excitation],
        # the excitation points' shapes of every eigenpair, for sign-free comparisons on simple modes ([position][pair][3])
        "summary_shapes": np.round(r.summary_shapes.astype(np.float64), 9).tolist(),
    }
    out = os.path.join(HERE, f"oracle_eigs_{name}.json")
    with open(out, "w") as f:
        json.dump(rec, f)
        f.write("\n")
    print(f"{name}: {len(tets)} tets, {rec['mesh']['dof']} dof, {nev} pairs in {secs:.1f} s on {threads} threads -> {out}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="+")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    for w in a.workloads:
        run(w, a.threads)
