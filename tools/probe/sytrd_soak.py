"""Soak of the multi-workgroup tridiagonalisation's tagged exchange beside other work of the same process (round 5).
One thread runs lab/mh_soak.hip's restated kernel over and over on a fixed matrix, every collected value checked against a recorded
undisturbed run; the other threads run the disturbing work.

    python tools/probe/sytrd_soak.py <transport> <launches> <aggressor> [m]

transport: bit 0 one slot per 128-byte line; bits 1-2 access (0 16-byte sc1, 1 8-byte agent atomics, 2 sc0 sc1, 3 sc1 + buffer_inv);
           bits 4-5 slot memory (0 hipMalloc, 1 uncached, 2 fine-grained); bits 6-7 LDS allocation bytes (0 51336, 1 53760, 2 65536, 3 64000)
aggressor: none | solveN (one thread solving N pairs over and over) | solveNxT (T threads) | lab names (see AGGRESSORS)"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes  # noqa: E402
from tools import lab  # noqa: E402

DEV = np.dtype([("launch", "<u4"), ("group", "<u4"), ("step", "<u4"), ("lane", "<u4"), ("kind", "<u4"), ("spins", "<u4"), ("xcc", "<u4"), ("pad", "<u4"), ("got", "<u8"), ("expected", "<u8"),
                ("reread_lo", "<u8"), ("reread_hi", "<u8")])
SHARED = np.dtype([("n_deviations", "<u4"), ("n_bad_launches", "<u4"), ("first_bad_launch", "<u4"), ("first_bad_index", "<u4"), ("n_gave_up", "<u4"), ("n_split_xcc", "<u4"), ("n_bad_split", "<u4"),
                   ("launches", "<u4"), ("xcc_of", "<u4", 16), ("bad_xcc_sets", "<u4", (16, 16)), ("wg_trace", "<u8", (16, 6)), ("bad_traces", "<u8", (16, 16, 6)), ("longest_gap", "<u8"), ("longest_gap_launch", "<u8"),
                   ("launches_with_moves", "<u8"), ("n_records", "<u4"), ("pad2", "<u4"), ("dev", DEV, 256)])


def run(transport, launches, aggressor, m=240, verbose=True):
    L = lab.lib()
    L.mhl_sytrd_soak.restype = C.c_int
    L.mhl_sytrd_soak.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_uint64]
    L.mhl_sytrd_soak_bytes.restype = C.c_uint64
    assert L.mhl_sytrd_soak_bytes() == SHARED.itemsize, (L.mhl_sytrd_soak_bytes(), SHARED.itemsize)
    victim_ctx = api.Context(0)
    out = np.zeros(1, dtype=SHARED)
    recorded, go, done = C.c_int(0), C.c_int(0), threading.Event()
    rc = [None]

    def victim():
        rc[0] = L.mhl_sytrd_soak(victim_ctx.h, transport, m, launches, C.byref(recorded), C.byref(go), out.ctypes.data_as(C.c_void_p), SHARED.itemsize)
        done.set()

    workers, solved = [], [0]
    if aggressor.startswith("solve"):
        spec = aggressor[5:].split("x")
        pairs, nthreads = int(spec[0]), int(spec[1]) if len(spec) > 1 else 1
        boxes = [meshes.jittered_box(12, 1000 + i) + (meshes.MATERIALS[meshes.MATERIAL_ORDER[i % 7]],) for i in range(6)]

        def solver(k):
            ctx = api.Context(0)
            i = k
            while not done.is_set():
                p, t, mat = boxes[i % len(boxes)]
                s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*mat))
                try:
                    s.eigs(pairs, residual_tol=1e-5)
                    solved[0] += 1
                except Exception as e:  # noqa: BLE001
                    print("aggressor solve failed:", repr(e)[:200])
                s.close()
                i += 1

        workers = [threading.Thread(target=solver, args=(k,)) for k in range(nthreads)]
    elif aggressor.startswith("hostmem"):
        mb = int(aggressor[7:] or 64)

        def churn(k):
            ctx = api.Context(0)
            while not done.is_set():
                big = np.random.rand(mb * 1000000 // 8)  # a fresh mmap'd array, pinned by the copy below, unmapped when freed
                mesh = api.Mesh(ctx, big[: big.size // 3 * 3].reshape(-1, 3), np.array([[0, 1, 2, 3]], dtype=np.uint32))
                mesh.close() if hasattr(mesh, "close") else None
                del big, mesh
                solved[0] += 1

        workers = [threading.Thread(target=churn, args=(0,))]
    elif aggressor.startswith("proc"):
        import subprocess
        pairs = int(aggressor[4:])
        child = subprocess.Popen([sys.executable, __file__, "--solver", str(pairs)])

        def reaper():
            done.wait()
            child.terminate()
            child.wait()

        workers = [threading.Thread(target=reaper)]
        time.sleep(8.0)  # the child has imported and built its first system
    elif aggressor.startswith("kind"):
        spec = aggressor[4:].split("x")
        kind, nthreads = int(spec[0]), int(spec[1]) if len(spec) > 1 else 1
        L.mhl_soak_aggressor.restype = C.c_int
        L.mhl_soak_aggressor.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        stop = C.c_int(0)

        def lab_work(k):
            ctx = api.Context(0)
            cnt = C.c_uint64(0)
            rc_a = L.mhl_soak_aggressor(ctx.h, kind, C.byref(stop), C.byref(cnt))
            if rc_a != 0:
                print("aggressor failed:", rc_a, (ctx.L.mh_last_error(ctx.h) or b"").decode())
            solved[0] += cnt.value

        def stopper():
            done.wait()
            stop.value = 1

        workers = [threading.Thread(target=lab_work, args=(k,)) for k in range(nthreads)] + [threading.Thread(target=stopper)]
    elif aggressor != "none":
        raise SystemExit(f"unknown aggressor {aggressor}")
    tv = threading.Thread(target=victim)
    tv.start()
    while not recorded.value and not done.is_set():
        time.sleep(0.01)
    [w.start() for w in workers]
    if workers:
        time.sleep(1.0)  # the aggressors have built their first systems
    t0 = time.time()
    go.value = 1
    tv.join()
    dt = time.time() - t0
    [w.join() for w in workers]
    if rc[0] != 0:
        print("victim failed:", rc[0], (victim_ctx.L.mh_last_error(victim_ctx.h) or b"").decode())
        return None
    o = out[0]
    print(f"transport {transport:#04x} m {m} aggressor {aggressor}: {o['launches']} launches in {dt:.1f} s ({solved[0]} aggressor solves), bad {o['n_bad_launches']} "
          f"(first launch {o['first_bad_launch']}, first index {o['first_bad_index']}), deviations {o['n_deviations']}, gave up {o['n_gave_up']}, "
          f"launches with workgroups on several XCCs {o['n_split_xcc']} (bad among them {o['n_bad_split']})")
    print(f"  longest pause between two steps of a workgroup: {o['longest_gap'] / 100:.1f} us (launch {o['longest_gap_launch']}); launches in which a workgroup's HW_ID changed: {o['launches_with_moves']}")
    if verbose and o["n_deviations"]:
        dev = o["dev"][: min(256, o["n_records"])]
        order = np.lexsort((dev["lane"], dev["group"], dev["step"], dev["launch"]))
        shown = 0
        last = None
        for d in dev[order]:
            key = (d["launch"], d["step"])
            if key != last and shown >= 12:
                break
            last = key
            shown += 1
            got, exp = np.array([d["got"]], dtype="<u8").view("<f8")[0], np.array([d["expected"]], dtype="<u8").view("<f8")[0]
            print(f"  launch {d['launch']} step {d['step']} group {d['group']} (xcc {d['xcc']}) lane {d['lane']} kind {d['kind']} spins {d['spins']} at +{d['pad'] / 100:.1f} us: got {got!r} ({d['got']:#018x}) expected {exp!r} "
                  f"({d['expected']:#018x}) reread {d['reread_lo']:#018x} {d['reread_hi']:#018x}")
        for i in range(min(16, o["n_bad_launches"])):
            print("  bad launch XCCs:", list(o["bad_xcc_sets"][i]))
            for g in range(16):
                t = o["bad_traces"][i][g]
                print(f"    group {g:2d}: longest pause {t[0] / 100:9.1f} us before step {t[1]:3d}; HW_ID {t[2]:#010x} -> {t[3]:#010x}, changes {t[4]}, first at step {t[5] if t[5] != 0xffffffff else -1}")
    if o["n_records"]:
        import json
        dev = o["dev"][: min(256, o["n_records"])]
        first = {}
        for d in dev:
            key = int(d["launch"])
            if key not in first or d["pad"] < first[key]["pad"]:
                first[key] = {"launch": key, "group": int(d["group"]), "step": int(d["step"]), "lane": int(d["lane"]), "kind": int(d["kind"]), "pad": int(d["pad"])}
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/soak_first_deviations.json", "w") as f:
            json.dump(sorted(first.values(), key=lambda r: r["launch"]), f)
    return o


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--solver":
    pairs = int(sys.argv[2])
    boxes = [meshes.jittered_box(12, 1000 + i) + (meshes.MATERIALS[meshes.MATERIAL_ORDER[i % 7]],) for i in range(6)]
    ctx = api.Context(0)
    i = 0
    while True:
        p, t, mat = boxes[i % len(boxes)]
        s = api.System(ctx, api.Mesh(ctx, p, t), api.material(*mat))
        s.eigs(pairs, residual_tol=1e-5)
        s.close()
        i += 1
elif __name__ == "__main__":
    transport = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    aggressor = sys.argv[3] if len(sys.argv) > 3 else "solve120"
    m = int(sys.argv[4]) if len(sys.argv) > 4 else 240
    run(transport, launches, aggressor, m)
