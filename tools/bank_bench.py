"""Resonator-bank benchmark: 1024 objects x 256 modes at 48 kHz, 512-frame blocks,
every object struck at block 0 and every 64th block (256 strikes per block over four blocks: the event ring holds 256).
Prints ms per block, x real time and mode-samples/s of the device render through the C++ mirror's RenderModal."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import bank as hipbank

SR, BLOCK, POINTS = 48000.0, 512, 4


def modes_for(o, n_modes):
    k = np.arange(n_modes, dtype=np.float32)
    freqs = (np.float32(40.0) * (k + 1) * np.float32(1.031) * np.float32(1.0 + 0.001 * o)).astype(np.float32)
    t60s = (np.float32(2.0) / (k + 1)).astype(np.float32)
    shapes = np.zeros((POINTS, n_modes, 3), np.float32)
    for p in range(POINTS):
        a = ((k + 1) * np.float32(0.37) + np.float32(p)).astype(np.float32)
        shapes[p] = np.stack([np.sin(a), np.cos(a * np.float32(1.7)), np.sin(a * np.float32(2.3))], -1) * np.float32(0.01)
    return freqs, t60s, shapes


def run(objects=1024, modes=256, blocks=192, renderers=4):
    a = argparse.Namespace(objects=objects, modes=modes, blocks=blocks, renderers=renderers)
    pos = np.array([[p * 0.01, 0.0, 0.02 if p % 2 else 0.0] for p in range(POINTS)], np.float32)
    idx = np.array([[p, p + 1, p + 2] for p in range(POINTS - 2)], np.uint32).reshape(-1)
    sc = hipbank.Scene(SR, 0)
    sc.set_renderers(a.renderers)
    for o in range(a.objects):
        f, t, sh = modes_for(o, a.modes)
        slot = sc.add_object(o, sh, pos, idx)
        sc.tune_object(slot, f, t)
        sc.set_gains(slot, 1.0, 1.0)
    sc.install()
    out = np.zeros(BLOCK, np.float32)
    sc.render(out)

    def strike(block):
        phase = block % 64
        lo, hi = phase * 256, min(a.objects, (phase + 1) * 256)
        for o in range(lo, hi):
            ev = hipbank.Event(0, o, 0, 1.0, 0.5, 0.0, 1.0 / 300.0, 20.0, 0.0, 0.0, 0.0, 0.0)
            assert sc.L.mhx_enqueue(sc.h, ev)
    times, live_modes = [], []
    peak = 0.0
    for b in range(a.blocks):
        strike(b)
        out[:] = 0
        t0 = time.perf_counter()
        sc.render(out)
        times.append(time.perf_counter() - t0)
        peak = max(peak, float(np.abs(out).max()))
        tuned, live, ring = sc.object_state()
        live_modes.append(int(live[ring != 0].sum()))
    assert np.isfinite(peak) and peak > 0
    t = np.array(times[8:])
    ms = 1e3 * t.mean()
    sc.close()
    return ({"workload": f"bank {a.objects}x{a.modes} @48k, {BLOCK}-frame blocks, {a.renderers} renderers", "blocks": len(t),
                      "ms_per_block": ms, "ms_per_block_p99": 1e3 * float(np.quantile(t, 0.99)), "x_real_time": BLOCK / SR / t.mean(),
                      "mode_samples_per_s": a.objects * a.modes * BLOCK / t.mean(), "live_modes_mean": float(np.mean(live_modes[8:])),
                      "live_mode_samples_per_s": float(np.mean(live_modes[8:])) * BLOCK / t.mean(), "peak": peak})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=1024)
    ap.add_argument("--modes", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=192)
    ap.add_argument("--renderers", type=int, default=4)
    a = ap.parse_args()
    print(json.dumps(run(a.objects, a.modes, a.blocks, a.renderers)))


if __name__ == "__main__":
    main()
