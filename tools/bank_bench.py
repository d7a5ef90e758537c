"""Resonator-bank benchmark: 1024 objects x 256 modes at 48 kHz, 512-frame blocks, through
the C++ mirror's RenderModal (libmodalhost -> libmodalhip), real click filters included.  Two phases, reported apart:

  all_live      every object is kept excited (a 2048-sample force pulse, re-struck every fourth block, 256 strikes per block --
                the event ring's capacity), so all 262 144 modes are rendered in every block: the worst case.
  steady_state  config 5's pattern: every object struck once at the start of each 64-block period (256 per block over four
                blocks) and left to decay, so audibility culling leaves a few per cent of the modes live on average.

The resonator kernel (k_bank_modes) is HIP-event timed inside the library; `roofline_bank` is its rendered mode-samples x 11
flop / kernel time against the fp32 vector peak (157.3 TFLOP/s, MI355X_MICROARCH.md) -- the kernel is issue/latency bound,
not HBM bound (160 flop/B)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import bank as hipbank  # noqa: E402

SR, BLOCK, POINTS = 48000.0, 512, 4
FP32_VECTOR_PEAK_TFLOPS = 157.3


def modes_for(o, n_modes):
    k = np.arange(n_modes, dtype=np.float32)
    freqs = (np.float32(40.0) * (k + 1) * np.float32(1.031) * np.float32(1.0 + 0.001 * o)).astype(np.float32)
    t60s = (np.float32(2.0) / (k + 1)).astype(np.float32)
    shapes = np.zeros((POINTS, n_modes, 3), np.float32)
    for p in range(POINTS):
        a = ((k + 1) * np.float32(0.37) + np.float32(p)).astype(np.float32)
        shapes[p] = np.stack([np.sin(a), np.cos(a * np.float32(1.7)), np.sin(a * np.float32(2.3))], -1) * np.float32(0.01)
    return freqs, t60s, shapes


def click_filter(sc, radius=0.05, mass=1.0):
    out = np.zeros(3, np.float32)
    sc.L.mhx_recoil_click_filter(radius, 4.0 / 3.0 * np.pi * radius ** 3, mass, SR, out.ctypes.data)
    return out


def build(objects, modes, renderers):
    pos = np.array([[p * 0.01, 0.0, 0.02 if p % 2 else 0.0] for p in range(POINTS)], np.float32)
    idx = np.array([[p, p + 1, p + 2] for p in range(POINTS - 2)], np.uint32).reshape(-1)
    sc = hipbank.Scene(SR, 0)
    sc.set_renderers(renderers)
    for o in range(objects):
        f, t, sh = modes_for(o, modes)
        slot = sc.add_object(o, sh, pos, idx)
        sc.tune_object(slot, f, t)
        sc.set_gains(slot, 1.0, 1.0)
    sc.install()
    sc.render(np.zeros(BLOCK, np.float32))
    return sc


def phase(sc, blocks, skip, strike):
    """Render `blocks` blocks, calling strike(block) before each; statistics over the blocks after the first `skip`."""
    out = np.zeros(BLOCK, np.float32)
    times, live_modes, peak = [], [], 0.0
    sc.time_kernels(True)
    for b in range(blocks):
        if b == skip:
            sc.time_kernels(True)  # restart the kernel totals where the statistics start
        strike(b)
        out[:] = 0
        t0 = time.perf_counter()
        sc.render(out)
        times.append(time.perf_counter() - t0)
        peak = max(peak, float(np.abs(out).max()))
        tuned, live, ring = sc.object_state()
        live_modes.append(int(live[ring != 0].sum()))
    k = sc.kernel_stats(2)
    sc.time_kernels(False)
    assert np.isfinite(peak) and peak > 0
    t = np.array(times[skip:])
    flops_rate = k["work"] / (k["total_ms"] * 1e-3) / 1e12 if k["total_ms"] > 0 else 0.0
    return {"blocks": len(t), "ms_per_block": 1e3 * float(t.mean()), "ms_per_block_p99": 1e3 * float(np.quantile(t, 0.99)), "x_real_time": BLOCK / SR / float(t.mean()),
            "live_modes_mean": float(np.mean(live_modes[skip:])), "rendered_mode_samples_per_s": k["work"] / 11.0 / float(t.sum()),
            "kernel_us_per_block": 1e3 * k["total_ms"] / max(1, k["launches"]), "peak": peak,
            "roofline_bank": {"bound": "fp32 issue", "kernel": "k_bank_modes<float>", "achieved": flops_rate, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": flops_rate / FP32_VECTOR_PEAK_TFLOPS, "flop_per_mode_sample": 11}}


def run(objects=1024, modes=256, blocks=72, renderers=4):
    sc = build(objects, modes, renderers)
    sc.L.mhx_set_max_impacts(sc.h, 4 * objects)  # a re-strike arrives while the previous impact's click still rings out
    b0, a1, a2 = click_filter(sc)
    per_block = 256  # the event ring's capacity

    def strike_some(lo, hi, pulse_samples):
        step = np.float32(1.0 / pulse_samples)
        for o in range(lo, min(objects, hi)):
            ev = hipbank.Event(0, o, 0, 1.0, 0.5, 0.0, step, 2 * step, SR, b0, a1, a2)
            assert sc.L.mhx_enqueue(sc.h, ev)

    rounds = (objects + per_block - 1) // per_block  # blocks needed to strike everything once

    def keep_everything_excited(block):  # each object re-struck every `rounds` blocks with a pulse that long
        ph = block % rounds
        strike_some(ph * per_block, (ph + 1) * per_block, rounds * BLOCK)
    all_live = phase(sc, max(blocks // 2, 3 * rounds), rounds, keep_everything_excited)
    assert all_live["live_modes_mean"] == objects * modes, all_live["live_modes_mean"]
    # silence everything before the second phase (Silence events, a ring-full per block)
    silent = np.zeros(BLOCK, np.float32)
    for o in range(objects):
        if o % per_block == 0 and o:
            sc.render(silent)
        ev = hipbank.Event(1, o, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)  # Silence
        assert sc.L.mhx_enqueue(sc.h, ev)
    sc.render(silent)

    def period_strike(block):  # config 5: everything once per 64-block period
        ph = block % 64
        if ph < rounds:
            strike_some(ph * per_block, (ph + 1) * per_block, 300)
    steady = phase(sc, max(blocks, 64), 0, period_strike)
    sc.close()
    return {"workload": f"bank {objects}x{modes} @48k, {BLOCK}-frame blocks, {renderers} renderers, real click filters", "all_live": all_live, "steady_state": steady,
            "x_real_time": steady["x_real_time"], "x_real_time_all_live": all_live["x_real_time"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=1024)
    ap.add_argument("--modes", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=72)
    ap.add_argument("--renderers", type=int, default=4)
    a = ap.parse_args()
    print(json.dumps(run(a.objects, a.modes, a.blocks, a.renderers)))


if __name__ == "__main__":
    main()
