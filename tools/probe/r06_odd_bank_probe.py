"""Odd inputs of the modal BANK, device and oracle side by side through the test harness (tests/bank_harness.py): every signal must be the oracle's sample for sample
(NaN where it has NaN), every refusal the oracle's refusal."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bank_harness as H
from oracle import pyoracle
pyoracle.build(); pyoracle.lib()

def run(name, build, drive, use_double=False):
    out = []
    for side in ("oracle", "device"):
        try:
            scene = build(side)
            sig = drive(scene)
            out.append(sig)
        except Exception as e:  # noqa: BLE001
            out.append(f"ERROR {type(e).__name__}: {str(e)[:120]}")
    o, d = out
    if isinstance(o, str) or isinstance(d, str):
        verdict = "both refuse" if isinstance(o, str) and isinstance(d, str) else "DIFFERENT BEHAVIOUR"
        print(f"{name}: {verdict}\n    oracle: {o if isinstance(o, str) else 'signal'}\n    device: {d if isinstance(d, str) else 'signal'}", flush=True)
        return
    same = np.array_equal(o, d, equal_nan=True)
    print(f"{name}: {'sample-exact' if same else 'DIFFERENT'} ({len(o)} samples, peak {np.nanmax(np.abs(o)) if len(o) else 0:.3e}, NaN {int(np.isnan(o).sum())}, non-zero {int((o != 0).sum())}"
          + ("" if same else f", first difference at {int(np.argmax(~((o == d) | (np.isnan(o) & np.isnan(d)))))}: {o[np.argmax(~((o == d) | (np.isnan(o) & np.isnan(d))))]} vs {d[np.argmax(~((o == d) | (np.isnan(o) & np.isnan(d))))]}") + ")", flush=True)

def scene(side, objects=3, modes=24, t60=0.4, renderers=2, sr=H.SAMPLE_RATE, mo=None, use_double=False):
    return H.OracleScene(pyoracle, objects, modes, t60, renderers, sample_rate=sr, modes=mo, use_double=use_double) if side == "oracle" else H.DeviceScene(objects, modes, t60, renderers, sample_rate=sr, modes=mo, use_double=use_double)

def strikes(s, imps, **kw):
    for k, imp in enumerate(imps):
        s.enqueue(H.impact_event(pyoracle, k % len(s.objects), imp, ex_pos=k % H.SAMPLE_POINTS, **kw))

for frames in (1, 7, 100, 513, 1024, 4096):
    run(f"blocks of {frames} frames", scene, lambda s, f=frames: (strikes(s, [0.3, -0.2, 0.5]), s.render(3, f))[1])
run("an object of one mode", lambda side: scene(side, modes=1), lambda s: (strikes(s, [0.3, 0.2]), s.render(2, 512))[1])
run("an object of no modes", lambda side: scene(side, modes=0), lambda s: (strikes(s, [0.3, 0.2]), s.render(2, 512))[1])
for imp in (0.0, 1e20, -1e-30, float("nan"), float("inf")):
    run(f"impulse {imp}", scene, lambda s, i=imp: (strikes(s, [i, 0.1]), s.render(2, 512))[1])
run("excitation position out of range", scene, lambda s: (s.enqueue(H.impact_event(pyoracle, 0, 0.3, ex_pos=99)), s.render(2, 512))[1])
run("object out of range", scene, lambda s: (s.enqueue(H.impact_event(pyoracle, 77, 0.3)), s.render(2, 512))[1])
run("600 impacts in one block", scene, lambda s: (strikes(s, [0.01 * (1 + k % 7) for k in range(600)]), s.render(3, 512))[1])
for step in (0.0, -1.0, 1e9, float("nan")):
    run(f"pulse step {step}", scene, lambda s, st=step: (strikes(s, [0.3, 0.2], pulse_step=st), s.render(2, 512))[1])
def odd_modes(**kw):
    mo = H.make_modes(24, 0.4)
    for k, v in kw.items(): mo[k] = np.asarray(v(mo[k]), np.float32)
    return mo
run("t60 of zero and negative", lambda side: scene(side, mo=odd_modes(t60s=lambda t: np.where(np.arange(len(t)) % 3 == 0, 0.0, np.where(np.arange(len(t)) % 3 == 1, -t, t)))), lambda s: (strikes(s, [0.3, 0.2]), s.render(2, 512))[1])
run("frequencies of zero, above Nyquist, NaN", lambda side: scene(side, mo=odd_modes(freqs=lambda f: np.where(np.arange(len(f)) % 4 == 0, 0.0, np.where(np.arange(len(f)) % 4 == 1, 30000.0, np.where(np.arange(len(f)) % 4 == 2, np.nan, f))))), lambda s: (strikes(s, [0.3, 0.2]), s.render(2, 512))[1])
for sr in (8000.0, 192000.0):
    run(f"sample rate {sr:g}", lambda side, r=sr: scene(side, sr=r), lambda s, r=sr: (strikes(s, [0.3, 0.2], sample_rate=r), s.render(2, 512))[1])
run("fp64 bank, 513-frame blocks", lambda side: scene(side, use_double=True), lambda s: (strikes(s, [0.3, -0.2, 0.5]), s.render(3, 513))[1])
