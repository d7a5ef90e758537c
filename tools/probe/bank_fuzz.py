"""Odd bank shapes against the oracle bank, sample for sample.   python tools/probe/bank_fuzz.py"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from tests import bank_harness as bh
from oracle import pyoracle as oracle

def strike_all(sc, impulse=1.0, **kw):
    for o in sc.objects:
        sc.enqueue(bh.impact_event(oracle, o, impulse, **kw))

def case(name, renderers, objects, modes, frames, blocks, modes_list=None, use_double=False):
    def run(make):
        sc = make()
        strike_all(sc)
        a = sc.render(blocks // 2, frames)
        for o in sc.objects[::2]:
            sc.enqueue(bh.impact_event(oracle, o, -0.4, 1, 1.0 / 90.0))
        b = sc.render(blocks - blocks // 2, frames)
        return np.concatenate([a, b])
    try:
        ref = run(lambda: bh.OracleScene(oracle, objects, modes, 0.2, renderers, modes=modes_list) if modes_list else bh.OracleScene(oracle, objects, modes, 0.2, renderers))
        got = run(lambda: bh.DeviceScene(objects, modes, 0.2, renderers, modes=modes_list, use_double=use_double) if modes_list else bh.DeviceScene(objects, modes, 0.2, renderers, use_double=use_double))
        same = np.array_equal(ref.astype(got.dtype), got) if not use_double else np.allclose(ref, got, rtol=0, atol=0) or np.abs(ref - got).max() < 1e-6 * np.abs(ref).max()
        print(f"{name:40s} {'EXACT' if np.array_equal(ref.astype(got.dtype), got) else ('close' if same else 'DIFFERENT')}  max |ref| {np.abs(ref).max():.3e}  max diff {np.abs(ref - got).max():.2e}", flush=True)
    except Exception as e:
        print(f"{name:40s} error: {str(e)[:160]}", flush=True)

case("1 object, 1 mode", 1, 1, 1, 512, 4)
case("3 objects, 7 modes", 2, 3, 7, 512, 4)
case("2 objects, 300 modes", 1, 2, 300, 512, 4)
case("2 objects, 513 modes", 2, 2, 513, 512, 4)
case("frames = 1", 1, 2, 16, 1, 20)
case("frames = 17", 2, 3, 33, 17, 12)
case("frames = 1000", 2, 3, 33, 1000, 4)
case("frames = 2048", 1, 2, 64, 2048, 2)
case("renderers 8 > objects 3", 8, 3, 20, 512, 4)
ml = [bh.make_modes(k, 0.2) for k in (1, 9, 64, 65, 200)]
case("mixed mode counts 1/9/64/65/200", 3, 5, 0, 512, 6, modes_list=ml)
