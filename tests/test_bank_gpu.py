"""GPU parity tests of the synthesis half.  The HIP bank is driven through the C++ mirror of the reference API
(libmodalhost.so) and, for fp64, through the C ABI directly; signals must equal the CPU oracle's sample for sample
(same expression trees, same summation order, -ffp-contract=off on both sides)."""
import ctypes as C

import numpy as np
import pytest

from tests import bank_harness as bh

pytestmark = pytest.mark.gpu


def _strike_all(scene, oracle, impulse=1.0, **kw):
    for o in scene.objects:
        scene.enqueue(bh.impact_event(oracle, o, impulse, **kw))


@pytest.mark.parametrize("renderers,objects,modes,frames,blocks", [(1, 1, 64, 512, 6), (1, 5, 37, 512, 6), (4, 16, 64, 512, 8), (3, 7, 130, 333, 9)])
def test_signal_is_sample_exact(oracle, renderers, objects, modes, frames, blocks):
    def run(make):
        sc = make()
        _strike_all(sc, oracle)
        a = sc.render(blocks // 2, frames)
        # a second, different strike mid-way on a subset (two impacts overlap on those objects)
        for o in sc.objects[::2]:
            sc.enqueue(bh.impact_event(oracle, o, -0.4, 1, 1.0 / 90.0))
        b = sc.render(blocks - blocks // 2, frames)
        return np.concatenate([a, b]), sc
    ref, so = run(lambda: bh.OracleScene(oracle, objects, modes, 0.2, renderers))
    got, sg = run(lambda: bh.DeviceScene(objects, modes, 0.2, renderers))
    assert np.abs(ref).max() > 0
    assert np.array_equal(ref, got), np.abs(ref - got).max()
    assert np.array_equal(so.bank.column("StateRe"), sg.bank.column("StateRe"))
    assert np.array_equal(so.bank.column("StateIm"), sg.bank.column("StateIm"))
    for a, b in zip(so.bank.object_state(), sg.bank.object_state()):
        assert np.array_equal(a, b)
    assert abs(so.bank.modal_energy - sg.bank.modal_energy) <= 1e-12 * max(so.bank.modal_energy, 1e-300)


@pytest.mark.parametrize("renderers,objects,modes,frames,blocks", [(4, 16, 64, 512, 8), (1, 3, 40, 512, 30), (3, 7, 130, 333, 9)])
def test_fp64_bank_through_the_mirror_is_sample_exact(oracle, renderers, objects, modes, frames, blocks):
    """BASELINE north star 'sample-exact at fp64', behind the reference's API: ModalBank64 / ModalAudio64 (double columns,
    impacts and output) against the oracle's double bank -- several renderers, overlapping impacts, real click filters,
    decay to the audible prefix."""
    def run(make):
        sc = make()
        _strike_all(sc, oracle)
        a = sc.render(blocks // 2, frames)
        for o in sc.objects[::2]:
            sc.enqueue(bh.impact_event(oracle, o, -0.4, 1, 1.0 / 90.0))
        b = sc.render(blocks - blocks // 2, frames)
        return np.concatenate([a, b]), sc
    ref, so = run(lambda: bh.OracleScene(oracle, objects, modes, 0.1, renderers, use_double=True))
    got, sg = run(lambda: bh.DeviceScene(objects, modes, 0.1, renderers, use_double=True))
    assert ref.dtype == got.dtype == np.float64 and np.abs(ref).max() > 0
    assert np.array_equal(ref, got), np.abs(ref - got).max()
    for name in ("CoeffRe", "CoeffIm", "RadiationGain", "OutPhaseIm", "OutPhaseRe", "StateRe", "StateIm"):
        assert np.array_equal(so.bank.column(name), sg.bank.column(name)), name
    for a, b in zip(so.bank.object_state(), sg.bank.object_state()):
        assert np.array_equal(a, b)
    assert so.bank.active_impacts == sg.bank.active_impacts


def test_click_filter_and_impact_retirement_match_the_oracle(oracle):
    """The per-impact recoil click (DF-II-T biquad driven by AccelAmp x force, ModalAudio.cpp:526-531) in isolation: an
    impulse-free strike (J = 0) leaves only the click in the output.  Sample-exact against the oracle at two click gains,
    and the impact retires in the same block on both sides once |z1| + |z2| < 1e-12 (:557-561)."""
    for gain in (1.0, 0.37):
        sigs, counts = [], []
        for make in (lambda: bh.OracleScene(oracle, 2, 16, 0.05, 2), lambda: bh.DeviceScene(2, 16, 0.05, 2)):
            sc = make()
            sc.bank.set_click_gain(gain)
            ev = bh.impact_event(oracle, sc.objects[1], 1.0, 0, 1.0 / 37.0)
            ev.jx = ev.jy = ev.jz = 0.0
            sc.enqueue(ev)
            parts, n = [], []
            for _ in range(6):
                parts.append(sc.render(1, 256))
                n.append(sc.bank.active_impacts)
            sigs.append(np.concatenate(parts))
            counts.append(n)
        assert np.abs(sigs[0]).max() > 0 and np.array_equal(sigs[0], sigs[1]), np.abs(sigs[0] - sigs[1]).max()
        assert counts[0] == counts[1] and counts[0][0] == 1 and counts[0][-1] == 0, counts


def test_many_impacts_culling_and_silence(oracle):
    """More than four impacts on one object (the register fast path overflows), decay to silence, a re-strike from
    culled state, and a Silence event."""
    def run(make):
        sc = make()
        for k in range(7):
            sc.enqueue(bh.impact_event(oracle, sc.objects[0], 0.3 + 0.1 * k, k % 4, 1.0 / (40.0 + 17 * k)))
        sc.enqueue(bh.impact_event(oracle, sc.objects[1], 1.0))
        parts = [sc.render(40, 512)]
        states = [sc.bank.object_state()]
        sc.enqueue(bh.impact_event(oracle, sc.objects[1], 0.5, 2))
        parts.append(sc.render(3, 512))
        ev = bh.impact_event(oracle, sc.objects[1], 0.0)
        ev.kind = 1  # Silence
        sc.enqueue(ev)
        parts.append(sc.render(2, 512))
        states.append(sc.bank.object_state())
        return np.concatenate(parts), states, sc
    ref, st_o, so = run(lambda: bh.OracleScene(oracle, 3, 40, 0.05, 2))
    got, st_g, sg = run(lambda: bh.DeviceScene(3, 40, 0.05, 2))
    assert np.array_equal(ref, got), np.abs(ref - got).max()
    for a, b in zip(st_o, st_g):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert (st_g[0][2] == 0).all()  # everything fell silent after 0.43 s at T60 <= 0.05 s
    assert so.bank.active_impacts == sg.bank.active_impacts


def test_tune_and_shapes_match_oracle(oracle):
    from mesheditor_amd import bank as hipbank
    modes = bh.make_modes(12, 0.3)
    freqs = modes["freqs"].copy()
    freqs[3], freqs[7] = np.nan, 30000.0
    t60s = modes["t60s"].copy()
    t60s[10:] = 0.0
    bo, bg = oracle.Bank(44100.0), hipbank.Scene(44100.0)
    for b in (bo, bg):
        s = b.add_object(9, modes["shapes"], modes["positions"], modes["indices"])
        b.tune_object(s, freqs, t60s, 1.5)
        b.set_gains(s, 0.7, 0.9)
        b.install()
    for name in ("CoeffRe", "CoeffIm", "RadiationGain", "RadiationArea", "DeflectionGain", "OutPhaseIm", "OutPhaseRe", "QuadCompliance", "QuadDriveScale",
                 "ShapeX", "ShapeY", "ShapeZ", "RadiantRadius", "DeflectionScale", "OutGain", "ListenerGain"):
        assert np.array_equal(bo.column(name), bg.column(name)), name
    for a, b in zip(bo.object_state(), bg.object_state()):
        assert np.array_equal(a, b)
    # in-place retune and shape overwrite of the live bank, then render: still sample-exact
    new_shapes = (modes["shapes"] * np.float32(1.25)).astype(np.float32)
    out_o, out_g = np.zeros(512, np.float32), np.zeros(512, np.float32)
    for b, out in ((bo, out_o), (bg, out_g)):
        b.render(np.zeros(512, np.float32))
        b.tune_object(0, modes["freqs"] * np.float32(1.1), modes["t60s"], live=True)
        assert b.set_shapes(0, new_shapes)
        assert not b.set_shapes(0, new_shapes[:, :5])
        b.enqueue(bh.impact_event(oracle, 0, 1.0))
        b.render(out)
    assert np.abs(out_o).max() > 0 and np.array_equal(out_o, out_g)


def test_render_properties_on_device(oracle):
    """The reference's ModalRenderTest properties (tests/ModalRenderTest.cpp:21-68) on the HIP path."""
    both = [bh.impact_event(oracle, 0, 1.0, 0, 1.0 / 300.0), bh.impact_event(oracle, 0, -0.4, 1, 1.0 / 90.0)]

    def render(events):
        sc = bh.DeviceScene(1, 64, 0.2, 1)
        for e in events:
            e.object = sc.objects[0]
            sc.enqueue(e)
        return sc.render(8, bh.BLOCK)
    a, b, together = render(both[:1]), render(both[1:]), render(both)
    assert np.abs(together - (a + b)).max() <= np.abs(together).max() * 1e-5

    def threads(n):
        sc = bh.DeviceScene(16, 64, 0.2, n)
        _strike_all(sc, oracle)
        return sc.render(32, bh.BLOCK)
    single, split = threads(1), threads(4)
    assert np.abs(single).max() > 0 and np.abs(single - split).max() < np.abs(single).max() * 1e-5

    from mesheditor_amd import bank as hipbank
    tau, radius, mass, impulse = 5e-4, 0.05, 1.0, 0.5
    volume = 4.0 / 3.0 * np.pi * radius ** 3

    def peak_at(rate):
        sc = bh.DeviceScene(1, 64, 0.2, 1, sample_rate=rate)
        step = np.float32(1.0 / (tau * rate))
        click = np.zeros(3, np.float32)
        hipbank.lib().mhx_recoil_click_filter(radius, volume, mass, rate, click.ctypes.data)
        ref = np.zeros(3, np.float32)
        oracle.lib().mo_recoil_click_filter(radius, volume, mass, rate, ref.ctypes.data)
        assert np.array_equal(click, ref)
        sc.enqueue(oracle.Event(0, sc.objects[0], 0, 0.0, 0.0, 0.0, step, 2 * step, np.float32(impulse) * np.float32(rate), click[0], click[1], click[2]))
        return np.abs(sc.render(int(np.ceil(4 * tau * rate / bh.BLOCK)), bh.BLOCK)).max()
    slow, fast = peak_at(48000.0), peak_at(96000.0)
    assert slow > 0 and abs(fast / slow - 1.0) < 2e-2


def test_fp64_bank_sample_exact(oracle):
    """BASELINE north star: resonator output sample-exact at fp64.  Drives the C ABI directly with the small host loop
    RenderModal performs (activate impacts, deal to one renderer, audible-prefix bookkeeping)."""
    from mesheditor_amd import _lib, api
    L = _lib.lib()
    ctx = api.Context(0)
    n_obj, n_modes, frames, blocks = 3, 40, 512, 30
    modes = bh.make_modes(n_modes, 0.1)
    ob = oracle.Bank(48000.0, use_double=True)
    for o in range(n_obj):
        s = ob.add_object(o, modes["shapes"], modes["positions"], modes["indices"])
        ob.tune_object(s, modes["freqs"], modes["t60s"])
        ob.set_gains(s, 1.0, 1.0)
    ob.install()
    cols = {k: ob.column(k) for k in ("CoeffRe", "CoeffIm", "RadiationGain", "OutPhaseIm", "OutPhaseRe")}
    sx, sy, sz = (ob.column(k).astype(np.float32) for k in ("ShapeX", "ShapeY", "ShapeZ"))
    mode_count = np.full(n_obj, n_modes, np.uint32)
    mode_offset = (np.arange(n_obj) * n_modes).astype(np.uint32)
    shape_offset = (np.arange(n_obj) * n_modes * 4).astype(np.uint32)
    h = C.c_void_p()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(L.mh_bank_create(ctx.h, 1, n_obj, n_obj * n_modes, len(sx), p(mode_offset), p(mode_count), p(shape_offset), p(sx), p(sy), p(sz), C.byref(h)))
    ctx.check(L.mh_bank_set_coefficients(h, 0, n_obj * n_modes, *[p(cols[k]) for k in ("CoeffRe", "CoeffIm", "RadiationGain", "OutPhaseIm", "OutPhaseRe")]))
    ob.render(np.zeros(frames))
    events = [(o, 1.0 - 0.2 * o, o % 4, 1.0 / (100.0 + 50 * o)) for o in range(n_obj)]
    impacts = (_lib.Impact * len(events))()
    for i, (o, imp, ex, step) in enumerate(events):
        ob.enqueue(bh.impact_event(oracle, o, imp, ex, step, click=False))
        st = np.float32(step)
        theta = 2 * np.pi * np.float64(st)
        impacts[i] = _lib.Impact(o, ex, int(np.ceil(1.0 / np.float64(st))), 0, np.float32(imp), np.float32(0.5 * imp), 0.0, 1.0, 0.0, np.cos(theta), np.sin(theta),
                                 20.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    n_imp = len(events)
    live = mode_count.copy()
    ringing = np.ones(n_obj, bool)
    ones = np.ones(n_obj, np.float32)
    for blk in range(blocks):
        ref = np.zeros(frames)
        ob.render(ref)
        dealt = np.array([o for o in range(n_obj) if ringing[o]], np.uint32)
        excited = [any(impacts[i].object == o for i in range(n_imp)) for o in dealt]
        rc = np.array([n_modes if e else live[o] for o, e in zip(dealt, excited)], np.uint32)
        tuned = np.full(len(dealt), n_modes, np.uint32)
        off = np.array([0, len(dealt)], np.uint32)
        out = np.zeros(frames)
        en, lv, sil, me = np.zeros(len(dealt)), np.zeros(len(dealt), np.uint32), np.zeros(len(dealt), np.uint8), np.zeros(len(dealt))
        ctx.check(L.mh_bank_render(h, frames, 1.0, n_imp, impacts, 1, p(off), p(dealt), p(rc), p(tuned), p(ones), p(ones), p(out), p(en), p(lv), p(sil), p(me)))
        assert np.array_equal(ref, out), (blk, np.abs(ref - out).max())
        for k, o in enumerate(dealt):
            if sil[k]:
                ringing[o], live[o] = False, n_modes
            else:
                live[o] = n_modes if excited[k] else lv[k]
        # retire drained impacts (ModalAudio.cpp:557-561), swap-with-last order as the reference
        i = n_imp
        while i > 0:
            i -= 1
            if impacts[i].samples_left == 0 and abs(impacts[i].click_z1) + abs(impacts[i].click_z2) < 1e-12:
                impacts[i] = impacts[n_imp - 1]
                n_imp -= 1
    tuned_o, live_o, ring_o = ob.object_state()
    assert np.array_equal(ring_o.astype(bool), ringing) and np.array_equal(live_o, live)
    L.mh_bank_destroy(h)
    ctx.close()


def test_full_size_bank(oracle):
    """BASELINE configs[4]: 1024 objects x 256 modes @ 48 kHz, one strike per object: three blocks sample-exact against
    the oracle, then real-time factor of the device render."""
    import time
    n_obj, n_modes = 1024, 256
    # config 5's bank: every object its own frequencies (f_k x (1 + 0.001 o)), as the bench
    modes = [bh.make_modes(n_modes, 2.0, freq_scale=1.0 + 0.001 * o) for o in range(n_obj)]
    so = bh.OracleScene(oracle, n_obj, n_modes, 2.0, 4, modes=modes)
    sg = bh.DeviceScene(n_obj, n_modes, 2.0, 4, modes=modes)
    # the SPSC queue holds 256 events: strike in waves of 256 per block, as a caller would
    ref_parts, got_parts = [], []
    for wave in range(4):
        for sc in (so, sg):
            for o in sc.objects[wave * 256:(wave + 1) * 256]:
                assert sc.enqueue(bh.impact_event(oracle, o, 1.0))
        ref_parts.append(so.render(1, bh.BLOCK))
        got_parts.append(sg.render(1, bh.BLOCK))
    ref, got = np.concatenate(ref_parts), np.concatenate(got_parts)
    assert np.abs(ref).max() > 0 and np.array_equal(ref, got), np.abs(ref - got).max()
    t0 = time.perf_counter()
    nblk = 20
    sig = sg.render(nblk, bh.BLOCK)
    dt = time.perf_counter() - t0
    assert np.isfinite(sig).all()
    print("full-size bank: %.2f ms per 512-frame block, x%.1f real time" % (1e3 * dt / nblk, nblk * bh.BLOCK / 48000.0 / dt))
