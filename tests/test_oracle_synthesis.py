"""Pins the CPU oracle (synthesis half + contact model) against the reference's own tests:
ModalRenderTest's three self-comparing properties (tests/ModalRenderTest.cpp:21-68 with the harness of
tests/ModalBench.h:19-81), the ContactModelTest known answers (tests/ContactModelTest.cpp:42-138) and the
impulse-response formula of the KHR_audio_rigid_bodies spec.  CPU only.
"""
import numpy as np
import pytest

from tests import bank_harness as bh


def test_excitations_superpose_linearly(oracle):
    """tests/ModalRenderTest.cpp:21-37"""
    both = [bh.impact_event(oracle, 0, 1.0, 0, 1.0 / 300.0), bh.impact_event(oracle, 0, -0.4, 1, 1.0 / 90.0)]

    def render(events):
        scene = bh.OracleScene(oracle, 1, 64, 0.2, 1)
        for e in events:
            e.object = scene.objects[0]
            scene.enqueue(e)
        return scene.render(8, bh.BLOCK)

    a, b, together = render(both[:1]), render(both[1:]), render(both)
    assert np.abs(a).max() > 0 and np.abs(b).max() > 0
    assert np.abs(together - (a + b)).max() <= np.abs(together).max() * 1e-5


def test_thread_count_independence(oracle):
    """tests/ModalRenderTest.cpp:40-49"""
    def render(renderers):
        scene = bh.OracleScene(oracle, 16, 64, 0.2, renderers)
        for o in scene.objects:
            scene.enqueue(bh.impact_event(oracle, o, 1.0))
        return scene.render(32, bh.BLOCK)

    single, split = render(1), render(4)
    assert np.abs(single).max() > 0
    assert np.abs(single - split).max() < np.abs(single).max() * 1e-5


def test_click_peak_independent_of_sample_rate(oracle):
    """tests/ModalRenderTest.cpp:53-68"""
    tau, radius, mass, impulse = 5e-4, 0.05, 1.0, 0.5
    volume = 4.0 / 3.0 * np.pi * radius ** 3

    def peak_at(rate):
        scene = bh.OracleScene(oracle, 1, 64, 0.2, 1, sample_rate=rate)
        step = np.float32(1.0 / (tau * rate))
        click = np.zeros(3, np.float32)
        oracle.lib().mo_recoil_click_filter(radius, volume, mass, rate, click.ctypes.data)
        ev = oracle.Event(0, scene.objects[0], 0, 0.0, 0.0, 0.0, step, 2 * step, np.float32(impulse) * np.float32(rate), click[0], click[1], click[2])
        scene.enqueue(ev)
        blocks = int(np.ceil(4 * tau * rate / bh.BLOCK))
        return np.abs(scene.render(blocks, bh.BLOCK)).max()

    slow, fast = peak_at(48000.0), peak_at(96000.0)
    assert slow > 0 and abs(fast / slow - 1.0) < 2e-2


def test_block_length_independence_and_culling(oracle):
    """State is carried across blocks (any block length gives the same signal) and the audible prefix shrinks
    until the object falls silent (ModalAudio.cpp:132-147)."""
    def render(frames, blocks):
        scene = bh.OracleScene(oracle, 2, 40, 0.05, 1)
        for o in scene.objects:
            scene.enqueue(bh.impact_event(oracle, o, 1.0))
        return scene, scene.render(blocks, frames)

    s1, a = render(512, 64)
    s2, b = render(256, 128)
    # the audible-prefix cull acts at block ends, so block length moves the signal by at most a few culled modes,
    # each below sqrt(SilentEnergy) = 1e-6 in amplitude
    assert np.abs(a - b).max() <= 5e-6
    tuned, live, ringing = s1.bank.object_state()
    assert (tuned == 40).all()
    # T60 = 0.05/(k+1) s: after 0.68 s everything has decayed below 1e-12 and the objects are silent
    assert (ringing == 0).all() and (live == tuned).all() and s1.bank.active_impacts == 0
    assert np.abs(s1.bank.column("StateRe")).max() == 0.0


def test_single_mode_impulse_response(oracle):
    """KHR_audio_rigid_bodies: a unit impulse on one mode rings as a*exp(-d t)*sin(2 pi f t) (spec README:259-273);
    the coupled-form recurrence reproduces it with the mode's radiation gain as amplitude."""
    rate, f, t60 = 48000.0, 440.0, 0.5
    bank = oracle.Bank(rate)
    shapes = np.zeros((4, 1, 3), np.float32)
    shapes[:, 0, :] = [0.0, 1.0, 0.0]  # the sample strip lies in the xz plane, so its normal is y
    pos, idx = bh.sample_strip()
    o = bank.add_object(0, shapes, pos, idx)
    bank.tune_object(o, [f], [t60])
    bank.set_gains(o, 1.0, 1.0)
    bank.install()
    out = np.zeros(bh.BLOCK, np.float32)
    bank.render(out)
    # one-sample pulse of unit sum: PulseStep = 1 -> SamplesLeft = 1, force = gamma/2 * (1 - cos 2pi) = 0 ... use step 1/2
    ev = oracle.Event(0, o, 0, 0.0, 1.0, 0.0, 0.5, 1.0, 0.0, 0.0, 0.0, 0.0)
    bank.enqueue(ev)
    sig = np.zeros(4096, np.float32)
    for blk in range(8):
        bank.render(sig[blk * 512:(blk + 1) * 512])
    gain = bank.column("RadiationGain")[0]
    cre, cim = bank.column("CoeffRe")[0], bank.column("CoeffIm")[0]
    pim, pre = bank.column("OutPhaseIm")[0], bank.column("OutPhaseRe")[0]
    # force curve: samples 0,1 carry gamma/2*(1-cos(pi)) = 1 and gamma/2*(1-cos(2pi)) = 0
    z, expect = 0j, []
    c = complex(cre, cim)
    for s in range(4096):
        z = z * c + (gain * 1.0 if s == 0 else 0.0)
        expect.append(pim * z.imag + pre * z.real)
    expect = np.array(expect)
    assert np.abs(sig - expect).max() < 1e-4 * np.abs(expect).max()
    decay = np.exp(-(np.log(1000.0) / t60) / rate)
    assert abs(abs(c) - decay) < 1e-4 and abs(np.angle(c) - 2 * np.pi * f / rate) < 1e-6


def test_tune_mutes_and_trims(oracle):
    bank = oracle.Bank(48000.0)
    modes = bh.make_modes(6, 0.2)
    o = bank.add_object(7, modes["shapes"], modes["positions"], modes["indices"])
    freqs = np.array([100.0, np.nan, 300.0, 23999.5, 500.0, 600.0], np.float32)
    t60s = np.array([0.2, 0.2, 0.2, 0.2, 0.0, -1.0], np.float32)
    bank.tune_object(o, freqs, t60s)
    cre = bank.column("CoeffRe", live=False)
    assert cre[1] == 0 and cre[3] == 0 and cre[4] == 0 and cre[5] == 0 and cre[0] != 0 and cre[2] != 0
    bank.install()
    tuned, live, _ = bank.object_state()
    assert tuned[0] == 3 and live[0] == 3  # only the trailing muted block is trimmed
    assert bank.column("OutPhaseIm")[1] == 1.0 and bank.column("RadiationGain")[1] == 0.0


def test_fp64_bank_tracks_fp32(oracle):
    f32 = bh.OracleScene(oracle, 4, 32, 0.2, 1)
    f64 = bh.OracleScene(oracle, 4, 32, 0.2, 1, use_double=True)
    for sc in (f32, f64):
        for o in sc.objects:
            sc.enqueue(bh.impact_event(oracle, o, 1.0))
    a, b = f32.render(8, bh.BLOCK), f64.render(8, bh.BLOCK)
    assert b.dtype == np.float64 and np.abs(a - b).max() < 1e-4 * np.abs(b).max()


# ---- contact model: tests/ContactModelTest.cpp ----
NULL_STRIKER = dict(density=1e6, young=1e30, poisson=0.0, tip_radius=1e6, length=1e6)
POLYMER = (1000.0, 1e9, 0.3, 0.0, 0.0)
CERAMIC0 = (2700.0, 7.2e10, 0.19, 0.0, 0.0)


def contact_time(oracle, mass, inv_inertia_diag, material, curvature, area=0.0, speed=1.0, scale=1.0, arm=(0, 0, 0), striker=NULL_STRIKER):
    L = oracle.lib()
    inv = np.diag([inv_inertia_diag] * 3).astype(np.float32).reshape(-1)
    armv, dirv = np.array(arm, np.float32), np.array([0, 0, 1], np.float32)
    smass = L.mo_striker_mass(striker["density"], striker["tip_radius"], striker["length"])
    smat = oracle.material(striker["density"], striker["young"], striker["poisson"])
    return L.mo_estimate_contact_time(mass, inv.ctypes.data, armv.ctypes.data, dirv.ctypes.data, speed, oracle.material(*material), curvature, area,
                                      smat, 1.0 / np.float32(striker["tip_radius"]), 1.0 / smass, scale, 0.0)


def near(a, b, tol=1e-6):
    return abs(a - b) <= tol * max(1.0, abs(b))


def test_contact_hertz_and_limits(oracle):
    L = oracle.lib()
    tau = contact_time(oracle, 1.0, 1.0, POLYMER, 100)
    assert near(tau, 1.744e-3, 2e-2)
    assert contact_time(oracle, 1.0, 1.0, POLYMER, 100, arm=(0.2, 0, 0)) < tau
    t = lambda s: contact_time(oracle, 1.0, 1.0, POLYMER, 100, 0, 1, s)
    assert near(t(2.0), 2 * t(1.0)) and near(t(100.0), 5e-2) and near(t(1e-6), 2e-5)
    inv_mod = 0.91 / 1e9
    tl = lambda curv, area, speed: contact_time(oracle, 1.0, 0.0, POLYMER, curv, area, speed)
    hertz = 2.868 * (inv_mod ** 2 * 100) ** 0.2
    assert near(tl(100, 0.0, 1.0), hertz, 1e-3)
    area = 1e-4
    punch = np.pi * np.sqrt(inv_mod / (2 * np.sqrt(area / np.pi)))
    assert near(tl(0.0, area, 1.0), punch, 1e-3)
    assert near(tl(100, 0.0, 32.0) / tl(100, 0.0, 1.0), 32.0 ** -0.2, 1e-3)
    assert near(tl(0.0, area, 32.0) / tl(0.0, area, 1.0), 1.0, 1e-3)
    assert near(L.mo_saturation_penetration(10.0, 1e-5), 3.183e-5, 1e-3)


def test_contact_saturation(oracle):
    L = oracle.lib()
    t = lambda area, speed: contact_time(oracle, 0.5, 0.0, CERAMIC0, 10, area, speed)
    assert near(t(1e-5, 0.1), t(0.0, 0.1)) and near(t(1.0, 3.0), t(0.0, 3.0))
    assert t(1e-5, 3.0) > t(0.0, 3.0)
    assert t(1e-5, 3.0) > np.pi * np.sqrt(0.5 / L.mo_punch_stiffness(0.91 / 7.2e10, 1e-5))
    assert near(t(1.7e-5, 1.0), t(1.5e-5, 1.0), 1e-3)
    hertz_ratio, sat_ratio = t(0.0, 3.0) / t(0.0, 0.1), t(1e-5, 3.0) / t(1e-5, 0.1)
    assert near(hertz_ratio, 30.0 ** -0.2, 1e-3) and hertz_ratio < sat_ratio < 1.0
    light = dict(density=7850.0, young=2.0e11, poisson=0.29, tip_radius=0.01, length=0.05)
    heavy = dict(light, length=5.0)
    assert contact_time(oracle, 1000.0, 0.0, CERAMIC0, 5, striker=light) < contact_time(oracle, 1000.0, 0.0, CERAMIC0, 5, striker=heavy)


def test_inverse_inertia_round_trip(oracle):
    """tests/ContactModelTest.cpp:42-53"""
    q = np.array([0.3, 0.1, -0.5, 0.8], np.float32)
    q /= np.linalg.norm(q)
    diag = np.array([2.0, 5.0, 9.0], np.float32)
    inv = np.zeros(9, np.float32)
    oracle.lib().mo_inverse_inertia_tensor(diag.ctypes.data, q.ctypes.data, inv.ctypes.data)
    w, x, y, z = q.astype(np.float64)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    inertia = R @ np.diag([2.0, 5.0, 9.0]) @ R.T
    prod = inertia @ inv.reshape(3, 3).T  # column-major -> transpose
    assert np.abs(prod - np.eye(3)).max() < 1e-4
