"""The modal render harness of the reference's audio tests, restated (tests/ModalBench.h:14-81):
SampleStrip / MakeModes / ImpactEvent / ModalScene.  OracleScene drives the CPU oracle; tests/test_bank_gpu.py
drives the HIP bank through the same interface.
"""
import numpy as np

SAMPLE_RATE = 48000.0
BLOCK = 512
SAMPLE_POINTS = 4


def sample_strip():
    """tests/ModalBench.h:19-24: a strip of alternating depth so consecutive triples form triangles."""
    pos = np.array([[p * 0.01, 0.0, 0.02 if p % 2 else 0.0] for p in range(SAMPLE_POINTS)], np.float32)
    idx = np.array([[p, p + 1, p + 2] for p in range(SAMPLE_POINTS - 2)], np.uint32).reshape(-1)
    return pos, idx


def make_modes(mode_count, longest_t60, shape_scale=1.0, freq_scale=1.0):
    """tests/ModalBench.h:26-40 (float arithmetic as the reference)."""
    k = np.arange(mode_count, dtype=np.float32)
    freqs = (np.float32(40.0) * (k + np.float32(1)) * np.float32(1.031) * np.float32(freq_scale)).astype(np.float32)
    t60s = (np.float32(longest_t60) / (k + np.float32(1))).astype(np.float32)
    pos, idx = sample_strip()
    shapes = np.zeros((SAMPLE_POINTS, mode_count, 3), np.float32)
    for p in range(SAMPLE_POINTS):
        a = ((k + np.float32(1)) * np.float32(0.37) + np.float32(p)).astype(np.float32)
        s = np.stack([np.sin(a), np.cos(a * np.float32(1.7)), np.sin(a * np.float32(2.3))], -1).astype(np.float32)
        shapes[p] = s * np.float32(0.01) * np.float32(shape_scale)
    return {"freqs": freqs, "t60s": t60s, "shapes": shapes, "positions": pos, "indices": idx}


def click_coefficients(oracle, sample_rate=SAMPLE_RATE, radius=0.05, mass=1.0):
    """RecoilClickFilter of a sphere of `radius`, `mass` (ModalAudio.h:92-99) -- input data for an event; the mirror's own
    filter routine is compared with the oracle's elsewhere (test_abi_cpu, test_render_properties_on_device)."""
    out = np.zeros(3, np.float32)
    oracle.lib().mo_recoil_click_filter(radius, 4.0 / 3.0 * np.pi * radius ** 3, mass, sample_rate, out.ctypes.data)
    return out


def impact_event(oracle, obj, impulse, ex_pos=0, pulse_step=1.0 / 300.0, click=True, sample_rate=SAMPLE_RATE):
    """tests/ModalBench.h:42-44, plus -- unless click=False -- the recoil click a real strike carries
    (AudioSystem.cpp:441-458: ClickB0/A1/A2 = RecoilClickFilter, AccelAmp = impulse * SR), so the per-impact biquad and
    its ring-out retirement (ModalAudio.cpp:526-531, 557-561) are part of every signal that is compared."""
    if not click:
        return oracle.Event(0, obj, ex_pos, impulse, 0.5 * impulse, 0.0, pulse_step, 20.0, 0.0, 0.0, 0.0, 0.0)
    b0, a1, a2 = click_coefficients(oracle, sample_rate)
    return oracle.Event(0, obj, ex_pos, impulse, 0.5 * impulse, 0.0, pulse_step, 20.0, np.float32(abs(impulse)) * np.float32(sample_rate), b0, a1, a2)


class OracleScene:
    """tests/ModalBench.h:47-81 over the CPU oracle."""

    def __init__(self, oracle, object_count, mode_count, longest_t60, renderers, sample_rate=SAMPLE_RATE, use_double=False, modes=None):
        self.dtype = np.float64 if use_double else np.float32
        self.bank = oracle.Bank(sample_rate, use_double)
        self.bank.set_renderers(renderers)
        modes = modes or make_modes(mode_count, longest_t60)
        self.objects = []
        for o in range(object_count):
            mo = modes[o] if isinstance(modes, (list, tuple)) else modes  # one description for all, or one per object
            slot = self.bank.add_object(o, mo["shapes"], mo["positions"], mo["indices"])
            self.bank.tune_object(slot, mo["freqs"], mo["t60s"])
            self.bank.set_gains(slot, 1.0, 1.0)
            self.objects.append(slot)
        self.bank.install()
        self.bank.render(np.zeros(BLOCK, self.dtype))

    def enqueue(self, ev):
        return self.bank.enqueue(ev)

    def render(self, blocks, frames):
        sig = np.zeros(blocks * frames, self.dtype)
        for b in range(blocks):
            self.bank.render(sig[b * frames:(b + 1) * frames])
        return sig


class DeviceScene:
    """tests/ModalBench.h:47-81 over the HIP bank (libmodalhost.so -> libmodalhip.so)."""

    def __init__(self, object_count, mode_count, longest_t60, renderers, sample_rate=SAMPLE_RATE, modes=None, device=0, use_double=False):
        from mesheditor_amd import bank as hipbank
        self.dtype = np.float64 if use_double else np.float32
        self.bank = hipbank.Scene(sample_rate, device, use_double)
        self.bank.set_renderers(renderers)
        modes = modes or make_modes(mode_count, longest_t60)
        self.objects = []
        for o in range(object_count):
            mo = modes[o] if isinstance(modes, (list, tuple)) else modes  # one description for all, or one per object
            slot = self.bank.add_object(o, mo["shapes"], mo["positions"], mo["indices"])
            self.bank.tune_object(slot, mo["freqs"], mo["t60s"])
            self.bank.set_gains(slot, 1.0, 1.0)
            self.objects.append(slot)
        self.bank.install()
        self.bank.render(np.zeros(BLOCK, self.dtype))

    def enqueue(self, ev):
        return self.bank.enqueue(ev)

    def render(self, blocks, frames):
        sig = np.zeros(blocks * frames, self.dtype)
        for b in range(blocks):
            self.bank.render(sig[b * frames:(b + 1) * frames])
        return sig
