"""Python binding of libmodalhost.so: the C++ mirror of the reference's modal bank API (AddModalObject,
TuneModalObject, InstallModalBank, EnqueueModalEvent, RenderModal, ...) running on the MI355X through libmodalhip.
Used by the tests and the bank benchmark; no CPU fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libmodalhost.so")
_LIB = None

COLUMNS = ["CoeffRe", "CoeffIm", "StateRe", "StateIm", "RadiationGain", "RadiationArea", "DeflectionGain", "OutPhaseIm", "OutPhaseRe",
           "QuadCompliance", "QuadDriveScale", "ShapeX", "ShapeY", "ShapeZ", "OutGain", "ListenerGain", "RadiantRadius", "DeflectionScale"]


class Event(C.Structure):  # ModalEvent, src/audio/ModalAudio.h:28-37
    _fields_ = [("kind", C.c_uint32), ("object", C.c_uint32), ("ex_pos", C.c_uint32), ("jx", C.c_float), ("jy", C.c_float), ("jz", C.c_float),
                ("pulse_step", C.c_float), ("pulse_gamma", C.c_float), ("accel_amp", C.c_float), ("click_b0", C.c_float), ("click_a1", C.c_float),
                ("click_a2", C.c_float)]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing: run __graft_entry__.build() -- there is no CPU fallback")
    from . import _lib as core
    core.lib()  # libmodalhip first (dependency, also checks its exports)
    L = C.CDLL(SO_PATH)
    vp, u32, i32, f32, f64 = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_double
    sig = {
        "mhx_last_error": (C.c_char_p, []), "mhx_scene_create": (vp, [f32, i32]), "mhx_scene_create_f64": (vp, [f32, i32]), "mhx_scene_destroy": (None, [vp]),
        "mhx_add_object": (u32, [vp, u32, u32, u32, vp, vp, u32, vp]), "mhx_tune_object": (None, [vp, i32, u32, u32, vp, vp, f32]),
        "mhx_set_shapes": (i32, [vp, i32, u32, u32, u32, vp]), "mhx_set_gains": (None, [vp, i32, u32, f32, f32]), "mhx_install": (i32, [vp]),
        "mhx_set_renderers": (None, [vp, u32]), "mhx_set_click_gain": (None, [vp, f32]), "mhx_set_max_impacts": (None, [vp, u32]),
        "mhx_enqueue": (i32, [vp, C.POINTER(Event)]), "mhx_render": (i32, [vp, vp, u32]), "mhx_num_objects": (u32, [vp]),
        "mhx_active_impacts": (u32, [vp]), "mhx_modal_energy": (f64, [vp]), "mhx_render_share": (f32, [vp]), "mhx_find_object": (i32, [vp, u32]),
        "mhx_time_kernels": (i32, [vp, i32]), "mhx_kernel_class_stats": (i32, [vp, i32, C.POINTER(C.c_uint64), C.POINTER(f64), C.POINTER(f64)]),
        "mhx_column": (u32, [vp, i32, i32, vp]), "mhx_object_state": (None, [vp, vp, vp, vp]),
        "mhx_recoil_click_filter": (None, [f64, f64, f64, f64, vp]), "mhx_recoil_object_filter": (None, [f64, f64, f64, vp]),
        "mhx_estimate_contact_time": (f64, [f64, vp, vp, vp, f64, vp, f64, f64, vp, f64, f64, f64, f64]),
        "mhx_striker_mass": (f64, [f64, f32, f32]), "mhx_inverse_inertia_tensor": (None, [vp, vp, vp]),
        "mhx_saturation_penetration": (f64, [f64, f64]), "mhx_punch_stiffness": (f64, [f64, f64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Scene:
    """ModalAudio + the bank under construction, as the reference's test harness drives them (tests/ModalBench.h:47-81)."""

    def __init__(self, sample_rate=48000.0, device=0, use_double=False):
        """use_double: the fp64 bank (ModalBank64 / ModalAudio64) -- columns, impacts and output samples in double."""
        self.L = lib()
        self.dtype = np.float64 if use_double else np.float32
        self.h = (self.L.mhx_scene_create_f64 if use_double else self.L.mhx_scene_create)(sample_rate, device)

    def close(self):
        if self.h:
            self.L.mhx_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_object(self, entity, shapes, positions, indices):
        sh, pos, idx = np.ascontiguousarray(shapes, np.float32), np.ascontiguousarray(positions, np.float32), np.ascontiguousarray(indices, np.uint32)
        return self.L.mhx_add_object(self.h, entity, sh.shape[1], sh.shape[0], _p(sh), _p(pos), len(idx), _p(idx))

    def tune_object(self, obj, freqs, t60s, radius_scale=1.0, live=False):
        f, t = np.ascontiguousarray(freqs, np.float32), np.ascontiguousarray(t60s, np.float32)
        self.L.mhx_tune_object(self.h, int(live), obj, min(len(f), len(t)), _p(f), _p(t), radius_scale)

    def set_shapes(self, obj, shapes, live=True):
        sh = np.ascontiguousarray(shapes, np.float32)
        return bool(self.L.mhx_set_shapes(self.h, int(live), obj, sh.shape[1], sh.shape[0], _p(sh)))

    def set_gains(self, obj, out_gain, listener_gain=1.0, live=False):
        self.L.mhx_set_gains(self.h, int(live), obj, out_gain, listener_gain)

    def install(self):
        if self.L.mhx_install(self.h):
            raise RuntimeError(self.L.mhx_last_error().decode())

    def set_renderers(self, n):
        self.L.mhx_set_renderers(self.h, n)

    def set_click_gain(self, g):
        self.L.mhx_set_click_gain(self.h, g)

    def enqueue(self, ev):
        e = Event(*[getattr(ev, n) for n, _ in Event._fields_])
        return bool(self.L.mhx_enqueue(self.h, C.byref(e)))

    def render(self, out):
        assert out.dtype == self.dtype and out.flags["C_CONTIGUOUS"]
        if self.L.mhx_render(self.h, _p(out), len(out)):
            raise RuntimeError(self.L.mhx_last_error().decode())

    def time_kernels(self, enable=True):
        """HIP-event timing of the bank's kernels on its device context (measurement aid)."""
        if self.L.mhx_time_kernels(self.h, int(enable)):
            raise RuntimeError(self.L.mhx_last_error().decode())

    def kernel_stats(self, kernel_class=2):
        """{"launches", "total_ms", "work"} of a kernel class since time_kernels(True); class 2 = the resonator kernel, work in flops."""
        n, ms, work = C.c_uint64(0), C.c_double(0), C.c_double(0)
        if self.L.mhx_kernel_class_stats(self.h, kernel_class, C.byref(n), C.byref(ms), C.byref(work)):
            raise RuntimeError(self.L.mhx_last_error().decode())
        return {"launches": n.value, "total_ms": ms.value, "work": work.value}

    def column(self, name, live=True):
        which = COLUMNS.index(name)
        n = self.L.mhx_column(self.h, int(live), which, None)
        out = np.zeros(n)
        self.L.mhx_column(self.h, int(live), which, _p(out))
        return out

    def object_state(self):
        n = self.L.mhx_num_objects(self.h)
        tuned, live, ring = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self.L.mhx_object_state(self.h, _p(tuned), _p(live), _p(ring))
        return tuned, live, ring

    @property
    def active_impacts(self):
        return self.L.mhx_active_impacts(self.h)

    @property
    def modal_energy(self):
        return self.L.mhx_modal_energy(self.h)

    @property
    def render_share(self):
        return self.L.mhx_render_share(self.h)
