"""ctypes loader of libmodalhip.so.  Fails loudly when the HIP library is missing: there is no fallback path."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libmodalhip.so")
_LIB = None

MH_OK, MH_EINVAL, MH_EHIP, MH_ECANCELLED, MH_ENOTCONVERGED, MH_EFACTOR, MH_EEMPTY = range(7)


class Material(C.Structure):
    _fields_ = [("density", C.c_double), ("young_modulus", C.c_double), ("poisson_ratio", C.c_double), ("alpha", C.c_double), ("beta", C.c_double)]


class SolverConfig(C.Structure):
    _fields_ = [("min_mode_freq", C.c_float), ("max_mode_freq", C.c_float), ("num_modes", C.c_uint32), ("num_fem_modes", C.c_uint32),
                ("tolerance", C.c_double), ("warm_tolerance", C.c_double), ("max_restarts", C.c_uint32), ("has_fundamental", C.c_int32),
                ("fundamental_freq", C.c_float)]


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("mass_props", "quad_mesh", "assemble", "sample_excite", "factorize", "iterate", "op_solve", "extract")] + \
               [(n, C.c_uint32) for n in ("dofs", "stiffness_nonzeros", "op_applications", "restarts", "sytrd_redos", "pairs_at_floor")] + [("rr_selfcheck", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MassProps(C.Structure):
    _fields_ = [("mass", C.c_double), ("center_of_mass", C.c_float * 3), ("inertia_diagonal", C.c_float * 3), ("inertia_orientation_wxyz", C.c_float * 4)]


class Impact(C.Structure):
    _fields_ = [("object", C.c_uint32), ("ex_pos", C.c_uint32), ("samples_left", C.c_uint32), ("reserved", C.c_uint32)] + \
               [(n, C.c_double) for n in ("jx", "jy", "jz", "phase_re", "phase_im", "rot_re", "rot_im", "gamma", "accel_amp",
                                          "click_b0", "click_a1", "click_a2", "click_z1", "click_z2")]


def build(force=False):
    """Compile libmodalhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    lab_src, lab_so = os.path.join(src, "lab"), os.path.join(_HERE, "libmodalhip_lab.so")  # the lab bench (tools/lab.py) is built alongside
    deps = [os.path.join(d, f) for d in (src, lab_src) for f in os.listdir(d) if f.endswith((".hip", ".cpp", ".h"))] + [os.path.join(_HERE, "..", "include", "modalhip.h")]
    if force or not os.path.exists(SO_PATH) or not os.path.exists(lab_so) or any(os.path.getmtime(d) > min(os.path.getmtime(SO_PATH), os.path.getmtime(lab_so)) for d in deps):
        subprocess.check_call(["make", "-s", "-j4", "-C", src])
    return SO_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` -- there is no CPU fallback")
    L = C.CDLL(SO_PATH)
    vp, u32, i32, f64 = C.c_void_p, C.c_uint32, C.c_int, C.c_double
    pp = C.POINTER(vp)
    sig = {
        "mh_context_create": (i32, [i32, pp]), "mh_context_destroy": (None, [vp]), "mh_last_error": (C.c_char_p, [vp]),
        "mh_context_synchronize": (i32, [vp]), "mh_context_time_kernels": (i32, [vp, i32]),
        "mh_context_kernel_stats": (i32, [vp, C.POINTER(C.c_uint64), C.POINTER(f64), C.POINTER(f64)]), "mh_context_kernel_class_stats": (i32, [vp, i32, C.POINTER(C.c_uint64), C.POINTER(f64), C.POINTER(f64)]), "mh_context_stream": (vp, [vp]), "mh_default_config": (None, [C.POINTER(SolverConfig)]),
        "mh_mesh_create": (i32, [vp, u32, vp, u32, vp, pp]), "mh_mesh_destroy": (None, [vp]),
        "mh_assemble": (i32, [vp, vp, C.POINTER(Material), pp]), "mh_system_destroy": (None, [vp]),
        "mh_system_dims": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64)]),
        "mh_system_element_nodes": (i32, [vp, vp]), "mh_system_export_blocks": (i32, [vp, vp, vp, vp, vp]),
        "mh_abi_struct_sizes": (None, [vp]),
        "mh_system_matvec": (i32, [vp, i32, vp, vp, u32]),
        "mh_system_shift_invert": (i32, [vp, C.c_double, vp, vp, u32, C.c_double, u32, C.POINTER(u32), C.POINTER(C.c_double)]),
        "mh_nearest_points": (i32, [vp, vp, u32, vp, vp]),
        "mh_eigs": (i32, [vp, u32, f64, f64, u32, vp, u32, u32, vp, vp, vp, C.POINTER(Profile)]),
        "mh_system_gather_shapes": (i32, [vp, u32, vp, u32, vp]), "mh_system_basis": (i32, [vp, u32, vp]),
        "mh_system_eigenvectors": (i32, [vp, u32, vp]),
        "mh_system_residual_report": (i32, [vp, vp, vp]),
        "mh_compute_mass_properties": (i32, [u32, vp, u32, vp, f64, vp, f64, C.POINTER(MassProps)]),
        "mh_postprocess_modes": (i32, [u32, vp, u32, vp, C.c_float, C.POINTER(Material), C.POINTER(SolverConfig), C.POINTER(u32), vp, vp, vp, C.POINTER(C.c_float)]),
        "mh_rescale_modes": (i32, [u32, vp, u32, vp, C.POINTER(Material), C.POINTER(Material), C.POINTER(SolverConfig), C.POINTER(i32), C.POINTER(u32), vp, vp, vp, C.POINTER(C.c_float)]),
        "mh_bank_create": (i32, [vp, i32, u32, u32, u32, vp, vp, vp, vp, vp, vp, pp]), "mh_bank_destroy": (None, [vp]),
        "mh_bank_set_coefficients": (i32, [vp, u32, u32, vp, vp, vp, vp, vp]), "mh_bank_set_shapes": (i32, [vp, u32, u32, vp, vp, vp]),
        "mh_bank_zero_state": (i32, [vp, u32, u32]),
        "mh_bank_render": (i32, [vp, u32, C.c_float, u32, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "mh_bank_read_state": (i32, [vp, u32, u32, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export what include/modalhip.h declares
        fn.restype, fn.argtypes = res, args
    L._declared = sorted(sig)
    _LIB = L
    return L
