"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libmodal_oracle.so (the CPU restatement of the reference's modal path).
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by mesheditor_amd.
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Material(C.Structure):
    _fields_ = [("density", C.c_double), ("young_modulus", C.c_double), ("poisson_ratio", C.c_double),
                ("alpha", C.c_double), ("beta", C.c_double)]


class SolverConfig(C.Structure):
    _fields_ = [("min_mode_freq", C.c_float), ("max_mode_freq", C.c_float), ("num_modes", C.c_uint32),
                ("num_fem_modes", C.c_uint32), ("tolerance", C.c_double), ("warm_tolerance", C.c_double),
                ("max_restarts", C.c_uint32), ("has_fundamental", C.c_int32), ("fundamental_freq", C.c_float)]


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("mass_props", "quad_mesh", "assemble", "sample_excite", "factorize",
                                           "iterate", "op_solve", "extract")] + \
               [(n, C.c_uint32) for n in ("dofs", "stiffness_nonzeros", "op_applications", "restarts")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Event(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("object", C.c_uint32), ("ex_pos", C.c_uint32),
                ("jx", C.c_float), ("jy", C.c_float), ("jz", C.c_float),
                ("pulse_step", C.c_float), ("pulse_gamma", C.c_float), ("accel_amp", C.c_float),
                ("click_b0", C.c_float), ("click_a1", C.c_float), ("click_a2", C.c_float)]


def available_cores():
    """CPUs this process can actually use: its affinity mask, cut to the cgroup CPU quota when there is one (a container on a
    256-core host may be entitled to 16: os.cpu_count() alone would size thread teams sixteen times too large)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def build(force=False):
    so = os.path.join(_HERE, "libmodal_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmodal_oracle.so")
        if not os.path.exists(so):
            so = build()
        # idle team members sleep instead of spinning: on a host that is busy with anything else, spinning OpenMP teams turn a
        # 0.1 s solve into 13 s (measured here with one other single-threaded job running); read by libgomp when it first loads
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(so)
        vp, u32, f32p, f64p, u32p = C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint32)
        L.mo_mesh2modes.restype = vp
        L.mo_mesh2modes.argtypes = [u32, vp, u32, vp, C.POINTER(Material), u32, vp, vp, C.POINTER(SolverConfig), vp, u32, u32, C.c_int, vp]
        L.mo_assemble.restype = vp
        L.mo_assemble.argtypes = [u32, vp, u32, vp, C.POINTER(Material)]
        for name in ("mo_result_free", "mo_system_free", "mo_bank_free"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = None
        for name in ("mo_result_num_modes", "mo_result_num_positions", "mo_result_num_eigenpairs", "mo_result_num_excitations", "mo_result_num_summary_points",
                     "mo_result_basis_rows", "mo_result_basis_cols", "mo_system_dofs", "mo_system_node_count", "mo_system_kept_tets",
                     "mo_bank_num_objects", "mo_bank_num_modes", "mo_bank_active_impacts"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = u32
        L.mo_result_modes.argtypes = [vp, vp, vp, vp, vp, vp]
        L.mo_result_summary.argtypes = [vp, vp, vp]
        L.mo_result_mass_props.argtypes = [vp, vp, vp, vp, vp]
        L.mo_result_profile.argtypes = [vp, C.POINTER(Profile)]
        L.mo_result_sample_point_of_excitation.argtypes = [vp, vp]
        L.mo_result_basis.argtypes = [vp, vp]
        L.mo_system_kept_tet_indices.argtypes = [vp, vp]
        L.mo_system_element_nodes.argtypes = [vp, vp]
        L.mo_system_nnz.argtypes = [vp, C.c_int]
        L.mo_system_nnz.restype = C.c_uint64
        L.mo_system_csc.argtypes = [vp, C.c_int, vp, vp, vp]
        L.mo_quad_basis.argtypes = [vp, vp]
        L.mo_system_eigs.argtypes = [vp, u32, u32, C.c_double, C.c_double, u32, vp, vp, C.POINTER(Profile)]
        L.mo_system_eigs.restype = C.c_int
        L.mo_system_matvec.argtypes = [vp, C.c_int, vp, vp]
        L.mo_postprocess_modes.argtypes = [u32, vp, u32, vp, C.c_float, C.POINTER(Material), C.POINTER(SolverConfig), vp, vp, vp, vp]
        L.mo_postprocess_modes.restype = u32
        L.mo_rescale_modes.argtypes = [u32, vp, u32, vp, C.POINTER(Material), C.POINTER(Material), C.POINTER(SolverConfig), vp, vp, vp, vp]
        L.mo_rescale_modes.restype = u32
        L.mo_mass_properties.argtypes = [u32, vp, u32, vp, C.c_double, vp, C.c_double, vp, vp, vp, vp]
        L.mo_default_config.argtypes = [C.POINTER(SolverConfig)]
        L.mo_set_threads.argtypes = [C.c_int]
        L.mo_max_threads.restype = C.c_int
        # synthesis
        L.mo_bank_create.restype = vp
        L.mo_bank_create.argtypes = [C.c_float, C.c_int]
        L.mo_bank_add_object.argtypes = [vp, u32, u32, u32, vp, vp, u32, vp]
        L.mo_bank_add_object.restype = u32
        L.mo_bank_tune_object.argtypes = [vp, C.c_int, u32, u32, vp, vp, C.c_float]
        L.mo_bank_set_shapes.argtypes = [vp, C.c_int, u32, u32, u32, vp]
        L.mo_bank_set_shapes.restype = C.c_int
        L.mo_bank_set_gains.argtypes = [vp, C.c_int, u32, C.c_float, C.c_float]
        L.mo_bank_install.argtypes = [vp]
        L.mo_bank_set_renderers.argtypes = [vp, u32]
        L.mo_bank_set_click_gain.argtypes = [vp, C.c_float]
        L.mo_bank_set_max_impacts.argtypes = [vp, u32]
        L.mo_bank_enqueue.argtypes = [vp, C.POINTER(Event)]
        L.mo_bank_enqueue.restype = C.c_int
        L.mo_bank_render_f32.argtypes = [vp, vp, u32]
        L.mo_bank_render_f64.argtypes = [vp, vp, u32]
        L.mo_bank_modal_energy.argtypes = [vp]
        L.mo_bank_modal_energy.restype = C.c_double
        L.mo_bank_events_dropped.argtypes = [vp]
        L.mo_bank_events_dropped.restype = C.c_uint64
        L.mo_bank_column.argtypes = [vp, C.c_int, C.c_int, vp]
        L.mo_bank_column.restype = u32
        L.mo_bank_object_state.argtypes = [vp, vp, vp, vp]
        L.mo_recoil_object_filter.argtypes = [C.c_double, C.c_double, C.c_double, vp]
        L.mo_recoil_click_filter.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, vp]
        L.mo_striker_mass.argtypes = [C.c_double, C.c_float, C.c_float]
        L.mo_striker_mass.restype = C.c_double
        for name, n in (("mo_contact_patch_radius", 3), ("mo_static_penetration", 2), ("mo_saturation_penetration", 2), ("mo_punch_stiffness", 2)):
            getattr(L, name).argtypes = [C.c_double] * n
            getattr(L, name).restype = C.c_double
        L.mo_inverse_inertia_tensor.argtypes = [vp, vp, vp]
        L.mo_reduced_contact_mass.argtypes = [C.c_double, vp, vp, vp, C.c_double]
        L.mo_reduced_contact_mass.restype = C.c_double
        L.mo_estimate_contact_time.argtypes = [C.c_double, vp, vp, vp, C.c_double, C.POINTER(Material), C.c_double, C.c_double,
                                               C.POINTER(Material), C.c_double, C.c_double, C.c_double, C.c_double]
        L.mo_estimate_contact_time.restype = C.c_double
        # OpenMP team of the sparse factorisation / solves / Lanczos kernels: a modest team by default.  The loops are
        # short (a front, a panel column); on a many-core host (the GPU box has 256) a team of every core spends its time in
        # fork/join -- measured: a 20 s solve became 650 s.  bench.py sets the team size explicitly for its timed rows.
        L.mo_set_threads(max(1, min(available_cores(), 16)))
        _LIB = L
    return _LIB


def set_threads(n):
    """OpenMP team of the sparse factorisation / solves (1 = the reference's single job thread)."""
    lib().mo_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def material(density, young, poisson, alpha=0.0, beta=0.0):
    return Material(density, young, poisson, alpha, beta)


def default_config(**kw):
    cfg = SolverConfig()
    lib().mo_default_config(C.byref(cfg))
    for k, v in kw.items():
        if k == "fundamental_freq":
            cfg.has_fundamental, cfg.fundamental_freq = (0, 0.0) if v is None else (1, v)
        else:
            setattr(cfg, k, v)
    return cfg


@dataclass
class ModalResult:
    freqs: np.ndarray
    t60s: np.ndarray
    shapes: np.ndarray  # [position][mode][3]
    positions: np.ndarray
    original_fundamental: float
    eigenvalues: np.ndarray
    summary_shapes: np.ndarray  # [position][eigenpair][3]
    mass: float
    center_of_mass: np.ndarray
    inertia_diagonal: np.ndarray
    inertia_orientation_wxyz: np.ndarray
    profile: dict
    sample_point_of_excitation: np.ndarray
    basis: np.ndarray = field(default=None)  # n x cols (column-major as Fortran array)


def mesh2modes(points, tets, mat, excite_positions, baked_scale=(1.0, 1.0, 1.0), config=None, seed_basis=None, keep_basis=False):
    L = lib()
    points = np.ascontiguousarray(points, dtype=np.float64)
    tets = np.ascontiguousarray(tets, dtype=np.uint32)
    ex = np.ascontiguousarray(excite_positions, dtype=np.float32)
    scale = np.asarray(baked_scale, dtype=np.float32)
    cfg = config or default_config()
    seed, rows, cols = None, 0, 0
    if seed_basis is not None:
        seed = np.asfortranarray(seed_basis, dtype=np.float32)
        rows, cols = seed.shape
    h = L.mo_mesh2modes(len(points), _p(points), len(tets), _p(tets), C.byref(mat), len(ex), _p(ex), _p(scale), C.byref(cfg),
                        _p(seed), rows, cols, int(keep_basis), None)
    try:
        k, npos, nev, nex = (L.mo_result_num_modes(h), L.mo_result_num_positions(h), L.mo_result_num_eigenpairs(h), L.mo_result_num_excitations(h))
        freqs, t60s = np.zeros(k, np.float32), np.zeros(k, np.float32)
        shapes, positions = np.zeros((npos, k, 3), np.float32), np.zeros((npos, 3), np.float32)
        orig = C.c_float(0)
        L.mo_result_modes(h, _p(freqs), _p(t60s), _p(shapes), _p(positions), C.byref(orig))
        ev, sshapes = np.zeros(nev), np.zeros((L.mo_result_num_summary_points(h), nev, 3), np.float32)
        L.mo_result_summary(h, _p(ev), _p(sshapes))
        mass = C.c_double(0)
        com, inertia, quat = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(4, np.float32)
        L.mo_result_mass_props(h, C.byref(mass), _p(com), _p(inertia), _p(quat))
        prof = Profile()
        L.mo_result_profile(h, C.byref(prof))
        remap = np.zeros(nex, np.uint32)
        L.mo_result_sample_point_of_excitation(h, _p(remap))
        basis = None
        br, bc = L.mo_result_basis_rows(h), L.mo_result_basis_cols(h)
        if br and bc:
            basis = np.zeros((br, bc), np.float32, order="F")
            L.mo_result_basis(h, _p(basis))
        return ModalResult(freqs, t60s, shapes, positions, orig.value, ev, sshapes, mass.value, com, inertia, quat,
                           prof.as_dict(), remap, basis)
    finally:
        L.mo_result_free(h)


class System:
    """FilterDegenerate + BuildQuadMesh + AssembleQuadratic on the CPU."""

    def __init__(self, points, tets, mat):
        self.L = lib()
        points = np.ascontiguousarray(points, dtype=np.float64)
        tets = np.ascontiguousarray(tets, dtype=np.uint32)
        self.h = self.L.mo_assemble(len(points), _p(points), len(tets), _p(tets), C.byref(mat))
        self.n = self.L.mo_system_dofs(self.h)
        self.node_count = self.L.mo_system_node_count(self.h)
        self.kept_tets = self.L.mo_system_kept_tets(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mo_system_free(self.h)
            self.h = None

    def element_nodes(self):
        out = np.zeros((self.kept_tets, 10), np.uint32)
        self.L.mo_system_element_nodes(self.h, _p(out))
        return out

    def kept_tet_indices(self):
        out = np.zeros(self.kept_tets, np.uint32)
        self.L.mo_system_kept_tet_indices(self.h, _p(out))
        return out

    def csc_lower(self, which):
        """scipy CSC of the lower triangle of K (0) or M (1)."""
        import scipy.sparse as sp
        nnz = self.L.mo_system_nnz(self.h, which)
        colptr, rows, vals = np.zeros(self.n + 1, np.int64), np.zeros(nnz, np.int32), np.zeros(nnz)
        self.L.mo_system_csc(self.h, which, _p(colptr), _p(rows), _p(vals))
        return sp.csc_matrix((vals, rows, colptr), shape=(self.n, self.n))

    def full(self, which):
        import scipy.sparse as sp
        low = self.csc_lower(which)
        return (low + sp.tril(low, -1).T).tocsr()

    def eigs(self, nev, ncv=None, sigma=-(2 * np.pi * 20.0) ** 2, tol=1e-8, max_restarts=100, vectors=True):
        ncv = ncv or min(max(nev + 20, 20), self.n)
        ev = np.zeros(nev)
        vec = np.zeros((self.n, nev), order="F") if vectors else None
        prof = Profile()
        rc = self.L.mo_system_eigs(self.h, nev, ncv, sigma, tol, max_restarts, _p(ev), _p(vec), C.byref(prof))
        if rc != 0:
            raise RuntimeError(f"oracle eigensolve failed rc={rc}")
        return ev, vec, prof.as_dict()

    def matvec(self, which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        self.L.mo_system_matvec(self.h, which, _p(x), _p(y))
        return y


def quad_basis():
    mass, grad = np.zeros((10, 10)), np.zeros((10, 4, 10, 4))
    lib().mo_quad_basis(_p(mass), _p(grad))
    return mass, grad


def postprocess_modes(eigenvalues, shapes, shape_scale, mat, cfg):
    ev = np.ascontiguousarray(eigenvalues, dtype=np.float64)
    sh = np.ascontiguousarray(shapes, dtype=np.float32)
    npos, n = sh.shape[0], len(ev)
    freqs, t60s, out = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((npos, n, 3), np.float32)
    orig = C.c_float(0)
    k = lib().mo_postprocess_modes(n, _p(ev), npos, _p(sh), shape_scale, C.byref(mat), C.byref(cfg), _p(freqs), _p(t60s), _p(out), C.byref(orig))
    return freqs[:k].copy(), t60s[:k].copy(), out.reshape(-1)[: npos * k * 3].reshape(npos, k, 3).copy(), orig.value


def rescale_modes(eigenvalues, summary_shapes, solved, edited, cfg):
    ev = np.ascontiguousarray(eigenvalues, dtype=np.float64)
    sh = np.ascontiguousarray(summary_shapes, dtype=np.float32)
    npos, n = sh.shape[0], len(ev)
    freqs, t60s, out = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((npos, n, 3), np.float32)
    orig = C.c_float(0)
    k = lib().mo_rescale_modes(n, _p(ev), npos, _p(sh), C.byref(solved), C.byref(edited), C.byref(cfg), _p(freqs), _p(t60s), _p(out), C.byref(orig))
    if k == 0xFFFFFFFF:
        return None
    return freqs[:k].copy(), t60s[:k].copy(), out.reshape(-1)[: npos * k * 3].reshape(npos, k, 3).copy(), orig.value


def mass_properties(points, tets, density, scale=(1, 1, 1), length_to_si=1.0):
    points = np.ascontiguousarray(points, dtype=np.float64)
    tets = np.ascontiguousarray(tets, dtype=np.uint32)
    sc = np.asarray(scale, np.float32)
    mass = C.c_double(0)
    com, inertia, quat = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(4, np.float32)
    lib().mo_mass_properties(len(points), _p(points), len(tets), _p(tets), density, _p(sc), length_to_si, C.byref(mass), _p(com), _p(inertia), _p(quat))
    return mass.value, com, inertia, quat


class Bank:
    """The reference's ModalAudio + ModalBank life cycle on the CPU (fp32 as the reference, or fp64)."""
    COLUMNS = ["CoeffRe", "CoeffIm", "StateRe", "StateIm", "RadiationGain", "RadiationArea", "DeflectionGain", "OutPhaseIm",
               "OutPhaseRe", "QuadCompliance", "QuadDriveScale", "ShapeX", "ShapeY", "ShapeZ", "OutGain", "ListenerGain",
               "RadiantRadius", "DeflectionScale"]

    def __init__(self, sample_rate=48000.0, use_double=False):
        self.L = lib()
        self.dbl = use_double
        self.h = self.L.mo_bank_create(sample_rate, int(use_double))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mo_bank_free(self.h)
            self.h = None

    def add_object(self, entity, shapes, positions, indices):
        sh = np.ascontiguousarray(shapes, np.float32)
        pos = np.ascontiguousarray(positions, np.float32)
        idx = np.ascontiguousarray(indices, np.uint32)
        return self.L.mo_bank_add_object(self.h, entity, sh.shape[1], sh.shape[0], _p(sh), _p(pos), len(idx), _p(idx))

    def tune_object(self, obj, freqs, t60s, radius_scale=1.0, live=False):
        f, t = np.ascontiguousarray(freqs, np.float32), np.ascontiguousarray(t60s, np.float32)
        self.L.mo_bank_tune_object(self.h, int(live), obj, min(len(f), len(t)), _p(f), _p(t), radius_scale)

    def set_shapes(self, obj, shapes, live=True):
        sh = np.ascontiguousarray(shapes, np.float32)
        return bool(self.L.mo_bank_set_shapes(self.h, int(live), obj, sh.shape[1], sh.shape[0], _p(sh)))

    def set_gains(self, obj, out_gain, listener_gain=1.0, live=False):
        self.L.mo_bank_set_gains(self.h, int(live), obj, out_gain, listener_gain)

    def install(self):
        self.L.mo_bank_install(self.h)

    def set_renderers(self, n):
        self.L.mo_bank_set_renderers(self.h, n)

    def set_click_gain(self, g):
        self.L.mo_bank_set_click_gain(self.h, g)

    def set_max_impacts(self, n):
        self.L.mo_bank_set_max_impacts(self.h, n)

    def enqueue(self, ev):
        return bool(self.L.mo_bank_enqueue(self.h, C.byref(ev)))

    def render(self, out):
        if self.dbl:
            assert out.dtype == np.float64
            self.L.mo_bank_render_f64(self.h, _p(out), len(out))
        else:
            assert out.dtype == np.float32
            self.L.mo_bank_render_f32(self.h, _p(out), len(out))

    def column(self, name, live=True):
        which = self.COLUMNS.index(name)
        n = self.L.mo_bank_column(self.h, int(live), which, None)
        out = np.zeros(n)
        self.L.mo_bank_column(self.h, int(live), which, _p(out))
        return out

    def object_state(self):
        n = self.L.mo_bank_num_objects(self.h)
        tuned, live, ring = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self.L.mo_bank_object_state(self.h, _p(tuned), _p(live), _p(ring))
        return tuned, live, ring

    @property
    def active_impacts(self):
        return self.L.mo_bank_active_impacts(self.h)

    @property
    def modal_energy(self):
        return self.L.mo_bank_modal_energy(self.h)
