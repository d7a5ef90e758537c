"""ctypes binding of libmodalhip_lab.so (mesheditor_amd/csrc/lab): measurement and experiment entry points that are NOT part of
the path's ABI -- timing loops around the product's kernels, the tridiagonalisation variants called directly, the matrix-free
element-by-element operator.  Used by tests, tools/ and bench.py's secondary figures; the product never loads it."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO_PATH = os.path.join(ROOT, "mesheditor_amd", "libmodalhip_lab.so")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        from mesheditor_amd import _lib as core
        core.lib()  # the product library first (the lab bench links it)
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(SO_PATH)
        vp, u32, i32, f64p = C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_double)
        L.mhl_system_bench_spmm.restype, L.mhl_system_bench_spmm.argtypes = i32, [vp, u32, u32, f64p, f64p]
        L.mhl_system_bench_elementwise.restype, L.mhl_system_bench_elementwise.argtypes = i32, [vp, u32, u32, f64p]
        L.mhl_system_elementwise_matvec.restype, L.mhl_system_elementwise_matvec.argtypes = i32, [vp, vp, vp, u32]
        L.mhl_context_bench_dense.restype, L.mhl_context_bench_dense.argtypes = i32, [vp, i32, C.c_uint64, u32, u32, u32, f64p]
        L.mhl_context_tridiagonalize.restype, L.mhl_context_tridiagonalize.argtypes = i32, [vp, i32, u32, vp, vp, vp, u32, f64p]
        L.mhl_context_bench_stream.restype, L.mhl_context_bench_stream.argtypes = i32, [vp, C.c_uint64, u32, f64p, f64p]
        L.mhl_context_gram.restype, L.mhl_context_gram.argtypes = i32, [vp, C.c_uint64, vp, u32, vp, u32, vp]
        L.mhl_context_pool_stats.restype, L.mhl_context_pool_stats.argtypes = i32, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.mhl_context_potrf_inverse.restype, L.mhl_context_potrf_inverse.argtypes = i32, [vp, u32, vp, vp, vp, vp, vp]
        L.mhl_context_spd_inverse.restype, L.mhl_context_spd_inverse.argtypes = i32, [vp, u32, vp, vp, u32, f64p]
        L.mhl_context_small_gemm.restype, L.mhl_context_small_gemm.argtypes = i32, [vp, i32, i32, u32, u32, u32, C.c_double, vp, u32, vp, u32, C.c_double, vp, u32, u32, f64p]
        L.mhl_context_tridiagonalize_full.restype, L.mhl_context_tridiagonalize_full.argtypes = i32, [vp, i32, u32, vp, vp, vp, vp, vp, u32, f64p]
        L.mhl_graph_aggregates.restype, L.mhl_graph_aggregates.argtypes = u32, [vp, vp, u32, u32, u32, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def bench_spmm(system, width, reps=20):
    """(average ms, algorithmic bytes) of one fp64 K x product over an n x width panel."""
    ms, by = C.c_double(0), C.c_double(0)
    system.ctx.check(lib().mhl_system_bench_spmm(system.h, width, reps, C.byref(ms), C.byref(by)))
    return ms.value, by.value


def bench_elementwise(system, width, reps=20):
    ms = C.c_double(0)
    system.ctx.check(lib().mhl_system_bench_elementwise(system.h, width, reps, C.byref(ms)))
    return ms.value


def elementwise_matvec(system, x):
    """(K - sigma M) x at the reference's shift, element by element without the assembled matrix."""
    x = np.asfortranarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    y = np.zeros_like(x, order="F")
    system.ctx.check(lib().mhl_system_elementwise_matvec(system.h, _p(x), _p(y), x.shape[1]))
    return y


def bench_dense(ctx, kind, n, wa, wb, reps=10):
    ms = C.c_double(0)
    ctx.check(lib().mhl_context_bench_dense(ctx.h, kind, n, wa, wb, reps, C.byref(ms)))
    return ms.value


def tridiagonalize(ctx, a, variant=0, reps=1):
    """(d, e, average ms) of the Householder tridiagonalisation of a symmetric matrix (order <= 256)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    m = a.shape[0]
    d, e, ms = np.zeros(m), np.zeros(m - 1), C.c_double(0)
    ctx.check(lib().mhl_context_tridiagonalize(ctx.h, variant, m, _p(a), _p(d), _p(e), reps, C.byref(ms)))
    return d, e, ms.value


def tridiagonalize_full(ctx, a, variant=2, reps=1):
    """(d, e, reflectors [column-major m x m, LAPACK's lower storage], tau, average ms); variant 2 = the wide kernel (order <= 768)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    m = a.shape[0]
    d, e, refl, tau, ms = np.zeros(m), np.zeros(m - 1), np.zeros((m, m)), np.zeros(m), C.c_double(0)
    ctx.check(lib().mhl_context_tridiagonalize_full(ctx.h, variant, m, _p(a), _p(d), _p(e), _p(refl), _p(tau), reps, C.byref(ms)))
    return d, e, refl.T.copy(), tau, ms.value  # (the device's column-major image read as rows: transposed back)


def gram(ctx, x, y):
    """X^T Y through the solver's Gram kernel (x: n x wa, y: n x wb)."""
    xc, yc = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
    n, wa = xc.shape
    wb = yc.shape[1]
    g = np.zeros((wb, wa))  # column-major wa x wb
    ctx.check(lib().mhl_context_gram(ctx.h, n, _p(xc), wa, _p(yc), wb, _p(g)))
    return g.T.copy()


def bench_stream(ctx, nbytes=4 << 30, reps=10):
    """(copy GB/s counting read + written bytes, read GB/s) of streaming kernels over `nbytes`: the device's measured HBM ceilings."""
    c, r = C.c_double(0), C.c_double(0)
    ctx.check(lib().mhl_context_bench_stream(ctx.h, nbytes, reps, C.byref(c), C.byref(r)))
    return c.value, r.value


def pool_stats(ctx):
    """(bytes held from the device, bytes idle in the cache, cap on the idle bytes) of the context's device pool."""
    r, i, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    ctx.check(lib().mhl_context_pool_stats(ctx.h, C.byref(r), C.byref(i), C.byref(c)))
    return r.value, i.value, c.value


def potrf_inverse(ctx, a, dscale):
    """(L, L^-1, info[2]) of the Cholesky-QR step's one-launch kernel: L = diag(1 / dscale) chol(a) (rows with dscale <= 0: identity rows)."""
    af = np.asfortranarray(a, dtype=np.float64)
    d = np.ascontiguousarray(dscale, dtype=np.float64)
    l, linv, info = np.zeros_like(af, order="F"), np.zeros_like(af, order="F"), np.zeros(2, np.int32)
    ctx.check(lib().mhl_context_potrf_inverse(ctx.h, af.shape[0], _p(af), _p(d), _p(l), _p(linv), _p(info)))
    return l, linv, info


def spd_inverse(ctx, a, reps=1):
    """(a^-1, average ms) through the coarse set-up's one-workgroup Gauss-Jordan kernel (order <= 128)."""
    af = np.asfortranarray(a, dtype=np.float64)
    out, ms = np.zeros_like(af, order="F"), C.c_double(0)
    ctx.check(lib().mhl_context_spd_inverse(ctx.h, af.shape[0], _p(af), _p(out), reps, C.byref(ms)))
    return out, ms.value


def small_gemm(ctx, a, b, c=None, ta=False, tb=False, alpha=1.0, beta=0.0, reps=1):
    """(alpha op(a) op(b) + beta c, average ms) through the Rayleigh-Ritz step's small-product kernel; a, b, c are numpy matrices
    (handed over column-major)."""
    af, bf = np.asfortranarray(a, dtype=np.float64), np.asfortranarray(b, dtype=np.float64)
    M, K = (af.shape[1], af.shape[0]) if ta else af.shape
    N = bf.shape[0] if tb else bf.shape[1]
    cf = np.asfortranarray(np.zeros((M, N)) if c is None else c, dtype=np.float64).copy(order="F")
    ms = C.c_double(0)
    ctx.check(lib().mhl_context_small_gemm(ctx.h, int(ta), int(tb), M, N, K, alpha, _p(af), af.shape[0], _p(bf), bf.shape[0], beta, _p(cf), M, reps, C.byref(ms)))
    return cf, ms.value


def graph_aggregates(row_ptr, col, target=16, max_order=6144):
    """(aggregate of every node, aggregate count) for a CSR node graph with its diagonal entries (host code, no device)."""
    rp = np.ascontiguousarray(row_ptr, np.uint32)
    cl = np.ascontiguousarray(col, np.uint32)
    n = len(rp) - 1
    out = np.zeros(n, np.uint32)
    na = lib().mhl_graph_aggregates(_p(rp), _p(cl), n, target, max_order, _p(out))
    return out, int(na)
