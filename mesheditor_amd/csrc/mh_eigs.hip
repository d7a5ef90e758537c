// Generalised eigensolve K x = lambda M x for the lowest pairs, on the device.
//
// Replaces the reference's Spectra SymGEigsShiftSolver over an Accelerate sparse Cholesky (cold branch,
// src/audio/mesh2modes.cpp:470,485-491; CholeskyShiftInvert.cpp) and its warm-started SubspaceIterate (:339-428)
// by a block LOBPCG (Knyazev 2001; basis handling after Hetmaniuk & Lehoucq 2006) on the same shifted pencil
// (A, M), A = K - sigma M, sigma = -(2 pi MinModeFreq)^2 (:460), preconditioned by one symmetric three-level cycle:
//   level 2  P2 operator, Chebyshev-Jacobi smoothing (SpMM-bound), exact corrections on sliver patches (mh_patch.hip)
//   level 1  its exact Galerkin restriction to the corner nodes (P1 subset of P2), gamma cycles
//   level 0  rigid-body modes of graph-grown node aggregates; explicit inverse by our block Gauss-Jordan elimination (mh_build_hierarchy)
// Tall-skinny products: k_gram_blocked / k_combine (fp64 MFMA, mh_dense.hip), the vendor dgemm only for blocks of >= 400 basis columns.
// The Rayleigh-Ritz problem: our tridiagonalisation (k_sytrd_regs up to order 256, k_sytrd_wide above), partial spectrum by multisection + inverse iteration
// (k_tridiag_lowest, k_tridiag_invit), one-launch back-transformation (k_apply_q); rocSOLVER only as the fall-back of a failed check.
#include "mh_common.h"

#include <algorithm>
#include <cmath>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <numeric>
#include <type_traits>

__global__ void k_shift_values(const double *kval, const double *mval, size_t nblocks, double sigma, double *aval);
__global__ void k_diag_inverse(const uint32_t *row_ptr, const uint32_t *col, const double *aval, uint32_t nnodes, double *dinv);
__global__ void k_coarse_matrix(const uint32_t *row_ptr, const uint32_t *col, const double *aval, const double *tmat, const uint32_t *agg_of, const uint32_t *agg_ptr,
                                const uint32_t *agg_nodes, uint32_t nagg, double *a0);
__global__ void k_fix_coarse_diag(double *a0, uint32_t n0, double rel);

namespace {
constexpr int TB = 256;

// Concurrency contract of a solve.  Measured on this stack (ROCm 7.2, MI355X) with three host threads solving on their
// own contexts: rocsolver_dpotrf of the SAME 3 690 x 3 690 coarse operator (identical input checksums) returned
// info != 0 in ~10 % of the calls -- also when every rocSOLVER call ran under one lock, and specifically whenever
// another stream was running the register-blocked Gram kernel (not the SpMM, basis-update or assembly kernels); the
// operator buffer itself was never touched while idle.  rocSOLVER's dense factorisation is disturbed by unrelated
// concurrent work; nothing else of the solve is (tools/concurrent_solves.py: ~500 concurrent solves from 2-8 threads, every
// eigenvalue bit-identical to the serial run).  So:
//   * the coarse operator is inverted by our own block Gauss-Jordan elimination (mh_build_hierarchy: one-workgroup
//     diagonal-block inverses + rocBLAS dgemm), which needs no isolation -- and is faster than potrf + potri;
//   * solves of different contexts iterate side by side.  Each iteration holds the device phase lock SHARED, as do the other
//     entry points that launch work (assembly, nearest points, shape gathers, bank rendering);
//   * what still goes through a rocSOLVER factorisation -- the tiny-system dense eigensolve (sygvd) -- takes the lock
//     EXCLUSIVELY after a device-wide synchronisation: it runs alone;
//   * the iteration itself calls no rocSOLVER factorisation (round 4: mh_potrf factors blocks wider than 128 columns in 128-column blocks
//     with our kernels and rocBLAS level 3; rocsolver_dpotrf lost rank beside other solves);
//   * (round 5) solves of EVERY block width overlap.  Rounds 2-4 had kept blocks wider than 128 columns under the exclusive lock because
//     other contexts' solves broke beside them; the cause was a barrier of k_sytrd_multi / k_sytrd_wide that hipcc had left without its
//     LDS wait, exposed only beside rocBLAS's LDS-bound dsymm kernel, which only wide blocks launch (mh_common.h: mh_lds_writes_landed;
//     DESIGN.md section 6).  The failures recorded above for rocSOLVER's potrf have the same signature (wrong only beside an LDS-heavy
//     kernel of another stream); that library is not ours to fix, so what goes through it stays under the exclusive lock.
// 2.0x the serial throughput on a batch of 30k-tet meshes with three threads, 2.5x on 4k-tet meshes with eight.
// MH_CONCURRENT_SOLVES=0 restores one-solve-at-a-time (g_solve_mutex).
std::mutex g_solve_mutex;
// The solver's run-time switches, read ONCE per process (first use) into one immutable struct.  These are all there are
// (DESIGN.md section 10 documents them); the alternatives that were measured and lost are gone from the code, the record of
// each is in DESIGN.md and profiles/.
//   MH_VERBOSE=1            per-iteration log on stderr
//   MH_PRECOND_FP64=1       double-precision smoothers (default: single precision with double residuals between levels)
//   MH_CYCLE=d2,d1,g,ratio  shape of the preconditioner cycle: Chebyshev degrees of the P2 and P1 smoothers, P1 cycles per
//                           application, spectrum ratio lmax / lmin the smoothers target; 0 or missing keeps a built-in value
//   MH_TEST=...             test hooks, comma separated: sytrd_giveup (treat every multi-workgroup tridiagonalisation as timed out: the
//                           fall-backs to the one-workgroup kernel / the library's syevd run), no_tridiag_wide (orders above 256: the
//                           fall-back of the partial-spectrum stage -- the library's divide and conquer and ormtr -- runs)
//   (the A/B hooks of round 4 -- wide tridiagonalisation, small products, pivot look-ahead, power-iteration norms, wide Gram -- are gone
//   with their measurements recorded in profiles/r04_setup_ab.txt, r04_dense_kernels.txt, r04_gram_cuts.txt and DESIGN.md section 10)
//   MH_TEST=redzone / farzone (mh_common.h): 64 KB guard zones around every pool array, checked at release / 32 MB of unchecked slack
// and, read elsewhere: MH_CONCURRENT_SOLVES=0 (one solve at a time), MH_AGG (aggregate size target), MH_PATCH_Q (sliver-patch
// threshold), MH_POOL_CAP_MB (idle device-pool cap); MH_TEST also understands `poison` (NaN-filled pool allocations).
struct Switches {
    bool verbose = getenv("MH_VERBOSE") != nullptr;
    bool fp32_prec = !(getenv("MH_PRECOND_FP64") && atoi(getenv("MH_PRECOND_FP64")) != 0);
    int deg2 = 0, deg1 = 0, gamma = 0; // 0: the built-in cycle shape
    double cheb_ratio = 0.0, cheb_ratio1 = 0.0; // (ratio1: the P1 level's own interval, optional fifth value of MH_CYCLE)
    bool test_sytrd_giveup = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "sytrd_giveup");
    bool no_tridiag_wide = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "no_tridiag_wide");
    bool test_last_resort = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "last_resort"); // every solve of more than 12 288 unknowns goes straight to the last resort
    bool test_coarse_pivot = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "coarse_pivot"); // the first coarse elimination of a system reports a non-positive pivot
    bool test_selfcheck_fail = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "selfcheck_fail"); // the first solve's Rayleigh-Ritz self-check reports a failure
    bool no_poly_start = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "no_poly_start"); // (A/B hook of round 5: the cold start block as rounds 1-4 had it)
    Switches() {
        if (const char *c = getenv("MH_CYCLE")) {
            double v[5] = {0, 0, 0, 0, 0};
            sscanf(c, "%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4]);
            deg2 = std::max(0, int(v[0])), deg1 = std::max(0, int(v[1])), gamma = std::max(0, int(v[2]));
            cheb_ratio = v[3];
            cheb_ratio1 = v[4];
        }
    }
};
const Switches &switches() {
    static const Switches s; // C++11 magic static: initialised once, thread-safe
    return s;
}
// Design constants (each was once a switch; the losing side of every comparison is recorded in DESIGN.md section 10)
constexpr int kPowerIterations = 20; // spectral-bound estimate of the smoothers (12 steps left the bound 15 % low on the scan meshes -- more than the 1.1 safety factor -- and one patch-threshold setting then failed to converge; the steps run beside the coarse elimination)
constexpr uint32_t kPrecondColumns = 256; // (narrower slabs measured slower on the 215-pair solves: 128 -> +2 %, 80 -> +5 %, 64 -> +9 %) // widest panel of one preconditioner application (the single-precision wide-load products: 64 lanes x 4)
constexpr uint32_t kSkipP = 4;       // no conjugate directions in the first iterations of a cold start
constexpr uint32_t kGuardPercent = 10; // guard vectors: max(15, 10 % of the wanted pairs)
constexpr float kFlatShape = 1e-4f; // element shape measure below which a mesh counts as having flat cells: double-precision smoothers from the start
constexpr size_t kDenseLastResort = 12288; // unknowns up to which a solve that did not converge is redone as one dense eigensolve (2 x 1.2 GB, seconds)

// readers-writer lock with writer priority (glibc's shared_mutex prefers readers: iterating solves would starve a factorisation)
struct PhaseLock {
    std::mutex m;
    std::condition_variable cv;
    int readers = 0, writers_waiting = 0;
    bool writer = false;
    void lock_shared() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !writer && writers_waiting == 0; });
        ++readers;
    }
    void unlock_shared() {
        std::unique_lock<std::mutex> l(m);
        if (--readers == 0) cv.notify_all();
    }
    void lock() {
        std::unique_lock<std::mutex> l(m);
        ++writers_waiting;
        cv.wait(l, [&] { return !writer && readers == 0; });
        --writers_waiting;
        writer = true;
    }
    void unlock() {
        std::unique_lock<std::mutex> l(m);
        writer = false;
        cv.notify_all();
    }
};
// One lock per DEVICE (round 6): what a factorisation must be alone with is its own GPU's queue -- a tiny dense solve on GPU 0 has no
// business stalling the solves of GPUs 1-7 of the same process (modal::SolveBatch with several devices per process; one process per GPU
// never sees more than one).  hipDeviceSynchronize below drains the current device only, which is the one locked.
constexpr int kMaxDevices = 64;
PhaseLock g_phase_of_device[kMaxDevices];
PhaseLock &phase_of(int device) { return g_phase_of_device[unsigned(device) % kMaxDevices]; }
const bool g_concurrent = !(getenv("MH_CONCURRENT_SOLVES") && atoi(getenv("MH_CONCURRENT_SOLVES")) == 0);
struct SharedPhase { // no-ops in the serialised mode (and for a solve that holds the device exclusively: enabled = false)
    PhaseLock &phase;
    bool held = false, enabled = true;
    explicit SharedPhase(int device, bool on = true) : phase(phase_of(device)), enabled(on) { acquire(); }
    ~SharedPhase() { release(); }
    void acquire() { if (g_concurrent && enabled && !held) { phase.lock_shared(); held = true; } }
    void release() { if (held) { phase.unlock_shared(); held = false; } }
};
struct ExclusivePhase {
    PhaseLock &phase;
    bool held = false;
    explicit ExclusivePhase(int device) : phase(phase_of(device)) {
        if (g_concurrent) {
            phase.lock();
            held = true;
            (void)hipSetDevice(device);
            (void)hipDeviceSynchronize(); // everything the iterating solves had queued ON THIS DEVICE has drained: it is ours
        }
    }
    ~ExclusivePhase() { if (held) phase.unlock(); }
};

struct Timer {
    mh_context *ctx;
    hipEvent_t a, b;
    explicit Timer(mh_context *c) : ctx(c) {
        HIP_CHECK(hipEventCreate(&a));
        HIP_CHECK(hipEventCreate(&b));
        HIP_CHECK(hipEventRecord(a, ctx->stream));
    }
    double stop() {
        HIP_CHECK(hipEventRecord(b, ctx->stream));
        HIP_CHECK(hipEventSynchronize(b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        return ms * 1e-3;
    }
    ~Timer() {
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    }
};

// ---- elementwise panel kernels (n x w row-major, flat index) ------------------------------------------------
__global__ void k_random_panel(double *__restrict__ x, size_t count, uint64_t seed) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    auto mix = [](uint64_t z) {
        z += 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    const uint64_t a = mix(seed + 2 * i), b = mix(seed + 2 * i + 1);
    const double u1 = (double(a >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = double(b >> 11) * (1.0 / 9007199254740992.0);
    x[i] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

// Chebyshev first step: r = b - t (t optional), d = dinv * r / theta, x = d or x += d.
template<typename T>
__global__ void k_cheb_init(const T *__restrict__ b, const T *__restrict__ t, const T *__restrict__ dinv, T inv_theta, T *__restrict__ r, T *__restrict__ d,
                            T *__restrict__ x, int accumulate, size_t rows, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const T rv = t ? b[i] - t[i] : b[i];
    const T dv = dinv[i / w] * rv * inv_theta;
    r[i] = rv;
    d[i] = dv;
    x[i] = accumulate ? x[i] + dv : dv;
}
// Single-precision panels have pitches that are multiples of 4: the smoother's elementwise steps move 16 bytes per lane.
typedef float f4_t __attribute__((ext_vector_type(4)));
__global__ void k_cheb_init_v4(const f4_t *__restrict__ b, const f4_t *__restrict__ t, const float *__restrict__ dinv, float inv_theta, f4_t *__restrict__ r,
                               f4_t *__restrict__ d, f4_t *__restrict__ x, int accumulate, size_t rows, uint32_t w4) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w4) return;
    const f4_t rv = t ? b[i] - t[i] : b[i];
    const f4_t dv = (dinv[i / w4] * inv_theta) * rv;
    r[i] = rv;
    d[i] = dv;
    x[i] = accumulate ? x[i] + dv : dv;
}
__global__ void k_cheb_last_v4(const f4_t *__restrict__ t, const float *__restrict__ dinv, float c1, float c2, const f4_t *__restrict__ r, const f4_t *__restrict__ d,
                               const f4_t *__restrict__ x, double *__restrict__ z, size_t rows, uint32_t w4, uint32_t wo) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w4) return;
    const size_t row = i / w4;
    const uint32_t c = uint32_t(i % w4) * 4;
    const f4_t rv = r[i] - t[i];
    const f4_t zv = x[i] + (c1 * d[i] + (c2 * dinv[row]) * rv);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c + e < wo) z[row * wo + c + e] = double(zv[e]);
}
__global__ void k_prolong_p1_v4(const f4_t *__restrict__ x1, const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb, f4_t *__restrict__ x2, uint32_t nnodes,
                                uint32_t w4) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nnodes) * 3 * w4) return;
    const uint32_t c = uint32_t(i % w4), comp = uint32_t((i / w4) % 3), node = uint32_t(i / (size_t(3) * w4));
    const uint32_t a = pa[node], bb = pb[node];
    const f4_t va = x1[(size_t(3) * a + comp) * w4 + c];
    x2[i] += a == bb ? va : 0.5f * (va + x1[(size_t(3) * bb + comp) * w4 + c]);
}
// The same with zero initial iterate, reading the double-precision right-hand side at its own pitch ws and leaving its
// single-precision copy b32 (pitch w, zero padded) for the later steps: conversion and first step in one pass.
__global__ void k_cheb_init_convert(const double *__restrict__ src, uint32_t ws, const float *__restrict__ dinv, float inv_theta, float *__restrict__ b32,
                                    float *__restrict__ r, float *__restrict__ d, float *__restrict__ x, size_t rows, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const size_t row = i / w;
    const uint32_t c = uint32_t(i % w);
    const float rv = c < ws ? float(src[row * ws + c]) : 0.f;
    const float dv = dinv[row] * rv * inv_theta;
    b32[i] = rv;
    r[i] = rv;
    d[i] = dv;
    x[i] = dv;
}
template<typename T>
__global__ void k_cheb_step(const T *__restrict__ t, const T *__restrict__ dinv, T c1, T c2, T *__restrict__ r, T *__restrict__ d, T *__restrict__ x, size_t rows,
                            uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const T rv = r[i] - t[i];
    const T dv = c1 * d[i] + c2 * dinv[i / w] * rv;
    r[i] = rv;
    d[i] = dv;
    x[i] += dv;
}
// Last Chebyshev step of the cycle: the iterate leaves in double, at the caller's (unpadded) pitch wo <= w.
template<typename T>
__global__ void k_cheb_last(const T *__restrict__ t, const T *__restrict__ dinv, T c1, T c2, const T *__restrict__ r, const T *__restrict__ d, const T *__restrict__ x,
                            double *__restrict__ z, size_t rows, uint32_t w, uint32_t wo) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const size_t row = i / w;
    const uint32_t c = uint32_t(i % w);
    if (c >= wo) return;
    const T rv = r[i] - t[i];
    z[row * wo + c] = double(x[i] + (c1 * d[i] + c2 * dinv[row] * rv));
}
template<typename S, typename D> __global__ void k_convert(const S *__restrict__ src, D *__restrict__ dst, size_t count) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = D(src[i]);
}
// dst (rows x wd) <- src (rows x ws), converting; columns beyond the source's are zero, beyond the destination's dropped
template<typename S, typename D> __global__ void k_convert_pitch(const S *__restrict__ src, uint32_t ws, D *__restrict__ dst, uint32_t wd, size_t rows) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * wd) return;
    const size_t row = i / wd;
    const uint32_t c = uint32_t(i % wd);
    dst[i] = c < ws ? D(src[row * ws + c]) : D(0);
}

// r1 = P^T (b - t): corner value plus half of every incident edge's midside value.
template<typename T>
__global__ void k_restrict_p1(const T *__restrict__ b, const T *__restrict__ t, const uint32_t *__restrict__ p1_corner, const uint32_t *__restrict__ eptr,
                              const uint32_t *__restrict__ emid, T *__restrict__ r1, uint32_t npts, uint32_t w, uint32_t wb = 0) {
    // wb: pitch of b when it differs from the pitch w of t and r1 (columns >= wb of b read as zero)
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(npts) * 3 * w) return;
    const uint32_t c = uint32_t(i % w), comp = uint32_t((i / w) % 3), p = uint32_t(i / (size_t(3) * w));
    const uint32_t pb = wb ? wb : w;
    auto res = [&](uint32_t node) {
        const size_t row = size_t(3) * node + comp;
        return (c < pb ? b[row * pb + c] : T(0)) - t[row * w + c];
    };
    T s = res(p1_corner[p]);
    T h = 0;
    for (uint32_t e = eptr[p]; e < eptr[p + 1]; ++e) h += res(emid[e]);
    r1[i] = s + T(0.5) * h;
}
// x2 += P x1
template<typename T>
__global__ void k_prolong_p1(const T *__restrict__ x1, const uint32_t *__restrict__ pa, const uint32_t *__restrict__ pb, T *__restrict__ x2, uint32_t nnodes,
                             uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nnodes) * 3 * w) return;
    const uint32_t c = uint32_t(i % w), comp = uint32_t((i / w) % 3), node = uint32_t(i / (size_t(3) * w));
    const uint32_t a = pa[node], bb = pb[node];
    const T va = x1[(size_t(3) * a + comp) * w + c];
    x2[i] += a == bb ? va : T(0.5) * (va + x1[(size_t(3) * bb + comp) * w + c]);
}
// r0 = T^T (b - t) per aggregate (6 rows each), one thread per (aggregate dof, column), the aggregate's nodes in list order; the
// coarse level stays double
template<typename T>
__global__ void k_restrict_agg(const T *__restrict__ b, const T *__restrict__ t, const double *__restrict__ tmat, double *__restrict__ r0,
                               const uint32_t *__restrict__ agg_ptr, const uint32_t *__restrict__ agg_nodes, uint32_t nagg, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nagg) * 6 * w) return;
    const uint32_t c = uint32_t(i % w), q = uint32_t((i / w) % 6), a = uint32_t(i / (size_t(6) * w));
    double s = 0;
    for (uint32_t l = agg_ptr[a]; l < agg_ptr[a + 1]; ++l) {
        const uint32_t nd = agg_nodes[l];
        const double *tm = tmat + 18 * size_t(nd);
        for (int p = 0; p < 3; ++p) {
            const size_t o = (size_t(3) * nd + p) * w + c;
            s += tm[6 * p + q] * double(b[o] - t[o]);
        }
    }
    r0[i] = s;
}
// x1 += T x0
template<typename T>
__global__ void k_prolong_agg(const double *__restrict__ x0, const double *__restrict__ tmat, T *__restrict__ x1, const uint32_t *__restrict__ agg_of, uint32_t npts,
                              uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(npts) * 3 * w) return;
    const uint32_t c = uint32_t(i % w), p = uint32_t((i / w) % 3), nd = uint32_t(i / (size_t(3) * w));
    const uint32_t a = agg_of[nd];
    const double *tm = tmat + 18 * size_t(nd) + 6 * p;
    double s = 0;
    for (int q = 0; q < 6; ++q) s += tm[q] * x0[(size_t(6) * a + q) * w + c];
    x1[i] += T(s);
}

// R[:, k] = AX[:, idx[k]] - theta[idx[k]] * MX[:, idx[k]]
__global__ void k_residual(const double *__restrict__ ax, const double *__restrict__ mx, const double *__restrict__ theta, const uint32_t *__restrict__ idx,
                           double *__restrict__ r, size_t rows, uint32_t b, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const uint32_t k = uint32_t(i % w);
    const size_t row = i / w;
    const uint32_t src = idx ? idx[k] : k;
    r[i] = ax[row * b + src] - theta[src] * mx[row * b + src];
}

// Residuals of all b columns and the three column norms the convergence test needs, in one pass over the panels:
// R = AX - theta MX, partial[blk][0..2][c] = sums of r^2, (Mx)^2, x^2 over the block's rows (fixed order).  ||x||^2 is only
// used by the rounding-floor clause, i.e. for columns with |theta| < near_limit (the rigid-body pairs): X is read for those
// columns alone (a fifth of its cache lines instead of a quarter of the kernel's traffic), the others report 0.
__global__ void k_residual_norms(const double *__restrict__ ax, const double *__restrict__ mx, const double *__restrict__ xx, const double *__restrict__ theta,
                                 double near_limit, double *__restrict__ r, size_t rows, uint32_t b, uint32_t rows_per_block, double *__restrict__ partial,
                                 const double *__restrict__ dinv) {
    // dinv (optional): the norms are taken in the Jacobi scaling -- sums of r^2 / D, (Mx)^2 / D, x^2 D (see converged_or_locked)
    const uint32_t c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= b) return;
    const size_t r0 = size_t(blockIdx.x) * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    const double th = theta[c];
    const bool need_x = fabs(th) < near_limit;
    double sr = 0, sm = 0, sx = 0;
    for (size_t row = r0; row < r1; ++row) {
        const size_t i = row * b + c;
        const double m = mx[i], res = ax[i] - th * m;
        const double wgt = dinv ? dinv[row] : 1.0;
        r[i] = res;
        sr += res * res * wgt;
        sm += m * m * wgt;
        if (need_x) {
            const double x = xx[i];
            sx += dinv ? x * x / wgt : x * x;
        }
    }
    double *p = partial + size_t(blockIdx.x) * 3 * b;
    p[c] = sr;
    p[b + c] = sm;
    p[2 * size_t(b) + c] = sx;
}

// Column sums of squares, two deterministic stages.
__global__ void k_colsumsq_partial(const double *__restrict__ x, size_t rows, uint32_t w, double *__restrict__ partial, uint32_t rows_per_block) {
    const uint32_t c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= w) return;
    const size_t r0 = size_t(blockIdx.x) * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double s = 0;
    for (size_t r = r0; r < r1; ++r) {
        const double v = x[r * w + c];
        s += v * v;
    }
    partial[size_t(blockIdx.x) * w + c] = s;
}
// Column sums (not of squares) of an n x w panel in row blocks: first stage for the per-node norm partials of the fused residual
__global__ void k_colsum_partial(const double *__restrict__ x, size_t rows, uint32_t w, double *__restrict__ partial, uint32_t rows_per_block) {
    const uint32_t c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= w) return;
    const size_t r0 = size_t(blockIdx.x) * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double s = 0;
    for (size_t r = r0; r < r1; ++r) s += x[r * w + c];
    partial[size_t(blockIdx.x) * w + c] = s;
}
// one 256-thread block per column: strided partial sums, then a fixed binary tree in LDS
__global__ void k_colsumsq_final(const double *__restrict__ partial, uint32_t nblocks, uint32_t w, double *__restrict__ out) {
    __shared__ double s[256];
    const uint32_t c = blockIdx.x;
    double acc = 0;
    for (uint32_t b = threadIdx.x; b < nblocks; b += 256) acc += partial[size_t(b) * w + c];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (int(threadIdx.x) < h) s[threadIdx.x] += s[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = s[0];
}
__global__ void k_scale_cols_inv_sqrt(double *__restrict__ x, const double *__restrict__ sumsq, size_t rows, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const double s = sumsq[i % w];
    x[i] *= s > 0 ? rsqrt(s) : 0.0;
}
__global__ void k_dinv_mul(const double *__restrict__ t, const double *__restrict__ dinv, double *__restrict__ v, size_t rows, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < rows * w) v[i] = dinv[i / w] * t[i];
}

// ---- small dense helpers (column-major, leading dimension ld) ------------------------------------------------
__global__ void k_set_identity_blocks(double *__restrict__ ga, double *__restrict__ gm, const double *__restrict__ theta, uint32_t b, uint32_t ld) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b) return;
    ga[size_t(i) * ld + i] = theta[i];
    gm[size_t(i) * ld + i] = 1.0;
}
// dst block (w x w at leading dimension ld) <- src (w x w, ld w), or identity when src is null
__global__ void k_place_block(double *__restrict__ dst, uint32_t ld, const double *__restrict__ src, uint32_t w) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * w) return;
    const uint32_t r = i % w, c = i / w;
    dst[size_t(c) * ld + r] = src ? src[size_t(c) * w + r] : (r == c ? 1.0 : 0.0);
}
// d = 1/sqrt(diag G); Gs = D G D written to the w x w block that follows G in memory (ld = w for both)
__global__ void k_scale_gram(double *__restrict__ g, uint32_t w, uint32_t ld, double *__restrict__ dscale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * w) return;
    const uint32_t r = i % w, c = i / w;
    const double dr = g[size_t(r) * ld + r], dc = g[size_t(c) * ld + c];
    const double sr = dr > 0 ? rsqrt(dr) : 0.0, sc = dc > 0 ? rsqrt(dc) : 0.0;
    if (r == c) dscale[r] = sr;
    g[size_t(w) * ld + size_t(c) * ld + r] = g[size_t(c) * ld + r] * sr * sc;
}
// L <- diag(1/d) L (lower triangle), so that W L^-T applies the column scaling as well
__global__ void k_unscale_chol(double *__restrict__ l, uint32_t w, uint32_t ld, const double *__restrict__ dscale) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * w) return;
    const uint32_t r = i % w, c = i / w;
    if (r < c) { l[size_t(c) * ld + r] = 0; return; }
    const double d = dscale[r];
    l[size_t(c) * ld + r] = d > 0 ? l[size_t(c) * ld + r] / d : (r == c ? 1.0 : 0.0);
}
// Cp[r][k] = r < b ? 0 : C[r][idx[k]]
__global__ void k_build_cp(const double *__restrict__ c, const uint32_t *__restrict__ idx, uint32_t b, uint32_t m, uint32_t w, uint32_t ldc,
                           double *__restrict__ cp, uint32_t ldp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * w) return;
    const uint32_t r = i % m, k = i / m;
    cp[size_t(k) * ldp + r] = r < b ? 0.0 : c[size_t(idx ? idx[k] : k) * ldc + r];
}
__global__ void k_gather_cols(const double *__restrict__ src, const uint32_t *__restrict__ idx, double *__restrict__ dst, size_t rows, uint32_t wsrc, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    dst[i] = src[(i / w) * wsrc + idx[i % w]];
}
// dst[:, idx[k]] = src[:, k]
__global__ void k_scatter_cols(const double *__restrict__ src, const uint32_t *__restrict__ idx, double *__restrict__ dst, size_t rows, uint32_t wdst, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    dst[(i / w) * wdst + idx[i % w]] = src[i];
}
// dst[:, idx[k]] = src[:, k] for a source of pitch wsrc
__global__ void k_scatter_cols_pitch(const double *__restrict__ src, uint32_t wsrc, const uint32_t *__restrict__ idx, double *__restrict__ dst, size_t rows, uint32_t wdst,
                                     uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    dst[(i / w) * wdst + idx[i % w]] = src[(i / w) * wsrc + i % w];
}
// dst (rows x cols at leading dimension ld) -= src (leading dimension lds)
__global__ void k_sub_block(double *__restrict__ dst, uint32_t ld, const double *__restrict__ src, uint32_t lds, uint32_t rows, uint32_t cols) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const uint32_t r = i % rows, c = i / rows;
    dst[size_t(c) * ld + r] -= src[size_t(c) * lds + r];
}
// bw (w x w at leading dimension ld) += -u - u^T + v  (u, v: w x w, leading dimension w)
__global__ void k_wblock_fix(double *__restrict__ bw, uint32_t ld, const double *__restrict__ u, const double *__restrict__ v, uint32_t w) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * w) return;
    const uint32_t r = i % w, c = i / w;
    bw[size_t(c) * ld + r] += v[size_t(c) * w + r] - u[size_t(c) * w + r] - u[size_t(r) * w + c];
}
// seed basis (column-major float, reference DOF order) -> leading columns of a row-major internal-order panel
__global__ void k_load_seed(const float *__restrict__ seed, const uint32_t *__restrict__ perm, uint32_t nnodes, uint32_t ncols, uint32_t b, double *__restrict__ x) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nnodes) * 3 * ncols) return;
    const uint32_t c = uint32_t(i % ncols), comp = uint32_t((i / ncols) % 3), node = uint32_t(i / (size_t(3) * ncols));
    x[(size_t(3) * node + comp) * b + c] = double(seed[size_t(c) * (size_t(3) * nnodes) + size_t(3) * perm[node] + comp]);
}
// Exact rigid-body modes of every connected body (three translations, three rotations about the body's centroid) into panel
// columns col0 + 6 c .. col0 + 6 c + 5 for body c: a node carries its own body's modes and zeros in the others' columns
__global__ void k_inject_rbm(const double *__restrict__ xyz, const uint32_t *__restrict__ component, const double *__restrict__ centroid, uint32_t ncomp, uint32_t nnodes,
                             double *__restrict__ x, uint32_t b, uint32_t col0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnodes) return;
    const uint32_t mine = component[i];
    const double rx = xyz[3 * size_t(i)] - centroid[3 * size_t(mine)], ry = xyz[3 * size_t(i) + 1] - centroid[3 * size_t(mine) + 1], rz = xyz[3 * size_t(i) + 2] - centroid[3 * size_t(mine) + 2];
    for (uint32_t c = 0; c < ncomp; ++c) {
        double *r0 = x + (size_t(3) * i) * b + col0 + 6 * c, *r1 = r0 + b, *r2 = r1 + b;
        const double on = c == mine ? 1.0 : 0.0;
        r0[0] = on; r1[0] = 0; r2[0] = 0;
        r0[1] = 0; r1[1] = on; r2[1] = 0;
        r0[2] = 0; r1[2] = 0; r2[2] = on;
        r0[3] = 0; r1[3] = -rz * on; r2[3] = ry * on; // e_x x r
        r0[4] = rz * on; r1[4] = 0; r2[4] = -rx * on; // e_y x r
        r0[5] = -ry * on; r1[5] = rx * on; r2[5] = 0; // e_z x r
    }
}
// Bounding box of the nodes as six order-preserving integer keys (atomicMin / atomicMax apply): box[0..2] minima, box[3..5] maxima.
__device__ __forceinline__ unsigned long long mh_ordered_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double mh_from_ordered_key(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
__global__ void k_bbox(const double *__restrict__ xyz, uint32_t nnodes, unsigned long long *__restrict__ box) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int a = 0; a < 3; ++a) {
        double lo = i < nnodes ? xyz[3 * size_t(i) + a] : 1e300, hi = i < nnodes ? xyz[3 * size_t(i) + a] : -1e300;
        for (int off = 32; off > 0; off >>= 1) lo = fmin(lo, __shfl_xor(lo, off, 64)), hi = fmax(hi, __shfl_xor(hi, off, 64));
        if ((threadIdx.x & 63) == 0) {
            atomicMin(box + a, mh_ordered_key(lo));
            atomicMax(box + 3 + a, mh_ordered_key(hi));
        }
    }
}
// Smooth start vectors of a cold solve: the lowest modes of an elastic body are smooth displacement fields, so the start block holds
// (besides the exact rigid-body modes) the six uniform-strain fields and the fields  e_k L_a(x) L_b(y) L_c(z)  in Legendre polynomials
// over the bounding box for a host-made list of exponent triples (packed a | b << 8 | c << 16; BlockLobpcg::start orders them by total
// degree and caps the degree along a thin axis).  Columns: strains, then per triple the three components.  One thread per node.
__global__ void k_inject_poly(const double *__restrict__ xyz, const double *__restrict__ box, uint32_t nnodes, const uint32_t *__restrict__ triples, uint32_t ntriples,
                              double *__restrict__ x, uint32_t b, uint32_t col0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnodes) return;
    double t[3], L[3][7];
    for (int a = 0; a < 3; ++a) {
        const double lo = box[a], hi = box[3 + a];
        const double half = fmax(0.5 * (hi - lo), 1e-300);
        t[a] = (xyz[3 * size_t(i) + a] - 0.5 * (hi + lo)) / half;
        L[a][0] = 1.0, L[a][1] = t[a];
        for (int d = 2; d < 7; ++d) L[a][d] = ((2 * d - 1) * t[a] * L[a][d - 1] - (d - 1) * L[a][d - 2]) / d;
    }
    double *row[3] = {x + (size_t(3) * i) * b + col0, x + (size_t(3) * i + 1) * b + col0, x + (size_t(3) * i + 2) * b + col0};
    uint32_t col = 0;
    auto put = [&](double v0, double v1, double v2) {
        row[0][col] = v0, row[1][col] = v1, row[2][col] = v2;
        ++col;
    };
    // uniform strains (the three rotations among the nine linear fields are rigid-body modes already)
    put(t[0], 0, 0), put(0, t[1], 0), put(0, 0, t[2]), put(t[1], t[0], 0), put(t[2], 0, t[0]), put(0, t[2], t[1]);
    for (uint32_t q = 0; q < ntriples; ++q) {
        const uint32_t tr = triples[q];
        const double v = L[0][tr & 0xff] * L[1][(tr >> 8) & 0xff] * L[2][(tr >> 16) & 0xff];
        put(v, 0, 0), put(0, v, 0), put(0, 0, v);
    }
}
__global__ void k_copy_cols(const double *__restrict__ src, uint32_t wsrc, double *__restrict__ dst, uint32_t wdst, size_t rows, uint32_t ncols) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= rows * ncols) return;
    dst[(i / ncols) * wdst + i % ncols] = src[(i / ncols) * wsrc + i % ncols];
}
// dense path: BSR -> dense column-major (both A-values and M scalars)
__global__ void k_bsr_to_dense(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const double *__restrict__ v9, const double *__restrict__ ms,
                               uint32_t nnodes, double *__restrict__ a, double *__restrict__ m) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nnodes) return;
    const size_t n = size_t(3) * nnodes;
    for (uint32_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
        const uint32_t c = col[p];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) a[(size_t(3) * c + j) * n + size_t(3) * r + i] = v9[9 * size_t(p) + 3 * i + j];
            m[(size_t(3) * c + i) * n + size_t(3) * r + i] = ms[p];
        }
    }
}
__global__ void k_colmajor_to_panel(const double *__restrict__ z, size_t n, uint32_t ncols, double *__restrict__ x) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n * ncols) return;
    x[i] = z[size_t(i % ncols) * n + i / ncols];
}

// upper triangle <- lower triangle (column-major n x n, ld)
__global__ void k_symmetrize_lower(double *__restrict__ a, uint32_t n, uint32_t ld) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(n) * n) return;
    const uint32_t r = uint32_t(i % n), c = uint32_t(i / n);
    if (r < c) a[size_t(c) * ld + r] = a[size_t(r) * ld + c];
}

unsigned grid1(size_t count) { return div_up(count, TB); }

// ---- rocBLAS wrappers over row-major panels ------------------------------------------------------------------
// G (wa x wb, column-major, ld) = Xa^T Yb for panels Xa (n x wa), Yb (n x wb): hand-written MFMA kernel (mh_dense.hip)
void gram(mh_context *ctx, size_t n, const double *xa, uint32_t wa, const double *yb, uint32_t wb, double *g, uint32_t ld) {
    mh_gram(ctx, n, xa, wa, yb, wb, g, ld);
}
// Z (n x wz) = alpha * X (n x wa) * C (wa x wz, column-major ld) + beta * Z
void panel_mul(mh_context *ctx, size_t n, const double *x, uint32_t wa, const double *c, uint32_t ld, double *z, uint32_t wz, double alpha, double beta) {
    if (!wz) return;
    if (!wa) return;
    ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_transpose, rocblas_operation_none, wz, rocblas_int(n), wa, &alpha, c, ld, x, wa, &beta, z, wz));
}
// W (n x w) <- W L^-T  (L lower, column-major)
void panel_trsm(mh_context *ctx, size_t n, double *wp, uint32_t w, const double *l, uint32_t ld) {
    const double one = 1;
    ROCBLAS_CHECK(rocblas_dtrsm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, w, rocblas_int(n), &one, l, ld, wp, w));
}


// The step's order-m products: our 32 x 32-tile MFMA kernel (mh_small_gemm: the library's 128 x 128 macro tile leaves 4 .. 36 workgroups
// on the device at these orders); anything larger goes to the library.
void small_dgemm(mh_context *ctx, rocblas_operation ta, rocblas_operation tb, rocblas_int M, rocblas_int N, rocblas_int K, const double *alpha, const double *a, rocblas_int lda, const double *b,
                 rocblas_int ldb, const double *beta, double *c, rocblas_int ldc) {
    if (M <= 1024 && N <= 1024 && K <= 4096)
        mh_small_gemm(ctx, ta != rocblas_operation_none, tb != rocblas_operation_none, uint32_t(M), uint32_t(N), uint32_t(K), *alpha, a, uint32_t(lda), b, uint32_t(ldb), *beta, c, uint32_t(ldc));
    else
        ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, ta, tb, M, N, K, alpha, a, lda, b, ldb, beta, c, ldc));
}

// Small generalised symmetric eigenproblem gA c = theta gM c (lower triangles given, order m, ld m):
// Cholesky reduction + rocSOLVER syevd (rocSOLVER's sygvd reduces with an unblocked sygs2 that launches O(m) tiny
// kernels; this form measured 2.3x faster at m = 225).  On return gA holds the gM-orthonormal eigenvectors.
// max |gM - I| over the lower triangle, as an ordered integer so that atomicMax applies (non-negative doubles)
__global__ void k_identity_defect(const double *__restrict__ g, uint32_t m, unsigned long long *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double d = 0;
    if (i < m * m) {
        const uint32_t r = i % m, c = i / m;
        if (r >= c) d = fabs(g[i] - (r == c ? 1.0 : 0.0));
    }
    for (int off = 32; off > 0; off >>= 1) d = fmax(d, __shfl_xor(d, off, 64));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(out, (unsigned long long)__double_as_longlong(d));
}

// E = gM - I as a full symmetric matrix from gM's lower triangle
__global__ void k_defect_matrix(const double *__restrict__ g, uint32_t m, double *__restrict__ e) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * m) return;
    const uint32_t r = i % m, c = i / m;
    e[i] = (r >= c ? g[i] : g[size_t(r) * m + c]) - (r == c ? 1.0 : 0.0);
}
// S = I - E/2 + 3/8 E^2: the inverse square root of I + E to second order
__global__ void k_inverse_sqrt_series(const double *__restrict__ e, const double *__restrict__ e2, uint32_t m, double *__restrict__ s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * m) return;
    s[i] = (i % m == i / m ? 1.0 : 0.0) - 0.5 * e[i] + 0.375 * e2[i];
}

// Self-check of a whole Rayleigh-Ritz step (tridiagonalisation, reflectors, partial spectrum, back-transformation at once): for three sampled
// pairs (first, middle, last) the residual max |A z - theta z| / (max_i sum_c |a_ic z_c| + |theta| max |z|) against the SAVED symmetric matrix, folded into
// one device word by atomicMax (non-negative doubles order as integers; NaN maps to the largest value).  Never read back inside the
// iteration: the solve reads the word once at its end (BlockLobpcg::finish) and fails loudly when it is not at rounding level.
__global__ void __launch_bounds__(256) k_rr_selfcheck(const double *__restrict__ a, uint32_t m, const double *__restrict__ z, uint32_t ldz, const double *__restrict__ theta, uint32_t ncols,
                                                      unsigned long long *__restrict__ worst) {
    __shared__ double zs[768], red[3][256];
    const uint32_t tid = threadIdx.x, j = blockIdx.x == 0 ? 0 : (blockIdx.x == 1 ? ncols / 2 : ncols - 1);
    for (uint32_t i = tid; i < m; i += 256) zs[i] = z[size_t(j) * ldz + i];
    __syncthreads();
    const double th = theta[j];
    double rmax = 0, azmax = 0, zmax = 0;
    bool nan = false;
    for (uint32_t i = tid; i < m; i += 256) {
        double az = 0, abs_az = 0; // abs_az = sum_c |a_ic| |z_c|: the scale rounding errors of the step's arithmetic live on (a pair far below
        for (uint32_t c = 0; c < m; ++c) { // ||A|| -- the rigid-body pairs beside a random start block -- carries an absolute error of eps ||A||)
            const double t = a[size_t(c) * m + i] * zs[c];
            az += t, abs_az += fabs(t);
        }
        const double r = az - th * zs[i];
        nan = nan || !(r == r);
        rmax = fmax(rmax, fabs(r)), azmax = fmax(azmax, abs_az), zmax = fmax(zmax, fabs(zs[i]));
    }
    red[0][tid] = nan ? INFINITY : rmax, red[1][tid] = azmax, red[2][tid] = zmax;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < uint32_t(h))
            for (int q = 0; q < 3; ++q) red[q][tid] = fmax(red[q][tid], red[q][tid + h]);
        __syncthreads();
    }
    if (tid == 0) {
        const double rel = red[0][0] / fmax(red[1][0] + fabs(th) * red[2][0], 1e-300);
        atomicMax(worst, (unsigned long long)__double_as_longlong(rel == rel ? rel : INFINITY));
    }
}
void rr_selfcheck(mh_context *ctx, const double *saved, uint32_t m, const double *z, uint32_t ldz, const double *theta, uint32_t ncols) {
    if (!ncols) return;
    if (!ctx->rr_check) {
        ctx->rr_check = static_cast<unsigned long long *>(ctx->pool.alloc(256));
        HIP_CHECK(hipMemsetAsync(ctx->rr_check, 0, 256, ctx->stream));
    }
    k_rr_selfcheck<<<3, 256, 0, ctx->stream>>>(saved, m, z, ldz, theta, ncols, ctx->rr_check);
    KERNEL_CHECK();
}

int rr_solve(mh_context *ctx, double *gA, double *gM, uint32_t m, double *evals, double *ework, DevArray<int> &info, uint32_t nwant = 0, bool gm_is_identity = false,
             std::vector<double> *host_evals = nullptr) { // host_evals: receives the nwant lowest eigenvalues when the partial-spectrum path delivered them (they travel with its quality read-back: no second synchronisation for them); left empty otherwise
    // nwant: only the nwant lowest pairs are needed (the active Ritz vectors): lets the tridiagonal stage compute a partial spectrum
    const double one = 1, zero = 0;
    int hinfo = 0;
    k_symmetrize_lower<<<grid1(size_t(m) * m), TB, 0, ctx->stream>>>(gA, m, m);
    KERNEL_CHECK();
    // The basis is built M-orthonormal (X and P by construction, W by projection + Cholesky-QR), so gM is the identity
    // up to the orthogonalisation error.  When that error is below 1e-11 the pencil is solved as a standard problem:
    // no Cholesky reduction (potrf + three trsm, ~1.2 ms of a ~4 ms solve at order 225).  Otherwise the full reduction.
    bool identity = gm_is_identity, series = false; // (gm_is_identity: every block of gM was SET by the caller, none measured: no defect to look for, no read-back)
    if (!identity) {
        static_assert(sizeof(unsigned long long) == sizeof(double), "defect word");
        unsigned long long *defect = reinterpret_cast<unsigned long long *>(ework);
        HIP_CHECK(hipMemsetAsync(defect, 0, sizeof(unsigned long long), ctx->stream));
        k_identity_defect<<<grid1(size_t(m) * m), TB, 0, ctx->stream>>>(gM, m, defect);
        KERNEL_CHECK();
        unsigned long long bits = 0;
        HIP_CHECK(hipMemcpyAsync(&bits, defect, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        double d;
        memcpy(&d, &bits, sizeof(d));
        identity = d < 1e-11;
        const bool verbose = switches().verbose;
        if (verbose) fprintf(stderr, "[rr] m %u identity defect %.2e\n", m, d);
        // A small defect (one Cholesky-QR pass of an ill-conditioned W leaves 1e-10 .. 1e-8) is absorbed by the series
        // S = (I + E)^(-1/2) = I - E/2 + 3/8 E^2 + O(E^3): S gA S z = theta z, c = S z.  Four order-m products instead of the
        // Cholesky reduction's factorisation and three triangular solves (~1.2 ms of single-workgroup kernels).
        series = !identity && d < 1e-7;
    }
    DevArray<double> sroot, stmp;
    if (series) {
        sroot.reset(ctx, size_t(m) * m);
        stmp.reset(ctx, size_t(m) * m * 2);
        double *e = stmp.get(), *e2 = stmp.get() + size_t(m) * m;
        k_defect_matrix<<<grid1(size_t(m) * m), TB, 0, ctx->stream>>>(gM, m, e);
        KERNEL_CHECK();
        small_dgemm(ctx, rocblas_operation_none, rocblas_operation_none, m, m, m, &one, e, m, e, m, &zero, e2, m);
        k_inverse_sqrt_series<<<grid1(size_t(m) * m), TB, 0, ctx->stream>>>(e, e2, m, sroot);
        KERNEL_CHECK();
        small_dgemm(ctx, rocblas_operation_none, rocblas_operation_none, m, m, m, &one, sroot, m, gA, m, &zero, e, m);
        small_dgemm(ctx, rocblas_operation_none, rocblas_operation_none, m, m, m, &one, e, m, sroot, m, &zero, gA, m);
        k_symmetrize_lower<<<grid1(size_t(m) * m), TB, 0, ctx->stream>>>(gA, m, m);
        KERNEL_CHECK();
        identity = true;
    }
    if (!identity) {
        mh_potrf(ctx, gM, m, m, info);
        info.download(&hinfo, 1);
        if (hinfo != 0) return hinfo;
        ROCBLAS_CHECK(rocblas_dtrsm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, m, m, &one, gM, m, gA, m));
        ROCBLAS_CHECK(rocblas_dtrsm(ctx->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, m, m, &one, gM, m, gA, m));
    }
    if (m >= 8 && m <= 256) {
        // syevd by parts: the tridiagonalisation (70 % of rocSOLVER's syevd at this order) in one workgroup of ours, then
        // rocSOLVER's divide and conquer on T and the back-transformation Z <- Q Z
        DevArray<double> z(ctx, size_t(m) * m), tau(ctx, m);
        const bool partial = nwant && nwant < m;
        DevArray<double> zl(ctx, partial ? size_t(m) * nwant : 0); // the first attempt's vectors: z still holds the saved matrix then
        double *zres = z.get();
        uint32_t ncols = m;
        // The multi-workgroup reduction can give up (a workgroup stalled by co-resident work past the poll bound): its output is
        // then garbage.  z is free until the tridiagonal stage writes it, so it keeps a copy of the symmetric matrix, and a
        // give-up redoes the step with the one-workgroup kernel; only a failure of that one fails the solve.
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (attempt == 0) HIP_CHECK(hipMemcpyAsync(z.get(), gA, size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            else HIP_CHECK(hipMemcpyAsync(gA, z.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            mh_sytrd_small(ctx, gA, m, evals, ework, tau, attempt == 0 ? -1 : 0); // gA is fully symmetric here (k_symmetrize_lower above / the reduction)
            // the give-up flag travels with the next read-back of this step (a workgroup stalled past the poll bound by co-resident work)
            int sytrd_gave_up = 0;
            const bool flagged = attempt == 0 && ctx->sytrd_flag;
            if (flagged) HIP_CHECK(hipMemcpyAsync(&sytrd_gave_up, ctx->sytrd_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            // only the nwant lowest pairs are needed: our multisection + inverse iteration (mh_tridiag_lowest) instead of
            // the full divide and conquer, accepted when its residual check passes
            bool done = false;
            if (partial) {
                DevArray<double> wv(ctx, 2 * m + 8), ufac(ctx, size_t(3) * m * nwant);
                double *zout = attempt == 0 ? zl.get() : z.get();
                if (mh_tridiag_lowest(ctx, evals, ework, m, nwant, wv, zout, m, ufac, wv.get() + m, wv.get() + m + 8)) {
                    double qv[5] = {1, 0, 0, 0, 0};
                    std::vector<double> lam_host(host_evals ? nwant : 0);
                    HIP_CHECK(hipMemcpyAsync(qv, wv.get() + m, sizeof(qv), hipMemcpyDeviceToHost, ctx->stream));
                    if (host_evals) HIP_CHECK(hipMemcpyAsync(lam_host.data(), wv.get(), size_t(nwant) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    if (flagged && switches().test_sytrd_giveup) sytrd_gave_up = 1;
                    if (sytrd_gave_up) {
                        ++ctx->sytrd_redos;
                        continue;
                    }
                    const double quality = qv[0];
                    const bool verbose = switches().verbose;
                    if (verbose)
                        fprintf(stderr, "[rr] tridiagonal m %u lowest %u: residual / ||T|| %.2e; us: multisection %.0f, inverse iteration %.0f, Gram-Schmidt %.0f, output %.0f\n", m,
                                nwant, quality, qv[1] * 0.01, qv[2] * 0.01, qv[3] * 0.01, qv[4] * 0.01);
                    if (quality < 1e-10) {
                        if (host_evals) host_evals->swap(lam_host);
                        HIP_CHECK(hipMemcpyAsync(evals, wv.get(), size_t(nwant) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                        zres = zout;
                        ncols = nwant;
                        done = true;
                    }
                }
            }
            if (!done) {
                if (flagged) HIP_CHECK(hipStreamSynchronize(ctx->stream)); // the flag is known before garbage could reach the library
                if (flagged && switches().test_sytrd_giveup) sytrd_gave_up = 1;
                if (sytrd_gave_up) {
                    ++ctx->sytrd_redos;
                    continue;
                }
                ROCBLAS_CHECK(rocsolver_dstedc(ctx->blas, rocblas_evect_tridiagonal, m, evals, ework, z, m, info));
                info.download(&hinfo, 1);
                if (hinfo != 0) return hinfo;
            }
            break;
        }
        mh_apply_q(ctx, gA, tau, m, zres, m, ncols);
        if (zres == zl.get()) rr_selfcheck(ctx, z.get(), m, zres, m, evals, ncols); // (z still holds the saved matrix on this path)
        HIP_CHECK(hipMemcpyAsync(gA, zres, size_t(m) * ncols * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        // no synchronisation: the workspaces return to the context's pool, whose blocks are only ever used on this same stream
    } else {
        // Orders 257 .. 768 (the 200-mode configuration iterates at 3 x 240): the tridiagonalisation is 55 % of rocSOLVER's syevd
        // there (8.3 of 15 ms at order 720); k_sytrd_wide does it in 4.4 ms across 48 workgroups, then the library's divide and
        // conquer on T and its back-transformation.  A give-up (see above) falls back to the library's syevd on the saved matrix.
        bool done = false;
        if (m > 256 && m <= 768 && !ctx->exchange_disabled) {
            DevArray<double> z(ctx, size_t(m) * m), saved(ctx, size_t(m) * m), tau(ctx, m);
            HIP_CHECK(hipMemcpyAsync(saved.get(), gA, size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            mh_sytrd_wide(ctx, gA, m, evals, ework, tau);
            int gave_up = 0;
            HIP_CHECK(hipMemcpyAsync(&gave_up, ctx->sytrd_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (switches().test_sytrd_giveup) gave_up = 1;
            if (gave_up) ++ctx->sytrd_redos;
            if (gave_up) {
                HIP_CHECK(hipMemcpyAsync(gA, saved.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            } else if (nwant && nwant < m && nwant <= 256 && !switches().no_tridiag_wide && [&] {
                           // only the nwant lowest pairs are needed: multisection, inverse iteration and Cholesky-QR (mh_tridiag_lowest_wide)
                           // instead of the full divide and conquer (stedc + ormtr: ~5 ms of library launches at order 720), accepted when
                           // its residual check passes; Z <- Q Z by our one-launch kernel
                           DevArray<double> work(ctx, size_t(5) * nwant * m + size_t(2) * nwant * nwant + 8), lam(ctx, nwant);
                           double quality = 1.0;
                           if (!mh_tridiag_lowest_wide(ctx, evals, ework, m, nwant, lam, z, m, work, info, &quality)) return false;
                           if (switches().verbose) fprintf(stderr, "[rr] tridiagonal m %u lowest %u (wide): residual / ||T|| %.2e\n", m, nwant, quality);
                           if (!(quality < 1e-10)) return false;
                           HIP_CHECK(hipMemcpyAsync(evals, lam.get(), size_t(nwant) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                           mh_apply_q(ctx, gA, tau, m, z, m, nwant);
                           rr_selfcheck(ctx, saved.get(), m, z, m, evals, nwant);
                           HIP_CHECK(hipMemcpyAsync(gA, z.get(), size_t(m) * nwant * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                           HIP_CHECK(hipStreamSynchronize(ctx->stream)); // (work and lam go back to the pool)
                           return true;
                       }()) {
                done = true;
            } else {
                ROCBLAS_CHECK(rocsolver_dstedc(ctx->blas, rocblas_evect_tridiagonal, m, evals, ework, z, m, info));
                info.download(&hinfo, 1);
                if (hinfo != 0) return hinfo;
                ROCBLAS_CHECK(rocsolver_dormtr(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, m, m, gA, m, tau, z, m));
                HIP_CHECK(hipMemcpyAsync(gA, z.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                HIP_CHECK(hipStreamSynchronize(ctx->stream)); // the library's internal workspace use is not ours to reason about: z goes back to the pool after it is done
                done = true;
            }
        }
        if (!done) {
            ROCBLAS_CHECK(rocsolver_dsyevd(ctx->blas, rocblas_evect_original, rocblas_fill_lower, m, gA, m, evals, ework, info));
            info.download(&hinfo, 1);
            if (hinfo != 0) return hinfo;
        }
    }
    if (series) { // c = S z
        small_dgemm(ctx, rocblas_operation_none, rocblas_operation_none, m, m, m, &one, sroot, m, gA, m, &zero, stmp, m);
        HIP_CHECK(hipMemcpyAsync(gA, stmp.get(), size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream)); // the workspaces go back to the pool on return
    } else if (!identity)
        ROCBLAS_CHECK(rocblas_dtrsm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, m, m, &one, gM, m, gA, m));
    return 0;
}

void colsumsq(mh_context *ctx, const double *x, size_t rows, uint32_t w, double *out, DevArray<double> &scratch) {
    const uint32_t rpb = 256;
    const uint32_t nb = div_up(rows, rpb);
    if (scratch.count < size_t(nb) * w) scratch.reset(ctx, size_t(nb) * w);
    dim3 grid(nb, div_up(w, 64));
    k_colsumsq_partial<<<grid, 64, 0, ctx->stream>>>(x, rows, w, scratch, rpb);
    KERNEL_CHECK();
    k_colsumsq_final<<<w, 256, 0, ctx->stream>>>(scratch, nb, w, out);
    KERNEL_CHECK();
}

// ---- multilevel preconditioner ---------------------------------------------------------------------------------
// T = precision of the two smoothed levels (float by default: the smoothers are gather-bound, so halving the bytes
// nearly halves their time, and the outer iteration -- residuals, Rayleigh-Ritz, convergence test -- stays double).
// The aggregate level is always solved in double: it carries the near-null (rigid-body) components.
template<typename T> struct Precond {
    mh_system *sys;
    mh_context *ctx;
    uint32_t wmax;
    int deg2{2}, deg1{5}, gamma{3}; // (the constructor picks the cycle shape by the kind of mesh)
    double ratio{8.0}, ratio1{0.0}; // ratio1 > 0: the P1 level's own interval [lmax / ratio1, lmax]
    DevArray<T> rin, z2, d2, t2, r2, r1, x1, d1, t1, rr1;
    DevArray<double> r0, x0, x0_partial;
    static constexpr uint32_t COARSE_SLICES = 8;
    DevArray<double> t2d, r1d, t1d; // single-precision smoothers: the residuals handed down a level stay double
    DevArray<T> patch_y; // y_e = (A_ee)^-1 R_e v of the sliver patches (mh_patch.hip), both levels share it
    static constexpr bool kDouble = std::is_same<T, double>::value;
    static uint32_t pitch(uint32_t w) { return kDouble ? w : (w + 3u) & ~3u; } // 16-byte panel rows for the wide-load SpMM
    Precond(mh_system *s, uint32_t w_in) : sys(s), ctx(s->ctx), wmax(w_in) {
        const uint32_t w = pitch(w_in);
        const size_t n2 = size_t(3) * s->n_nodes, n1 = size_t(3) * s->n_points, n0 = size_t(6) * s->n_agg;
        d2.reset(ctx, n2 * w); t2.reset(ctx, n2 * w); r2.reset(ctx, n2 * w); z2.reset(ctx, n2 * w);
        if (!kDouble) {
            rin.reset(ctx, n2 * w);
            t2d.reset(ctx, n2 * w);
            r1d.reset(ctx, n1 * w); t1d.reset(ctx, n1 * w);
        }
        r1.reset(ctx, n1 * w); x1.reset(ctx, n1 * w); d1.reset(ctx, n1 * w); t1.reset(ctx, n1 * w); rr1.reset(ctx, n1 * w);
        r0.reset(ctx, n0 * w);
        x0.reset(ctx, n0 * w);
        x0_partial.reset(ctx, n0 * w * COARSE_SLICES);
        patch_y.reset(ctx, std::max(s->patches2.scratch_rows(), s->patches1.scratch_rows()) * w);
        const Switches &sw = switches();
        // Three cycle shapes, each the measured best of its class (round 5; MH_CYCLE = deg2, deg1, gamma, ratio[, ratio1] overrides; profiles/r05_cycle_by_body.txt):
        //  * a mesh with SLIVER PATCHES: P2 Chebyshev degree 5 over [lmax / 60, lmax]; P1: see below (three cycles of degree 5 over the same ratio until the end of round 5).  The P1 space
        //    represents the smooth error of such a mesh poorly (measured two-grid bound, exact coarse solve, 30k-tet skillet scan: condition 34 with two
        //    steps over [lmax/8, lmax], 12 with four over [lmax/30, lmax]); with the patches scaled by their overlap (mh_patch.hip) five steps over
        //    [lmax/60, lmax] are the measured best on the four scan workloads: 24 / 26 / 40 / 40 iterations (4 over lmax/30: 26 / 33 / 44 / 48).
        //  * a SURFACE-DOMINATED body without patches -- fewer than 4.5 tetrahedra per mesh point (a bulk fill has 5-6.7; a plate two cells thick 3.9,
        //    a UV sphere's fill 4.2, the reference's test bars 2.4-3.7): the same long P2 smoother (the 215-pair Kuhn plate 860 -> 674 ms, 22 -> 17
        //    iterations; the 48 x 24 UV sphere 64 -> 53 ms, 28 -> 21; the thin bar 24 -> 19 ms), and ONE P1 cycle of degree 16 over [lmax / 250, lmax]
        //    instead of three of degree 5 (plate 674 -> 643 ms, sphere 54 -> 50.5).
        //  * a BULK body (Kuhn cubes, jittered boxes): the short P2 smoother (degree 2 over [lmax / 8, lmax]: the long one costs 8-18 % there) and one
        //    P1 cycle of degree 12 over [lmax / 100, lmax] instead of three of degree 5 over [lmax / 8, lmax]: the P1 level's launches are latency-bound
        //    (~25 us each whatever the size) and one long sequence over a wide interval smooths better than three short ones around the coarse solve --
        //    26^3 / 17^3 / 12^3 cubes 133.0 -> 122.3 / 54.7 -> 49.8 / 32.8 -> 31.4 ms, the batch of 64 boxes 1.65 -> 1.54 s (with [lmax / 8, lmax] the
        //    same single cycle LOSES 10 %: the interval is what makes it work).
        const bool surface_dominated = s->kept_tets < 4.5 * s->n_points;
        if (s->patches2.any()) {
            deg2 = 5;
            ratio = 60.0;
            // one P1 cycle of degree 28 over [lmax / 800, lmax]: the single cycle of degree 16 over lmax / 250 that suits the patch-free bodies lost 6-8 % on
            // the 65-pair solves of the 100k-tet scans (and won 4-5 % on the 215-pair ones); longer and wider it is level with the three short cycles or
            // ahead on all nine patch workloads -- scan_s100k_repaired 336 -> 317 ms, config3_s100k_repaired 1 093 -> 1 007, config3_s30k 464 -> 438, the
            // unrepaired scan_s100k 570 -> 579, ball and 30k-tet scans within 1 %
            // -- except the fills DENSE in patches at blocks up to 128 columns (unrepaired scans with and without interior points, 9-28 patches per
            // thousand tetrahedra: 65-pair solves +2 ... +8 % with it; the repaired fills have 0.3-0.5 per thousand), which keep three cycles of degree 5
            const bool few_patches = uint64_t(s->patches2.n_bad_elements) * 500 < s->kept_tets;
            if (w_in > 128 || few_patches) deg1 = 28, gamma = 1, ratio1 = 800.0;
        } else if (surface_dominated) {
            deg2 = 5;
            ratio = 60.0;
            deg1 = 16, gamma = 1, ratio1 = 250.0;
        } else {
            deg1 = 12, gamma = 1, ratio1 = 100.0;
        }
        if (sw.deg2 > 0) deg2 = sw.deg2;
        if (sw.deg1 > 0) deg1 = sw.deg1;
        if (sw.gamma > 0) gamma = sw.gamma;
        if (sw.cheb_ratio > 0) ratio = std::max(1.5, sw.cheb_ratio), ratio1 = 0.0; // (a given ratio holds for both levels unless a fifth value follows)
        if (sw.cheb_ratio1 > 0) ratio1 = std::max(1.5, sw.cheb_ratio1);
    }
    void spmm(const BsrLevel &lvl, const T *x, T *y, uint32_t w) {
        if constexpr (kDouble) mh_spmm(ctx, lvl, lvl.aval, x, y, nullptr, nullptr, w);
        else mh_spmm_f32(ctx, lvl, x, y, w);
    }
    static const T *dinv_of(const BsrLevel &lvl) {
        if constexpr (kDouble) return lvl.dinv.get();
        else return lvl.dinv32.get();
    }
    const PatchSet &patches_of(const BsrLevel &lvl) const { return lvl.id == 2 ? sys->patches2 : sys->patches1; }
    // deg Chebyshev-Jacobi steps on lvl; when z_out is given the last step writes the iterate there (in double)
    void cheb(const BsrLevel &lvl, int deg, const T *b, T *x, bool zero_init, T *r, T *d, T *t, uint32_t w, double *z_out = nullptr, uint32_t w_out = 0,
              const double *b_src = nullptr, uint32_t w_src = 0, T *b_copy = nullptr) { // b_src: (single-precision cycle) convert the right-hand side on the way
        const size_t rows = size_t(3) * lvl.n_nodes;
        const double lmax = lvl.lmax, lmin = lvl.lmax / (lvl.id == 1 && ratio1 > 0 ? ratio1 : ratio);
        const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sig = theta / delta;
        double rho = 1.0 / sig;
        const T *dinv = dinv_of(lvl);
        if (!zero_init) spmm(lvl, x, t, w);
        if constexpr (!kDouble) {
            if (b_src && zero_init) k_cheb_init_convert<<<grid1(rows * w), TB, 0, ctx->stream>>>(b_src, w_src, dinv, float(1.0 / theta), b_copy, r, d, x, rows, w);
            else k_cheb_init_v4<<<grid1(rows * (w / 4)), TB, 0, ctx->stream>>>(reinterpret_cast<const f4_t *>(b), zero_init ? nullptr : reinterpret_cast<const f4_t *>(t), dinv,
                                                                               float(1.0 / theta), reinterpret_cast<f4_t *>(r), reinterpret_cast<f4_t *>(d),
                                                                               reinterpret_cast<f4_t *>(x), zero_init ? 0 : 1, rows, w / 4);
        } else {
            k_cheb_init<T><<<grid1(rows * w), TB, 0, ctx->stream>>>(b, zero_init ? nullptr : t, dinv, T(1.0 / theta), r, d, x, zero_init ? 0 : 1, rows, w);
        }
        KERNEL_CHECK();
        // the smoother's scaling is M^-1 = D^-1 + the sliver patches: the kernels above and below apply the D^-1 part, every step is
        // followed by the patch part of the same residual (r holds it; the last step's is r - t, formed on the fly)
        const PatchSet &ps = patches_of(lvl);
        mh_apply_patches<T>(ctx, ps, r, nullptr, w, T(1.0 / theta), d, x, nullptr, 0, patch_y.get());
        constexpr bool fuse_steps = true; // the smoothing step as the epilogue of the product that forms A d (neutral in time, one panel pass less)
        T *cur = d, *alt = t; // the direction lives in `cur`; `alt` takes the product, or (fused step) the next direction
        for (int k = 1; k < deg; ++k) {
            const double rho_new = 1.0 / (2 * sig - rho);
            const bool last = z_out && k + 1 == deg;
            bool fused = false;
            if constexpr (!kDouble) {
                if (fuse_steps && !last) fused = mh_spmm_f32_cheb_step(ctx, lvl, cur, alt, r, x, dinv, float(rho_new * rho), float(2 * rho_new / delta), w);
            }
            if (fused) {
                std::swap(cur, alt);
                mh_apply_patches<T>(ctx, ps, r, nullptr, w, T(2 * rho_new / delta), cur, x, nullptr, 0, patch_y.get());
            } else {
                spmm(lvl, cur, alt, w);
                if (!last) {
                    k_cheb_step<T><<<grid1(rows * w), TB, 0, ctx->stream>>>(alt, dinv, T(rho_new * rho), T(2 * rho_new / delta), r, cur, x, rows, w);
                } else if constexpr (kDouble) {
                    k_cheb_last<T><<<grid1(rows * w), TB, 0, ctx->stream>>>(alt, dinv, T(rho_new * rho), T(2 * rho_new / delta), r, cur, x, z_out, rows, w, w_out);
                } else {
                    k_cheb_last_v4<<<grid1(rows * (w / 4)), TB, 0, ctx->stream>>>(reinterpret_cast<const f4_t *>(alt), dinv, float(rho_new * rho), float(2 * rho_new / delta),
                                                                             reinterpret_cast<const f4_t *>(r), reinterpret_cast<const f4_t *>(cur),
                                                                             reinterpret_cast<const f4_t *>(x), z_out, rows, w / 4, w_out);
                }
                KERNEL_CHECK();
                if (!last) mh_apply_patches<T>(ctx, ps, r, nullptr, w, T(2 * rho_new / delta), cur, x, nullptr, 0, patch_y.get());
                else mh_apply_patches<T>(ctx, ps, r, alt, w, T(2 * rho_new / delta), nullptr, nullptr, z_out, w_out, patch_y.get());
            }
            rho = rho_new;
        }
        if (z_out && deg == 1) {
            k_convert_pitch<T, double><<<grid1(rows * w_out), TB, 0, ctx->stream>>>(x, w, z_out, w_out, rows);
            KERNEL_CHECK();
        }
    }
    // z = B r for an n2 x w panel
    void apply(const double *r_in, double *z_out, uint32_t w_in) {
        mh_finish_hierarchy(sys);
        const uint32_t w = pitch(w_in);
        const uint32_t nn = sys->n_nodes, np = sys->n_points, na = sys->n_agg;
        const size_t n2 = size_t(3) * nn, n1 = size_t(3) * np, n0 = size_t(6) * na;
        const T *r;
        if constexpr (kDouble) r = r_in;
        else r = rin.get(); // filled by the first smoothing step below
        T *z = z2.get();
        if constexpr (kDouble) cheb(sys->L2, deg2, r, z, true, r2, d2, t2, w);
        else cheb(sys->L2, deg2, r, z, true, r2, d2, t2, w, nullptr, 0, r_in, w_in, rin.get());
        // Residual for the next level.  With single-precision smoothers it is formed in double (double A, double r,
        // the float iterate): a float residual carries an error of 6e-8 ||A|| ||z|| that the coarse solves amplify by
        // the condition number, which stalls the lowest modes of thin, ill-conditioned bodies at ~1e-4.
        if constexpr (kDouble) {
            spmm(sys->L2, z, t2, w);
            k_restrict_p1<double><<<grid1(n1 * w), TB, 0, ctx->stream>>>(r, t2.get(), sys->p1_corner, sys->p1_edge_ptr, sys->p1_edge_mid, r1.get(), np, w);
            KERNEL_CHECK();
        } else {
            mh_spmm_mixed(ctx, sys->L2, z, t2d, w);
            k_restrict_p1<double><<<grid1(n1 * w), TB, 0, ctx->stream>>>(r_in, t2d.get(), sys->p1_corner, sys->p1_edge_ptr, sys->p1_edge_mid, r1d.get(), np, w, w_in); // r_in at its own pitch
            KERNEL_CHECK();
            k_convert<double, T><<<grid1(n1 * w), TB, 0, ctx->stream>>>(r1d.get(), r1.get(), n1 * w);
            KERNEL_CHECK();
        }
        for (int g = 0; g < gamma; ++g) {
            cheb(sys->L1, deg1, r1, x1, g == 0, rr1, d1, t1, w);
            if constexpr (kDouble) {
                spmm(sys->L1, x1, t1, w);
                k_restrict_agg<double><<<grid1(n0 * w), TB, 0, ctx->stream>>>(r1.get(), t1.get(), sys->agg_t, r0, sys->agg_ptr, sys->agg_nodes, na, w);
            } else {
                mh_spmm_mixed(ctx, sys->L1, x1, t1d, w);
                k_restrict_agg<double><<<grid1(n0 * w), TB, 0, ctx->stream>>>(r1d.get(), t1d.get(), sys->agg_t, r0, sys->agg_ptr, sys->agg_nodes, na, w);
            }
            KERNEL_CHECK();
            // r0 is (6 na) x w row-major = w x (6 na) column-major: x0 = r0 * A0^-1 (A0^-1 symmetric, explicit)
            mh_short_product(ctx, n0, sys->a0, uint32_t(n0), r0, w, x0, x0_partial, COARSE_SLICES);
            k_prolong_agg<T><<<grid1(n1 * w), TB, 0, ctx->stream>>>(x0, sys->agg_t, x1.get(), sys->agg_of, np, w);
            KERNEL_CHECK();
            cheb(sys->L1, deg1, r1, x1, false, rr1, d1, t1, w);
        }
        if constexpr (kDouble) k_prolong_p1<T><<<grid1(n2 * w), TB, 0, ctx->stream>>>(x1.get(), sys->parent_a, sys->parent_b, z, nn, w);
        else k_prolong_p1_v4<<<grid1(n2 * (w / 4)), TB, 0, ctx->stream>>>(reinterpret_cast<const f4_t *>(x1.get()), sys->parent_a, sys->parent_b, reinterpret_cast<f4_t *>(z), nn, w / 4);
        KERNEL_CHECK();
        cheb(sys->L2, deg2, r, z, false, r2, d2, t2, w, z_out, w_in); // (an unsymmetric cycle -- no post-smoothing, or one step of it -- was measured in round 6: 14 -> 19 / 16 iterations, 107 -> 130 / 116 ms on the 100k-tet cube)
    }
};

double estimate_lmax(mh_context *ctx, BsrLevel &lvl, const PatchSet &ps) {
    const uint32_t w = 8;
    const size_t rows = size_t(3) * lvl.n_nodes;
    DevArray<double> v(ctx, rows * w), t(ctx, rows * w), nrm(ctx, w), scratch, py(ctx, ps.scratch_rows() * w);
    k_random_panel<<<grid1(rows * w), TB, 0, ctx->stream>>>(v, rows * w, 0x5eedull);
    KERNEL_CHECK();
    colsumsq(ctx, v, rows, w, nrm, scratch);
    k_scale_cols_inv_sqrt<<<grid1(rows * w), TB, 0, ctx->stream>>>(v, nrm, rows, w);
    KERNEL_CHECK();
    // (Twice the steps on the P1 level, whose Chebyshev sequences are the long ones since round 5, were measured: the estimate rises, the interval
    // with it, and the solves get 1-3 % slower -- UV sphere 20 -> 23 iterations -- on top of 0.5-1.6 ms of set-up; the measured margin of 1.1 x the
    // 20-step estimate is 6-9 % on the workloads, and a bound that falls short is caught by the retry in eigs_impl.)
    constexpr int power_its = kPowerIterations;
    for (int it = 0; it < power_its; ++it) {
        mh_spmm(ctx, lvl, lvl.aval, v, t, nullptr, nullptr, w);
        k_dinv_mul<<<grid1(rows * w), TB, 0, ctx->stream>>>(t, lvl.dinv, v, rows, w);
        KERNEL_CHECK();
        mh_apply_patches<double>(ctx, ps, t, nullptr, w, 1.0, v, nullptr, nullptr, 0, py.get()); // v = M^-1 t with the sliver patches
        // The columns are rescaled every fourth step only (the iterate grows by at most lmax <= ~20 per step: 1.6e5 in four), and
        // before the last step, whose norm is the estimate: the column norms and the rescaling were 45 % of a step's 390 us, and
        // this loop -- not the coarse elimination beside it -- is what `factorize` waits for.
        const bool last = it + 1 == power_its, rescale = !last && (it % 4 == 3 || it + 2 == power_its);
        if (last || rescale) colsumsq(ctx, v, rows, w, nrm, scratch);
        if (rescale) {
            k_scale_cols_inv_sqrt<<<grid1(rows * w), TB, 0, ctx->stream>>>(v, nrm, rows, w);
            KERNEL_CHECK();
        }
    }
    auto h = nrm.to_host();
    double m = 0;
    for (double s : h) m = std::max(m, std::sqrt(s));
    // MH_TEST=lmax_low: the bound 12 % below its value, i.e. BELOW the spectrum's end -- what a power iteration that has not converged would
    // deliver; the long Chebyshev sequences then amplify the top of the spectrum and the iteration stalls (the retry in eigs_impl is tested with it)
    static const bool low = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "lmax_low");
    return 1.1 * m * (low ? 0.88 : 1.0);
}
} // namespace

void mh_finish_hierarchy(mh_system *sys) {
    if (!sys->coarse_pending) return;
    mh_context *ctx = sys->ctx;
    sys->coarse_pending = false;
    HIP_CHECK(hipStreamWaitEvent(ctx->stream, sys->coarse_done, 0));
    int hinfo = 0;
    sys->coarse_info.download(&hinfo, 1); // (on the main stream, i.e. after the elimination)
    for (auto &ws : sys->coarse_ws) ws.reset(ctx, 0);
    for (PatchSet *ps : {&sys->patches2, &sys->patches1}) { // sliver patches that were dropped (block not safely positive definite)
        if (!ps->any() || !ps->dropped.count) continue;
        int dropped[2] = {0, 0};
        ps->dropped.download(dropped, 2);
        sys->dropped_patches[ps->npe == 10 ? 0 : 1] = uint32_t(dropped[0]);
        if (dropped[0] && switches().verbose)
            fprintf(stderr, "[lobpcg] %d of %u + %u sliver patches / clusters of the %s level dropped (e.g. patch %d; clusters count from 1000000): their nodes keep the diagonal scaling only\n", dropped[0], ps->n_patches, ps->n_clusters,
                    ps->npe == 10 ? "P2" : "P1", dropped[1] - 1);
    }
    if (switches().test_coarse_pivot && sys->coarse_lift == 0) hinfo = 1; // (test hook: the first elimination of every system reports a lost pivot)
    if (hinfo != 0) {
        sys->hierarchy_ready = false;
        mh_throw(MH_EFACTOR, "coarse operator not positive definite (pivot %d of a diagonal block): shift must be negative", hinfo);
    }
}

void mh_build_hierarchy(mh_system *sys, double sigma, bool defer) {
    mh_context *ctx = sys->ctx;
    if (sys->hierarchy_ready && sys->sigma_built == sigma) {
        if (!defer) mh_finish_hierarchy(sys);
        return;
    }
    mh_finish_hierarchy(sys); // (a rebuild at another shift while the last elimination was never waited for)
    hipStream_t main_stream = ctx->stream;
    for (BsrLevel *lvl : {&sys->L2, &sys->L1}) {
        lvl->aval.reset(ctx, lvl->n_blocks * 9);
        lvl->dinv.reset(ctx, size_t(3) * lvl->n_nodes);
        k_shift_values<<<grid1(lvl->n_blocks * 9), TB, 0, ctx->stream>>>(lvl->kval, lvl->mval, lvl->n_blocks, sigma, lvl->aval);
        KERNEL_CHECK();
        k_diag_inverse<<<grid1(lvl->n_nodes), TB, 0, ctx->stream>>>(lvl->row_ptr, lvl->col, lvl->aval, lvl->n_nodes, lvl->dinv);
        KERNEL_CHECK();
        mh_build_patch_inverses(ctx, *lvl, lvl->id == 2 ? sys->patches2 : sys->patches1);
    }
    const size_t n0 = size_t(6) * sys->n_agg;
    sys->a0.reset(ctx, n0 * n0);
    sys->a0.zero();
    k_coarse_matrix<<<sys->n_agg, 64, 0, ctx->stream>>>(sys->L1.row_ptr, sys->L1.col, sys->L1.aval, sys->agg_t, sys->agg_of, sys->agg_ptr, sys->agg_nodes, sys->n_agg, sys->a0);
    KERNEL_CHECK();
    // (a mesh with flat cells: the Galerkin product cancels entries of 1e17 down to rigid-body terms of 1e10 and below -- its rounding is
    // ~1e-9 of the diagonal, enough to cost the coarse operator its definiteness: "coarse operator not positive definite" on one stretched
    // UV sphere of the round-6 soak.  The diagonal is lifted by that much instead of by 1e-12.)
    k_fix_coarse_diag<<<grid1(n0), TB, 0, ctx->stream>>>(sys->a0, uint32_t(n0), std::max(std::max(sys->coarse_lift, getenv("MH_COARSE_LIFT") ? atof(getenv("MH_COARSE_LIFT")) : 0.0), sys->worst_quality < kFlatShape ? 1e-8 : 1e-12));
    KERNEL_CHECK();
    DevArray<int> &info = sys->coarse_info;
    info.reset(ctx, 1);
    // The rest of the set-up -- the spectral bounds of both smoothers (power iterations: SpMMs that fill the GPU, with host
    // round trips) and the single-precision copies -- does not depend on the coarse inverse, whose elimination is a chain of
    // one-workgroup kernels and small products: the two run side by side on two streams.
    auto smoother_setup = [&] {
        for (BsrLevel *lvl : {&sys->L2, &sys->L1}) {
            lvl->lmax = estimate_lmax(ctx, *lvl, lvl->id == 2 ? sys->patches2 : sys->patches1);
            if (sys->lmax_widened) lvl->lmax *= 1.25; // (an earlier solve of this system needed the wider bound: eigs_impl)
            lvl->aval32.reset(ctx, lvl->n_blocks * 9);
            lvl->dinv32.reset(ctx, size_t(3) * lvl->n_nodes);
            k_convert<double, float><<<grid1(lvl->n_blocks * 9), TB, 0, ctx->stream>>>(lvl->aval.get(), lvl->aval32.get(), lvl->n_blocks * 9);
            KERNEL_CHECK();
            k_convert<double, float><<<grid1(size_t(3) * lvl->n_nodes), TB, 0, ctx->stream>>>(lvl->dinv.get(), lvl->dinv32.get(), size_t(3) * lvl->n_nodes);
            KERNEL_CHECK();
        }
    };
    {
        // Explicit inverse by block Gauss-Jordan elimination (no pivoting: the matrix is SPD), 128 columns per step:
        //   P = A_kk^-1 (one workgroup, in registers);  C = A(:, k);  R = P A(k, :);  A -= C R;  A(k, :) = R;  A(:, k) = -C P;  A_kk = P.
        // 2 n0^3 flops in n0 / 128 rank-128 updates of the whole matrix (rocBLAS dgemm) instead of potrf + potri's chains of
        // panel kernels: 17 ms -> 8 ms at n0 = 3 690 -- and, unlike rocsolver_dpotrf, undisturbed by concurrent streams, so a
        // solve needs no exclusive phase on the device.  The coarse solve becomes one dense product per application (2.5x
        // faster than two triangular solves at these sizes, and free of their O(n0 / 128) dependent launches).
        const uint32_t nb = 128;
        DevArray<double> &cblk = sys->coarse_ws[0], &rblk = sys->coarse_ws[1], &pnext = sys->coarse_ws[4];
        DevArray<double> *pinv2[2] = {&sys->coarse_ws[2], &sys->coarse_ws[3]};
        cblk.reset(ctx, n0 * nb);
        rblk.reset(ctx, n0 * nb);
        pinv2[0]->reset(ctx, size_t(nb) * nb);
        pinv2[1]->reset(ctx, size_t(nb) * nb);
        pnext.reset(ctx, size_t(nb) * nb);
        info.zero();
        const bool side = ctx->aux_stream_ready();
        struct StreamGuard { // whatever happens below, the context leaves on its own stream
            mh_context *c;
            hipStream_t s;
            bool swapped{false};
            ~StreamGuard() {
                if (c->stream != s) {
                    c->stream = s;
                    if (swapped) std::swap(c->blas, c->blas_aux);
                    (void)hipStreamSynchronize(c->aux_stream);
                    if (c->aux2_stream) (void)hipStreamSynchronize(c->aux2_stream);
                }
            }
        } guard{ctx, main_stream};
        hipEvent_t forked = nullptr;
        if (side) { // everything queued so far precedes the elimination
            HIP_CHECK(hipEventCreateWithFlags(&forked, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(forked, main_stream));
            HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, forked, 0));
            ctx->stream = ctx->aux_stream;
            std::swap(ctx->blas, ctx->blas_aux); // the second stream's own handle
            guard.swapped = true;
        }
        // One step ahead: the pivot block of step k + 1 is brought up to date FIRST (a 128 x 128 x 128 product of ours into a buffer of
        // its own) and inverted on a third stream while the rank-128 update of the whole matrix runs -- the one-workgroup inverse
        // (117 us) and the update (131 us) were 85 % of a step's 287 us, one after the other.
        const size_t steps_total = div_up(n0, size_t(nb));
        const bool ahead = side && ctx->aux2_stream_ready(2 * steps_total);
        // (the context's own events: an event destroyed while a stream still waits on it -- as these were at first, at the end of this
        // scope, milliseconds ahead of the device -- lets the wait through early on this runtime: one solve in ~50 then ran on a coarse
        // inverse built from a half-finished pivot inverse, under load only)
        hipEvent_t *ev_piv = ctx->ahead_ev.data(), *ev_inv = ctx->ahead_ev.data() + steps_total; // [step]: no event serves two steps
        const double one = 1, zero = 0, mone = -1;
        const rocblas_int ld = rocblas_int(n0);
        double *a = sys->a0.get();
        uint32_t step = 0;
        for (size_t k0 = 0; k0 < n0; k0 += nb, ++step) {
            const rocblas_int w = rocblas_int(std::min<size_t>(nb, n0 - k0));
            double *pinv = pinv2[step & 1]->get();
            if (step == 0 || !ahead) mh_spd_inverse_small(ctx, a + k0 * n0 + k0, uint32_t(n0), uint32_t(w), pinv, uint32_t(w), info);
            else HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev_inv[step], 0));
            HIP_CHECK(hipMemcpyAsync(cblk, a + k0 * n0, n0 * size_t(w) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, w, ld, w, &one, pinv, w, a + k0, ld, &zero, rblk, w));
            const size_t k1 = k0 + nb;
            if (ahead && k1 < n0) {
                const uint32_t wn = uint32_t(std::min<size_t>(nb, n0 - k1));
                HIP_CHECK(hipMemcpy2DAsync(pnext.get(), size_t(wn) * sizeof(double), a + k1 * n0 + k1, n0 * sizeof(double), size_t(wn) * sizeof(double), wn, hipMemcpyDeviceToDevice, ctx->stream));
                mh_small_gemm(ctx, false, false, wn, wn, uint32_t(w), -1.0, cblk.get() + k1, uint32_t(n0), rblk.get() + k1 * size_t(w), uint32_t(w), 1.0, pnext.get(), wn);
                HIP_CHECK(hipEventRecord(ev_piv[step], ctx->stream));
                HIP_CHECK(hipStreamWaitEvent(ctx->aux2_stream, ev_piv[step], 0));
                hipStream_t elimination = ctx->stream;
                ctx->stream = ctx->aux2_stream;
                try {
                    mh_spd_inverse_small(ctx, pnext.get(), wn, wn, pinv2[(step + 1) & 1]->get(), wn, info);
                } catch (...) {
                    ctx->stream = elimination;
                    throw;
                }
                ctx->stream = elimination;
                HIP_CHECK(hipEventRecord(ev_inv[step + 1], ctx->aux2_stream));
            }
            ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, ld, ld, w, &mone, cblk, ld, rblk, w, &one, a, ld));
            HIP_CHECK(hipMemcpy2DAsync(a + k0, n0 * sizeof(double), rblk.get(), size_t(w) * sizeof(double), size_t(w) * sizeof(double), n0, hipMemcpyDeviceToDevice, ctx->stream));
            ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, ld, w, w, &mone, cblk, ld, pinv, w, &zero, a + k0 * n0, ld));
            HIP_CHECK(hipMemcpy2DAsync(a + k0 * n0 + k0, n0 * sizeof(double), pinv, size_t(w) * sizeof(double), size_t(w) * sizeof(double), size_t(w), hipMemcpyDeviceToDevice, ctx->stream));
        }
        k_symmetrize_lower<<<grid1(n0 * n0), TB, 0, ctx->stream>>>(sys->a0, uint32_t(n0), uint32_t(n0));
        KERNEL_CHECK();
        if (side) { // back to the main stream for the smoothers' set-up while the elimination runs
            ctx->stream = main_stream;
            std::swap(ctx->blas, ctx->blas_aux);
        }
        // the elimination's end, as the main stream will wait for it (without a second stream it ran on the main one)
        if (!sys->coarse_done) HIP_CHECK(hipEventCreateWithFlags(&sys->coarse_done, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(sys->coarse_done, side ? ctx->aux_stream : main_stream));
        sys->coarse_pending = true;
        smoother_setup();
        if (side) (void)hipEventDestroy(forked);
    }
    sys->sigma_built = sigma;
    sys->hierarchy_ready = true;
    if (!defer) {
        mh_finish_hierarchy(sys);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
}

namespace {
// x[i][j] = scale_j * z[(first + reversed j)] : the `ncols` LAST columns of a column-major n x n matrix, last first, as a row-major
// panel, column j scaled by scale[j]
__global__ void k_last_columns_to_panel(const double *__restrict__ z, size_t n, uint32_t ncols, const double *__restrict__ scale, double *__restrict__ x) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n * ncols) return;
    const uint32_t j = uint32_t(i % ncols);
    x[i] = scale[j] * z[(n - 1 - j) * n + i / ncols];
}

// One dense generalised eigensolve on the device: tiny problems, and the last resort of a small one whose iteration did not
// converge.  `inverse` solves M x = nu A x instead of A x = theta M x and takes the LARGEST nu = 1 / theta: on a pencil whose
// stiffness is inflated by near-degenerate elements (||A|| / theta ~ 1e13 on a UV sphere's tetrahedra of four ring points) the
// direct form returns the low pairs with an absolute error of eps ||L^-1 A L^-T||, the inverse form -- like the reference's
// shift-invert -- with a relative one.
void dense_eigs(mh_system *sys, uint32_t nev, double sigma, double *eigenvalues, bool inverse = false) {
    mh_context *ctx = sys->ctx;
    const size_t n = size_t(3) * sys->n_nodes;
    auto &lvl = sys->L2;
    DevArray<double> a(ctx, n * n), m(ctx, n * n), d(ctx, n), e(ctx, n);
    DevArray<int> info(ctx, 1);
    a.zero();
    m.zero();
    lvl.aval.reset(ctx, lvl.n_blocks * 9);
    k_shift_values<<<grid1(lvl.n_blocks * 9), TB, 0, ctx->stream>>>(lvl.kval, lvl.mval, lvl.n_blocks, sigma, lvl.aval);
    KERNEL_CHECK();
    k_bsr_to_dense<<<grid1(lvl.n_nodes), TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval, lvl.mval, lvl.n_nodes, a, m);
    KERNEL_CHECK();
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ExclusivePhase alone_on_the_device(ctx->device);
    double *lhs = inverse ? m.get() : a.get(), *rhs = inverse ? a.get() : m.get(); // (vectors come back in lhs)
    ROCBLAS_CHECK(rocsolver_dsygvd(ctx->blas, rocblas_eform_ax, rocblas_evect_original, rocblas_fill_lower, rocblas_int(n), lhs, rocblas_int(n), rhs, rocblas_int(n), d, e, info));
    int hinfo = 0;
    info.download(&hinfo, 1);
    if (hinfo != 0) mh_throw(hinfo > int(n) ? MH_EFACTOR : MH_ENOTCONVERGED, "dense sygvd failed (info %d)", hinfo);
    std::vector<double> th(n);
    d.download(th.data(), n);
    sys->evecs.reset(ctx, n * nev);
    sys->evec_cols = nev;
    if (!inverse) {
        for (uint32_t i = 0; i < nev; ++i) eigenvalues[i] = th[i] + sigma;
        k_colmajor_to_panel<<<grid1(n * nev), TB, 0, ctx->stream>>>(lhs, n, nev, sys->evecs);
    } else {
        // nu ascending: the wanted pairs are the last ones, largest first; x^T A x = 1 as returned, so x^T M x = nu: scale by nu^-1/2
        std::vector<double> scale(nev);
        for (uint32_t i = 0; i < nev; ++i) {
            const double nu = th[n - 1 - i];
            if (!(nu > 0)) mh_throw(MH_ENOTCONVERGED, "dense inverse eigensolve: non-positive eigenvalue %g at position %u", nu, i);
            eigenvalues[i] = 1.0 / nu + sigma;
            scale[i] = 1.0 / std::sqrt(nu);
        }
        DevArray<double> dscale(ctx, nev);
        dscale.upload(scale.data(), nev);
        k_last_columns_to_panel<<<grid1(n * nev), TB, 0, ctx->stream>>>(lhs, n, nev, dscale, sys->evecs);
    }
    KERNEL_CHECK();
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
}
} // namespace

namespace {
__global__ void k_axpy_panel(double *__restrict__ y, const double *__restrict__ b, size_t count) { // y <- b - y
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) y[i] = b[i] - y[i];
}
__global__ void k_coldot_partial(const double *__restrict__ x, const double *__restrict__ y, size_t rows, uint32_t w, double *__restrict__ partial, uint32_t rows_per_block) {
    const uint32_t c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= w) return;
    const size_t r0 = size_t(blockIdx.x) * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double s = 0;
    for (size_t r = r0; r < r1; ++r) s += x[r * w + c] * y[r * w + c];
    partial[size_t(blockIdx.x) * w + c] = s;
}
// x += alpha p, r -= alpha ap with alpha_c = rz_c / pap_c (0 for a column that has converged exactly)
__global__ void k_cg_advance(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ p, const double *__restrict__ ap, const double *__restrict__ rz, const double *__restrict__ pap,
                             size_t count, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t c = uint32_t(i % w);
    const double alpha = pap[c] > 0 ? rz[c] / pap[c] : 0.0;
    x[i] += alpha * p[i];
    r[i] -= alpha * ap[i];
}
// p = z + beta p with beta_c = rz_new_c / rz_old_c
__global__ void k_cg_direction(double *__restrict__ p, const double *__restrict__ z, const double *__restrict__ rz_new, const double *__restrict__ rz_old, size_t count, uint32_t w) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t c = uint32_t(i % w);
    const double beta = rz_old[c] > 0 ? rz_new[c] / rz_old[c] : 0.0;
    p[i] = z[i] + beta * p[i];
}
} // namespace

namespace {
// z ~= A^-1 r for an n x w panel by `iters` steps of conjugate gradients preconditioned with the cycle, the columns independent runs in
// lockstep (no convergence test: no read-back).  The LAST RESORT's preconditioner (eigs_impl): where the cycle alone leaves the block
// iteration unstable -- a mesh with cells flat to 1e-9: what the cycle returns carries components along the flat cells' stiff directions
// that the Rayleigh-Ritz step cannot digest -- a few CG steps around it hand the eigensolver (nearly) the shift-invert direction A^-1 r of
// the reference's own method (src/audio/mesh2modes.cpp:441-512), at ten times the cost per iteration.
struct PanelCg {
    mh_system *sys;
    mh_context *ctx;
    size_t n;
    uint32_t wmax;
    DevArray<double> rr, zz, p, ap, rz, rz_new, pap, scratch;
    static constexpr uint32_t rpb = 256;
    PanelCg(mh_system *s, uint32_t w) : sys(s), ctx(s->ctx), n(size_t(3) * s->n_nodes), wmax(w), rr(s->ctx, n * w), zz(s->ctx, n * w), p(s->ctx, n * w), ap(s->ctx, n * w), rz(s->ctx, w), rz_new(s->ctx, w),
                                        pap(s->ctx, w), scratch(s->ctx, size_t(div_up(n, rpb)) * w) {}
    void dot(const double *u, const double *v, double *out, uint32_t w) {
        const uint32_t nb = uint32_t(div_up(n, rpb));
        k_coldot_partial<<<dim3(nb, div_up(w, 64)), 64, 0, ctx->stream>>>(u, v, n, w, scratch, rpb);
        KERNEL_CHECK();
        k_colsumsq_final<<<w, 256, 0, ctx->stream>>>(scratch, nb, w, out);
        KERNEL_CHECK();
    }
    template<typename Prec> void solve(Prec &prec, const double *r, double *z, uint32_t w, int iters) {
        hipStream_t st = ctx->stream;
        HIP_CHECK(hipMemsetAsync(z, 0, n * w * sizeof(double), st));
        HIP_CHECK(hipMemcpyAsync(rr.get(), r, n * w * sizeof(double), hipMemcpyDeviceToDevice, st));
        prec.apply(rr, zz, w);
        HIP_CHECK(hipMemcpyAsync(p.get(), zz.get(), n * w * sizeof(double), hipMemcpyDeviceToDevice, st));
        dot(rr, zz, rz, w);
        for (int k = 0; k < iters; ++k) {
            mh_spmm(ctx, sys->L2, sys->L2.aval, p, ap, nullptr, nullptr, w);
            dot(p, ap, pap, w);
            k_cg_advance<<<grid1(n * w), TB, 0, st>>>(z, rr, p, ap, rz, pap, n * w, w);
            KERNEL_CHECK();
            if (k + 1 == iters) break;
            prec.apply(rr, zz, w);
            dot(rr, zz, rz_new, w);
            k_cg_direction<<<grid1(n * w), TB, 0, st>>>(p, zz, rz_new, rz, n * w, w);
            KERNEL_CHECK();
            HIP_CHECK(hipMemcpyAsync(rz.get(), rz_new.get(), w * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
    }
};
} // namespace

namespace {
// Block LOBPCG on the pencil (A, M) = (K - sigma M, M) with hard locking: one object per solve; the
// stages of an iteration are its member functions, in the order run() calls them.
struct BlockLobpcg {
    // ---- the solve
    mh_system *sys;
    mh_context *ctx;
    hipStream_t st;
    const size_t n;
    const uint32_t nev, b, mmax;
    const double sigma, residual_tol;
    const uint32_t max_iters;
    const float *seed_basis;
    const uint32_t seed_rows, seed_cols;
    const volatile unsigned char *cancel;
    volatile float *progress;
    mh_profile &prof;
    mh_profile *profile;
    SharedPhase iterating; // released and re-taken at the top of every iteration so that a waiting factorisation gets in
    Timer t_iter;
    double precond_seconds = 0;
    double best_worst_active = 1e300; // smallest worst relative residual of the active columns seen so far
    int floor_strikes = 0;            // consecutive iterations in which the worst active residual sat 1e3 above it
    std::vector<double> hist_active;  // the worst active residual per iteration (the last resort's stall test)
    uint32_t pairs_at_floor = 0;      // wanted pairs handed on although their residual stayed above the tolerance (the last resort only; see converged_or_locked)
    // ---- panels (n x b) and small matrices
    DevArray<double> X, AX, MX, Xn, AXn, MXn, W, AW, MW, P, MP, Pn, MPn, R, Rw;
    DevArray<double> gA, gM, gM0, gA0, App, evals, ework, Cp, T1, H, H2, G, dscale, Linv, theta_d, rn_d, mn_d, scratch, Ct, norms_d, theta_act_d, Hp, Up, Vp, T1p;
    DevArray<double> Linv_p, T1p2; // the conjugate directions' own triangular inverse (Linv belongs to W until the basis update) and the product's output
    DevArray<uint32_t> idx_d, pos_d;
    // The residuals of the active columns as the image product's epilogue leaves them (mh_spmm_mapped): compact panel of pitch
    // res_pitch for the columns res_act, per-node norm partials, the reduced norms [2][res_pitch]
    DevArray<double> Rr, res_partial, res_blocks, res_norms_d;
    std::vector<double> rr_evals; // the Rayleigh-Ritz step's eigenvalues when rr_solve read them back itself
    std::vector<uint32_t> res_act, res_pos; // (res_pos: a member so that its asynchronous upload never outlives it)
    std::vector<double> res_norms;
    uint32_t res_pitch = 0;
    bool res_ready = false, rw_from_rr = false;
    DevArray<int> info; // [1]: conditioning report of mh_potrf_small
    int inner_cg = 0; // > 0: the preconditioner is that many CG steps around the (double-precision) cycle -- the last resort (eigs_impl)
    std::unique_ptr<Precond<float>> prec32;
    std::unique_ptr<Precond<double>> prec64;
    std::unique_ptr<PanelCg> panel_cg;
    // ---- switches (read once) and what follows from them
    const bool verbose = switches().verbose, fp32_prec = switches().fp32_prec;
    // W is orthogonalised against P in coefficient space (no M P panel, no tall projection against P) for blocks of up to 128 columns
    const bool pproj_ok;
    // A mesh with sliver elements (mh_patch.hip) measures its residuals in the Jacobi-scaled norm, ||r||_{D^-1} against
    // theta ||M x||_{D^-1}: the same quantity on a uniform mesh, but the rounding noise of the assembled operator sits on the slivers'
    // rows (entries up to 1e6 times their neighbours': eps ||A|| |x| there), where D is as large as the noise -- in the plain
    // 2-norm that noise alone exceeds the tolerance (95k-tet skillet scan: floor 6e-3 against 1e-5) and nothing ever converges.
    const bool scaled_norms;
    // ---- state across iterations
    bool warm = false;
    bool w_implicit = false;       // W itself left untransformed this iteration (see chol_orthonormalise)
    int last_spread = 1 << 20;     // 16 log2(max / min diagonal of the last Cholesky factor of a unit-diagonal Gram matrix)
    bool p_needs_explicit = false; // the implicit projection against P was refused (ill-conditioned Gram matrix): caller redoes it explicitly
    uint32_t wp = 0;               // width of P
    uint32_t iters = 0, nconv = 0, redos_at_start = 0;
    bool poly_allowed = true; // (cleared when the polynomial start block turned out rank deficient: start() then runs once more without it)
    std::vector<double> theta, rn, mn, xn, norms, theta_act, hist_worst;
    double anorm = 0;
    std::vector<uint32_t> act, order, hist_nconv;
    std::vector<uint8_t> locked; // hard locking: a converged column leaves the Rayleigh-Ritz basis for good
    bool converged = false;
    bool gm_identity = false; // this iteration's gM0 is exactly I (all blocks placed, none measured)
    bool p_implicit = false;  // this iteration: the basis is [X, (W - P Hp) L^-T, P] with W, P stored
    bool lazy_images = false;
    // ---- this iteration
    uint32_t w = 0, wa = 0, m = 0, wp_new = 0; // active columns (= width of W), their count as the X part, order of the small problem, width of the next P

    BlockLobpcg(mh_system *system, uint32_t nev_, uint32_t block, double sigma_, double residual_tol_, uint32_t max_iters_, const float *seed_basis_, uint32_t seed_rows_,
                uint32_t seed_cols_, const volatile unsigned char *cancel_, volatile float *progress_, mh_profile &prof_, mh_profile *profile_, int inner_cg_ = 0)
        : sys(system), ctx(system->ctx), st(system->ctx->stream), n(size_t(3) * system->n_nodes), nev(nev_), b(block), mmax(3 * block), sigma(sigma_), residual_tol(residual_tol_),
          max_iters(max_iters_), seed_basis(seed_basis_), seed_rows(seed_rows_), seed_cols(seed_cols_), cancel(cancel_), progress(progress_), prof(prof_), profile(profile_),
          iterating(system->ctx->device, true), t_iter(system->ctx), pproj_ok(block <= 128), scaled_norms(system->patches2.any()), theta(block), rn(block), mn(block), xn(block), norms(3 * size_t(block)),
          theta_act(block), order(block), locked(block, 0) {
        inner_cg = inner_cg_;
        for (DevArray<double> *panel : {&X, &AX, &MX, &Xn, &AXn, &MXn, &W, &AW, &MW, &P, &Pn, &R, &Rw}) panel->reset(ctx, n * b);
        if (!pproj_ok) // M P is only kept for blocks wider than 128 columns (the narrower ones project against P in coefficient space)
            for (DevArray<double> *panel : {&MP, &MPn}) panel->reset(ctx, n * b);
        for (DevArray<double> *small : {&gA, &gM, &gM0, &gA0}) small->reset(ctx, size_t(mmax) * mmax);
        for (DevArray<double> *small : {&App, &H, &H2, &Linv, &Hp, &Up, &Vp, &T1p}) small->reset(ctx, size_t(b) * b);
        Linv_p.reset(ctx, size_t(b) * b);
        T1p2.reset(ctx, size_t(mmax) * b);
        for (DevArray<double> *small : {&Cp, &T1}) small->reset(ctx, size_t(mmax) * b);
        for (DevArray<double> *small : {&evals, &ework}) small->reset(ctx, mmax);
        for (DevArray<double> *small : {&dscale, &theta_d, &rn_d, &mn_d, &theta_act_d}) small->reset(ctx, b);
        G.reset(ctx, size_t(b) * 2 * b);
        Ct.reset(ctx, size_t(mmax) * 2 * b);
        norms_d.reset(ctx, 3 * size_t(b));
        idx_d.reset(ctx, b);
        pos_d.reset(ctx, b);
        if (b <= 128) { // (the epilogue covers panels of up to 128 columns; wider blocks keep the separate residual pass)
            const uint32_t pmax = (b + 1u) & ~1u;
            Rr.reset(ctx, n * pmax);
            res_partial.reset(ctx, size_t(2) * sys->n_nodes * pmax);
            res_blocks.reset(ctx, size_t(div_up(sys->n_nodes, 48)) * 2 * pmax);
            res_norms_d.reset(ctx, size_t(2) * pmax);
            res_norms.resize(size_t(2) * pmax);
        }
        info.reset(ctx, 2);
        // Single-precision smoothers unless the mesh has FLAT cells (round 6): an element of shape 1e-9 puts entries 1e9 times its neighbours'
        // into the operator, and the residuals of its rows -- differences of such entries -- have no correct digit in fp32; the clusters'
        // inverses (condition number ~ 1 / shape) then turn that noise into the iterate.  Such a mesh starts in double precision.
        if (fp32_prec && !(sys->worst_quality < kFlatShape) && !inner_cg) prec32 = std::make_unique<Precond<float>>(sys, std::min(b, kPrecondColumns));
        else prec64 = std::make_unique<Precond<double>>(sys, std::min(b, kPrecondColumns));
        if (inner_cg) panel_cg = std::make_unique<PanelCg>(sys, std::min(b, kPrecondColumns));
        auto hd = sys->L2.dinv.to_host();
        double dmin = 1e300;
        for (double v : hd) dmin = std::min(dmin, v);
        anorm = sys->L2.lmax / dmin; // lambda_max(A) <= lambda_max(D^-1 A) * max diag(A)
    }

    // G = V^T M V, scaled to unit diagonal, Cholesky; V <- V L^-T (and the same for M V, A V).  False: the block is rank deficient.

    bool chol_orthonormalise(double *V, double *MV, double *AV, uint32_t w, bool transform_images = true, bool allow_implicit = false,
                             const double *hp = nullptr, uint32_t hp_rows = 0) {
        // hp (hp_rows x w, = P^T M V): V is to be taken as V - P hp without forming it: G -= hp^T hp (P is M-orthonormal).  Only
        // valid together with the implicit treatment of V; when that is refused nothing is transformed and p_needs_explicit is set.
        // allow_implicit: with a well-conditioned Gram matrix not even V is transformed (w_implicit is set): the caller works
        // with V L^-T through L^-1 on the small matrices
        // transform_images = false: M V is only read (for the Gram matrix); the caller keeps the images untransformed
        // G = V^T M V, scaled to unit diagonal, Cholesky; V <- V L^-T (and the same for MV, AV)
        gram(ctx, n, V, w, MV, w, G, w);
        if (hp) {
            const double minus = -1, plus = 1;
            small_dgemm(ctx, rocblas_operation_transpose, rocblas_operation_none, rocblas_int(w), rocblas_int(w), rocblas_int(hp_rows), &minus, hp,
                                        rocblas_int(hp_rows), hp, rocblas_int(hp_rows), &plus, G, rocblas_int(w));
        }
        k_scale_gram<<<grid1(size_t(w) * w), TB, 0, st>>>(G, w, w, dscale);
        KERNEL_CHECK();
        double *Gs = G.get() + size_t(w) * w;
        int hinfo = 0;
        bool fused_inverse = false; // Gs already unscaled and Linv already formed (orders <= 128)
        {
            last_spread = 1 << 20;
            if (w <= 128) {
                // one workgroup of ours: factor, unscale and invert in one launch (round 5; before: potrf, unscale, memset and the library's
                // trtri -- six to eight launches, ~130 us with their gaps, twice per iteration); MH_TEST=potrf_chain: the old chain
                static const bool chain = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "potrf_chain");
                fused_inverse = !chain;
                if (fused_inverse) mh_potrf_small_inverse(ctx, Gs, w, info, dscale, Linv);
                else mh_potrf_small(ctx, Gs, w, info);
                int both[2] = {0, 0};
                info.download(both, 2);
                hinfo = both[0];
                last_spread = both[1];
            } else { // (blocked, without rocSOLVER: its potrf is disturbed by concurrent solves -- mh_potrf)
                mh_potrf(ctx, Gs, w, w, info);
                info.download(&hinfo, 1);
            }
        }
        p_needs_explicit = false;
        if (hp && (hinfo != 0 || last_spread >= 16 * 8)) { // V - P hp is nearly dependent: project in the tall space instead
            p_needs_explicit = true;
            return true;
        }
        if (hinfo != 0) return false;
        if (!fused_inverse) {
            k_unscale_chol<<<grid1(size_t(w) * w), TB, 0, st>>>(Gs, w, w, dscale);
            KERNEL_CHECK();
        }
        if (w > 256) {
            panel_trsm(ctx, n, V, w, Gs, w);
            if (MV) panel_trsm(ctx, n, MV, w, Gs, w);
            if (AV) panel_trsm(ctx, n, AV, w, Gs, w);
        } else {
            // V <- V L^-T through the explicit small inverse and the MFMA basis-update kernel (in place: a workgroup
            // reads its rows before it writes them): L^-1 column-major IS the k-major coefficient matrix of V L^-T
            if (!fused_inverse) {
                HIP_CHECK(hipMemsetAsync(Linv, 0, size_t(w) * w * sizeof(double), st));
                ROCBLAS_CHECK(rocblas_dtrtri(ctx->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, w, Gs, w, Linv, w));
            }
            w_implicit = allow_implicit && !transform_images && last_spread < 16 * 8;
            if (!w_implicit) mh_combine(ctx, n, V, w, nullptr, 0, nullptr, 0, Linv, w, V, w, nullptr);
            if (MV && transform_images) mh_combine(ctx, n, MV, w, nullptr, 0, nullptr, 0, Linv, w, MV, w, nullptr);
            if (AV && transform_images) mh_combine(ctx, n, AV, w, nullptr, 0, nullptr, 0, Linv, w, AV, w, nullptr);
        }
        return true;
    }

    // z = B r for w columns.  The smoothers' wide-load products take panels of at most 256 columns (64 lanes x 4 floats): a wider block
    // (more than ~230 wanted pairs: beyond the reference editor's 128 + margin and BASELINE's 200) goes through in column slabs,
    // gathered into a compact panel and scattered back (two extra passes over the slab, a few per cent of the cycle's own traffic).
    void precondition(const double *r, double *z, uint32_t w) {
        auto one = [&](const double *rp, double *zp, uint32_t wc) {
            if (panel_cg) panel_cg->solve(*prec64, rp, zp, wc, inner_cg);
            else if (prec32) prec32->apply(rp, zp, wc);
            else prec64->apply(rp, zp, wc);
        };
        if (w <= kPrecondColumns) return one(r, z, w);
        const uint32_t slabs = div_up(w, kPrecondColumns), step = (div_up(w, slabs) + 3u) & ~3u;
        if (w > 1024) mh_throw(MH_EINVAL, "a block of %u columns exceeds the 1 024 the column maps cover", w);
        DevArray<double> rs(ctx, n * step), zs(ctx, n * step);
        const uint32_t *iota = mh_identity_map(ctx);
        for (uint32_t c0 = 0; c0 < w; c0 += step) {
            const uint32_t wc = std::min(step, w - c0);
            k_gather_cols<<<grid1(n * wc), TB, 0, st>>>(r, iota + c0, rs.get(), n, w, wc);
            KERNEL_CHECK();
            one(rs, zs, wc);
            k_scatter_cols<<<grid1(n * wc), TB, 0, st>>>(zs.get(), iota + c0, z, n, w, wc);
            KERNEL_CHECK();
        }
        HIP_CHECK(hipStreamSynchronize(st)); // the slabs return to the pool
    }

    // The initial block: seed columns (warm start), then Gaussian noise, the exact rigid-body modes, one smoothing pass;
    // M-orthonormalised and rotated into its Ritz vectors.
    void start() {
        redos_at_start = ctx->sytrd_redos;
        if (ctx->rr_check) HIP_CHECK(hipMemsetAsync(ctx->rr_check, 0, sizeof(unsigned long long), st));
        k_random_panel<<<grid1(n * b), TB, 0, st>>>(X, n * b, 20260710ull);
        KERNEL_CHECK();
        warm = seed_basis && seed_rows == n && seed_cols >= nev;
        if (warm) {
            const uint32_t ncols = std::min(seed_cols, b);
            DevArray<float> seed(ctx, n * ncols);
            seed.upload(seed_basis, n * ncols);
            k_load_seed<<<grid1(n * ncols), TB, 0, st>>>(seed, sys->perm, sys->n_nodes, ncols, b, X);
            KERNEL_CHECK();
            HIP_CHECK(hipStreamSynchronize(st));
        }
        // A free body's six rigid-body modes are exact eigenvectors (lambda = 0): start from them -- also on a warm start, where
        // they replace the seeded (single-precision) copies: a rigid mode known only to 1e-7 leaves A x = |sigma| M x as the
        // difference of terms 12 orders larger, and its Ritz value is then noise.  A mesh of several disconnected bodies (a scan
        // with stray fragments) has six such modes per body; all of them are seeded.
        const uint32_t n_rigid = 6 * sys->n_components;
        const bool seed_rigid = b >= n_rigid + 6;
        if (!seed_rigid && sys->n_components > 1)
            mh_throw(MH_EINVAL, "the mesh consists of %u disconnected bodies (%u zero modes): more than a block of %u columns can seed", sys->n_components, n_rigid, b);
        const auto inject_rigid_modes = [&] {
            k_inject_rbm<<<grid1(sys->n_nodes), TB, 0, st>>>(sys->node_xyz, sys->node_component, sys->component_centroid, sys->n_components, sys->n_nodes, X, b, 0);
            KERNEL_CHECK();
        };
        if (seed_rigid) inject_rigid_modes();
        // Smooth start (round 5): behind the rigid-body modes a cold block begins from low-degree polynomial displacement fields
        // (k_inject_poly: uniform strains, then every Legendre product of total degree 2, 3, ... while whole degrees fit) instead of noise
        // in three quarters of its columns; the last quarter stays random (what the polynomials cannot represent still has to be found).
        // They pass through the same B M x step below.  Same iteration counts, but pairs lock earlier: 3-4 % per solve on the Kuhn cubes
        // (interleaved A/B, profiles/r05_poly_start_ab.txt); nothing on the sphere and the ball.  Only bodies that are not thin along any
        // axis (every extent at least a quarter of the longest) get it: on plates and shells it bought nothing -- the wanted modes there
        // are bending modes of the walls -- and the Kuhn plate's block came out rank deficient (a few node layers through the thickness).
        bool with_poly = !warm && seed_rigid && !switches().no_poly_start && sys->n_components == 1 && poly_allowed;
        if (with_poly) {
            DevArray<unsigned long long> keys(ctx, 6);
            const unsigned long long init[6] = {~0ull, ~0ull, ~0ull, 0, 0, 0};
            keys.upload(init, 6);
            k_bbox<<<grid1(sys->n_nodes), TB, 0, st>>>(sys->node_xyz, sys->n_nodes, keys);
            KERNEL_CHECK();
            unsigned long long hk[6];
            keys.download(hk, 6);
            double box[6], longest = 0, shortest = 1e300;
            for (int a = 0; a < 6; ++a) {
                const unsigned long long u = (hk[a] >> 63) ? (hk[a] & 0x7fffffffffffffffull) : ~hk[a];
                memcpy(&box[a], &u, sizeof(double));
            }
            for (int a = 0; a < 3; ++a) longest = std::max(longest, box[3 + a] - box[a]), shortest = std::min(shortest, box[3 + a] - box[a]);
            with_poly = shortest >= 0.25 * longest;
            if (with_poly) {
                const uint32_t room = b - n_rigid, want = room - room / 4;
                std::vector<uint32_t> triples; // exponents packed a | b << 8 | c << 16
                for (uint32_t deg = 2; deg <= 6 && 6 + 3 * (triples.size() + (deg + 1) * (deg + 2) / 2) <= want; ++deg)
                    for (uint32_t a = deg + 1; a-- > 0;)
                        for (uint32_t bb = deg - a + 1; bb-- > 0;) triples.push_back(a | bb << 8 | (deg - a - bb) << 16);
                DevArray<double> box_d(ctx, 6);
                DevArray<uint32_t> tri_d(ctx, std::max<size_t>(1, triples.size()));
                box_d.upload(box, 6);
                if (!triples.empty()) tri_d.upload(triples.data(), triples.size());
                k_inject_poly<<<grid1(sys->n_nodes), TB, 0, st>>>(sys->node_xyz, box_d, sys->n_nodes, tri_d, uint32_t(triples.size()), X, b, n_rigid);
                KERNEL_CHECK();
                HIP_CHECK(hipStreamSynchronize(st)); // (the small arrays return to the pool; the uploads read stack memory)
            }
        }
        // A cold start begins from B M x for Gaussian noise x (one preconditioner application: the high-frequency content
        // of the noise is damped before the first Rayleigh-Ritz step), with the exact rigid-body modes put back: one
        // iteration fewer on every workload measured (18 -> 17 at S100k, 40 -> 39 on the ball, 17 -> 16 at S30k).
        if (!warm) {
            {
                mh_spmm(ctx, sys->L2, nullptr, X, nullptr, sys->L2.mval, MX, b);
                Timer tp(ctx);
                precondition(MX, Xn, b);
                precond_seconds += tp.stop();
                prof.op_applications += b;
                HIP_CHECK(hipMemcpyAsync(X, Xn.get(), n * b * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
            if (seed_rigid) inject_rigid_modes();
        }
        mh_spmm(ctx, sys->L2, nullptr, X, nullptr, sys->L2.mval, MX, b);
        if (!chol_orthonormalise(X, MX, nullptr, b)) {
            if (with_poly) { // a body the polynomial fields are dependent on (too few node layers along an axis): the random block instead
                if (verbose) fprintf(stderr, "[lobpcg] polynomial start block rank deficient: random block instead\n");
                poly_allowed = false;
                start();
                return;
            }
            mh_throw(MH_ENOTCONVERGED, "initial block is rank deficient");
        }
        mh_spmm(ctx, sys->L2, sys->L2.aval, X, AX, sys->L2.mval, MX, b);
        {
            gram(ctx, n, X, b, AX, b, gA, b);
            gram(ctx, n, X, b, MX, b, gM, b);
            // (the library's divide and conquer here: every pair of a start block's matrix, whose spectrum spans |sigma| .. ||A|| -- our partial-spectrum
            // kernels, accepted at a residual of 1e-10 ||T||, left the small pairs at 1e-8 .. 1e-7 in the step's self-check: measured in round 5)
            const int hinfo = rr_solve(ctx, gA, gM, b, evals, ework, info);
            if (hinfo != 0) mh_throw(MH_ENOTCONVERGED, "initial Rayleigh-Ritz failed (info %d)", hinfo);
            panel_mul(ctx, n, X, b, gA, b, Xn, b, 1.0, 0.0);
            panel_mul(ctx, n, AX, b, gA, b, AXn, b, 1.0, 0.0);
            panel_mul(ctx, n, MX, b, gA, b, MXn, b, 1.0, 0.0);
            std::swap(X, Xn); std::swap(AX, AXn); std::swap(MX, MXn);
            evals.download(theta.data(), b);
        }
    }

    // Residuals and column norms of the whole block, hard locking of the converged columns, the count of converged wanted pairs;
    // true when the iteration is over (converged, nothing left to iterate on, or out of iterations).  Also the safety net
    // of the single-precision smoothers.
    bool converged_or_locked(uint32_t it) {
        if (cancel && *cancel) mh_throw(MH_ECANCELLED, "cancelled");
        if (g_concurrent) { // an exclusive holder synchronises the device itself: no need to drain our queue first
            iterating.release();
            iterating.acquire();
        }
        bool fused = res_ready;
        if (fused) // the rounding-floor clause needs ||x|| of the pairs near the shift: with one of them still active the separate pass runs
            for (const uint32_t i : res_act) fused = fused && !(std::abs(theta[i]) < 10.0 * std::abs(sigma));
        res_ready = false;
        rw_from_rr = fused;
        if (fused) {
            res_norms_d.download(res_norms.data(), size_t(2) * res_pitch);
            for (size_t k = 0; k < res_act.size(); ++k) {
                rn[res_act[k]] = res_norms[k];
                mn[res_act[k]] = res_norms[res_pitch + k];
            }
        } else {
        theta_d.upload(theta.data(), b);
        {
            const uint32_t rpb = 256, nblk = div_up(n, rpb);
            if (scratch.count < size_t(nblk) * 3 * b) scratch.reset(ctx, size_t(nblk) * 3 * b);
            dim3 grid(nblk, div_up(b, 64));
            k_residual_norms<<<grid, 64, 0, st>>>(AX, MX, X, theta_d, 10.0 * std::abs(sigma), R, n, b, rpb, scratch, scaled_norms ? sys->L2.dinv.get() : nullptr);
            KERNEL_CHECK();
            k_colsumsq_final<<<3 * b, 256, 0, st>>>(scratch, nblk, 3 * b, norms_d); // partial rows are 3b wide
            KERNEL_CHECK();
            norms_d.download(norms.data(), 3 * size_t(b));
            std::copy(norms.begin(), norms.begin() + b, rn.begin());
            std::copy(norms.begin() + b, norms.begin() + 2 * b, mn.begin());
            std::copy(norms.begin() + 2 * b, norms.end(), xn.begin());
        }
        }
        act.clear();
        for (uint32_t i = 0; i < b; ++i) {
            // Converged: relative residual below tol, or at the rounding floor of forming A x (which is what
            // limits the rigid-body pairs: theta = |sigma| sits 10-12 orders below ||A||).
            const double rel = std::sqrt(rn[i]) / (std::abs(theta[i]) * std::sqrt(mn[i]));
            // The floor clause is for those pairs only (theta within 10x of |sigma|): an elastic pair of a stiff, sliver-heavy
            // mesh must not be accepted at a relative residual above the tolerance because ||A|| happens to be huge.
            const bool near_shift = std::abs(theta[i]) < 10.0 * std::abs(sigma);
            // (Jacobi-scaled norms: the rounding floor of (A x)_i is eps lmax(D^-1 A) D_ii |x|, i.e. eps lmax ||x||_D in the D^-1 norm)
            const double floor_norm = scaled_norms ? sys->L2.lmax * std::sqrt(xn[i]) : anorm * std::sqrt(xn[i]);
            // The factor: forming (A x)_i rounds to gamma_k sum_j |a_ij| |x_j| with k the row length (~100-250 entries of a P2 row), against
            // eps ||A|| ||x|| here.  The EXACT rigid-body vectors of the start block measure 37 ... 96 times eps ||A|| ||x|| on the quality-refined
            // 96 x 48 sphere (||A|| = 2e16 from one sliver; tol |sigma| lies 3e-4 / 1e-5 = 30 times BELOW that floor there); with 50 (rounds
            // 1-4) whether those pairs ever locked was decided by the rounding of the Rayleigh-Ritz step: 32 iterations with one
            // tridiagonalisation kernel, no convergence with the other two (tools/probe/sphere_iters_probe.py, profiles/r05_rigid_floor.txt).
            // A pair accepted here has an eigenvector error of 256 eps ||A|| / (lambda_7 - sigma) ~ 1e-7 at worst and an eigenvalue error of its square.
            // (it 0 of a cold start: the pairs near the shift ARE the exact rigid-body vectors of the start block -- null vectors of K whatever the
            // mesh -- and what they measure is the rounding of forming A x alone: 1 000 ... 3 000 times eps ||A|| ||x|| on a raw Delaunay fill with
            // cells at 3e-11, whose rows cancel entries of 1e18.  Left active they only collect noise from the search directions and the solve
            // never ends; the reference's factorisation hands the same six back at +-1e2 beside elastic values of 2e7)
            const double floor_factor = it == 0 && !warm && sys->worst_quality < kFlatShape ? 65536.0 : 256.0;
            const bool ok = rel < residual_tol || (near_shift && std::sqrt(rn[i]) < floor_factor * 2.2e-16 * floor_norm);
            if (ok) locked[i] = 1;
            if (!locked[i]) act.push_back(i);
        }
        // the nev smallest Ritz values must all belong to converged columns
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) { return theta[a] < theta[c]; });
        nconv = 0;
        for (uint32_t k = 0; k < nev; ++k) nconv += locked[order[k]];
        if (progress) *progress = 0.3f + 0.65f * float(nconv) / float(nev);
        if (verbose) {
            double worst = 0;
            for (uint32_t k = 0; k < nev; ++k) { const uint32_t i = order[k]; worst = std::max(worst, std::sqrt(rn[i]) / (std::abs(theta[i]) * std::sqrt(mn[i]))); }
            if (it == 0) fprintf(stderr, "[lobpcg] ||A|| <= %.3e, sliver patches %u + %u clusters (largest %u nodes; worst element shape %.2e), aggregates %u, lmax %.4f / %.4f\n", anorm, sys->patches2.n_patches, sys->patches2.n_clusters, sys->patches2.largest_cluster, double(sys->worst_quality), sys->n_agg, sys->L2.lmax, sys->L1.lmax);
            fprintf(stderr, "[lobpcg] it %3u conv %3u/%u active %zu wp %u worst %.2e  floor-ratio[0..7]:", it, nconv, nev, act.size(), wp, worst);
            for (uint32_t i = 0; i < std::min(8u, b); ++i) fprintf(stderr, " %.1f", xn[i] > 0 ? std::sqrt(rn[i]) / (2.2e-16 * (scaled_norms ? sys->L2.lmax : anorm) * std::sqrt(xn[i])) : 0.0);
            fprintf(stderr, "  theta0 %.6e\n", theta[0]);
        }
        iters = it;
        if (nconv >= nev || act.empty()) { converged = nconv >= nev; return true; }
        if (it == max_iters) return true;
        // Safety net of the single-precision smoothers: no newly converged pair and no 20 % drop of the worst
        // residual over 6 iterations switches the cycle to double precision for the rest of the solve.
        {
            double worst = 0;
            for (uint32_t k = 0; k < nev; ++k) { const uint32_t i = order[k]; worst = std::max(worst, std::sqrt(rn[i]) / (std::abs(theta[i]) * std::sqrt(mn[i]))); }
            hist_worst.push_back(worst);
            hist_nconv.push_back(nconv);
            const size_t h = hist_worst.size();
            // against the state six iterations ago (the first iterations are not monotone: the residuals of a
            // random block first rise)
            if (prec32 && it >= 10 && h > 6 && hist_nconv[h - 7] == nconv && worst > 0.5 * hist_worst[h - 7]) {
                if (verbose) fprintf(stderr, "[lobpcg] it %3u stagnation: switching the preconditioner to double precision\n", it);
                prec32.reset();
                prec64 = std::make_unique<Precond<double>>(sys, std::min(b, kPrecondColumns));
            }
            // A tolerance below what the mesh admits (the rounding floor of forming A x on a sliver-heavy fill: 1e-9 on the repaired
            // scan) leaves the last active columns at their floor; W then lies in span(X, P) to rounding, the Cholesky-QR spread grows
            // and the residuals of those columns start to GROW, doubling per iteration (tools/probe/tight_tolerance_probe.py).  Say so
            // instead of iterating into a rank failure.
            double worst_active = 0;
            for (uint32_t k = 0; k < nev; ++k) { // (the wanted pairs only: guard columns converge late by design)
                const uint32_t i = order[k];
                if (!locked[i]) worst_active = std::max(worst_active, std::sqrt(rn[i]) / (std::abs(theta[i]) * std::sqrt(mn[i])));
            }
            best_worst_active = std::min(best_worst_active, worst_active);
            // (three iterations in a row: one step's jump is also what a guard column looks like when its Ritz value drops into the
            // wanted range late -- a missed member of a multiplet arrives with a residual of 1e-2 .. 1e-1 beside pairs just above the tolerance)
            // The last resort on a mesh at the edge of double precision (raw Delaunay fills, cells at 1e-8 .. 1e-9, ||A|| / theta ~ 1e9): the last
            // few pairs stop a factor below ten above the tolerance and stay there -- the floor of forming A x.  The reference's factorisation
            // returns its pairs on such a mesh whatever their residuals are; an empty result for a residual of 1.8e-4 against 1e-4 (eigenvalue error
            // ~ its square) would be the worse answer.  Thirty iterations without a newly converged pair and without a tenth off the worst
            // residual, that residual within ten times the tolerance: the pairs are handed on, counted in mh_profile.pairs_at_floor and said so.
            hist_active.push_back(worst_active);
            if (inner_cg > 0 && it >= 40) {
                const size_t h = hist_active.size();
                if (h > 30 && hist_nconv[h - 31] == nconv && worst_active > 0.9 * hist_active[h - 31] && worst_active <= 10.0 * residual_tol) {
                    pairs_at_floor = nev > nconv ? nev - nconv : 0;
                    if (verbose) fprintf(stderr, "[lobpcg] it %3u: %u pairs have sat at a residual of %.2e (tolerance %.1e) for thirty iterations -- the rounding floor of this mesh; handed on as they are\n", it, pairs_at_floor, worst_active, residual_tol);
                    converged = true;
                    return true;
                }
            }
            floor_strikes = it >= 20 && worst_active > 1e3 * best_worst_active ? floor_strikes + 1 : 0;
            if (floor_strikes >= 3)
                mh_throw(MH_ENOTCONVERGED, "LOBPCG: %u of %u pairs converged; the others sit at the rounding floor (residual %.1e and growing, best %.1e, tolerance %.1e)", nconv, nev,
                         worst_active, best_worst_active, residual_tol);
        }
        return false;
    }

    // W = preconditioned residuals of the active columns, projected against X (and P) and M-orthonormalised, with its images A W, M W.
    void search_directions(uint32_t it) {
        w = uint32_t(act.size());
        idx_d.upload(act.data(), w);
        const double *residuals = Rw.get();
        if (rw_from_rr) { // the epilogue's compact panel holds the previous active set; today's is a subset of it
            if (act == res_act && res_pitch == w) {
                residuals = Rr.get(); // nothing was locked and the pitch is the count: the panel is the input as it stands
            } else {
                std::vector<uint32_t> &pos = res_pos;
                pos.assign(w, 0);
                size_t q = 0;
                for (uint32_t k = 0; k < w; ++k) {
                    while (q < res_act.size() && res_act[q] != act[k]) ++q;
                    if (q == res_act.size()) mh_throw(MH_EHIP, "active column %u was not active before", act[k]);
                    pos[k] = uint32_t(q);
                }
                pos_d.upload(pos.data(), w);
                k_gather_cols<<<grid1(n * w), TB, 0, st>>>(Rr, pos_d, Rw, n, res_pitch, w);
                KERNEL_CHECK();
            }
        } else {
            k_gather_cols<<<grid1(n * w), TB, 0, st>>>(R, idx_d, Rw, n, b, w);
            KERNEL_CHECK();
        }
        // The Rayleigh-Ritz basis is [X_active W P]; the active columns of X, A X, M X are addressed in place through
        // the index list idx_d (column maps of the Gram and basis-update kernels), never copied out.
        for (uint32_t k = 0; k < w; ++k) theta_act[k] = theta[act[k]];
        theta_act_d.upload(theta_act.data(), w);
        {
            Timer tp(ctx);
            precondition(residuals, W, w);
            precond_seconds += tp.stop();
            prof.op_applications += w;
        }
        // W <- (I - X X^T M - P P^T M) W (the coefficients come from M X and M P: no M W needed yet), then A W and M W in
        // one fused product, then M-orthonormalise W carrying both images along.  Against "M W first, project W and
        // M W, orthonormalise, then A W" this is one basis-update launch and one pass over the matrix fewer per iteration.
        bool ok = true;
        // One projection + Cholesky-QR pass suffices: the Rayleigh-Ritz step solves the full pencil (gA, gM), so the
        // basis only has to be well conditioned, not orthonormal to working precision.
        p_implicit = false;
        if (pproj_ok && wp) {
            // project against X in the tall space, against P in coefficient space
            gram(ctx, n, MX, b, W, w, H, b); // b x w
            mh_pack_stacked(ctx, H, b, nullptr, 0, w, -1.0, Ct);
            mh_combine(ctx, n, X, b, nullptr, 0, nullptr, 0, Ct, w, W, w, nullptr, true);
            mh_spmm(ctx, sys->L2, sys->L2.aval, W, AW, sys->L2.mval, MW, w);
            gram(ctx, n, P, wp, MW, w, Hp, wp); // Hp = P^T M W, wp x w
            lazy_images = true;
            w_implicit = false;
            ok = chol_orthonormalise(W, MW, AW, w, false, true, Hp, wp);
            if (!ok) mh_throw(MH_ENOTCONVERGED, "search directions lost rank at iteration %u", it);
            if (p_needs_explicit) { // rare: W - P Hp nearly dependent -> explicit projection, images again, ordinary Cholesky-QR
                mh_pack_stacked(ctx, Hp, wp, nullptr, 0, w, -1.0, Ct);
                mh_combine(ctx, n, P, wp, nullptr, 0, nullptr, 0, Ct, w, W, w, nullptr, true);
                mh_spmm(ctx, sys->L2, sys->L2.aval, W, AW, sys->L2.mval, MW, w);
                lazy_images = w <= 256;
                ok = chol_orthonormalise(W, MW, AW, w, !lazy_images, lazy_images);
                if (!ok) mh_throw(MH_ENOTCONVERGED, "search directions lost rank at iteration %u", it);
            } else {
                p_implicit = true;
            }
        } else { // no previous directions yet, or a block wider than 128 columns (M P is maintained there: update_basis)
            gram(ctx, n, MX, b, W, w, H, b); // b x w
            if (wp) gram(ctx, n, MP, wp, W, w, H2, wp);
            mh_pack_stacked(ctx, H, b, H2, wp, w, -1.0, Ct);
            mh_combine(ctx, n, X, b, P, wp, nullptr, 0, Ct, w, W, w, nullptr, true);
            mh_spmm(ctx, sys->L2, sys->L2.aval, W, AW, sys->L2.mval, MW, w);
            // Only W itself is multiplied by L^-T.  Its images keep their pre-orthonormalisation form (A W L^T, M W L^T): the
            // Gram blocks that involve them are corrected on the small matrices (B <- B L^-T), the recombination of M P folds
            // L^-T into its coefficient rows, and A X, M X are recomputed from the new Ritz vectors anyway -- two tall
            // basis-update launches fewer per iteration.
            lazy_images = w <= 256;
            w_implicit = false;
            ok = chol_orthonormalise(W, MW, AW, w, !lazy_images, lazy_images);
            if (!ok) mh_throw(MH_ENOTCONVERGED, "search directions lost rank at iteration %u", it);
        }
    }

    // The Gram matrices of [X_active W P] (most blocks known by construction), the small eigenproblem, the new Ritz values.
    void rayleigh_ritz(uint32_t it) {
        // Gram matrices of S = [X_active W P] (lower triangles), X block known: diag(theta) and I.  Locked columns are
        // not part of the basis any more (W was projected against them above): the small problem has order
        // 2w + wp instead of b + w + wp.
        wa = w;
        m = wa + w + wp;
        for (int attempt = 0; attempt < 2; ++attempt) {
            HIP_CHECK(hipMemsetAsync(gA, 0, size_t(m) * m * sizeof(double), st));
            HIP_CHECK(hipMemsetAsync(gM, 0, size_t(m) * m * sizeof(double), st));
            k_set_identity_blocks<<<grid1(wa), TB, 0, st>>>(gA, gM, theta_act_d, wa, m);
            KERNEL_CHECK();
            // W^T M X and P^T M W are zero by the projection that W just went through (measured 1e-14 .. 1e-12 in every run): they
            // are not formed.  W^T M W, which carries the Cholesky-QR's error, is measured unless the factor says it is tiny.
            const double unit_one = 1;
            auto left_corrected = [&](double *block, uint32_t cols) { // block (w x cols at leading dimension m) <- L^-1 block: W was not transformed
                if (w_implicit)
                    ROCBLAS_CHECK(rocblas_dtrmm(ctx->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, rocblas_int(w),
                                                rocblas_int(cols), &unit_one, Linv, rocblas_int(w), block, rocblas_int(m), block, rocblas_int(m)));
            };
            mh_gram(ctx, n, W, w, AX, wa, gA.get() + wa, m, b, idx_d);
            left_corrected(gA.get() + wa, wa);
            const double unit = 1;
            auto untransformed = [&](double *block, uint32_t rows) { // block (rows x w at leading dimension m) <- block L^-T
                if (lazy_images)
                    ROCBLAS_CHECK(rocblas_dtrmm(ctx->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, rocblas_int(rows),
                                                rocblas_int(w), &unit, Linv, rocblas_int(w), block, rocblas_int(m), block, rocblas_int(m)));
            };
            gram(ctx, n, W, w, AW, w, gA.get() + size_t(wa) * m + wa, m);
            if (p_implicit && wp) {
                // W' = W - P Hp:  P^T A W' = Bp - App Hp,  W'^T A W' = Bw - U - U^T + Hp^T App Hp with U = Hp^T Bp  (P^T A P = App, X^T A P = 0)
                const double plus = 1, nil = 0;
                double *bw = gA.get() + size_t(wa) * m + wa, *bp = gA.get() + size_t(wa) * m + wa + w;
                gram(ctx, n, P, wp, AW, w, bp, m);
                small_dgemm(ctx, rocblas_operation_transpose, rocblas_operation_none, rocblas_int(w), rocblas_int(w), rocblas_int(wp), &plus, Hp, rocblas_int(wp), bp,
                                            rocblas_int(m), &nil, Up, rocblas_int(w));
                small_dgemm(ctx, rocblas_operation_none, rocblas_operation_none, rocblas_int(wp), rocblas_int(w), rocblas_int(wp), &plus, App, rocblas_int(wp), Hp,
                                            rocblas_int(wp), &nil, T1p, rocblas_int(wp));
                small_dgemm(ctx, rocblas_operation_transpose, rocblas_operation_none, rocblas_int(w), rocblas_int(w), rocblas_int(wp), &plus, Hp, rocblas_int(wp), T1p,
                                            rocblas_int(wp), &nil, Vp, rocblas_int(w));
                k_sub_block<<<grid1(size_t(wp) * w), TB, 0, st>>>(bp, m, T1p, wp, wp, w); // Bp -= App Hp
                k_wblock_fix<<<grid1(size_t(w) * w), TB, 0, st>>>(bw, m, Up, Vp, w);
                KERNEL_CHECK();
            }
            untransformed(gA.get() + size_t(wa) * m + wa, w);
            left_corrected(gA.get() + size_t(wa) * m + wa, w);
            // W^T M W after the Cholesky-QR deviates from I by about eps * cond(G); it is measured unless the factor's diagonal
            // says cond(G) < 2^16 (deviation ~1e-11)
            const bool w_block_trusted = last_spread < 16 * 8 && !(p_implicit && wp == 0); // (a retry without P after an implicit P-projection: measure)
            gm_identity = w_block_trusted; // every block of gM was set, not measured: gM0 is the identity exactly
            if (w_block_trusted) {
                k_place_block<<<grid1(size_t(w) * w), TB, 0, st>>>(gM.get() + size_t(wa) * m + wa, m, nullptr, w);
                KERNEL_CHECK();
            } else {
                gram(ctx, n, W, w, MW, w, gM.get() + size_t(wa) * m + wa, m);
                untransformed(gM.get() + size_t(wa) * m + wa, w);
            }
            if (verbose) fprintf(stderr, "[lobpcg] it %3u Cholesky-QR diagonal spread 2^%.1f%s\n", it, last_spread / 16.0, w_block_trusted ? "" : " (W block measured)");
            if (wp) {
                if (!p_implicit) gram(ctx, n, P, wp, AW, w, gA.get() + size_t(wa) * m + wa + w, m); // (already formed and corrected above otherwise)
                untransformed(gA.get() + size_t(wa) * m + wa + w, wp);
                {
                    // P = S_prev Cp with Cp gM-orthonormal and gM-orthogonal to the Ritz coefficients Cx, and
                    // gA Cx = gM Cx Theta: hence P^T M P = I, P^T M X = P^T A X = 0 and P^T A P = Cp^T gA_prev Cp
                    // (formed last iteration from the small matrices) -- four tall Gram products saved.
                    k_place_block<<<grid1(size_t(wp) * wp), TB, 0, st>>>(gA.get() + size_t(wa + w) * m + wa + w, m, App, wp);
                    k_place_block<<<grid1(size_t(wp) * wp), TB, 0, st>>>(gM.get() + size_t(wa + w) * m + wa + w, m, nullptr, wp);
                    KERNEL_CHECK();
                }
            }
            k_symmetrize_lower<<<grid1(size_t(m) * m), TB, 0, st>>>(gA, m, m);
            KERNEL_CHECK();
            HIP_CHECK(hipMemcpyAsync(gA0, gA, size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_CHECK(hipMemcpyAsync(gM0, gM, size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, st));
            rr_evals.clear();
            const int hinfo = rr_solve(ctx, gA, gM, m, evals, ework, info, wa, gm_identity, &rr_evals);
            if (hinfo == 0) break;
            if (attempt == 1 || wp == 0) mh_throw(MH_ENOTCONVERGED, "Rayleigh-Ritz failed at iteration %u (info %d)", it, hinfo);
            wp = 0; // drop the previous directions and retry on [X W]
            m = wa + w;
        }
        if (rr_evals.size() == wa) std::copy(rr_evals.begin(), rr_evals.end(), theta_act.begin()); // (came back with the step's quality word)
        else evals.download(theta_act.data(), wa); // ascending Ritz values into the (ascending) active slots
        for (uint32_t k = 0; k < wa; ++k) theta[act[k]] = theta_act[k];
    }

    // The next conjugate directions in coefficient space (Cp), and P^T A P for the next iteration.
    void conjugate_directions(uint32_t it) {
        // New directions in coefficient space: the [W P] part of the active Ritz vectors, made
        // gM-orthonormal against the new X coefficients and among themselves.
        const double one = 1, zero = 0, mone = -1;
        wp_new = w;
        k_build_cp<<<grid1(size_t(m) * w), TB, 0, st>>>(gA, nullptr, wa, m, w, m, Cp, m);
        KERNEL_CHECK();
        // (gM0 Cp is Cp itself when gM0 is the identity by construction: the two symmetric products are skipped)
        const double *mcp = gm_identity ? Cp.get() : T1.get();
        if (!gm_identity) ROCBLAS_CHECK(rocblas_dsymm(ctx->blas, rocblas_side_left, rocblas_fill_lower, m, w, &one, gM0, m, Cp, m, &zero, T1, m));
        small_dgemm(ctx, rocblas_operation_transpose, rocblas_operation_none, wa, w, m, &one, gA, m, mcp, m, &zero, H, wa);
        small_dgemm(ctx, rocblas_operation_none, rocblas_operation_none, m, w, wa, &mone, gA, m, H, wa, &one, Cp, m);
        if (!gm_identity) ROCBLAS_CHECK(rocblas_dsymm(ctx->blas, rocblas_side_left, rocblas_fill_lower, m, w, &one, gM0, m, Cp, m, &zero, T1, m));
        small_dgemm(ctx, rocblas_operation_transpose, rocblas_operation_none, w, w, m, &one, Cp, m, mcp, m, &zero, G, w);
        k_scale_gram<<<grid1(size_t(w) * w), TB, 0, st>>>(G, w, w, dscale);
        KERNEL_CHECK();
        {
            double *Gs = G.get() + size_t(w) * w;
            int hinfo = 0;
            // Up to 128 columns: factor, unscale and invert in one launch, then Cp <- Cp L^-T as one small product (round 5; before: potrf,
            // unscale and the library's trsm, which is a trtri and several GEMM launches of its own -- a dozen launches on the serial path of
            // every iteration).  MH_TEST=potrf_chain: the old chain.
            static const bool chain = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "potrf_chain");
            const bool fused = w <= 128 && !chain;
            {
                if (fused) mh_potrf_small_inverse(ctx, Gs, w, info, dscale, Linv_p);
                else mh_potrf(ctx, Gs, w, w, info); // ours at every order (one workgroup up to 128 columns, 128-column blocks above)
                info.download(&hinfo, 1);
            }
            if (hinfo != 0) {
                wp_new = 0;
            } else if (fused) {
                mh_small_gemm(ctx, false, true, m, w, w, 1.0, Cp, m, Linv_p, w, 0.0, T1p2, m);
                HIP_CHECK(hipMemcpyAsync(Cp, T1p2, size_t(m) * w * sizeof(double), hipMemcpyDeviceToDevice, st));
            } else {
                k_unscale_chol<<<grid1(size_t(w) * w), TB, 0, st>>>(Gs, w, w, dscale);
                KERNEL_CHECK();
                ROCBLAS_CHECK(rocblas_dtrsm(ctx->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, m, w, &one, Gs, w, Cp, m));
            }
        }
        // No conjugate directions in the first iterations of a cold start, while the block is far from the invariant
        // subspace: with residuals of order one the previous step carries no usable curvature information -- the iteration
        // count is the same without it (18 and 18 at S100k) -- and an iteration on [X W] costs a third less
        // (Rayleigh-Ritz of order 2w, no P Grams, narrower updates): 224 -> 213 ms per solve.
        if (!warm && it < kSkipP && !hist_worst.empty() && hist_worst.back() > 0.5) wp_new = 0;
        if (wp_new) { // App = Cp^T gA_prev Cp for the next iteration's P-P block
            small_dgemm(ctx, rocblas_operation_none, rocblas_operation_none, m, wp_new, m, &one, gA0, m, Cp, m, &zero, T1, m);
            small_dgemm(ctx, rocblas_operation_transpose, rocblas_operation_none, wp_new, wp_new, m, &one, Cp, m, T1, m, &zero, App, wp_new);
        }
    }

    // X <- S Cx, P <- S Cp, and the images the next iteration needs.
    void update_basis(uint32_t it) {
        // X <- S Cx, P <- S Cp (and the A-, M-images): one fused MFMA launch per image
        mh_pack_coefficients(ctx, gA, wa, Cp, wp_new, m, m, Ct);
        // X_active <- S Cx (mapped columns of X), P <- S Cp.  One launch per image works in place (a workgroup reads its 64
        // rows completely before writing them); more than 256 output columns take several launches over the same
        // inputs, so those go through a contiguous copy and a scatter.
        // Images of the new block.  A X and M X are formed from the new Ritz vectors by one fused product (written straight into
        // the active columns) instead of being recombined from [A X, A W, A P] and [M X, M W, M P]: one pass over the matrix
        // costs less than two passes over three tall panels each, the images carry no accumulated rounding, and A P is
        // not needed at all (P^T A P comes from the small matrices, above).  M P, which the next projection needs, is
        // one product with M alone.
        {
            const uint32_t pitch = (wa + 1u) & ~1u; // 16-byte rows for the wide-load product
            if (w_implicit) { // the basis holds W, not W L^-T: every coefficient row of the W part <- L^-T row (all columns)
                const double unit = 1;
                double *rows = Ct.get() + size_t(wa) * (wa + wp_new);
                ROCBLAS_CHECK(rocblas_dtrmm(ctx->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, rocblas_int(wa + wp_new),
                                            rocblas_int(w), &unit, Linv, rocblas_int(w), rows, rocblas_int(wa + wp_new), rows, rocblas_int(wa + wp_new)));
            }
            if (p_implicit && wp) { // ... and it holds W, not W - P Hp: the P rows take the difference, rows_P -= Hp rows_W (all columns)
                const double minus = -1, plus = 1;
                const rocblas_int pitchc = rocblas_int(wa + wp_new);
                double *rows_w = Ct.get() + size_t(wa) * (wa + wp_new), *rows_p = Ct.get() + size_t(wa + w) * (wa + wp_new);
                ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_transpose, pitchc, rocblas_int(wp), rocblas_int(w), &minus, rows_w, pitchc, Hp,
                                            rocblas_int(wp), &plus, rows_p, pitchc));
            }
            mh_combine(ctx, n, X, wa, W, w, P, wp, Ct, wa + wp_new, Xn, wa, Pn, false, b, idx_d, pitch);
            // M P_new (kept for blocks wider than 128 columns: the next projection needs it) from P_new itself: one pass over M.  Until
            // round 4 it was recombined from [M X, M W, M P] with the coefficients of P_new -- three tall panels read for one written:
            // 215-pair solves 1.7 ... 2.8 % slower, 129-pair solves 1 ... 1.5 % (and the image carried the recombination's rounding).
            if (wp_new && !pproj_ok) mh_spmm(ctx, sys->L2, nullptr, Pn, nullptr, sys->L2.mval, MPn, wp_new);
            if (pitch == wa) k_scatter_cols<<<grid1(n * wa), TB, 0, st>>>(Xn.get(), idx_d, X.get(), n, b, wa);
            else k_scatter_cols_pitch<<<grid1(n * wa), TB, 0, st>>>(Xn.get(), pitch, idx_d, X.get(), n, b, wa);
            KERNEL_CHECK();
            if (Rr.count) {
                // the new Ritz values sit in `evals` in the active slots' order: the residuals and their norms leave with the images
                mh_spmm_mapped(ctx, sys->L2, sys->L2.aval, Xn, AX, sys->L2.mval, MX, pitch, b, wa, idx_d, evals, Rr, res_partial, scaled_norms ? sys->L2.dinv.get() : nullptr);
                const uint32_t rpb = 48, nblk = div_up(sys->n_nodes, rpb), w2 = 2 * pitch; // (short row blocks: ~9 k waves instead of ~2 k for the 190 MB of partials)
                dim3 grid(nblk, div_up(w2, 64));
                k_colsum_partial<<<grid, 64, 0, st>>>(res_partial, sys->n_nodes, w2, res_blocks, rpb); // rows of res_partial: [node][2][pitch]
                KERNEL_CHECK();
                k_colsumsq_final<<<w2, 256, 0, st>>>(res_blocks, nblk, w2, res_norms_d);
                KERNEL_CHECK();
                res_act = act;
                res_pitch = pitch;
                res_ready = true;
            } else {
                mh_spmm_mapped(ctx, sys->L2, sys->L2.aval, Xn, AX, sys->L2.mval, MX, pitch, b, wa, idx_d);
            }
        }
        std::swap(P, Pn); std::swap(MP, MPn);
        wp = wp_new;
    }

    void finish(double *eigenvalues) {
        prof.restarts = iters;
        prof.op_solve = precond_seconds;
        if (!converged) mh_throw(MH_ENOTCONVERGED, "LOBPCG: %u of %u pairs converged in %u iterations", nconv, nev, iters);
        for (uint32_t k = 0; k < nev; ++k) eigenvalues[k] = theta[order[k]] + sigma;
        sys->plain_residual = -1.0;
        if (scaled_norms) {
            // The pairs were accepted in the Jacobi-scaled norm (above): say what that means in the 2-norm relative residual
            // ||K x - lambda M x||_2 / (|lambda - sigma| ||M x||_2) the caller's tolerance is phrased in -- one pass, once per solve.
            theta_d.upload(theta.data(), b);
            const uint32_t rpb = 256, nblk = div_up(n, rpb);
            if (scratch.count < size_t(nblk) * 3 * b) scratch.reset(ctx, size_t(nblk) * 3 * b);
            k_residual_norms<<<dim3(nblk, div_up(b, 64)), 64, 0, st>>>(AX, MX, X, theta_d, 10.0 * std::abs(sigma), R, n, b, rpb, scratch, nullptr);
            KERNEL_CHECK();
            k_colsumsq_final<<<3 * b, 256, 0, st>>>(scratch, nblk, 3 * b, norms_d);
            KERNEL_CHECK();
            norms_d.download(norms.data(), 3 * size_t(b));
            double worst = 0;
            for (uint32_t k = 0; k < nev; ++k) {
                const uint32_t i = order[k];
                if (std::abs(theta[i]) < 10.0 * std::abs(sigma)) continue; // (the rigid-body pairs sit at the rounding floor by construction)
                worst = std::max(worst, std::sqrt(norms[i]) / (std::abs(theta[i]) * std::sqrt(norms[b + i])));
            }
            sys->plain_residual = worst;
            if (verbose) fprintf(stderr, "[lobpcg] accepted in the Jacobi-scaled norm at %.1e; worst elastic pair in the plain 2-norm: %.2e\n", residual_tol, worst);
        }
        sys->evecs.reset(ctx, n * nev);
        sys->evec_cols = nev;
        idx_d.upload(order.data(), nev);
        k_gather_cols<<<grid1(n * nev), TB, 0, st>>>(X, idx_d, sys->evecs, n, b, nev);
        KERNEL_CHECK();
        HIP_CHECK(hipStreamSynchronize(st));
        prof.iterate = t_iter.stop();
        read_health();
        sys->profile = prof;
        if (profile) *profile = prof;
        if (switches().test_selfcheck_fail && !ctx->exchange_disabled) prof.rr_selfcheck = 1.0; // (test hook: the redo in eigs_impl)
        // (1e-6, not the 1e-8 of round 5: a step whose matrix spans |sigma| .. ||A|| with an ill-conditioned block -- the first steps on two bodies joined at
        // ONE vertex, whose smoothed columns all lean on the same three hinge modes -- leaves 2e-8 honestly, and the pairs a solve returns are accepted
        // on their TRUE residuals, not on the step's; what the check is for, a wrong launch of an exchange kernel, leaves 1e-3 and more)
        if (!(prof.rr_selfcheck < 1e-6))
            mh_throw(MH_EHIP, "Rayleigh-Ritz self-check failed: a step's eigenpairs leave a relative residual of %.2e against the step's own matrix", prof.rr_selfcheck);
    }
    // the solve's health counters: redone Rayleigh-Ritz steps and the worst sampled self-check residual (k_rr_selfcheck) since start()
    void read_health() {
        prof.sytrd_redos = ctx->sytrd_redos - redos_at_start;
        prof.pairs_at_floor = pairs_at_floor;
        prof.rr_selfcheck = 0.0;
        if (ctx->rr_check) {
            unsigned long long bits = 0;
            HIP_CHECK(hipMemcpyAsync(&bits, ctx->rr_check, sizeof(bits), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            memcpy(&prof.rr_selfcheck, &bits, sizeof(double));
        }
    }

    void run(double *eigenvalues) {
        start();
        for (uint32_t it = 0; it <= max_iters; ++it) {
            if (converged_or_locked(it)) break;
            search_directions(it);
            rayleigh_ritz(it);
            conjugate_directions(it);
            update_basis(it);
        }
        finish(eigenvalues);
    }
};
} // namespace

static void eigs_impl(mh_system *sys, uint32_t nev, double sigma, double residual_tol, uint32_t max_iters, const float *seed_basis, uint32_t seed_rows,
                      uint32_t seed_cols, const volatile unsigned char *cancel, volatile float *progress, double *eigenvalues, mh_profile *profile) {
    {
        mh_context *ctx = sys->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = size_t(3) * sys->n_nodes;
        if (nev >= n) mh_throw(MH_EINVAL, "nev %u must be below the %zu unknowns", nev, n);
        if (!(sigma < 0)) mh_throw(MH_EFACTOR, "shift must be negative for a positive-definite shifted operator");
        // A mesh point that no kept tetrahedron uses has empty rows in K and M: K - sigma M is singular there and the reference's
        // Cholesky factorisation fails ("Modal shift-invert factorization failed.", CholeskyShiftInvert.cpp:44) -- so does this solve.
        if (sys->unreferenced_points) mh_throw(MH_EFACTOR, "%u mesh point(s) belong to no tetrahedron: the shifted operator is singular", sys->unreferenced_points);
        mh_profile prof = sys->profile;
        prof.dofs = uint32_t(n);
        // the reference counts the lower triangle of K (Eigen nonZeros of the lower-stored matrix, mesh2modes.cpp:615)
        prof.stiffness_nonzeros = uint32_t((sys->L2.n_blocks - sys->n_nodes) / 2 * 9 + uint64_t(6) * sys->n_nodes);
        if (cancel && *cancel) mh_throw(MH_ECANCELLED, "cancelled");
        // guard vectors: at least 15 (measured at S100k, nev = 65: 10 -> 23 iterations / 338 ms, 15 -> 19 / 317 ms,
        // 31 -> 15 / 338 ms), block rounded up to whole 16-column MFMA tiles
        // (every further disconnected body brings six more zero modes: the block grows by as many columns)
        // (215 pairs: 240 columns -> 23 / 24 / 21 iterations on the three 215-pair workloads, 256 columns -> 21 / 21 / 19 but +2 ... +4 % time)
        uint32_t guards = std::max(15u, nev * kGuardPercent / 100);
        if (const char *g = getenv("MH_GUARDS")) guards = uint32_t(std::max(1, atoi(g))); // (A/B hook: tools/probe/guard_sweep.sh)
        uint32_t b = (nev + guards + 6u * (sys->n_components - 1u) + 15u) / 16u * 16u;
        if (n <= 768 || n < size_t(5) * b) {
            Timer t(ctx);
            dense_eigs(sys, nev, sigma, eigenvalues);
            prof.iterate = t.stop();
            prof.restarts = 1;
            sys->profile = prof;
            if (profile) *profile = prof;
        } else {
            // Solves of any block width share the device with other contexts' work.  (Rounds 2-4 ran blocks wider than 128 columns alone, under
            // an exclusive process-wide lock: other contexts' solves broke beside them.  Round 5 found the cause -- not the wide path
            // itself but what ran beside it: rocBLAS's LDS-bound dsymm kernel, which only wide blocks call, on the same CU as a workgroup
            // of k_sytrd_multi, whose barrier at the top of the column loop hipcc had left without its LDS wait; mh_common.h:
            // mh_lds_writes_landed, DESIGN.md section 6.  With the wait in place the lock is gone.)
            const float *seed = seed_basis; // (a warm start that fails is retried cold: build_and_solve)
            uint32_t srows = seed_rows, scols = seed_cols;
            const auto build_and_solve = [&] {
            {
                Timer t(ctx);
                mh_build_hierarchy(sys, sigma, true); // (the coarse elimination may still run: the first preconditioner application waits for it)
                prof.factorize = t.stop();
            }
            if (progress) *progress = 0.3f;
            try {
                try {
                    try {
                        if (switches().test_last_resort) mh_throw(MH_ENOTCONVERGED, "MH_TEST=last_resort");
                        try {
                            BlockLobpcg solver(sys, nev, b, sigma, residual_tol, max_iters, seed, srows, scols, cancel, progress, prof, profile);
                            solver.run(eigenvalues);
                        } catch (const MhError &e) {
                            // A seeded basis that does not serve -- its columns in another order than the solve left them (the exact rigid-body vectors
                            // then stand beside their seeded copies: a rank-deficient block), zeros, a NaN, one column forty-five times -- is no
                            // reason to go through the fall-backs below with it, or to return nothing: the reference's SubspaceIterate takes what
                            // it is given.  Once more from a cold start, and the fall-backs after that are cold as well.
                            if (!seed || e.code != MH_ENOTCONVERGED || max_iters < 50) throw;
                            if (switches().verbose) fprintf(stderr, "[lobpcg] %s -- the seeded basis did not serve: once more from a cold start\n", e.what());
                            seed = nullptr, srows = 0, scols = 0;
                            prof = sys->profile;
                            prof.dofs = uint32_t(n);
                            BlockLobpcg solver(sys, nev, b, sigma, residual_tol, max_iters, seed, srows, scols, cancel, progress, prof, profile);
                            solver.run(eigenvalues);
                        }
                    } catch (const MhError &e) {
                        // A Rayleigh-Ritz step whose eigenpairs do not fit the step's own matrix (k_rr_selfcheck: sampled per step, read once at
                        // the end) means a dense kernel delivered wrong numbers -- in rounds 2-4 the tagged exchange beside an LDS-bound
                        // neighbour.  Rather than hand the caller MH_EHIP after a whole solve's time: once more with the
                        // exchange kernel out of the path (orders 257-768 go to the library's syevd); the context remembers.
                        if (e.code != MH_EHIP || !strstr(e.what(), "self-check") || ctx->exchange_disabled) throw;
                        if (switches().verbose) fprintf(stderr, "[lobpcg] %s -- once more without the exchange kernels\n", e.what());
                        ctx->exchange_disabled = true;
                        prof = sys->profile;
                        prof.dofs = uint32_t(n);
                        BlockLobpcg solver(sys, nev, b, sigma, residual_tol, max_iters, seed, srows, scols, cancel, progress, prof, profile);
                        solver.run(eigenvalues);
                    }
                } catch (const MhError &e) {
                    // A Chebyshev smoother whose interval ends below lmax(D^-1 A) amplifies the top of the spectrum instead of damping it -- by
                    // T_deg(1 + 2 x overshoot): twelve-fold per smoothing at degree 16 for 2 %.  The bound is a power-iteration estimate times 1.1
                    // (measured margin on the workloads: 6-9 %); should it fall short on some mesh, the iteration stalls or loses rank.  One retry
                    // with both levels' bounds widened by a quarter (costs the smoothers a few per cent of their efficiency, nothing else).
                    // (not for a mesh with flat cells: what fails there is not the bound -- the last resort below is next)
                    if (e.code != MH_ENOTCONVERGED || max_iters < 50 || sys->lmax_widened || switches().test_last_resort || sys->worst_quality < kFlatShape) throw;
                    if (switches().verbose) fprintf(stderr, "[lobpcg] %s -- once more with the smoothers' spectral bounds widened by 25 %%\n", e.what());
                    sys->L1.lmax *= 1.25;
                    sys->L2.lmax *= 1.25;
                    sys->lmax_widened = true;
                    BlockLobpcg solver(sys, nev, b, sigma, residual_tol, max_iters, seed, srows, scols, cancel, progress, prof, profile);
                    solver.run(eigenvalues);
                }
            } catch (const MhError &e) {
                // Last resort of a LARGER system (round 6): the block iteration once more with (nearly) the reference's own search directions --
                // A^-1 r by eight conjugate-gradient steps around the double-precision cycle instead of the cycle alone (PanelCg).  Ten times
                // the cost per iteration, a handful of iterations; reached only by meshes the two attempts above gave up on (cells flat to
                // 1e-9 in a caller's own mesh: the front end's fills have none since round 6).  MH_ENOTCONVERGED is what is left when this
                // fails too.  MH_TEST=last_resort sends every solve here (tests).
                if (e.code == MH_ENOTCONVERGED && n > kDenseLastResort && max_iters >= 50) {
                    if (switches().verbose) fprintf(stderr, "[lobpcg] %s -- last resort: conjugate-gradient search directions\n", e.what());
                    prof = sys->profile;
                    prof.dofs = uint32_t(n);
                    // (with the smoothers' bounds a quarter wider, if they are not already: the conjugate gradients around the cycle need it to be
                    // positive definite, i.e. no end of the spectrum above the bound -- on the 128 x 64 sphere's flat fill 19 iterations with the
                    // wider bounds against 59 without)
                    if (!sys->lmax_widened) {
                        sys->L1.lmax *= 1.25;
                        sys->L2.lmax *= 1.25;
                        sys->lmax_widened = true;
                    }
                    // (eight steps first; a mesh they do not do -- ||A|| / theta ~ 1e11: the cycle is a poor preconditioner of the flat cells' rows -- gets
                    // forty: each outer iteration is then nearly an exact inverse iteration, at fifty cycles' cost)
                    static const int first_steps = getenv("MH_LAST_RESORT_CG") ? std::max(1, atoi(getenv("MH_LAST_RESORT_CG"))) : 8;
                    for (const int steps : {first_steps, 5 * first_steps}) {
                        try {
                            BlockLobpcg solver(sys, nev, b, sigma, residual_tol, steps == first_steps ? max_iters : std::min<uint32_t>(max_iters, 120), seed, srows, scols, cancel, progress, prof, profile, steps);
                            solver.run(eigenvalues);
                            return;
                        } catch (const MhError &again) {
                            if (again.code != MH_ENOTCONVERGED) throw;
                            if (steps == first_steps) {
                                if (switches().verbose) fprintf(stderr, "[lobpcg] the last resort with %d conjugate-gradient steps: %s -- once more with %d\n", steps, again.what(), 5 * steps);
                                prof = sys->profile;
                                prof.dofs = uint32_t(n);
                                continue;
                            }
                            mh_throw(MH_ENOTCONVERGED, "%s (the last resort too: %s)", e.what(), again.what()); // what the caller reads first is why the solve itself stopped
                        }
                    }
                    return;
                }
                // Last resort of a SMALL system whose iteration stalled (measured: a UV sphere's surface filled without interior
                // points -- a quarter of the tetrahedra flat to 1e-8, ||A|| / theta ~ 1e13): one dense eigensolve in the inverse
                // form.  O(n^3), seconds at the size limit: better than no modes for an editor primitive.
                // (not when the caller's own iteration limit is what stopped it: MaxRestarts exceeded stays the reference's empty result)
                if (e.code != MH_ENOTCONVERGED || n > kDenseLastResort || max_iters < 50) throw;
                if (switches().verbose) fprintf(stderr, "[lobpcg] %s -- dense eigensolve of order %zu instead\n", e.what(), n);
                Timer t(ctx);
                dense_eigs(sys, nev, sigma, eigenvalues, true);
                prof = sys->profile;
                prof.dofs = uint32_t(n);
                prof.iterate += t.stop();
                prof.restarts = max_iters + 1;
                sys->profile = prof;
                if (profile) *profile = prof;
            }
            };
            // The shift is negative (checked above), so K - sigma M IS positive definite and so is every Galerkin coarse operator of it in exact
            // arithmetic: a coarse elimination that meets a non-positive pivot has lost it to rounding (flat cells: entries of 1e17 cancelling).
            // That is not the caller's "factorization failed": the diagonal lift of the coarse operator goes up a thousandfold, twice at most.
            for (int lifted = 0;; ++lifted) {
                try {
                    build_and_solve();
                    break;
                } catch (const MhError &e) {
                    if (e.code != MH_EFACTOR || !strstr(e.what(), "coarse operator") || lifted >= 2) throw;
                    sys->coarse_lift = (sys->coarse_lift > 0 ? sys->coarse_lift : 1e-9) * 1e3;
                    sys->hierarchy_ready = false;
                    if (switches().verbose) fprintf(stderr, "[lobpcg] %s -- the coarse operator's diagonal lifted by %.0e, once more\n", e.what(), sys->coarse_lift);
                }
            }
        }
    }
}

// ---- the shift-invert operator of the reference as an operation: x = (K - sigma M)^-1 b ---------------------------------------------
// The reference offers its Cholesky shift-invert to Spectra as an operator concept (src/audio/CholeskyShiftInvert.h:11-30: rows, cols,
// set_shift, perform_op, solve_panel).  There is no factorisation here; the same operation is served by preconditioned conjugate
// gradients on the panel -- w independent runs in lockstep, the eigensolver's three-level cycle (double-precision smoothers) as the
// preconditioner -- to a relative residual rel_tol per column.  A caller that drives its own Lanczos through this pays a full iterative
// solve per application (~25 cycle applications for 1e-11): it exists for interface completeness, mh_eigs is the fast path.

// b, x: n x w row-major panels in the internal numbering.  Returns the iterations taken; *worst_rel = the worst column's ||b - A x|| / ||b||.
uint32_t mh_shift_invert_panel(mh_system *sys, double sigma, const double *b, double *x, uint32_t w, double rel_tol, uint32_t max_iters, double *worst_rel) {
    mh_context *ctx = sys->ctx;
    hipStream_t st = ctx->stream;
    const size_t n = size_t(3) * sys->n_nodes;
    if (!(sigma < 0)) mh_throw(MH_EFACTOR, "shift-invert: the shift must be negative (K - sigma M positive definite)");
    if (w == 0 || w > 64) mh_throw(MH_EINVAL, "shift-invert panel: %u columns outside 1..64", w);
    std::unique_lock<std::mutex> one_solve_at_a_time(g_solve_mutex, std::defer_lock); // MH_CONCURRENT_SOLVES=0 serialises this entry point like mh_eigs
    if (!g_concurrent) one_solve_at_a_time.lock();
    SharedPhase solving(ctx->device);
    mh_build_hierarchy(sys, sigma); // (finished: the coarse inverse is waited for)
    Precond<double> prec(sys, w);
    DevArray<double> r(ctx, n * w), z(ctx, n * w), p(ctx, n * w), ap(ctx, n * w), rz(ctx, w), rz_new(ctx, w), pap(ctx, w), rn(ctx, w), scratch;
    const uint32_t rpb = 256, nb = uint32_t(div_up(n, rpb));
    scratch.reset(ctx, size_t(nb) * w);
    const dim3 grid(nb, div_up(w, 64));
    auto dot = [&](const double *u, const double *v, double *out) {
        k_coldot_partial<<<grid, 64, 0, st>>>(u, v, n, w, scratch, rpb);
        KERNEL_CHECK();
        k_colsumsq_final<<<w, 256, 0, st>>>(scratch, nb, w, out);
        KERNEL_CHECK();
    };
    HIP_CHECK(hipMemsetAsync(x, 0, n * w * sizeof(double), st));
    std::vector<double> bn(w), rnh(w);
    dot(b, b, rn);
    rn.download(bn.data(), w);
    auto worst_of = [&] {
        double worst = 0;
        for (uint32_t c = 0; c < w; ++c) worst = std::max(worst, bn[c] > 0 ? std::sqrt(rnh[c] / bn[c]) : 0.0);
        return worst;
    };
    uint32_t it = 0;
    double worst = 1.0, previous = 1e300;
    // Restarted from the TRUE residual: the recurrence's r drifts from b - A x by eps ||A|| ||x||, and ||x|| is large where the shifted
    // operator is nearly singular (the rigid-body components at a small |sigma|); a restart takes the drift out, and the solve ends when
    // the true residual is below the tolerance or stops improving (its floor eps ||A|| ||x|| / ||b|| -- a direct solve's residual too).
    for (int restart = 0; restart < 6 && it < max_iters; ++restart) {
        mh_spmm(ctx, sys->L2, sys->L2.aval, x, ap, nullptr, nullptr, w);
        HIP_CHECK(hipMemcpyAsync(r.get(), ap.get(), n * w * sizeof(double), hipMemcpyDeviceToDevice, st));
        k_axpy_panel<<<grid1(n * w), TB, 0, st>>>(r, b, n * w); // r <- b - A x
        KERNEL_CHECK();
        dot(r, r, rn);
        rn.download(rnh.data(), w);
        worst = worst_of();
        if (!(worst == worst)) mh_throw(MH_ENOTCONVERGED, "shift-invert: the conjugate gradients broke down (restart %d)", restart);
        if (worst <= rel_tol || worst > 0.5 * previous) break;
        previous = worst;
        prec.apply(r, z, w);
        HIP_CHECK(hipMemcpyAsync(p.get(), z.get(), n * w * sizeof(double), hipMemcpyDeviceToDevice, st));
        dot(r, z, rz);
        for (; it < max_iters;) {
            mh_spmm(ctx, sys->L2, sys->L2.aval, p, ap, nullptr, nullptr, w);
            dot(p, ap, pap);
            k_cg_advance<<<grid1(n * w), TB, 0, st>>>(x, r, p, ap, rz, pap, n * w, w);
            KERNEL_CHECK();
            dot(r, r, rn);
            rn.download(rnh.data(), w);
            ++it;
            const double rec = worst_of();
            if (!(rec == rec)) mh_throw(MH_ENOTCONVERGED, "shift-invert: the conjugate gradients broke down at iteration %u", it);
            if (rec <= 0.5 * rel_tol) break;
            prec.apply(r, z, w);
            dot(r, z, rz_new);
            k_cg_direction<<<grid1(n * w), TB, 0, st>>>(p, z, rz_new, rz, n * w, w);
            KERNEL_CHECK();
            HIP_CHECK(hipMemcpyAsync(rz.get(), rz_new.get(), w * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
    }
    // what is reported and judged is the TRUE residual of the x that is returned, not the one a restart began with (the inner loop may
    // have ended on max_iters, or the sixth restart may just have finished)
    mh_spmm(ctx, sys->L2, sys->L2.aval, x, ap, nullptr, nullptr, w);
    HIP_CHECK(hipMemcpyAsync(r.get(), ap.get(), n * w * sizeof(double), hipMemcpyDeviceToDevice, st));
    k_axpy_panel<<<grid1(n * w), TB, 0, st>>>(r, b, n * w);
    KERNEL_CHECK();
    dot(r, r, rn);
    rn.download(rnh.data(), w);
    worst = worst_of();
    if (worst_rel) *worst_rel = worst;
    // accepted: the tolerance asked for, or -- where eps ||A|| ||x|| / ||b|| lies above it (a direct solve's residual too) -- 1e-8
    if (!(worst <= std::max(rel_tol, 1e-8))) mh_throw(MH_ENOTCONVERGED, "shift-invert: relative residual %.2e after %u iterations (asked %.1e)", worst, it, rel_tol);
    HIP_CHECK(hipStreamSynchronize(st));
    return it;
}

std::mutex &mh_solve_mutex() { return g_solve_mutex; }
void mh_phase_shared_lock(int device) { if (g_concurrent) phase_of(device).lock_shared(); }
void mh_phase_shared_unlock(int device) { if (g_concurrent) phase_of(device).unlock_shared(); }

extern "C" int mh_eigs(mh_system *sys, uint32_t nev, double sigma, double residual_tol, uint32_t max_iters, const float *seed_basis, uint32_t seed_rows,
                       uint32_t seed_cols, const volatile unsigned char *cancel, volatile float *progress, double *eigenvalues, mh_profile *profile) {
    if (!sys || !eigenvalues || nev == 0) return MH_EINVAL;
    try {
        std::unique_lock<std::mutex> one_solve_at_a_time(g_solve_mutex, std::defer_lock);
        if (!g_concurrent) one_solve_at_a_time.lock();
        eigs_impl(sys, nev, sigma, residual_tol, max_iters, seed_basis, seed_rows, seed_cols, cancel, progress, eigenvalues, profile);
        HIP_CHECK(hipStreamSynchronize(sys->ctx->stream)); // nothing of this solve is in flight when the next one starts
        return MH_OK;
    } catch (const std::exception &e) {
        return mh_guard(sys->ctx, e);
    }
}
