// libmodalhip_lab.so: measurement and experiment entry points that are NOT part of the path's ABI (include/modalhip.h has
// none of them): timing loops around the product's own kernels, the tridiagonalisation variants called directly, and the
// matrix-free element-by-element operator (lab/mh_elem.hip: built, measured 2.1-2.6x slower than the BSR product, kept here for
// the record).  Links libmodalhip.so and reaches its internals through mh_common.h; tests, tools/ and bench.py's secondary
// figures load it explicitly (tools/lab.py), the product never does.
#include "../mh_common.h"
#include "modalhip_lab.h"

#include <algorithm>

void mh_elementwise_apply(mh_context *ctx, const mh_system *sys, double sigma, const double *x, double *y, uint32_t w); // lab/mh_elem.hip

namespace {
constexpr int TB = 256;
// reference DOF order (column-major n x width) <-> internal row-major panel
__global__ void k_lab_ref_to_panel(const double *__restrict__ x, const uint32_t *__restrict__ perm, uint32_t nnodes, uint32_t width, double *__restrict__ panel) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nnodes) * 3 * width) return;
    const uint32_t c = uint32_t(i % width), comp = uint32_t((i / width) % 3), node = uint32_t(i / (size_t(3) * width));
    panel[i] = x[size_t(c) * (size_t(3) * nnodes) + size_t(3) * perm[node] + comp];
}
__global__ void k_lab_panel_to_ref(const double *__restrict__ panel, const uint32_t *__restrict__ perm, uint32_t nnodes, uint32_t width, double *__restrict__ y) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nnodes) * 3 * width) return;
    const uint32_t c = uint32_t(i % width), comp = uint32_t((i / width) % 3), node = uint32_t(i / (size_t(3) * width));
    y[size_t(c) * (size_t(3) * nnodes) + size_t(3) * perm[node] + comp] = panel[i];
}
} // namespace

namespace {
typedef float f4_stream __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream_copy(const f4_stream *__restrict__ src, f4_stream *__restrict__ dst, size_t count) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}
// U independent 16-byte loads in flight per lane before the first store (the one-load-per-iteration form above leaves the memory
// system with too few bytes in flight: 4.5 TB/s against the 6.3 the device reaches); NT: non-temporal loads and stores
template<int U, bool NT, int T = 256, bool NTL = NT> __global__ void __launch_bounds__(T) k_stream_copy_u(const f4_stream *__restrict__ src, f4_stream *__restrict__ dst, size_t count) {
    const size_t stride = size_t(gridDim.x) * T;
    for (size_t base = size_t(blockIdx.x) * T + threadIdx.x; base < count; base += stride * U) {
        f4_stream v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + size_t(u) * stride;
            if (i < count) v[u] = NTL ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + size_t(u) * stride;
            if (i < count) {
                if (NT) __builtin_nontemporal_store(v[u], dst + i);
                else dst[i] = v[u];
            }
        }
    }
}
template<int U, bool NT> __global__ void __launch_bounds__(256) k_stream_read_u(const f4_stream *__restrict__ src, size_t count, float *__restrict__ out) {
    const size_t stride = size_t(gridDim.x) * 256;
    f4_stream acc = {0, 0, 0, 0};
    for (size_t base = size_t(blockIdx.x) * 256 + threadIdx.x; base < count; base += stride * U) {
        f4_stream v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + size_t(u) * stride;
            v[u] = i < count ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : f4_stream{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, s);
}
__global__ void __launch_bounds__(256) k_stream_read(const f4_stream *__restrict__ src, size_t count, float *__restrict__ out) {
    f4_stream acc = {0, 0, 0, 0};
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += size_t(gridDim.x) * 256) acc += src[i];
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, s); // (a workgroup's four waves: the value is never read)
}
} // namespace

extern "C" {
// The rigid-body level's graph aggregation (host code, no device): CSR graph of a level's node blocks in, aggregate of every node
// out; returns the aggregate count.
uint32_t mhl_graph_aggregates(const uint32_t *row_ptr, const uint32_t *col, uint32_t n, uint32_t target, uint32_t max_order, uint32_t *agg_of) {
    std::vector<uint32_t> rp(row_ptr, row_ptr + n + 1), cl(col, col + row_ptr[n]), out;
    const uint32_t na = mh_graph_aggregates(rp, cl, n, target, max_order, out);
    std::copy(out.begin(), out.end(), agg_of);
    return na;
}

// y = (K - sigma M) x at the reference's shift, element by element without the assembled matrix (atomic scatter: equal to
// mh_system_matvec(which = 2) up to rounding, not bit-reproducible).  x, y: column-major n x width, the reference's DOF order.
int mhl_system_elementwise_matvec(mh_system *s, const double *x, double *y, uint32_t width) {
    if (!s || !x || !y || width == 0) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = size_t(3) * s->n_nodes;
        DevArray<double> xr(ctx, n * width), xp(ctx, n * width), yp(ctx, n * width);
        xr.upload(x, n * width);
        k_lab_ref_to_panel<<<div_up(n * width, TB), TB, 0, ctx->stream>>>(xr, s->perm, s->n_nodes, width, xp);
        KERNEL_CHECK();
        mh_elementwise_apply(ctx, s, -15791.367041742974, xp, yp, width);
        k_lab_panel_to_ref<<<div_up(n * width, TB), TB, 0, ctx->stream>>>(yp, s->perm, s->n_nodes, width, xr.get());
        KERNEL_CHECK();
        xr.download(y, n * width);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mhl_system_bench_spmm(mh_system *s, uint32_t width, uint32_t reps, double *avg_ms, double *algorithmic_bytes) {
    if (!s || width == 0 || reps == 0 || !avg_ms) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = size_t(3) * s->n_nodes;
        DevArray<double> x(ctx, n * width), y(ctx, n * width);
        std::vector<double> hx(n * width);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = double((i * 2654435761u) % 1000) * 1e-3 - 0.5;
        x.upload(hx.data(), hx.size());
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int warm = 0; warm < 2; ++warm) mh_spmm(ctx, s->L2, s->L2.kval, x, y, nullptr, nullptr, width);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        for (uint32_t r = 0; r < reps; ++r) mh_spmm(ctx, s->L2, s->L2.kval, x, y, nullptr, nullptr, width);
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = ms / reps;
        if (algorithmic_bytes) *algorithmic_bytes = 76.0 * double(s->L2.n_blocks) + 4.0 * (double(s->n_nodes) + 1) + 16.0 * double(n) * width;
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

// The single-precision smoother's fused product-and-Chebyshev step (mh_spmm_f32_cheb_step) on level 2 (P2) or 1 (P1) over an n x width
// panel, `slabs` launches of width / slabs columns each on separate contiguous panels: does a wide block cost more per column than
// narrow ones?  (Needs mh_build_hierarchy: the fp32 copies of the operators; a solve of the system leaves them behind.)
int mhl_system_bench_cheb_step(mh_system *s, int level, uint32_t width, uint32_t slabs, uint32_t reps, double *avg_ms) {
    if (!s || !avg_ms || width == 0 || slabs == 0 || width % (4 * slabs) || reps == 0 || (level != 1 && level != 2)) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const BsrLevel &lvl = level == 2 ? s->L2 : s->L1;
        if (!lvl.aval32.get() || !lvl.dinv32.get()) mh_throw(MH_EINVAL, "bench_cheb_step: no single-precision operator (solve the system once first)");
        const size_t n = size_t(3) * lvl.n_nodes;
        const uint32_t wc = width / slabs;
        std::vector<DevArray<float>> d(slabs), d2(slabs), r(slabs), x(slabs);
        std::vector<float> h(n * wc);
        for (size_t i = 0; i < h.size(); ++i) h[i] = float((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
        for (uint32_t q = 0; q < slabs; ++q) {
            for (auto *a : {&d[q], &d2[q], &r[q], &x[q]}) {
                a->reset(ctx, n * wc);
                a->upload(h.data(), h.size());
            }
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        auto run = [&] {
            for (uint32_t q = 0; q < slabs; ++q)
                if (!mh_spmm_f32_cheb_step(ctx, lvl, d[q], d2[q], r[q], x[q], lvl.dinv32, 0.3f, 0.1f, wc)) mh_throw(MH_EINVAL, "bench_cheb_step: width %u not served by the wide kernel", wc);
        };
        run();
        run();
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        for (uint32_t rep = 0; rep < reps; ++rep) run();
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = ms / reps;
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mhl_system_bench_elementwise(mh_system *s, uint32_t width, uint32_t reps, double *avg_ms) {
    if (!s || width == 0 || reps == 0 || !avg_ms) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = size_t(3) * s->n_nodes;
        DevArray<double> x(ctx, n * width), y(ctx, n * width);
        std::vector<double> hx(n * width);
        for (size_t i = 0; i < hx.size(); ++i) hx[i] = double((i * 2654435761u) % 1000) * 1e-3 - 0.5;
        x.upload(hx.data(), hx.size());
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int warm = 0; warm < 2; ++warm) mh_elementwise_apply(ctx, s, -15791.367041742974, x, y, width);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        for (uint32_t r = 0; r < reps; ++r) mh_elementwise_apply(ctx, s, -15791.367041742974, x, y, width);
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = ms / reps;
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mhl_context_bench_dense(mh_context *ctx, int kind, uint64_t n, uint32_t wa, uint32_t wb, uint32_t reps, double *avg_ms) {
    if (!ctx || !avg_ms || n == 0 || wa == 0 || wb == 0 || reps == 0 || kind < 0 || kind > 1) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        std::lock_guard<std::mutex> lock(mh_solve_mutex());
        DevArray<double> x(ctx, n * wa), y(ctx, n * wb), g(ctx, size_t(wa + wb) * (wa + wb)), z(ctx, n * wa);
        std::vector<double> h(n * std::max(wa, wb));
        for (size_t i = 0; i < h.size(); ++i) h[i] = double((i * 2654435761u) % 1000) * 1e-3 - 0.5;
        x.upload(h.data(), n * wa);
        y.upload(h.data(), n * wb);
        g.upload(h.data(), g.count);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        auto run = [&]() {
            if (kind == 0) mh_gram(ctx, n, x, wa, y, wb, g, wa);
            else mh_combine(ctx, n, x, wa, y, wb, nullptr, 0, g, wa, z, wa, nullptr);
        };
        run();
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        for (uint32_t r = 0; r < reps; ++r) run();
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *avg_ms = ms / reps;
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mhl_context_tridiagonalize_full(mh_context *ctx, int variant, uint32_t m, const double *a, double *d, double *e, double *reflectors, double *tau, uint32_t reps, double *avg_ms);
int mhl_context_tridiagonalize(mh_context *ctx, int variant, uint32_t m, const double *a, double *d, double *e, uint32_t reps, double *avg_ms) {
    return mhl_context_tridiagonalize_full(ctx, variant, m, a, d, e, nullptr, nullptr, reps, avg_ms);
}

// variant 2: the wide kernel (orders up to 768).  reflectors (m x m, LAPACK's lower storage) and tau (m) are returned when asked for.
int mhl_context_tridiagonalize_full(mh_context *ctx, int variant, uint32_t m, const double *a, double *d, double *e, double *reflectors, double *tau, uint32_t reps, double *avg_ms) {
    if (!ctx || !a || !d || !e || m < 2 || variant < 0 || variant > 3 || m > (variant == 2 ? 768u : 256u)) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        MhSharedPhase not_during_a_factorisation(ctx->device); // (no process-wide lock: calls on different contexts are meant to overlap)
        DevArray<double> da(ctx, size_t(m) * m), work(ctx, size_t(m) * m), dd(ctx, m), de(ctx, m), dtau(ctx, m);
        da.upload(a, size_t(m) * m);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        float total = 0;
        for (uint32_t r = 0; r < std::max(1u, reps); ++r) {
            HIP_CHECK(hipMemcpyAsync(work, da, size_t(m) * m * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            HIP_CHECK(hipEventRecord(e0, ctx->stream));
            if (variant == 2) mh_sytrd_wide(ctx, work, m, dd, de, dtau);
            else mh_sytrd_small(ctx, work, m, dd, de, dtau, variant);
            HIP_CHECK(hipEventRecord(e1, ctx->stream));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        dd.download(d, m);
        de.download(e, m - 1);
        if (reflectors) work.download(reflectors, size_t(m) * m);
        if (tau) dtau.download(tau, m);
        if (mh_sytrd_gave_up(ctx)) mh_throw(MH_EHIP, "tridiagonalisation: a workgroup timed out waiting for the others' values");
        if (avg_ms) *avg_ms = total / std::max(1u, reps);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

// The Cholesky-QR step's one-launch kernel: l <- diag(1 / dscale) chol(a) (lower, zeros above), linv <- l^-1; info[0] = 0 or the failing column,
// info[1] = the diagonal spread report.  Column-major host arrays of order w <= 128.
int mhl_context_potrf_inverse(mh_context *ctx, uint32_t w, const double *a, const double *dscale, double *l, double *linv, int *info2) {
    if (!ctx || !a || !dscale || !l || !linv || !info2 || w < 1 || w > 128) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        MhSharedPhase not_during_a_factorisation(ctx->device);
        DevArray<double> da(ctx, size_t(w) * w), dd(ctx, w), dl(ctx, size_t(w) * w);
        DevArray<int> info(ctx, 2);
        da.upload(a, size_t(w) * w);
        dd.upload(dscale, w);
        mh_potrf_small_inverse(ctx, da, w, info, dd, dl);
        da.download(l, size_t(w) * w);
        dl.download(linv, size_t(w) * w);
        info.download(info2, 2);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

// out = a^-1 for a symmetric positive definite a of order w <= 128 (column-major host arrays) through the coarse set-up's one-workgroup
// Gauss-Jordan kernel; the average of `reps` launches, HIP events around each.
int mhl_context_spd_inverse(mh_context *ctx, uint32_t w, const double *a, double *out, uint32_t reps, double *avg_ms) {
    if (!ctx || !a || !out || w < 1 || w > 128) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        MhSharedPhase not_during_a_factorisation(ctx->device);
        DevArray<double> da(ctx, size_t(w) * w), dout(ctx, size_t(w) * w);
        DevArray<int> info(ctx, 1);
        info.zero();
        da.upload(a, size_t(w) * w);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        float total = 0;
        for (uint32_t r = 0; r < std::max(1u, reps); ++r) {
            HIP_CHECK(hipEventRecord(e0, ctx->stream));
            mh_spd_inverse_small(ctx, da, w, w, dout, w, info);
            HIP_CHECK(hipEventRecord(e1, ctx->stream));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        dout.download(out, size_t(w) * w);
        int hinfo = 0;
        info.download(&hinfo, 1);
        if (avg_ms) *avg_ms = total / std::max(1u, reps);
        return hinfo ? MH_EFACTOR : MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

// C = alpha op(A) op(B) + beta C with the Rayleigh-Ritz step's small-product kernel (column-major host arrays; c in and out).
int mhl_context_small_gemm(mh_context *ctx, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, double alpha, const double *a, uint32_t lda, const double *b, uint32_t ldb, double beta,
                           double *c, uint32_t ldc, uint32_t reps, double *avg_ms) {
    if (!ctx || !a || !b || !c || !M || !N || !K || lda < (ta ? K : M) || ldb < (tb ? N : K) || ldc < M) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        MhSharedPhase not_during_a_factorisation(ctx->device);
        const size_t na = size_t(lda) * (ta ? M : K), nb = size_t(ldb) * (tb ? K : N), nc = size_t(ldc) * N;
        DevArray<double> da(ctx, na), db(ctx, nb), dc(ctx, nc), work(ctx, nc);
        da.upload(a, na);
        db.upload(b, nb);
        dc.upload(c, nc);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        float total = 0;
        for (uint32_t r = 0; r < std::max(1u, reps); ++r) {
            HIP_CHECK(hipMemcpyAsync(work, dc, nc * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            HIP_CHECK(hipEventRecord(e0, ctx->stream));
            mh_small_gemm(ctx, ta != 0, tb != 0, M, N, K, alpha, da, lda, db, ldb, beta, work, ldc);
            HIP_CHECK(hipEventRecord(e1, ctx->stream));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total += ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        work.download(c, nc);
        if (avg_ms) *avg_ms = total / std::max(1u, reps);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

// The context's device pool: bytes held from the device, bytes of them idle in the cache, and the cap on the idle part.
int mhl_context_pool_stats(mh_context *ctx, uint64_t *reserved, uint64_t *idle, uint64_t *cap) {
    if (!ctx || !reserved || !idle || !cap) return MH_EINVAL;
    *reserved = ctx->pool.bytes_reserved;
    *idle = ctx->pool.bytes_idle;
    *cap = ctx->pool.cap;
    return MH_OK;
}

// G (wa x wb, column-major) = X^T Y for row-major host panels X (n x wa), Y (n x wb): the solver's Gram kernel, for parity tests.
int mhl_context_gram(mh_context *ctx, uint64_t n, const double *x, uint32_t wa, const double *y, uint32_t wb, double *g) {
    if (!ctx || !x || !y || !g || !n || !wa || !wb) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        MhSharedPhase not_during_a_factorisation(ctx->device);
        DevArray<double> dx(ctx, n * wa), dy(ctx, n * wb), dg(ctx, size_t(wa) * wb);
        dx.upload(x, n * wa);
        dy.upload(y, n * wb);
        mh_gram(ctx, n, dx, wa, dy, wb, dg, wa);
        dg.download(g, size_t(wa) * wb);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

// Measured HBM ceilings of this device (SURVEY 8d asks for a measured copy-kernel ceiling beside the nominal 8 TB/s): a streaming copy
// (bytes read + bytes written per second) and a streaming read (a sum stored once per workgroup); 16-byte accesses, grid-stride, 16
// workgroups per CU.
int mhl_context_bench_stream(mh_context *ctx, uint64_t bytes, uint32_t reps, double *copy_gbs, double *read_gbs) {
    if (!ctx || !copy_gbs || !read_gbs || bytes < 4096 || !reps) return MH_EINVAL;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        MhSharedPhase not_during_a_factorisation(ctx->device);
        const size_t count = bytes / 16;
        const unsigned grid = unsigned(ctx->cu_count) * 32;
        DevArray<float> a(ctx, count * 4), b(ctx, count * 4), out(ctx, grid);
        a.zero();
        out.zero();
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        auto timed = [&](auto &&launch) {
            launch();
            HIP_CHECK(hipEventRecord(e0, ctx->stream));
            for (uint32_t r = 0; r < reps; ++r) launch();
            HIP_CHECK(hipEventRecord(e1, ctx->stream));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            return double(ms) * 1e-3 / reps;
        };
        // the best of a few forms of each (bytes in flight per lane, temporal or not, workgroups per CU): a ceiling, not a kernel of the path
        double t_copy = 1e30, t_read = 1e30;
        const f4_stream *src = reinterpret_cast<const f4_stream *>(a.get());
        f4_stream *dst = reinterpret_cast<f4_stream *>(b.get());
        for (unsigned per_cu : {4u, 8u, 16u, 32u}) {
            const unsigned g = std::min(grid, unsigned(ctx->cu_count) * per_cu);
            t_copy = std::min(t_copy, timed([&] { k_stream_copy<<<g, 256, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<4, false><<<g, 256, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<8, false><<<g, 256, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<4, true><<<g, 256, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<8, true><<<g, 256, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<16, false><<<g, 256, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<16, true><<<g, 256, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<8, true, 256, false><<<g, 256, 0, ctx->stream>>>(src, dst, count); })); // non-temporal stores only
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<8, false, 1024><<<g / 4 + 1, 1024, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { k_stream_copy_u<4, true, 1024><<<g / 4 + 1, 1024, 0, ctx->stream>>>(src, dst, count); }));
            t_copy = std::min(t_copy, timed([&] { (void)hipMemcpyAsync(dst, src, count * 16, hipMemcpyDeviceToDevice, ctx->stream); })); // the runtime's own copy
            t_read = std::min(t_read, timed([&] { k_stream_read<<<g, 256, 0, ctx->stream>>>(src, count, out.get()); }));
            t_read = std::min(t_read, timed([&] { k_stream_read_u<4, false><<<g, 256, 0, ctx->stream>>>(src, count, out.get()); }));
            t_read = std::min(t_read, timed([&] { k_stream_read_u<8, false><<<g, 256, 0, ctx->stream>>>(src, count, out.get()); }));
            t_read = std::min(t_read, timed([&] { k_stream_read_u<8, true><<<g, 256, 0, ctx->stream>>>(src, count, out.get()); }));
        }
        KERNEL_CHECK();
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *copy_gbs = 2.0 * double(count) * 16 / t_copy / 1e9;
        *read_gbs = double(count) * 16 / t_read / 1e9;
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

}
