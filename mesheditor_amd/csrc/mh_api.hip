// C ABI entry points of libmodalhip (include/modalhip.h): contexts, mesh upload, assembly, exports for parity,
// nearest-point sampling, result extraction.
#include "mh_common.h"

#include <algorithm>
#include <cmath>
#include <memory>

namespace {
constexpr int TB = 256;

__global__ void k_export_blocks(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const uint32_t *__restrict__ perm, uint32_t nnodes,
                                uint32_t *__restrict__ row_out, uint32_t *__restrict__ col_out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nnodes) return;
    for (uint32_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
        row_out[p] = perm[r];
        col_out[p] = perm[col[p]];
    }
}
// column-major n x w (reference DOF order) <-> row-major n x w (internal order)
__global__ void k_ref_to_panel(const double *__restrict__ xref, const uint32_t *__restrict__ perm, uint32_t nnodes, uint32_t w, double *__restrict__ panel) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nnodes) * 3 * w) return;
    const uint32_t c = uint32_t(i % w), comp = uint32_t((i / w) % 3), node = uint32_t(i / (size_t(3) * w));
    panel[i] = xref[size_t(c) * (size_t(3) * nnodes) + size_t(3) * perm[node] + comp];
}
template<typename S, typename D> __global__ void k_cast(const S *__restrict__ src, D *__restrict__ dst, size_t count) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = D(src[i]);
}
template<typename T>
__global__ void k_panel_to_ref(const double *__restrict__ panel, const uint32_t *__restrict__ perm, uint32_t nnodes, uint32_t wsrc, uint32_t w, T *__restrict__ xref) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nnodes) * 3 * w) return;
    const uint32_t c = uint32_t(i % w), comp = uint32_t((i / w) % 3), node = uint32_t(i / (size_t(3) * w));
    xref[size_t(c) * (size_t(3) * nnodes) + size_t(3) * perm[node] + comp] = T(panel[(size_t(3) * node + comp) * wsrc + c]);
}
__global__ void k_gather_shapes(const double *__restrict__ evecs, const uint32_t *__restrict__ inv_perm, const uint32_t *__restrict__ nodes, uint32_t n_nodes,
                                uint32_t wsrc, uint32_t ncols, float *__restrict__ out) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(n_nodes) * ncols * 3) return;
    const uint32_t comp = uint32_t(i % 3), c = uint32_t((i / 3) % ncols), k = uint32_t(i / (size_t(3) * ncols));
    out[i] = float(evecs[(size_t(3) * inv_perm[nodes[k]] + comp) * wsrc + c]);
}
// Nearest tet point per excitation position: strict '<' over ascending point index = first minimum.
__global__ void k_nearest(const double *__restrict__ pts, uint32_t npts, const float *__restrict__ pos, uint32_t n, uint32_t *__restrict__ nearest) {
    __shared__ double s_pts[TB * 3];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double px = 0, py = 0, pz = 0;
    if (i < n) { px = double(pos[3 * size_t(i)]); py = double(pos[3 * size_t(i) + 1]); pz = double(pos[3 * size_t(i) + 2]); }
    double best = 1.7976931348623157e308;
    uint32_t arg = 0;
    for (uint32_t base = 0; base < npts; base += TB) {
        const uint32_t cnt = min(uint32_t(TB), npts - base);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt * 3; k += TB) s_pts[k] = pts[3 * size_t(base) + k];
        __syncthreads();
        for (uint32_t k = 0; k < cnt; ++k) {
            const double dx = px - s_pts[3 * k], dy = py - s_pts[3 * k + 1], dz = pz - s_pts[3 * k + 2];
            const double d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; arg = base + k; }
        }
    }
    if (i < n) nearest[i] = arg;
}
// The same for a handful of positions (the app and the bench use P = 10): one workgroup per position, the points dealt over
// its threads, then an ordered reduction -- smallest distance, ties to the smallest index, i.e. the same first minimum.
__global__ void k_nearest_few(const double *__restrict__ pts, uint32_t npts, const float *__restrict__ pos, uint32_t *__restrict__ nearest) {
    __shared__ double s_d[TB];
    __shared__ uint32_t s_i[TB];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const double px = double(pos[3 * size_t(q)]), py = double(pos[3 * size_t(q) + 1]), pz = double(pos[3 * size_t(q) + 2]);
    double best = 1.7976931348623157e308;
    uint32_t arg = 0xffffffffu;
    for (uint32_t k = tid; k < npts; k += TB) {
        const double dx = px - pts[3 * size_t(k)], dy = py - pts[3 * size_t(k) + 1], dz = pz - pts[3 * size_t(k) + 2];
        const double d = dx * dx + dy * dy + dz * dz;
        if (d < best) { best = d; arg = k; }
    }
    s_d[tid] = best;
    s_i[tid] = arg;
    __syncthreads();
    for (uint32_t half = TB / 2; half > 0; half >>= 1) {
        if (tid < half) {
            const double od = s_d[tid + half];
            const uint32_t oi = s_i[tid + half];
            if (od < s_d[tid] || (od == s_d[tid] && oi < s_i[tid])) { s_d[tid] = od; s_i[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) nearest[q] = s_i[0] == 0xffffffffu ? 0u : s_i[0];
}
} // namespace

template<typename T> static int export_vectors(const mh_system *s, uint32_t n_cols, T *out) {
    if (!s || !out) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        if (n_cols > s->evec_cols) mh_throw(MH_EINVAL, "%u columns requested, %u solved", n_cols, s->evec_cols);
        const size_t n = size_t(3) * s->n_nodes;
        DevArray<T> ref(ctx, n * n_cols);
        k_panel_to_ref<T><<<div_up(n * n_cols, TB), TB, 0, ctx->stream>>>(s->evecs, s->perm, s->n_nodes, s->evec_cols, n_cols, ref.get());
        KERNEL_CHECK();
        ref.download(out, n * n_cols);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

extern "C" {
void mh_abi_struct_sizes(uint32_t out[4]) {
    if (!out) return;
    out[0] = uint32_t(sizeof(mh_profile)), out[1] = uint32_t(sizeof(mh_solver_config)), out[2] = uint32_t(sizeof(mh_material)), out[3] = uint32_t(sizeof(mh_mass_props));
}
int mh_context_create(int device, mh_context **out) {
    if (!out) return MH_EINVAL;
    *out = nullptr;
    auto *ctx = new mh_context;
    try {
        int count = 0;
        HIP_CHECK(hipGetDeviceCount(&count));
        if (count <= 0) mh_throw(MH_EHIP, "no HIP device: libmodalhip has no CPU fallback");
        if (device < 0 || device >= count) mh_throw(MH_EINVAL, "device %d out of range (%d visible)", device, count);
        ctx->device = device;
        if (hipDeviceGetAttribute(&ctx->cu_count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->cu_count <= 0) ctx->cu_count = 256;
        HIP_CHECK(hipSetDevice(device));
        {
            size_t free_bytes = 0, total_bytes = 0;
            if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) total_bytes = 0, (void)hipGetLastError();
            ctx->pool.set_cap_for_device(total_bytes);
        }
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ROCBLAS_CHECK(rocblas_create_handle(&ctx->blas));
        ROCBLAS_CHECK(rocblas_set_stream(ctx->blas, ctx->stream));
        ROCBLAS_CHECK(rocblas_set_pointer_mode(ctx->blas, rocblas_pointer_mode_host));
        ROCBLAS_CHECK(rocblas_set_atomics_mode(ctx->blas, rocblas_atomics_not_allowed)); // bit-reproducible library reductions
        *out = ctx;
        return MH_OK;
    } catch (const std::exception &e) {
        const int code = mh_guard(ctx, e);
        fprintf(stderr, "modalhip: %s\n", e.what());
        if (ctx->blas) rocblas_destroy_handle(ctx->blas);
        if (ctx->blas_aux) rocblas_destroy_handle(ctx->blas_aux);
        if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
        if (ctx->aux2_stream) (void)hipStreamDestroy(ctx->aux2_stream);
        for (hipEvent_t ev : ctx->ahead_ev) (void)hipEventDestroy(ev);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return code;
    }
}
void mh_context_destroy(mh_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->blas) rocblas_destroy_handle(ctx->blas);
    if (ctx->blas_aux) rocblas_destroy_handle(ctx->blas_aux);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->aux2_stream) (void)hipStreamDestroy(ctx->aux2_stream);
    for (hipEvent_t e : ctx->ahead_ev) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
const char *mh_last_error(const mh_context *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }
int mh_context_synchronize(mh_context *ctx) {
    if (!ctx) return MH_EINVAL;
    try {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}
void *mh_context_stream(mh_context *ctx) { return ctx ? ctx->stream : nullptr; }
int mh_context_time_kernels(mh_context *ctx, int enable) {
    if (!ctx) return MH_EINVAL;
    try {
        mh_timer_flush(ctx);
        ctx->time_kernels = enable != 0;
        for (auto &t : ctx->totals) t = {};
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}
int mh_context_kernel_class_stats(mh_context *ctx, int kernel_class, uint64_t *launches, double *total_ms, double *total_work) {
    if (!ctx || kernel_class < 0 || kernel_class >= MH_KERNEL_CLASSES) return MH_EINVAL;
    try {
        mh_timer_flush(ctx);
        const auto &t = ctx->totals[kernel_class];
        if (launches) *launches = t.launches;
        if (total_ms) *total_ms = t.ms;
        if (total_work) *total_work = t.work;
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}
int mh_context_kernel_stats(mh_context *ctx, uint64_t *launches, double *total_ms, double *total_bytes) {
    return mh_context_kernel_class_stats(ctx, MH_KERNEL_SPMM, launches, total_ms, total_bytes);
}
void mh_default_config(mh_solver_config *c) {
    if (c) *c = mh_solver_config{20.f, 16000.f, 30, 45, 1e-8, 1e-4, 100, 0, 0.f};
}

int mh_mesh_create(mh_context *ctx, uint32_t n_points, const double *points_xyz, uint32_t n_tets, const uint32_t *tets, mh_mesh **out) {
    if (!ctx || !out || (n_points && !points_xyz) || (n_tets && !tets)) return MH_EINVAL;
    *out = nullptr;
    try {
        MhSharedPhase not_during_a_factorisation(ctx->device);
        HIP_CHECK(hipSetDevice(ctx->device));
        for (size_t i = 0; i < size_t(n_tets) * 4; ++i)
            if (tets[i] >= n_points) mh_throw(MH_EINVAL, "tet %zu references point %u of %u", i / 4, tets[i], n_points);
        auto mesh = std::make_unique<mh_mesh>();
        mesh->ctx = ctx;
        mesh->n_points = n_points;
        mesh->n_tets = n_tets;
        mesh->points.reset(ctx, size_t(n_points) * 3);
        mesh->tets.reset(ctx, size_t(n_tets) * 4);
        if (n_points) mesh->points.upload(points_xyz, size_t(n_points) * 3);
        if (n_tets) mesh->tets.upload(tets, size_t(n_tets) * 4);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = mesh.release();
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}
void mh_mesh_destroy(mh_mesh *m) { delete m; }

int mh_assemble(mh_context *ctx, const mh_mesh *mesh, const mh_material *material, mh_system **out) {
    if (!ctx || !mesh || !material || !out) return MH_EINVAL;
    *out = nullptr;
    try {
        MhSharedPhase not_during_a_factorisation(ctx->device);
        HIP_CHECK(hipSetDevice(ctx->device));
        auto sys = std::make_unique<mh_system>();
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        HIP_CHECK(hipEventRecord(e0, ctx->stream));
        mh_build_system(ctx, mesh, *material, sys.get());
        HIP_CHECK(hipEventRecord(e1, ctx->stream));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        sys->profile.assemble = ms * 1e-3; // QuadMesh + Assemble of the reference's profile, fused on the device
        sys->profile.dofs = 3 * sys->n_nodes;
        *out = sys.release();
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}
void mh_system_destroy(mh_system *s) { delete s; }

int mh_system_dims(const mh_system *s, uint32_t *dofs, uint32_t *node_count, uint32_t *kept_tets, uint64_t *node_blocks) {
    if (!s) return MH_EINVAL;
    if (dofs) *dofs = 3 * s->n_nodes;
    if (node_count) *node_count = s->n_nodes;
    if (kept_tets) *kept_tets = s->kept_tets;
    if (node_blocks) *node_blocks = s->L2.n_blocks;
    return MH_OK;
}
int mh_system_element_nodes(const mh_system *s, uint32_t *out) {
    if (!s || !out) return MH_EINVAL;
    try {
        s->elem_nodes_ref.download(out, size_t(s->kept_tets) * 10);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(s->ctx, e); }
}
int mh_system_export_blocks(const mh_system *s, uint32_t *row_node, uint32_t *col_node, double *k_blocks, double *m_blocks) {
    if (!s) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t nb = s->L2.n_blocks;
        DevArray<uint32_t> r(ctx, nb), c(ctx, nb);
        k_export_blocks<<<div_up(s->n_nodes, TB), TB, 0, ctx->stream>>>(s->L2.row_ptr, s->L2.col, s->perm, s->n_nodes, r, c);
        KERNEL_CHECK();
        if (row_node) r.download(row_node, nb);
        if (col_node) c.download(col_node, nb);
        if (k_blocks) s->L2.kval.download(k_blocks, nb * 9);
        if (m_blocks) s->L2.mval.download(m_blocks, nb);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}
int mh_system_matvec(mh_system *s, int which, const double *x, double *y, uint32_t width) {
    if (!s || !x || !y || width == 0 || which < 0 || which > 4) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = size_t(3) * s->n_nodes;
        DevArray<double> xr(ctx, n * width), xp(ctx, n * width), yp(ctx, n * width);
        xr.upload(x, n * width);
        k_ref_to_panel<<<div_up(n * width, TB), TB, 0, ctx->stream>>>(xr, s->perm, s->n_nodes, width, xp);
        KERNEL_CHECK();
        if (which == 0) mh_spmm(ctx, s->L2, s->L2.kval, xp, yp, nullptr, nullptr, width);
        else if (which == 1) mh_spmm(ctx, s->L2, nullptr, xp, nullptr, s->L2.mval, yp, width);
        else if (which == 3 || which == 4) { // the preconditioner's products of the shifted operator: single precision / double A x single panel
            std::lock_guard<std::mutex> lock(mh_solve_mutex());
            mh_build_hierarchy(s, -15791.367041742974);
            DevArray<float> xf(ctx, n * width), yf(ctx, n * width);
            k_cast<double, float><<<div_up(n * width, TB), TB, 0, ctx->stream>>>(xp.get(), xf.get(), n * width);
            KERNEL_CHECK();
            if (which == 3) {
                mh_spmm_f32(ctx, s->L2, xf, yf, width);
                k_cast<float, double><<<div_up(n * width, TB), TB, 0, ctx->stream>>>(yf.get(), yp.get(), n * width);
                KERNEL_CHECK();
            } else {
                mh_spmm_mixed(ctx, s->L2, xf, yp, width);
            }
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        } else { // which == 2: the shifted operator A = K - sigma M of the eigensolver at the reference's shift
            std::lock_guard<std::mutex> lock(mh_solve_mutex());
            mh_build_hierarchy(s, -15791.367041742974);
            mh_spmm(ctx, s->L2, s->L2.aval, xp, yp, nullptr, nullptr, width);
        }
        k_panel_to_ref<double><<<div_up(n * width, TB), TB, 0, ctx->stream>>>(yp, s->perm, s->n_nodes, width, width, xr.get());
        KERNEL_CHECK();
        xr.download(y, n * width);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mh_system_shift_invert(mh_system *s, double sigma, const double *b, double *x, uint32_t width, double rel_tol, uint32_t max_iters, uint32_t *iterations, double *worst_relative_residual) {
    if (!s || !b || !x || width == 0) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n = size_t(3) * s->n_nodes;
        if (!(rel_tol > 0)) rel_tol = 1e-11;
        if (!max_iters) max_iters = 200;
        uint32_t its = 0;
        double worst = 0;
        // (slabs of 64 columns: the preconditioner's panels are sized by the slab)
        for (uint32_t c0 = 0; c0 < width; c0 += 64) {
            const uint32_t wc = std::min(64u, width - c0);
            DevArray<double> ref(ctx, n * wc), bp(ctx, n * wc), xp(ctx, n * wc);
            ref.upload(b + size_t(c0) * n, n * wc);
            k_ref_to_panel<<<div_up(n * wc, TB), TB, 0, ctx->stream>>>(ref, s->perm, s->n_nodes, wc, bp);
            KERNEL_CHECK();
            double slab_worst = 0;
            its = std::max(its, mh_shift_invert_panel(s, sigma, bp, xp, wc, rel_tol, max_iters, &slab_worst));
            worst = std::max(worst, slab_worst);
            k_panel_to_ref<double><<<div_up(n * wc, TB), TB, 0, ctx->stream>>>(xp, s->perm, s->n_nodes, wc, wc, ref.get());
            KERNEL_CHECK();
            ref.download(x + size_t(c0) * n, n * wc);
        }
        if (iterations) *iterations = its;
        if (worst_relative_residual) *worst_relative_residual = worst;
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mh_nearest_points(mh_context *ctx, const mh_mesh *mesh, uint32_t n, const float *positions_xyz, uint32_t *nearest) {
    if (!ctx || !mesh || (n && (!positions_xyz || !nearest))) return MH_EINVAL;
    if (n == 0) return MH_OK;
    try {
        MhSharedPhase not_during_a_factorisation(ctx->device);
        HIP_CHECK(hipSetDevice(ctx->device));
        DevArray<float> pos(ctx, size_t(n) * 3);
        DevArray<uint32_t> out(ctx, n);
        pos.upload(positions_xyz, size_t(n) * 3);
        if (n <= 64) k_nearest_few<<<n, TB, 0, ctx->stream>>>(mesh->points, mesh->n_points, pos, out);
        else k_nearest<<<div_up(n, TB), TB, 0, ctx->stream>>>(mesh->points, mesh->n_points, pos, n, out);
        KERNEL_CHECK();
        out.download(nearest, n);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mh_system_gather_shapes(const mh_system *s, uint32_t n_nodes, const uint32_t *nodes, uint32_t n_cols, float *shapes) {
    if (!s || !nodes || !shapes) return MH_EINVAL;
    mh_context *ctx = s->ctx;
    try {
        MhSharedPhase not_during_a_factorisation(ctx->device);
        HIP_CHECK(hipSetDevice(ctx->device));
        if (n_cols > s->evec_cols) mh_throw(MH_EINVAL, "%u columns requested, %u solved", n_cols, s->evec_cols);
        for (uint32_t i = 0; i < n_nodes; ++i)
            if (nodes[i] >= s->n_nodes) mh_throw(MH_EINVAL, "node %u out of range", nodes[i]);
        if (n_nodes == 0 || n_cols == 0) return MH_OK;
        DevArray<uint32_t> nd(ctx, n_nodes);
        DevArray<float> out(ctx, size_t(n_nodes) * n_cols * 3);
        nd.upload(nodes, n_nodes);
        k_gather_shapes<<<div_up(size_t(n_nodes) * n_cols * 3, TB), TB, 0, ctx->stream>>>(s->evecs, s->inv_perm, nd, n_nodes, s->evec_cols, n_cols, out);
        KERNEL_CHECK();
        out.download(shapes, size_t(n_nodes) * n_cols * 3);
        return MH_OK;
    } catch (const std::exception &e) { return mh_guard(ctx, e); }
}

int mh_system_basis(const mh_system *s, uint32_t n_cols, float *basis) { return export_vectors<float>(s, n_cols, basis); }
int mh_system_eigenvectors(const mh_system *s, uint32_t n_cols, double *vectors) { return export_vectors<double>(s, n_cols, vectors); }
int mh_system_residual_report(const mh_system *s, double *worst_plain_residual, uint32_t *dropped_patches) {
    if (!s) return MH_EINVAL;
    if (worst_plain_residual) *worst_plain_residual = s->plain_residual;
    if (dropped_patches) dropped_patches[0] = s->dropped_patches[0], dropped_patches[1] = s->dropped_patches[1];
    return MH_OK;
}
}
