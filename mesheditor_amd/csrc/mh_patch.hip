// Sliver patches of the smoothers.
//
// A badly shaped tetrahedron (four nearly coplanar points: volume / edge^3 orders below a regular one) enters the operator as a
// near-constraint: entries 1e3 .. 1e6 times its neighbours', tying its ten (P2) or four (P1) nodes together.  Point Jacobi sees
// only the huge diagonal and moves those nodes by almost nothing per step, while the error it should remove -- the element
// moving rigidly with its surroundings deforming -- is cheap in energy: the Chebyshev-Jacobi smoother stalls on exactly the
// meshes a tetrahedralised scan produces (measured on the 30k-tet skillet scan, two-grid bound with an exact coarse solve:
// condition number 309 with point Jacobi, 34 with the patches below on the worst 3 % of the elements; tools/proto/smoothers.py).
// The cure is local and exact: for every element whose shape measure q = 6 sqrt(2) V / (rms edge)^3 falls below a threshold, the
// operator restricted to the element's nodes (30 x 30 or 12 x 12, a principal submatrix of the assembled A, hence SPD) is
// inverted once at set-up, and the smoother's diagonal scaling becomes
//        M^-1 = D^-1 + sum over sliver elements e of R_e^T (A_ee)^-1 R_e                                    (additive Schwarz)
// The existing smoother kernels keep computing the D^-1 part; the patch part is a separate correction after each step, in two
// small launches: (a) per patch, y_e = (A_ee)^-1 (R_e v) with the lanes over the panel's columns; (b) per node row touched by
// a patch, the sum of its entries of y in list order -- no atomics, bit-reproducible.  A mesh without slivers (every Kuhn
// workload) has no patches and runs exactly the code it ran before.
#include "mh_common.h"

namespace {
constexpr int TB = 256;

__global__ void k_element_quality(const double *__restrict__ pts, const uint32_t *__restrict__ elem_ref, uint32_t stride, uint32_t nt, float *__restrict__ q, float threshold,
                                  uint32_t *__restrict__ summary) {
    const uint32_t el_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = el_raw < nt;
    const uint32_t el = valid ? el_raw : nt - 1; // (idle lanes recompute the last element: the wave reductions below need every lane)
    double v[4][3];
    for (int a = 0; a < 4; ++a)
        for (int d = 0; d < 3; ++d) v[a][d] = pts[3 * size_t(elem_ref[size_t(el) * stride + a]) + d];
    const double bx = v[1][0] - v[0][0], by = v[1][1] - v[0][1], bz = v[1][2] - v[0][2];
    const double cx = v[2][0] - v[0][0], cy = v[2][1] - v[0][1], cz = v[2][2] - v[0][2];
    const double dx = v[3][0] - v[0][0], dy = v[3][1] - v[0][1], dz = v[3][2] - v[0][2];
    const double vol = fabs(dx * (by * cz - cy * bz) + dy * (bz * cx - cz * bx) + dz * (bx * cy - cx * by)) / 6;
    double e2 = 0;
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b)
            for (int d = 0; d < 3; ++d) e2 += (v[a][d] - v[b][d]) * (v[a][d] - v[b][d]);
    const double rms = sqrt(e2 / 6);
    const float shape = rms > 0 ? float(vol * 8.485281374238571 / (rms * rms * rms)) : 0.f; // 1 for the regular tetrahedron
    if (valid) q[el] = shape;
    uint32_t below = valid && shape < threshold ? 1u : 0u, lowest = __float_as_uint(shape);
    for (int off = 32; off > 0; off >>= 1) {
        below += __shfl_xor(below, off, 64);
        lowest = min(lowest, uint32_t(__shfl_xor(lowest, off, 64)));
    }
    if ((threadIdx.x & 63) == 0) { // (integer atomics: the order of arrival does not change the result)
        if (below) atomicAdd(summary, below);
        atomicMin(summary + 1, lowest);
    }
}

// One 64-thread workgroup per patch: gather the (3 NPE)^2 principal submatrix from the BSR rows (columns ascending: binary
// search), invert it in LDS by Gauss-Jordan elimination without pivoting (SPD), store it row-major in both precisions.
template<int NPE>
__global__ void __launch_bounds__(64) k_patch_inverse(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const double *__restrict__ aval,
                                                     const uint32_t *__restrict__ pnodes, const double *__restrict__ weight, uint32_t npatches,
                                                     double *__restrict__ inv64, float *__restrict__ inv32, int *__restrict__ info) {
    constexpr int N = 3 * NPE, LD = N + 1;
    __shared__ double a[N * LD];
    __shared__ uint32_t nd[NPE];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    if (tid < NPE) nd[tid] = pnodes[size_t(p) * NPE + tid];
    __syncthreads();
    for (int pair = tid; pair < NPE * NPE; pair += 64) {
        const int ia = pair / NPE, ib = pair % NPE;
        const uint32_t r = nd[ia], c = nd[ib];
        uint32_t lo = row_ptr[r], hi = row_ptr[r + 1];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (col[mid] < c) lo = mid + 1;
            else hi = mid;
        }
        const bool found = lo < row_ptr[r + 1] && col[lo] == c; // (two nodes of one element always share a block; a repeated node would not)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) a[(3 * ia + i) * LD + 3 * ib + j] = found ? aval[9 * size_t(lo) + 3 * i + j] : 0.0;
    }
    __syncthreads();
    // a pivot that is not safely positive (below 1e-13 of the block's largest diagonal entry: a near-singular A_ee would put
    // entries of 1e+15 into the inverse and infinities into its single-precision copy) fails the patch
    __shared__ double floor_;
    if (tid == 0) {
        double dmax = 0;
        for (int k = 0; k < N; ++k) dmax = fmax(dmax, a[k * LD + k]);
        floor_ = 1e-13 * dmax;
    }
    __syncthreads();
    const double floor = floor_;
    bool bad = false;
    for (int k = 0; k < N; ++k) {
        const double piv = a[k * LD + k];
        if (!(piv > floor)) bad = true;
        const double d = 1.0 / piv;
        __syncthreads();
        if (int(tid) < N && int(tid) != k) a[k * LD + tid] *= d; // row k scaled (entry (k, k) waits)
        __syncthreads();
        if (int(tid) < N && int(tid) != k) {
            const double f = a[tid * LD + k];
            for (int j = 0; j < N; ++j)
                if (j != k) a[tid * LD + j] -= f * a[k * LD + j];
            a[tid * LD + k] = -f * d;
        }
        if (int(tid) == k) a[k * LD + k] = d;
        __syncthreads();
    }
    if (bad && tid == 0) {
        atomicAdd(info, 1);
        atomicMax(info + 1, int(p) + 1);
    }
    for (int e = tid; e < N * N; e += 64) {
        const double v = bad ? 0.0 : weight[p] * a[(e / N) * LD + e % N]; // a patch that failed contributes nothing (the diagonal scaling still covers its nodes)
        inv64[size_t(p) * N * N + e] = v;
        inv32[size_t(p) * N * N + e] = float(v);
    }
}

// (a) y_p = inv_p (R_p v), v = in - (minus ? minus : 0); one wave per (patch, 64-column tile), lanes = panel columns.
template<typename T, int NPE>
__global__ void __launch_bounds__(64) k_patch_solve(const T *__restrict__ in, const T *__restrict__ minus, uint32_t w, const uint32_t *__restrict__ pnodes,
                                                   const T *__restrict__ inv, T *__restrict__ y) {
    constexpr int N = 3 * NPE;
    const uint32_t p = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x;
    if (c >= w) return;
    T v[N];
#pragma unroll
    for (int a = 0; a < NPE; ++a) {
        const size_t row = size_t(3) * pnodes[size_t(p) * NPE + a];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const size_t o = (row + k) * w + c;
            v[3 * a + k] = minus ? in[o] - minus[o] : in[o];
        }
    }
    const T *m = inv + size_t(p) * N * N; // wave-uniform addresses: scalar loads
    T *out = y + size_t(p) * N * w + c;
    for (int i = 0; i < N; ++i) {
        T s = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) s += m[i * N + j] * v[j];
        out[size_t(i) * w] = s;
    }
}

// (b) per touched node row and column: s = coef * (its entries of y, in list order); d += s, x += s (either may be null), or
// z (double, pitch wz) += s for columns below wz.
template<typename T, int NPE>
__global__ void k_patch_gather(const T *__restrict__ y, uint32_t w, const uint32_t *__restrict__ touched, const uint32_t *__restrict__ t_ptr,
                               const uint32_t *__restrict__ t_patch, const uint32_t *__restrict__ t_local, uint32_t ntouched, T coef, T *__restrict__ d, T *__restrict__ x,
                               double *__restrict__ z, uint32_t wz) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(ntouched) * 3 * w) return;
    const uint32_t c = uint32_t(i % w), k = uint32_t((i / w) % 3), t = uint32_t(i / (size_t(3) * w));
    T s = 0;
    for (uint32_t l = t_ptr[t]; l < t_ptr[t + 1]; ++l) s += y[(size_t(t_patch[l]) * (3 * NPE) + 3 * t_local[l] + k) * w + c];
    s *= coef;
    const size_t row = size_t(3) * touched[t] + k;
    if (d) d[row * w + c] += s;
    if (x) x[row * w + c] += s;
    if (z && c < wz) z[row * wz + c] += double(s);
}
} // namespace

// Shape measure of every kept element; the elements below `threshold` become the patch list of both levels (host: a few
// thousand entries).  elem_ref: kept_tets x 10 in the reference numbering (corners first), points in the same numbering.
void mh_select_patches(mh_system *sys, float threshold) {
    mh_context *ctx = sys->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t nt = sys->kept_tets;
    sys->patches2 = PatchSet{};
    sys->patches1 = PatchSet{};
    sys->patches2.npe = 10;
    sys->patches1.npe = 4;
    if (!(threshold > 0) || nt == 0) return;
    DevArray<float> q(ctx, nt);
    DevArray<uint32_t> summary(ctx, 2); // [0]: elements below the threshold, [1]: the smallest shape measure (bits of a positive float order as integers)
    const uint32_t init[2] = {0u, 0x7f7fffffu};
    summary.upload(init, 2);
    k_element_quality<<<div_up(nt, TB), TB, 0, st>>>(sys->points, sys->elem_nodes_ref, 10, nt, q, threshold, summary);
    KERNEL_CHECK();
    uint32_t hs[2];
    summary.download(hs, 2);
    memcpy(&sys->worst_quality, &hs[1], sizeof(float));
    if (hs[0] == 0) return; // a well-shaped mesh (every Kuhn workload): eight bytes came back, nothing else happens
    const std::vector<float> hq = q.to_host();
    std::vector<uint32_t> bad;
    for (uint32_t e = 0; e < nt; ++e)
        if (hq[e] < threshold) bad.push_back(e);
    // node lists of the bad elements, both levels (internal numbering)
    std::vector<uint32_t> en(size_t(nt) * 10), ep(size_t(nt) * 4);
    HIP_CHECK(hipMemcpyAsync(en.data(), sys->elem_nodes.get(), en.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(ep.data(), sys->elem_p1.get(), ep.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    // Overlapping patches add up: where c of them share a node the sum overshoots c-fold and drags the smoother's spectral bound
    // along (lmax 21 against 8 on the 30k-tet scan filled with interior points, whose slivers come in clusters).  Each patch is
    // scaled by (the largest number of patches at any of its nodes)^-0.35: still symmetric positive definite.  Measured on the four
    // scan workloads (iterations, surface-refined / interior fills at 30k and 95k tets): unscaled 27 / 39 / 44 / 54; exponent -1
    // 26 / 40 / 54 / 57; -0.5 26 / 34 / 45 / 50; -0.35 26 / 33 / 44 / 48; a greedy independent set of patches instead 29 / 45 / 77 / 76.
    constexpr double overlap_exponent = -0.35;
    const auto fill = [&](PatchSet &ps, const std::vector<uint32_t> &elem, uint32_t npe) {
        const uint32_t np = uint32_t(bad.size());
        std::vector<uint32_t> nodes(size_t(np) * npe);
        std::vector<std::pair<uint32_t, uint32_t>> inc; // (node, patch * npe + local)
        for (uint32_t p = 0; p < np; ++p)
            for (uint32_t a = 0; a < npe; ++a) {
                nodes[size_t(p) * npe + a] = elem[size_t(bad[p]) * npe + a];
                inc.emplace_back(nodes[size_t(p) * npe + a], p * npe + a);
            }
        std::sort(inc.begin(), inc.end());
        std::vector<uint32_t> touched, ptr{0}, tp, tl;
        for (size_t k = 0; k < inc.size(); ++k) {
            if (k == 0 || inc[k].first != inc[k - 1].first) {
                if (k) ptr.push_back(uint32_t(k));
                touched.push_back(inc[k].first);
            }
            tp.push_back(inc[k].second / npe);
            tl.push_back(inc[k].second % npe);
        }
        ptr.push_back(uint32_t(inc.size()));
        std::vector<double> weight(np, 1.0);
        {
            std::vector<uint32_t> cover(touched.size(), 0);
            for (size_t t = 0; t + 1 < ptr.size(); ++t) cover[t] = ptr[t + 1] - ptr[t];
            std::vector<uint32_t> worst(np, 1);
            for (size_t t = 0; t + 1 < ptr.size(); ++t)
                for (uint32_t l = ptr[t]; l < ptr[t + 1]; ++l) worst[tp[l]] = std::max(worst[tp[l]], cover[t]);
            for (uint32_t p = 0; p < np; ++p) weight[p] = std::pow(double(worst[p]), overlap_exponent);
        }
        ps.weight.reset(ctx, np);
        ps.weight.upload(weight.data(), np);
        ps.n_patches = np;
        ps.n_touched = uint32_t(touched.size());
        ps.nodes.reset(ctx, nodes.size());
        ps.touched.reset(ctx, touched.size());
        ps.t_ptr.reset(ctx, ptr.size());
        ps.t_patch.reset(ctx, tp.size());
        ps.t_local.reset(ctx, tl.size());
        ps.nodes.upload(nodes.data(), nodes.size());
        ps.touched.upload(touched.data(), touched.size());
        ps.t_ptr.upload(ptr.data(), ptr.size());
        ps.t_patch.upload(tp.data(), tp.size());
        ps.t_local.upload(tl.data(), tl.size());
        HIP_CHECK(hipStreamSynchronize(st));
    };
    fill(sys->patches2, en, 10);
    fill(sys->patches1, ep, 4);
}

// The patch inverses of one level from its shifted operator (after k_shift_values).
void mh_build_patch_inverses(mh_context *ctx, const BsrLevel &lvl, PatchSet &ps) {
    if (!ps.n_patches) return;
    const size_t n = size_t(3) * ps.npe;
    ps.inv64.reset(ctx, size_t(ps.n_patches) * n * n);
    ps.inv32.reset(ctx, size_t(ps.n_patches) * n * n);
    ps.dropped.reset(ctx, 2); // read back by mh_finish_hierarchy, with the set-up's next synchronising download
    ps.dropped.zero();
    if (ps.npe == 10) k_patch_inverse<10><<<ps.n_patches, 64, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval, ps.nodes, ps.weight, ps.n_patches, ps.inv64, ps.inv32, ps.dropped);
    else k_patch_inverse<4><<<ps.n_patches, 64, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval, ps.nodes, ps.weight, ps.n_patches, ps.inv64, ps.inv32, ps.dropped);
    KERNEL_CHECK();
}

template<typename T>
void mh_apply_patches(mh_context *ctx, const PatchSet &ps, const T *in, const T *minus, uint32_t w, T coef, T *d, T *x, double *z, uint32_t wz, T *scratch) {
    if (!ps.n_patches) return;
    const T *inv;
    if constexpr (std::is_same<T, double>::value) inv = ps.inv64.get();
    else inv = ps.inv32.get();
    const dim3 grid(ps.n_patches, div_up(w, 64));
    const size_t rows = size_t(ps.n_touched) * 3 * w;
    if (ps.npe == 10) {
        k_patch_solve<T, 10><<<grid, 64, 0, ctx->stream>>>(in, minus, w, ps.nodes, inv, scratch);
        k_patch_gather<T, 10><<<div_up(rows, TB), TB, 0, ctx->stream>>>(scratch, w, ps.touched, ps.t_ptr, ps.t_patch, ps.t_local, ps.n_touched, coef, d, x, z, wz);
    } else {
        k_patch_solve<T, 4><<<grid, 64, 0, ctx->stream>>>(in, minus, w, ps.nodes, inv, scratch);
        k_patch_gather<T, 4><<<div_up(rows, TB), TB, 0, ctx->stream>>>(scratch, w, ps.touched, ps.t_ptr, ps.t_patch, ps.t_local, ps.n_touched, coef, d, x, z, wz);
    }
    KERNEL_CHECK();
}
template void mh_apply_patches<float>(mh_context *, const PatchSet &, const float *, const float *, uint32_t, float, float *, float *, double *, uint32_t, float *);
template void mh_apply_patches<double>(mh_context *, const PatchSet &, const double *, const double *, uint32_t, double, double *, double *, double *, uint32_t, double *);
