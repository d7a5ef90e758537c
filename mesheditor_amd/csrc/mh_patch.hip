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
struct PerDeviceOnce { // hipFuncSetAttribute once per (kernel, device)
    std::mutex m;
    bool done[64] = {};
    template<typename F> void run(int device, F &&f) {
        std::lock_guard<std::mutex> l(m);
        if (!done[unsigned(device) % 64]) f(), done[unsigned(device) % 64] = true;
    }
};

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
// ---- clusters: one exact inverse on the union of the nodes of badly shaped elements that hang together (PatchSet, mh_common.h) ----
// The principal submatrix of the level's operator on a cluster's rows, dense row-major of order N = rows of the cluster: one thread per
// pair of nodes (binary search of the BSR row), blockIdx.x = cluster, the pairs strided over blockIdx.y.
__global__ void k_cluster_block(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const double *__restrict__ aval, const uint32_t *__restrict__ crow,
                                const uint32_t *__restrict__ cptr, const uint64_t *__restrict__ iptr, double *__restrict__ out) {
    const uint32_t c = blockIdx.x, r0 = cptr[c], N = cptr[c + 1] - r0, nn = N / 3;
    double *a = out + iptr[c];
    for (uint64_t pair = uint64_t(blockIdx.y) * blockDim.x + threadIdx.x; pair < uint64_t(nn) * nn; pair += uint64_t(gridDim.y) * blockDim.x) {
        const uint32_t ia = uint32_t(pair / nn), ib = uint32_t(pair % nn);
        const uint32_t r = crow[r0 + 3 * ia] / 3, cc = crow[r0 + 3 * ib] / 3;
        uint32_t lo = row_ptr[r], hi = row_ptr[r + 1];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (col[mid] < cc) lo = mid + 1;
            else hi = mid;
        }
        const bool found = lo < row_ptr[r + 1] && col[lo] == cc;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) a[size_t(3 * ia + i) * N + 3 * ib + j] = found ? aval[9 * size_t(lo) + 3 * i + j] : 0.0;
    }
}

// Clusters of order <= 128: Gauss-Jordan in LDS, one workgroup each (as k_patch_inverse, order at run time); `list` = the clusters of this launch.
__global__ void __launch_bounds__(128) k_cluster_inverse_lds(const uint32_t *__restrict__ list, const uint32_t *__restrict__ cptr, const uint64_t *__restrict__ iptr, double *__restrict__ inv,
                                                           int *__restrict__ info) {
    extern __shared__ double lds_a[];
    const uint32_t c = list[blockIdx.x];
    const int N = int(cptr[c + 1] - cptr[c]), LD = N + 1, tid = threadIdx.x;
    double *g = inv + iptr[c];
    for (int e = tid; e < N * N; e += 128) lds_a[(e / N) * LD + e % N] = g[e];
    __shared__ double floor_;
    __syncthreads();
    if (tid == 0) {
        double dmax = 0;
        for (int k = 0; k < N; ++k) dmax = fmax(dmax, lds_a[k * LD + k]);
        floor_ = 1e-13 * dmax;
    }
    __syncthreads();
    const double floor = floor_;
    bool bad = false;
    for (int k = 0; k < N; ++k) {
        const double piv = lds_a[k * LD + k];
        if (!(piv > floor)) bad = true;
        const double d = 1.0 / piv;
        __syncthreads();
        if (tid < N && tid != k) lds_a[k * LD + tid] *= d;
        __syncthreads();
        if (tid < N && tid != k) {
            const double f = lds_a[tid * LD + k];
            for (int j = 0; j < N; ++j)
                if (j != k) lds_a[tid * LD + j] -= f * lds_a[k * LD + j];
            lds_a[tid * LD + k] = -f * d;
        }
        if (tid == k) lds_a[k * LD + k] = d;
        __syncthreads();
    }
    if (bad && tid == 0) {
        atomicAdd(info, 1);
        atomicMax(info + 1, int(c) + 1000001); // (cluster ids are reported above a million: told apart from element patches)
    }
    for (int e = tid; e < N * N; e += 128) g[e] = bad ? 0.0 : lds_a[(e / N) * LD + e % N];
}

// y_c = inv_c (R_c v), v = in - (minus ? minus : 0): 64 x 64 output tiles, K in slices of 32 through LDS, 4 x 4 values per thread.
template<typename T>
__global__ void __launch_bounds__(256) k_cluster_apply(const T *__restrict__ in, const T *__restrict__ minus, uint32_t w, const uint32_t *__restrict__ crow, const uint32_t *__restrict__ cptr,
                                                      const uint64_t *__restrict__ iptr, const uint32_t *__restrict__ tile_cluster, const uint32_t *__restrict__ tile_row0,
                                                      const T *__restrict__ inv, T *__restrict__ y, size_t y_slice) {
    constexpr int KT = 32;
    __shared__ T s_inv[64][KT + 1];
    __shared__ T s_v[KT][64 + 1];
    const uint32_t c = tile_cluster[blockIdx.x], row0 = tile_row0[blockIdx.x], r0 = cptr[c], N = cptr[c + 1] - r0, col0 = blockIdx.y * 64;
    const T *m = inv + iptr[c];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    // the K range in gridDim.z slices (whole KT chunks each), slice z to its own copy of y: a cluster of 2 400 rows has 39 row tiles -- too few
    // workgroups for the chip, each with a K loop of 77 chunks; the slices are added in a fixed order by k_cluster_scatter (bit-reproducible)
    const uint32_t kslice = ((N + gridDim.z - 1) / gridDim.z + KT - 1) / KT * KT, kb = blockIdx.z * kslice, ke = min(N, kb + kslice);
    y += size_t(blockIdx.z) * y_slice;
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    for (uint32_t k0 = kb; k0 < ke; k0 += KT) {
        for (int e = tid; e < 64 * KT; e += 256) {
            const uint32_t r = row0 + e / KT, k = k0 + e % KT;
            s_inv[e / KT][e % KT] = (r < N && k < ke) ? m[size_t(r) * N + k] : T(0);
        }
        for (int e = tid; e < KT * 64; e += 256) {
            const uint32_t k = k0 + e / 64, cc = col0 + e % 64;
            T v = 0;
            if (k < ke && cc < w) {
                const size_t o = size_t(crow[r0 + k]) * w + cc;
                v = minus ? in[o] - minus[o] : in[o];
            }
            s_v[e / 64][e % 64] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < KT; ++k) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = s_inv[ty * 4 + i][k];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = s_v[k][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t r = row0 + ty * 4 + i;
        if (r >= N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t cc = col0 + tx * 4 + j;
            if (cc < w) y[size_t(r0 + r) * w + cc] = acc[i][j];
        }
    }
}

// the clusters' rows are pairwise distinct: s = coef * y goes straight to its row; d += s, x += s (either may be null), or z (double, pitch wz) += s
template<typename T>
__global__ void k_cluster_scatter(const T *__restrict__ y, uint32_t w, const uint32_t *__restrict__ crow, uint32_t nrows, T coef, T *__restrict__ d, T *__restrict__ x, double *__restrict__ z,
                                  uint32_t wz, int slices) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nrows) * w) return;
    const uint32_t c = uint32_t(i % w);
    const size_t row = crow[i / w];
    T sum = 0;
    for (int q = 0; q < slices; ++q) sum += y[size_t(q) * nrows * w + i]; // (fixed order)
    const T s = coef * sum;
    if (d) d[row * w + c] += s;
    if (x) x[row * w + c] += s;
    if (z && c < wz) z[row * wz + c] += double(s);
}

__global__ void k_f64_to_f32(const double *__restrict__ in, float *__restrict__ out, size_t count) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < count) out[i] = float(in[i]);
}
__global__ void k_copy_block(const double *__restrict__ in, uint32_t ldi, double *__restrict__ out, uint32_t ldo, uint32_t rows, uint32_t cols) { // out[r][c] = in[r][c] (column-major, lds as given)
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(rows) * cols) return;
    const size_t c = i / rows, r = i % rows;
    out[c * ldo + r] = in[c * ldi + r];
}
__global__ void k_symmetrize(double *__restrict__ a, uint32_t n) { // both triangles = their mean
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(n) * n) return;
    const size_t r = i / n, c = i % n;
    if (c < r) {
        const double v = 0.5 * (a[r * n + c] + a[c * n + r]);
        a[r * n + c] = v;
        a[c * n + r] = v;
    }
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
    // Connected components of the bad elements (two of them hang together when they share a P2 node).  A component of ONE element stays an
    // element patch (30 x 30 / 12 x 12, the kernels above); a component of two or more -- a pole fan's needles, caps stacked on one another, the
    // slivers along a scan's sharp rim -- becomes a CLUSTER: one exact inverse on the union of its nodes (PatchSet, mh_common.h).
    // Clusters are for meshes with FLAT cells (worst shape below 1e-4): there they turn "no convergence" into the iteration count of a well-shaped
    // mesh (96 x 48 UV sphere's fill with 172 cells flat to 1e-8: 54 -> 23 iterations).  On the scan fills (worst shape ~1e-3, slivers strung along
    // the rims in components of up to 163 elements) they save 4-7 iterations of 41 and cost twice the time in dense products: element patches
    // there, as before.  MH_CLUSTERS=1 forces them, =0 forbids.
    static const int cluster_switch = getenv("MH_CLUSTERS") ? atoi(getenv("MH_CLUSTERS")) : -1;
    const bool flat_cells = sys->worst_quality < 1e-4f;
    bool use_clusters = cluster_switch == 1 || (cluster_switch != 0 && cluster_switch != 2 && flat_cells); // (2, experiment: the P1 level only)
    static const uint32_t cluster_cap = getenv("MH_CLUSTER_CAP") ? uint32_t(std::max(16, atoi(getenv("MH_CLUSTER_CAP")))) : 2048u; // nodes: an inverse of order 6 144 at most
    if (getenv("MH_DEBUG_PATCH")) fprintf(stderr, "[patch] bad %zu of %u, n_nodes %u n_points %u worst %.2e clusters %d\n", bad.size(), nt, sys->n_nodes, sys->n_points, double(sys->worst_quality), int(use_clusters));
    std::vector<uint32_t> comp(bad.size());
    {
        std::vector<uint32_t> parent(bad.size());
        for (uint32_t k = 0; k < bad.size(); ++k) parent[k] = k;
        const auto find = [&](uint32_t a) {
            while (parent[a] != a) a = parent[a] = parent[parent[a]];
            return a;
        };
        std::vector<int32_t> owner(sys->n_nodes, -1);
        for (uint32_t k = 0; k < bad.size(); ++k)
            for (uint32_t a = 0; a < 10; ++a) {
                const uint32_t v = en[size_t(bad[k]) * 10 + a];
                if (owner[v] < 0) owner[v] = int32_t(k);
                else {
                    const uint32_t ra = find(uint32_t(owner[v])), rb = find(k);
                    if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb); // (the root is the component's first element: deterministic)
                }
            }
        for (uint32_t k = 0; k < bad.size(); ++k) comp[k] = find(k);
    }
    std::vector<uint32_t> comp_size(bad.size(), 0);
    for (uint32_t k = 0; k < bad.size(); ++k) ++comp_size[comp[k]];
    std::vector<uint32_t> single; // positions in `bad` of the elements that stay element patches
    std::vector<std::vector<uint32_t>> multi; // per cluster component: positions in `bad`, ascending
    const auto fill = [&](PatchSet &ps, const std::vector<uint32_t> &elem, uint32_t npe, uint32_t n_level_nodes) {
        const uint32_t np = uint32_t(single.size());
        std::vector<uint32_t> nodes(size_t(np) * npe);
        std::vector<std::pair<uint32_t, uint32_t>> inc; // (node, patch * npe + local)
        for (uint32_t p = 0; p < np; ++p)
            for (uint32_t a = 0; a < npe; ++a) {
                nodes[size_t(p) * npe + a] = elem[size_t(bad[single[p]]) * npe + a];
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
        if (np) {
            std::vector<uint32_t> cover(touched.size(), 0);
            for (size_t t = 0; t + 1 < ptr.size(); ++t) cover[t] = ptr[t + 1] - ptr[t];
            std::vector<uint32_t> worst(np, 1);
            for (size_t t = 0; t + 1 < ptr.size(); ++t)
                for (uint32_t l = ptr[t]; l < ptr[t + 1]; ++l) worst[tp[l]] = std::max(worst[tp[l]], cover[t]);
            for (uint32_t p = 0; p < np; ++p) weight[p] = std::pow(double(worst[p]), overlap_exponent); // (1 with clusters: single components share no node)
        }
        ps.n_patches = np;
        ps.n_bad_elements = uint32_t(bad.size());
        ps.n_touched = uint32_t(touched.size());
        if (np) {
            ps.weight.reset(ctx, np);
            ps.weight.upload(weight.data(), np);
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
        }
        // clusters: the union of each component's nodes on this level, first come first kept; a component with more than cluster_cap nodes is
        // cut into pieces of that many (node-disjoint: block Jacobi over the pieces)
        std::vector<uint32_t> crow, cptr{0};
        std::vector<uint64_t> iptr{0};
        std::vector<uint32_t> tile_cluster, tile_row0;
        std::vector<uint8_t> seen(n_level_nodes, 0);
        uint32_t largest = 0;
        const auto close_piece = [&](std::vector<uint32_t> &piece) {
            if (piece.empty()) return;
            std::sort(piece.begin(), piece.end());
            for (const uint32_t v : piece)
                for (uint32_t k = 0; k < 3; ++k) crow.push_back(3 * v + k);
            const uint32_t order = uint32_t(3 * piece.size());
            const uint32_t c = uint32_t(cptr.size() - 1);
            for (uint32_t r0 = 0; r0 < order; r0 += 64) tile_cluster.push_back(c), tile_row0.push_back(r0);
            cptr.push_back(uint32_t(crow.size()));
            iptr.push_back(iptr.back() + uint64_t(order) * order);
            largest = std::max(largest, uint32_t(piece.size()));
            piece.clear();
        };
        for (const auto &members : multi) {
            std::vector<uint32_t> piece;
            for (const uint32_t k : members)
                for (uint32_t a = 0; a < npe; ++a) {
                    const uint32_t v = elem[size_t(bad[k]) * npe + a];
                    if (seen[v]) continue;
                    seen[v] = 1;
                    piece.push_back(v);
                    if (piece.size() >= cluster_cap) close_piece(piece);
                }
            close_piece(piece);
        }
        ps.n_clusters = uint32_t(cptr.size() - 1);
        ps.cluster_rows = uint32_t(crow.size());
        ps.cluster_tiles = uint32_t(tile_cluster.size());
        ps.largest_cluster = largest;
        ps.h_cluster_ptr = cptr;
        if (ps.n_clusters) {
            ps.cluster_row.reset(ctx, crow.size());
            ps.cluster_ptr.reset(ctx, cptr.size());
            ps.cluster_inv_ptr.reset(ctx, iptr.size());
            ps.tile_cluster.reset(ctx, tile_cluster.size());
            ps.tile_row0.reset(ctx, tile_row0.size());
            ps.cluster_row.upload(crow.data(), crow.size());
            ps.cluster_ptr.upload(cptr.data(), cptr.size());
            ps.cluster_inv_ptr.upload(iptr.data(), iptr.size());
            ps.tile_cluster.upload(tile_cluster.data(), tile_cluster.size());
            ps.tile_row0.upload(tile_row0.data(), tile_row0.size());
            ps.cinv64.reset(ctx, iptr.back());
            ps.cinv32.reset(ctx, iptr.back());
        }
        HIP_CHECK(hipStreamSynchronize(st));
    };
    const auto regroup = [&](bool clusters) { // which elements stay element patches, which hang together as clusters
        single.clear();
        multi.clear();
        std::vector<int32_t> slot(bad.size(), -1);
        for (uint32_t k = 0; k < bad.size(); ++k) {
            if (!clusters || comp_size[comp[k]] == 1) { single.push_back(k); continue; }
            if (slot[comp[k]] < 0) slot[comp[k]] = int32_t(multi.size()), multi.emplace_back();
            multi[size_t(slot[comp[k]])].push_back(k);
        }
    };
    regroup(use_clusters);
    if (getenv("MH_DEBUG_PATCH")) fprintf(stderr, "[patch] single %zu multi %zu\n", single.size(), multi.size());
    fill(sys->patches2, en, 10, sys->n_nodes);
    if (getenv("MH_DEBUG_PATCH")) fprintf(stderr, "[patch] P2 filled: clusters %u rows %u largest %u\n", sys->patches2.n_clusters, sys->patches2.cluster_rows, sys->patches2.largest_cluster);
    if (cluster_switch == 2) regroup(true);
    fill(sys->patches1, ep, 4, sys->n_points);
    if (getenv("MH_DEBUG_PATCH")) fprintf(stderr, "[patch] P1 filled: clusters %u rows %u largest %u\n", sys->patches1.n_clusters, sys->patches1.cluster_rows, sys->patches1.largest_cluster);
}

// SPD inverse in place, order n (column-major = row-major: symmetric), by the block Gauss-Jordan elimination of the coarse level
// (mh_eigs.hip: mh_build_hierarchy), 128 columns per step: P = A_kk^-1 (one workgroup), C = A(:, k), R = P A(k, :), A -= C R, A(k, :) = R,
// A(:, k) = -C P, A_kk = P.  Returns the first pivot block's failure (0: fine).
static int spd_inverse_in_place(mh_context *ctx, double *a, uint32_t n) {
    const uint32_t nb = 128;
    DevArray<double> cblk(ctx, size_t(n) * nb), rblk(ctx, size_t(n) * nb), pinv(ctx, size_t(nb) * nb);
    DevArray<int> info(ctx, 1);
    info.zero();
    const double one = 1, zero = 0, mone = -1;
    const rocblas_int ld = rocblas_int(n);
    for (uint32_t k0 = 0; k0 < n; k0 += nb) { // (ctx->blas is bound to ctx->stream: mh_context)
        const rocblas_int w = rocblas_int(std::min(nb, n - k0));
        mh_spd_inverse_small(ctx, a + size_t(k0) * n + k0, n, uint32_t(w), pinv, uint32_t(w), info);
        HIP_CHECK(hipMemcpyAsync(cblk, a + size_t(k0) * n, size_t(n) * size_t(w) * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, w, ld, w, &one, pinv, w, a + k0, ld, &zero, rblk, w));
        ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, ld, ld, w, &mone, cblk, ld, rblk, w, &one, a, ld));
        HIP_CHECK(hipMemcpy2DAsync(a + k0, size_t(n) * sizeof(double), rblk.get(), size_t(w) * sizeof(double), size_t(w) * sizeof(double), n, hipMemcpyDeviceToDevice, ctx->stream));
        ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, ld, w, w, &mone, cblk, ld, pinv, w, &zero, a + size_t(k0) * n, ld));
        HIP_CHECK(hipMemcpy2DAsync(a + size_t(k0) * n + k0, size_t(n) * sizeof(double), pinv.get(), size_t(w) * sizeof(double), size_t(w) * sizeof(double), size_t(w), hipMemcpyDeviceToDevice, ctx->stream));
    }
    k_symmetrize<<<div_up(size_t(n) * n, TB), TB, 0, ctx->stream>>>(a, n);
    KERNEL_CHECK();
    int h = 0;
    info.download(&h, 1); // (synchronises: the workspaces go back to the pool after the last kernel that reads them)
    return h;
}

// The patch inverses of one level from its shifted operator (after k_shift_values).
void mh_build_patch_inverses(mh_context *ctx, const BsrLevel &lvl, PatchSet &ps) {
    if (!ps.any()) return;
    ps.dropped.reset(ctx, 2); // read back by mh_finish_hierarchy, with the set-up's next synchronising download
    ps.dropped.zero();
    if (ps.n_patches) {
        const size_t n = size_t(3) * ps.npe;
        ps.inv64.reset(ctx, size_t(ps.n_patches) * n * n);
        ps.inv32.reset(ctx, size_t(ps.n_patches) * n * n);
        if (ps.npe == 10) k_patch_inverse<10><<<ps.n_patches, 64, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval, ps.nodes, ps.weight, ps.n_patches, ps.inv64, ps.inv32, ps.dropped);
        else k_patch_inverse<4><<<ps.n_patches, 64, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval, ps.nodes, ps.weight, ps.n_patches, ps.inv64, ps.inv32, ps.dropped);
        KERNEL_CHECK();
    }
    if (!ps.n_clusters) return;
    k_cluster_block<<<dim3(ps.n_clusters, 64), TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval, ps.cluster_row, ps.cluster_ptr, ps.cluster_inv_ptr, ps.cinv64);
    KERNEL_CHECK();
    // orders up to 128 in one launch (Gauss-Jordan in LDS, a workgroup each); the larger ones one after the other by the blocked elimination
    std::vector<uint32_t> small, large;
    for (uint32_t c = 0; c < ps.n_clusters; ++c) (ps.h_cluster_ptr[c + 1] - ps.h_cluster_ptr[c] <= 128 ? small : large).push_back(c);
    if (!small.empty()) {
        DevArray<uint32_t> list(ctx, small.size());
        list.upload(small.data(), small.size());
        static PerDeviceOnce attr;
        attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cluster_inverse_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); });
        k_cluster_inverse_lds<<<uint32_t(small.size()), 128, size_t(128) * 129 * sizeof(double), ctx->stream>>>(list, ps.cluster_ptr, ps.cluster_inv_ptr, ps.cinv64, ps.dropped);
        KERNEL_CHECK();
        HIP_CHECK(hipStreamSynchronize(ctx->stream)); // (`list` returns to the pool)
    }
    if (!large.empty()) {
        std::vector<uint64_t> iptr(ps.n_clusters + 1);
        ps.cluster_inv_ptr.download(iptr.data(), iptr.size());
        for (const uint32_t c : large) {
            const uint32_t order = ps.h_cluster_ptr[c + 1] - ps.h_cluster_ptr[c];
            if (spd_inverse_in_place(ctx, ps.cinv64.get() + iptr[c], order) != 0) { // not safely positive definite: the cluster contributes nothing
                HIP_CHECK(hipMemsetAsync(ps.cinv64.get() + iptr[c], 0, size_t(order) * order * sizeof(double), ctx->stream));
                const int report[2] = {1, int(c) + 1000001};
                int have[2] = {0, 0};
                ps.dropped.download(have, 2);
                have[0] += report[0], have[1] = std::max(have[1], report[1]);
                ps.dropped.upload(have, 2);
                HIP_CHECK(hipStreamSynchronize(ctx->stream));
            }
        }
    }
    k_f64_to_f32<<<div_up(ps.cinv64.count, TB), TB, 0, ctx->stream>>>(ps.cinv64, ps.cinv32, ps.cinv64.count);
    KERNEL_CHECK();
}

template<typename T>
void mh_apply_patches(mh_context *ctx, const PatchSet &ps, const T *in, const T *minus, uint32_t w, T coef, T *d, T *x, double *z, uint32_t wz, T *scratch) {
    if (ps.n_patches) {
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
    if (ps.n_clusters) { // (rows disjoint from the element patches' and from one another: the order of the two parts does not matter)
        const T *cinv;
        if constexpr (std::is_same<T, double>::value) cinv = ps.cinv64.get();
        else cinv = ps.cinv32.get();
        T *y = scratch + size_t(ps.n_patches) * 3 * ps.npe * w;
        // K slices: enough workgroups for the chip when the tiles alone are few (the largest clusters), one when there are plenty
        const uint32_t tiles = ps.cluster_tiles * div_up(w, 64);
        const int slices = ps.largest_cluster >= 256 && tiles < 2048 ? PatchSet::kClusterSlices : 1;
        k_cluster_apply<T><<<dim3(ps.cluster_tiles, div_up(w, 64), slices), 256, 0, ctx->stream>>>(in, minus, w, ps.cluster_row, ps.cluster_ptr, ps.cluster_inv_ptr, ps.tile_cluster, ps.tile_row0, cinv, y,
                                                                                                   size_t(ps.cluster_rows) * w);
        k_cluster_scatter<T><<<div_up(size_t(ps.cluster_rows) * w, TB), TB, 0, ctx->stream>>>(y, w, ps.cluster_row, ps.cluster_rows, coef, d, x, z, wz, slices);
        KERNEL_CHECK();
    }
}
template void mh_apply_patches<float>(mh_context *, const PatchSet &, const float *, const float *, uint32_t, float, float *, float *, double *, uint32_t, float *);
template void mh_apply_patches<double>(mh_context *, const PatchSet &, const double *, const double *, uint32_t, double, double *, double *, double *, uint32_t, double *);
