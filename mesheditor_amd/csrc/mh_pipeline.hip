// Mesh -> matrices, entirely on the device (gfx950).
//
//   FilterDegenerate      reference src/audio/mesh2modes.cpp:42-60   -> k_flag_tets + scan + compaction
//   BuildQuadMesh         :246-264  midside ids in first-encounter order -> sort edge keys, rank by first occurrence
//   ComputeElementBases   :137-165  -> k_element_basis (13 doubles per tet)
//   AssembleQuadratic     :273-327  -> sorted (row node, col node) pair list = sparsity pattern + per-block
//                                      contributor lists; one thread sums each 3x3 node block in a fixed order
//                                      (no atomics, bit-reproducible), 30x30 element tables staged in LDS.
// Nodes are renumbered internally along a Morton curve so that the SpMM's gathers of x hit L2; results are mapped
// back to the reference's numbering at the boundary.  The P1 (corner-node) operator is the exact Galerkin coarse
// operator of the P2 one (P1 is a subspace of P2), so it is assembled directly by the same kernel with linear tables.
#include "mh_common.h"

#include <optional>

#include <hipcub/hipcub.hpp>

namespace {
constexpr int TB = 256;
constexpr int EB = 14; // doubles per element-basis row: volume, four barycentric gradients, one pad (16-byte aligned rows)

// ---- small utilities -------------------------------------------------------------------------------------
struct CubTemp {
    DevArray<unsigned char> buf;
    void *ensure(mh_context *ctx, size_t bytes) {
        if (buf.count < bytes) buf.reset(ctx, bytes + bytes / 4 + 256);
        return buf.get();
    }
};

template<typename K, typename V>
void sort_pairs(mh_context *ctx, CubTemp &tmp, const K *kin, K *kout, const V *vin, V *vout, size_t n, int end_bit) {
    size_t bytes = 0;
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, n, 0, end_bit, ctx->stream));
    void *t = tmp.ensure(ctx, bytes);
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(t, bytes, kin, kout, vin, vout, n, 0, end_bit, ctx->stream));
}
void inclusive_sum(mh_context *ctx, CubTemp &tmp, const uint32_t *in, uint32_t *out, size_t n) {
    size_t bytes = 0;
    HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, in, out, n, ctx->stream));
    void *t = tmp.ensure(ctx, bytes);
    HIP_CHECK(hipcub::DeviceScan::InclusiveSum(t, bytes, in, out, n, ctx->stream));
}
int bit_width64(uint64_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b < 1 ? 1 : b;
}
uint32_t read_u32(mh_context *ctx, const uint32_t *dptr) {
    uint32_t v = 0;
    HIP_CHECK(hipMemcpyAsync(&v, dptr, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return v;
}

// ---- FilterDegenerate -------------------------------------------------------------------------------------
__global__ void k_flag_tets(const double *__restrict__ pts, const uint32_t *__restrict__ tets, uint32_t nt, uint32_t *__restrict__ flag) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    double p[4][3];
    for (int v = 0; v < 4; ++v) {
        const uint32_t id = tets[4 * t + v];
        for (int d = 0; d < 3; ++d) p[v][d] = pts[3 * size_t(id) + d];
    }
    double r[3][3];
    for (int e = 0; e < 3; ++e)
        for (int d = 0; d < 3; ++d) r[e][d] = p[e + 1][d] - p[0][d];
    const double cx = r[1][1] * r[2][2] - r[2][1] * r[1][2];
    const double cy = r[1][2] * r[2][0] - r[2][2] * r[1][0];
    const double cz = r[1][0] * r[2][1] - r[2][0] * r[1][1];
    const double det = fabs(r[0][0] * cx + r[0][1] * cy + r[0][2] * cz);
    double lmax_sq = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = i + 1; j < 4; ++j) {
            const double dx = p[i][0] - p[j][0], dy = p[i][1] - p[j][1], dz = p[i][2] - p[j][2];
            lmax_sq = fmax(lmax_sq, dx * dx + dy * dy + dz * dz);
        }
    flag[t] = det > 1e-12 * lmax_sq * sqrt(lmax_sq) ? 1u : 0u;
}

__global__ void k_compact_tets(const uint32_t *__restrict__ tets, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ incl,
                               uint32_t nt, uint32_t *__restrict__ kept) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt || !flag[t]) return;
    const uint32_t dst = incl[t] - 1;
    for (int v = 0; v < 4; ++v) kept[4 * size_t(dst) + v] = tets[4 * size_t(t) + v];
}

// ---- BuildQuadMesh ----------------------------------------------------------------------------------------
__constant__ int c_edge_corners[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};

__global__ void k_edge_keys(const uint32_t *__restrict__ tets, uint32_t nt, uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt * 6) return;
    const uint32_t t = i / 6, e = i % 6;
    const uint32_t a = tets[4 * size_t(t) + c_edge_corners[e][0]], b = tets[4 * size_t(t) + c_edge_corners[e][1]];
    keys[i] = (uint64_t(min(a, b)) << 32) | max(a, b);
    idx[i] = i;
}

template<typename K> __global__ void k_heads(const K *__restrict__ keys, size_t n, uint32_t *__restrict__ head) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_unique_edges(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ head,
                               const uint32_t *__restrict__ incl, uint32_t n, uint32_t *__restrict__ first_idx, uint64_t *__restrict__ edge_key,
                               uint32_t *__restrict__ iota) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    const uint32_t u = incl[i] - 1;
    first_idx[u] = idx[i]; // stable sort: the head of a group carries the smallest (element, edge) index
    edge_key[u] = keys[i];
    iota[u] = u;
}

__global__ void k_scatter_rank(const uint32_t *__restrict__ u_sorted, uint32_t ne, uint32_t *__restrict__ rank_of) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < ne) rank_of[u_sorted[r]] = r;
}

__global__ void k_assign_nodes(const uint32_t *__restrict__ tets, const uint32_t *__restrict__ idx_sorted, const uint32_t *__restrict__ incl,
                               const uint32_t *__restrict__ rank_of, uint32_t nt, uint32_t npts, uint32_t *__restrict__ elem_nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt * 6) {
        const uint32_t slot = idx_sorted[i];
        elem_nodes[size_t(slot / 6) * 10 + 4 + slot % 6] = npts + rank_of[incl[i] - 1];
    }
    if (i < nt * 4) elem_nodes[size_t(i / 4) * 10 + i % 4] = tets[i];
}

// Coordinates of every P2 node in the reference numbering (midside = edge midpoint) and its Morton key.
__global__ void k_node_xyz_keys(const double *__restrict__ pts, uint32_t npts, const uint64_t *__restrict__ edge_key, const uint32_t *__restrict__ rank_of,
                                uint32_t ne, double3 lo, double inv_extent, double *__restrict__ xyz_ref, uint64_t *__restrict__ mkey, uint32_t *__restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts + ne) return;
    double x, y, z;
    uint32_t node;
    if (i < npts) {
        node = i;
        x = pts[3 * size_t(i)]; y = pts[3 * size_t(i) + 1]; z = pts[3 * size_t(i) + 2];
    } else {
        const uint32_t u = i - npts;
        node = npts + rank_of[u];
        const uint32_t a = uint32_t(edge_key[u] >> 32), b = uint32_t(edge_key[u]);
        x = 0.5 * (pts[3 * size_t(a)] + pts[3 * size_t(b)]);
        y = 0.5 * (pts[3 * size_t(a) + 1] + pts[3 * size_t(b) + 1]);
        z = 0.5 * (pts[3 * size_t(a) + 2] + pts[3 * size_t(b) + 2]);
    }
    xyz_ref[3 * size_t(node)] = x; xyz_ref[3 * size_t(node) + 1] = y; xyz_ref[3 * size_t(node) + 2] = z;
    auto quant = [&](double v, double l) {
        double q = (v - l) * inv_extent * 2097151.0;
        q = fmin(fmax(q, 0.0), 2097151.0);
        return uint64_t(q);
    };
    auto spread = [](uint64_t v) { // 21 bits -> every third bit
        v &= 0x1fffffull;
        v = (v | v << 32) & 0x1f00000000ffffull;
        v = (v | v << 16) & 0x1f0000ff0000ffull;
        v = (v | v << 8) & 0x100f00f00f00f00full;
        v = (v | v << 4) & 0x10c30c30c30c30c3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    mkey[node] = spread(quant(x, lo.x)) | (spread(quant(y, lo.y)) << 1) | (spread(quant(z, lo.z)) << 2);
    ids[node] = node;
}

__global__ void k_invert_perm(const uint32_t *__restrict__ perm, uint32_t n, uint32_t npts, uint32_t *__restrict__ inv, uint32_t *__restrict__ is_corner) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    inv[perm[i]] = i;
    is_corner[i] = perm[i] < npts ? 1u : 0u;
}

__global__ void k_internal_nodes(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ inv, const uint32_t *__restrict__ corner_incl,
                                 const double *__restrict__ xyz_ref, const uint64_t *__restrict__ edge_key_by_rank, uint32_t n, uint32_t npts,
                                 double *__restrict__ node_xyz, uint32_t *__restrict__ parent_a, uint32_t *__restrict__ parent_b,
                                 uint32_t *__restrict__ p1_corner, double *__restrict__ p1_xyz) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t old = perm[i];
    for (int d = 0; d < 3; ++d) node_xyz[3 * size_t(i) + d] = xyz_ref[3 * size_t(old) + d];
    if (old < npts) {
        const uint32_t p1 = corner_incl[i] - 1;
        parent_a[i] = parent_b[i] = p1;
        p1_corner[p1] = i;
        for (int d = 0; d < 3; ++d) p1_xyz[3 * size_t(p1) + d] = xyz_ref[3 * size_t(old) + d];
    } else {
        const uint64_t key = edge_key_by_rank[old - npts];
        const uint32_t ia = inv[uint32_t(key >> 32)], ib = inv[uint32_t(key)];
        parent_a[i] = corner_incl[ia] - 1;
        parent_b[i] = corner_incl[ib] - 1;
    }
}

__global__ void k_edge_key_by_rank(const uint64_t *__restrict__ edge_key, const uint32_t *__restrict__ rank_of, uint32_t ne, uint64_t *__restrict__ out) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < ne) out[rank_of[u]] = edge_key[u];
}

__global__ void k_renumber_elements(const uint32_t *__restrict__ elem_ref, const uint32_t *__restrict__ inv, const uint32_t *__restrict__ corner_incl,
                                    uint32_t nt, uint32_t *__restrict__ elem_int, uint32_t *__restrict__ elem_p1) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt * 10) return;
    const uint32_t v = inv[elem_ref[i]];
    elem_int[i] = v;
    if (i % 10 < 4) elem_p1[size_t(i / 10) * 4 + i % 10] = corner_incl[v] - 1;
}

// Edge lists per P1 node (for the transposed interpolation): pairs (P1 parent, midside P2 node).
__global__ void k_edge_incidence(const uint32_t *__restrict__ parent_a, const uint32_t *__restrict__ parent_b, const uint32_t *__restrict__ perm,
                                 uint32_t n, uint32_t npts, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t old = perm[i];
    if (old < npts) return;
    const uint32_t e = old - npts;
    keys[2 * size_t(e)] = parent_a[i]; vals[2 * size_t(e)] = i;
    keys[2 * size_t(e) + 1] = parent_b[i]; vals[2 * size_t(e) + 1] = i;
}

// ptr[s] = first index whose key >= s (keys ascending), s in [0, nseg]
template<typename K> __global__ void k_segment_ptr(const K *__restrict__ keys, size_t n, uint32_t nseg, uint32_t *__restrict__ ptr) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nseg) return;
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (keys[mid] < K(s)) lo = mid + 1; else hi = mid;
    }
    ptr[s] = uint32_t(lo);
}

// ---- ComputeElementBases ----------------------------------------------------------------------------------
__global__ void k_element_basis(const double *__restrict__ pts, const uint32_t *__restrict__ tets, uint32_t nt, double *__restrict__ basis) {
    const uint32_t el = blockIdx.x * blockDim.x + threadIdx.x;
    if (el >= nt) return;
    double v[4][3];
    for (int a = 0; a < 4; ++a) {
        const uint32_t id = tets[4 * size_t(el) + a];
        for (int d = 0; d < 3; ++d) v[a][d] = pts[3 * size_t(id) + d];
    }
    // det = dot(d - a, cross(b - a, c - a))  (GetTetDeterminant, mesh2modes.cpp:64-66)
    const double bx = v[1][0] - v[0][0], by = v[1][1] - v[0][1], bz = v[1][2] - v[0][2];
    const double cx = v[2][0] - v[0][0], cy = v[2][1] - v[0][1], cz = v[2][2] - v[0][2];
    const double dx = v[3][0] - v[0][0], dy = v[3][1] - v[0][1], dz = v[3][2] - v[0][2];
    const double det = dx * (by * cz - cy * bz) + dy * (bz * cx - cz * bx) + dz * (bx * cy - cx * by);
    double *out = basis + EB * size_t(el); // EB = 14: 13 values + one pad, so that a row is seven aligned 16-byte words
    out[13] = 0.0;
    out[0] = fabs(det / 6);
    // Gradient of barycentric function i along j = signed 3x3 cofactor / det (mesh2modes.cpp:144-161).
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) {
            double col[2][3];
            int ni = 0;
            for (int ii = 0; ii < 4; ++ii) {
                if (ii == i) continue;
                int nj = 0;
                for (int jj = 0; jj < 3; ++jj) {
                    if (jj != j) { col[nj][ni] = v[ii][jj]; ++nj; }
                }
                ++ni;
            }
            const double crx = col[0][1] * col[1][2] - col[1][1] * col[0][2];
            const double cry = col[0][2] * col[1][0] - col[1][2] * col[0][0];
            const double crz = col[0][0] * col[1][1] - col[1][0] * col[0][1];
            const double sign = ((i + j) % 2 == 0) ? -1.0 : 1.0;
            out[1 + 3 * i + j] = sign * (crx + cry + crz) / det;
        }
    }
}

// ---- sparsity pattern + assembly --------------------------------------------------------------------------
template<int NN> __global__ void k_pairs(const uint32_t *__restrict__ en, uint32_t nt, uint64_t nnodes, uint64_t *__restrict__ keys, uint32_t *__restrict__ payload) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(nt) * NN * NN) return;
    const uint32_t t = uint32_t(i / (NN * NN)), ac = uint32_t(i % (NN * NN));
    const uint32_t a = ac / NN, c = ac % NN;
    keys[i] = uint64_t(en[size_t(t) * NN + a]) * nnodes + en[size_t(t) * NN + c];
    payload[i] = uint32_t(i);
}

__global__ void k_block_index(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ head, const uint32_t *__restrict__ incl, size_t n,
                              uint64_t nnodes, uint32_t *__restrict__ col, uint32_t *__restrict__ blk_row, uint32_t *__restrict__ seg) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { seg[incl[n - 1]] = uint32_t(n); return; }
    if (!head[i]) return;
    const uint32_t b = incl[i] - 1;
    col[b] = uint32_t(keys[i] % nnodes);
    blk_row[b] = uint32_t(keys[i] / nnodes);
    seg[b] = uint32_t(i);
}

// One thread per node block: sum the element contributions in contributor order.
// tables: mass[NN][NN] then grad[NN][4][NN][4] (doubles), staged in LDS.
template<int NN> __global__ void __launch_bounds__(TB) k_assemble(const uint32_t *__restrict__ seg, const uint32_t *__restrict__ payload, uint32_t nblocks,
                                                                 const double *__restrict__ basis, const double *__restrict__ tables, double rho, double lambda,
                                                                 double mu, double *__restrict__ kval, double *__restrict__ mval) {
    constexpr int NT = NN * NN + NN * 4 * NN * 4;
    __shared__ double s_tab[NT];
    for (int i = threadIdx.x; i < NT; i += TB) s_tab[i] = tables[i];
    __syncthreads();
    const double *s_mass = s_tab, *s_grad = s_tab + NN * NN;
    __shared__ double s_out[TB * 9];
    const uint32_t b = blockIdx.x * TB + threadIdx.x;
    const bool valid = b < nblocks;
    double k[3][3] = {}, m = 0;
    const uint32_t p0 = valid ? seg[b] : 0u, p1 = valid ? seg[b + 1] : 0u;
    for (uint32_t p = p0; p < p1; ++p) {
        const uint32_t pl = payload[p];
        const uint32_t t = pl / (NN * NN), ac = pl % (NN * NN), a = ac / NN, c = ac % NN;
        double eb[EB];
        {
            const double2 *src = reinterpret_cast<const double2 *>(basis + EB * size_t(t)); // seven 16-byte gathers instead of thirteen 8-byte ones:
#pragma unroll                                                                        // the kernel is bound by the address path of its gathers
            for (int i = 0; i < EB / 2; ++i) {
                const double2 v = src[i];
                eb[2 * i] = v.x, eb[2 * i + 1] = v.y;
            }
        }
        const double vol = eb[0];
        m += rho * vol * s_mass[a * NN + c];
        double g[3][3] = {};
        for (int kk = 0; kk < 4; ++kk) {
            for (int ll = 0; ll < 4; ++ll) {
                const double w = s_grad[((a * 4 + kk) * NN + c) * 4 + ll];
                if (w == 0) continue;
                for (int pp = 0; pp < 3; ++pp)
                    for (int qq = 0; qq < 3; ++qq) g[pp][qq] += w * (eb[1 + 3 * kk + pp] * eb[1 + 3 * ll + qq]);
            }
        }
        const double trace = g[0][0] + g[1][1] + g[2][2];
        for (int pp = 0; pp < 3; ++pp)
            for (int qq = 0; qq < 3; ++qq) k[pp][qq] += vol * (lambda * g[pp][qq] + mu * g[qq][pp] + (pp == qq ? mu * trace : 0.0));
    }
    // the workgroup's 256 blocks are contiguous in kval: through LDS (stride 9 words of 8 bytes: conflict-free) they leave as
    // whole runs of doubles instead of nine 72-byte-strided stores per lane
    for (int pp = 0; pp < 3; ++pp)
        for (int qq = 0; qq < 3; ++qq) s_out[9 * threadIdx.x + 3 * pp + qq] = k[pp][qq];
    if (valid) mval[b] = m;
    __syncthreads();
    const size_t first = size_t(blockIdx.x) * TB;
    const uint32_t count = uint32_t(min(size_t(TB), size_t(nblocks) - first)) * 9;
    for (uint32_t f = threadIdx.x; f < count; f += TB) kval[9 * first + f] = s_out[f];
}

// The same sums, organised by node row: one wave per row of node blocks.
//   * The contributors of a row are (element, a, c) for every element e around the row's node (a = the node's place in e,
//     c = 0 .. NN-1): NN per element, contiguous in the sorted contributor list.  Lanes take ONE contributor each, 64 per
//     round, so the expensive part -- the 4 x 4 x 3 x 3 contraction of the gradient table with the element's barycentric
//     gradients -- is perfectly balanced (in the one-thread-per-block form the lane of the diagonal block walks through
//     every element around the node, 7 .. 30 of them, while its neighbours idle after one to three).
//   * The element data a row needs (volume + four gradients = 13 doubles per element; the elements are exactly the
//     contributors of the row's diagonal block) is staged once per row in a wave-private LDS slice; the shape-function
//     tables live in LDS for the workgroup.
//   * A round's 64 contributions go through LDS to the lanes that own the blocks, which add them in contributor order --
//     the order the one-thread-per-block kernel uses, so both produce the same bits.
//   * The finished 3 x 3 blocks of a row leave through LDS as whole contiguous runs of doubles (coalesced), not as nine
//     72-byte-strided stores per lane.
template<int NN> __global__ void __launch_bounds__(256) k_assemble_rows(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const uint32_t *__restrict__ seg,
                                                                      const uint32_t *__restrict__ payload, uint32_t nrows, const double *__restrict__ basis,
                                                                      const double *__restrict__ tables, double rho, double lambda, double mu,
                                                                      double *__restrict__ kval, double *__restrict__ mval) {
    constexpr int NT = NN * NN + NN * 4 * NN * 4, WPB = 4, MAXE = 32;
    __shared__ double s_tab[NT];
    __shared__ double s_basis[WPB][MAXE * 13];
    __shared__ uint32_t s_tet[WPB][MAXE];
    __shared__ double s_x[WPB][64 * 10]; // a round's contributions [value][lane]; afterwards the row's blocks [block][9]
    for (int i = threadIdx.x; i < NT; i += 256) s_tab[i] = tables[i];
    __syncthreads();
    const double *s_mass = s_tab, *s_grad = s_tab + NN * NN;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * WPB + wave;
    if (row >= nrows) return; // no workgroup barrier below this line
    const uint32_t p0 = row_ptr[row], p1 = row_ptr[row + 1], nblk = p1 - p0;
    // Everything that depends only on the row's extent is requested at once (one memory round trip): the column and the
    // contributor offset of block `lane`.  A row of fewer than 64 blocks then gets every other offset from a neighbouring lane.
    const uint32_t c_lane = lane < nblk ? col[p0 + lane] : 0xffffffffu;
    const uint32_t s_lane = lane <= nblk ? seg[p0 + lane] : 0u;
    const bool narrow = nblk < 64;
    // the row's elements = contributors of its diagonal block
    uint32_t diag = 0;
    {
        const unsigned long long m = __ballot(c_lane == row);
        if (m) diag = uint32_t(__ffsll(m)) - 1;
        else // (a row of more than 64 blocks whose diagonal block lies beyond the first 64)
            for (uint32_t b0 = 64; b0 < nblk; b0 += 64) {
                const unsigned long long m2 = __ballot(b0 + lane < nblk && col[p0 + b0 + lane] == row);
                if (m2) {
                    diag = b0 + uint32_t(__ffsll(m2)) - 1;
                    break;
                }
            }
    }
    const uint32_t e0 = diag < 63 ? uint32_t(__shfl(int(s_lane), int(diag), 64)) : seg[p0 + diag];
    const uint32_t ne = (diag < 63 ? uint32_t(__shfl(int(s_lane), int(diag) + 1, 64)) : seg[p0 + diag + 1]) - e0;
    const bool staged = ne <= uint32_t(MAXE); // more elements around one node than the slice holds: read them from memory
    double *sb = s_basis[wave], *sx = s_x[wave];
    uint32_t *st = s_tet[wave];
    // the first round's contributor descriptors travel together with the element list (one more round trip)
    const uint32_t q_first = uint32_t(__shfl(int(s_lane), 0, 64)) + lane, q_end_first = narrow ? uint32_t(__shfl(int(s_lane), int(nblk), 64)) : seg[p0 + 64];
    const uint32_t pl_first = q_first < q_end_first ? payload[q_first] : 0u;
    if (staged) {
        if (lane < ne) st[lane] = payload[e0 + lane] / (NN * NN);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t f = lane; f < ne * 13; f += 64) sb[f] = basis[EB * size_t(st[f / 13]) + f % 13];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) { // 64 blocks of the row at a time (a row rarely has more)
        const uint32_t nb = min(64u, nblk - b0);
        const bool owner = lane < nb;
        uint32_t mine0, mine1, qa, qb;
        if (narrow) { // offsets from the neighbouring lanes
            const uint32_t next = uint32_t(__shfl_down(int(s_lane), 1, 64));
            mine0 = owner ? s_lane : 0u, mine1 = owner ? next : 0u;
            qa = uint32_t(__shfl(int(s_lane), 0, 64)), qb = uint32_t(__shfl(int(s_lane), int(nblk), 64));
        } else {
            mine0 = owner ? seg[p0 + b0 + lane] : 0u, mine1 = owner ? seg[p0 + b0 + lane + 1] : 0u;
            qa = seg[p0 + b0], qb = seg[p0 + b0 + nb];
        }
        double k[9] = {}, m = 0;
        for (uint32_t r = qa; r < qb; r += 64) {
            const uint32_t q = r + lane;
            double cv[10] = {};
            if (q < qb) {
                const uint32_t pl = (b0 == 0 && r == qa) ? pl_first : payload[q];
                const uint32_t t = pl / (NN * NN), ac = pl % (NN * NN), a = ac / NN, c = ac % NN;
                double eb[13];
                if (staged) {
                    uint32_t slot = 0;
                    for (uint32_t e = 0; e < ne; ++e) slot = st[e] == t ? e : slot;
#pragma unroll
                    for (int i = 0; i < 13; ++i) eb[i] = sb[13 * slot + i];
                } else {
#pragma unroll
                    for (int i = 0; i < 13; ++i) eb[i] = basis[EB * size_t(t) + i];
                }
                const double vol = eb[0];
                cv[9] = rho * vol * s_mass[a * NN + c];
                double g[3][3] = {};
                for (int kk = 0; kk < 4; ++kk) {
                    for (int ll = 0; ll < 4; ++ll) {
                        const double w = s_grad[((a * 4 + kk) * NN + c) * 4 + ll];
                        if (w == 0) continue;
                        for (int pp = 0; pp < 3; ++pp)
                            for (int qq = 0; qq < 3; ++qq) g[pp][qq] += w * (eb[1 + 3 * kk + pp] * eb[1 + 3 * ll + qq]);
                    }
                }
                const double trace = g[0][0] + g[1][1] + g[2][2];
                for (int pp = 0; pp < 3; ++pp)
                    for (int qq = 0; qq < 3; ++qq) cv[3 * pp + qq] = vol * (lambda * g[pp][qq] + mu * g[qq][pp] + (pp == qq ? mu * trace : 0.0));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the previous round has been consumed
#pragma unroll
            for (int e = 0; e < 10; ++e) sx[e * 64 + lane] = cv[e];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t lo = max(mine0, r), hi = min(mine1, r + 64);
            for (uint32_t qq = lo; qq < hi; ++qq) { // contributor order
                const uint32_t j = qq - r;
#pragma unroll
                for (int e = 0; e < 9; ++e) k[e] += sx[e * 64 + j];
                m += sx[9 * 64 + j];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (owner) {
#pragma unroll
            for (int e = 0; e < 9; ++e) sx[9 * lane + e] = k[e];
            mval[p0 + b0 + lane] = m;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double *out = kval + 9 * size_t(p0 + b0);
        for (uint32_t f = lane; f < 9 * nb; f += 64) out[f] = sx[f];
    }
}


// ---- level 0: rigid-body aggregates ------------------------------------------------------------------------
__global__ void k_aggregate_t(const double *__restrict__ p1_xyz, uint32_t npts, uint32_t agg_size, uint32_t nagg, double *__restrict__ tmat) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nagg) return;
    const uint32_t i0 = a * agg_size, i1 = (a == nagg - 1) ? npts : (a + 1) * agg_size;
    double c[3] = {0, 0, 0};
    for (uint32_t i = i0; i < i1; ++i)
        for (int d = 0; d < 3; ++d) c[d] += p1_xyz[3 * size_t(i) + d];
    const double cnt = double(i1 - i0);
    for (int d = 0; d < 3; ++d) c[d] /= cnt;
    double rn[3] = {0, 0, 0}; // squared norms of the three rotation columns
    for (uint32_t i = i0; i < i1; ++i) {
        const double rx = p1_xyz[3 * size_t(i)] - c[0], ry = p1_xyz[3 * size_t(i) + 1] - c[1], rz = p1_xyz[3 * size_t(i) + 2] - c[2];
        rn[0] += ry * ry + rz * rz; // |e_x x r|^2
        rn[1] += rx * rx + rz * rz;
        rn[2] += rx * rx + ry * ry;
    }
    const double st = 1.0 / sqrt(cnt);
    double sr[3];
    for (int q = 0; q < 3; ++q) sr[q] = rn[q] > 1e-300 ? 1.0 / sqrt(rn[q]) : 0.0;
    for (uint32_t i = i0; i < i1; ++i) {
        const double rx = p1_xyz[3 * size_t(i)] - c[0], ry = p1_xyz[3 * size_t(i) + 1] - c[1], rz = p1_xyz[3 * size_t(i) + 2] - c[2];
        double *t = tmat + 18 * size_t(i); // row-major 3 x 6
        for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) t[6 * p + q] = p == q ? st : 0.0;
        // columns 3..5: e_q x r
        t[6 * 0 + 3] = 0;            t[6 * 1 + 3] = -rz * sr[0]; t[6 * 2 + 3] = ry * sr[0];
        t[6 * 0 + 4] = rz * sr[1];   t[6 * 1 + 4] = 0;           t[6 * 2 + 4] = -rx * sr[1];
        t[6 * 0 + 5] = -ry * sr[2];  t[6 * 1 + 5] = rx * sr[2];  t[6 * 2 + 5] = 0;
    }
}
} // namespace

__global__ void k_shift_values(const double *__restrict__ kval, const double *__restrict__ mval, size_t nblocks, double sigma, double *__restrict__ aval) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= nblocks * 9) return;
    const size_t b = i / 9;
    const int e = int(i % 9);
    aval[i] = kval[i] - ((e == 0 || e == 4 || e == 8) ? sigma * mval[b] : 0.0);
}

__global__ void k_diag_inverse(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const double *__restrict__ aval, uint32_t nnodes, double *__restrict__ dinv) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nnodes) return;
    for (uint32_t p = row_ptr[r]; p < row_ptr[r + 1]; ++p) {
        if (col[p] == r) {
            dinv[3 * size_t(r)] = 1.0 / aval[9 * size_t(p)];
            dinv[3 * size_t(r) + 1] = 1.0 / aval[9 * size_t(p) + 4];
            dinv[3 * size_t(r) + 2] = 1.0 / aval[9 * size_t(p) + 8];
            return;
        }
    }
}

// A0 = sum over P1 node blocks (i, j) of T_i^T A_ij T_j; A0 dense column-major of order 6*nagg (zeroed by the caller).
// One 64-thread workgroup per aggregate; its first 36 threads own the (r, c) entries of that aggregate's 6-row band
// and walk the aggregate's node blocks in storage order, so every entry is summed in a fixed order by one thread: no
// atomics, bit-reproducible.
__global__ void __launch_bounds__(64) k_coarse_matrix(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const double *__restrict__ aval,
                                                     const double *__restrict__ tmat, uint32_t npts, uint32_t agg_size, uint32_t nagg, double *__restrict__ a0) {
    const uint32_t ai = blockIdx.x, t = threadIdx.x;
    if (t >= 36) return;
    const uint32_t r = t / 6, c = t % 6;
    const size_t n0 = size_t(6) * nagg;
    const uint32_t i0 = ai * agg_size, i1 = ai == nagg - 1 ? npts : (ai + 1) * agg_size;
    for (uint32_t i = i0; i < i1; ++i) {
        const double *ti = tmat + 18 * size_t(i);
        const double t0 = ti[r], t1 = ti[6 + r], t2 = ti[12 + r];
        for (uint32_t p = row_ptr[i]; p < row_ptr[i + 1]; ++p) {
            const uint32_t j = col[p];
            const uint32_t aj = min(j / agg_size, nagg - 1);
            const double *a = aval + 9 * size_t(p);
            const double *tj = tmat + 18 * size_t(j);
            // (T_i^T A_ij T_j)[r][c] = sum_k T_i[k][r] * (A_ij T_j)[k][c]
            const double at0 = a[0] * tj[c] + a[1] * tj[6 + c] + a[2] * tj[12 + c];
            const double at1 = a[3] * tj[c] + a[4] * tj[6 + c] + a[5] * tj[12 + c];
            const double at2 = a[6] * tj[c] + a[7] * tj[6 + c] + a[8] * tj[12 + c];
            a0[(size_t(6) * aj + c) * n0 + size_t(6) * ai + r] += t0 * at0 + t1 * at1 + t2 * at2;
        }
    }
}

__global__ void k_fix_coarse_diag(double *__restrict__ a0, uint32_t n0, double rel) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n0) return;
    double &v = a0[size_t(d) * n0 + d];
    v = v > 0 ? v * (1.0 + rel) : 1.0; // empty rigid-body column (degenerate aggregate): decouple it
}

// ------------------------------------------------------------------------------------------------------------
namespace {
template<int NN>
void build_level(mh_context *ctx, CubTemp &tmp, const uint32_t *elem, uint32_t nt, uint32_t nnodes, const double *basis, const double *tables_dev,
                 const mh_material &mat, BsrLevel &lvl) {
    const size_t npairs = size_t(nt) * NN * NN;
    DevArray<uint64_t> keys(ctx, npairs), keys_s(ctx, npairs);
    DevArray<uint32_t> pay(ctx, npairs), pay_s(ctx, npairs), head(ctx, npairs), incl(ctx, npairs);
    k_pairs<NN><<<div_up(npairs, TB), TB, 0, ctx->stream>>>(elem, nt, nnodes, keys, pay);
    KERNEL_CHECK();
    sort_pairs(ctx, tmp, keys.get(), keys_s.get(), pay.get(), pay_s.get(), npairs, bit_width64(uint64_t(nnodes) * nnodes));
    k_heads<<<div_up(npairs, TB), TB, 0, ctx->stream>>>(keys_s.get(), npairs, head.get());
    KERNEL_CHECK();
    inclusive_sum(ctx, tmp, head, incl, npairs);
    const uint32_t nb = read_u32(ctx, incl.get() + npairs - 1);
    lvl.n_nodes = nnodes;
    lvl.n_blocks = nb;
    lvl.col.reset(ctx, nb);
    lvl.row_ptr.reset(ctx, size_t(nnodes) + 1);
    lvl.kval.reset(ctx, size_t(nb) * 9);
    lvl.mval.reset(ctx, nb);
    DevArray<uint32_t> blk_row(ctx, nb), seg(ctx, size_t(nb) + 1);
    k_block_index<<<div_up(npairs + 1, TB), TB, 0, ctx->stream>>>(keys_s, head, incl, npairs, nnodes, lvl.col, blk_row, seg);
    KERNEL_CHECK();
    k_segment_ptr<uint32_t><<<div_up(size_t(nnodes) + 1, TB), TB, 0, ctx->stream>>>(blk_row.get(), nb, nnodes, lvl.row_ptr.get());
    KERNEL_CHECK();
    const double lambda = (mat.poisson_ratio * mat.young_modulus) / ((1 + mat.poisson_ratio) * (1 - 2 * mat.poisson_ratio));
    const double mu = mat.young_modulus / (2 * (1 + mat.poisson_ratio));
    {
        // SURVEY 8d's count for the assembly: per tet 16 B corner ids + 4 x 24 B coordinates + 40 B node ids read (the
        // element bases are built from them), 80 B (9 K values + 1 M value) written per node block
        std::optional<TimedLaunch> timed; // the quadratic level's launch is the one the roofline object reports
        if (NN == 10) timed.emplace(ctx, MH_KERNEL_ASSEMBLY, 152.0 * double(nt) + 80.0 * double(nb));
        // The row-wise form (k_assemble_rows: balanced contributor lanes, LDS-staged element data, coalesced block stores) was
        // built to replace the one-thread-per-block form and measured beside it (tools/ab_assembly.sh, S100k): 426 us against
        // 327 us -- its LDS round trips and fences cost more than the imbalance and the strided stores they remove.  It stays
        // selectable (MH_ASSEMBLE_BY_ROW=1); both produce the same bits.
        static const bool by_block = !(getenv("MH_ASSEMBLE_BY_ROW") && atoi(getenv("MH_ASSEMBLE_BY_ROW")) != 0);
        if (by_block) k_assemble<NN><<<div_up(nb, TB), TB, 0, ctx->stream>>>(seg, pay_s, nb, basis, tables_dev, mat.density, lambda, mu, lvl.kval, lvl.mval);
        else
            k_assemble_rows<NN><<<div_up(nnodes, 4), 256, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, seg, pay_s, nnodes, basis, tables_dev, mat.density, lambda, mu, lvl.kval,
                                                                           lvl.mval);
    }
    KERNEL_CHECK();
}

// Exact unit-volume integrals of the shape-function products (GetQuadBasis, mesh2modes.cpp:209-237), in closed form.
// With I1 = int l_i = 1/4, I2(i,i) = 1/10, I2(i,j) = 1/20, I3 and I4 from the same factorial formula
// int l^e dV / V = 6 prod(e!) / (sum(e) + 3)!.
double bary_integral(const int e[4]) {
    static const double fact[] = {1, 1, 2, 6, 24, 120, 720, 5040};
    return 6.0 * fact[e[0]] * fact[e[1]] * fact[e[2]] * fact[e[3]] / fact[e[0] + e[1] + e[2] + e[3] + 3];
}
struct Poly { // sum of coeff * l^exp terms
    std::vector<std::pair<double, std::array<int, 4>>> terms;
};
double integrate(const Poly &a, const Poly &b) {
    double s = 0;
    for (auto &ta : a.terms)
        for (auto &tb : b.terms) {
            int e[4];
            for (int i = 0; i < 4; ++i) e[i] = ta.second[i] + tb.second[i];
            s += ta.first * tb.first * bary_integral(e);
        }
    return s;
}
void quad_tables(std::vector<double> &t) { // mass[10][10], grad[10][4][10][4]
    const int EC[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    auto unit = [](int i) { std::array<int, 4> u{0, 0, 0, 0}; u[i] = 1; return u; };
    Poly n[10], dn[10][4];
    for (int i = 0; i < 4; ++i) {
        auto u2 = unit(i); u2[i] = 2;
        n[i].terms = {{2.0, u2}, {-1.0, unit(i)}}; // l(2l - 1)
        dn[i][i].terms = {{4.0, unit(i)}, {-1.0, {0, 0, 0, 0}}};
    }
    for (int e = 0; e < 6; ++e) {
        const int i = EC[e][0], j = EC[e][1];
        auto u = unit(i); u[j] += 1;
        n[4 + e].terms = {{4.0, u}}; // 4 l_i l_j
        dn[4 + e][i].terms = {{4.0, unit(j)}};
        dn[4 + e][j].terms = {{4.0, unit(i)}};
    }
    t.assign(100 + 1600, 0.0);
    for (int a = 0; a < 10; ++a)
        for (int c = 0; c < 10; ++c) {
            t[a * 10 + c] = integrate(n[a], n[c]);
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < 4; ++l)
                    if (!dn[a][k].terms.empty() && !dn[c][l].terms.empty()) t[100 + ((a * 4 + k) * 10 + c) * 4 + l] = integrate(dn[a][k], dn[c][l]);
        }
}
void linear_tables(std::vector<double> &t) { // mass[4][4] = (1 + delta)/20, grad[a][k][c][l] = delta_ak delta_cl
    t.assign(16 + 256, 0.0);
    for (int a = 0; a < 4; ++a)
        for (int c = 0; c < 4; ++c) {
            t[a * 4 + c] = (a == c ? 2.0 : 1.0) / 20.0;
            t[16 + ((a * 4 + a) * 4 + c) * 4 + c] = 1.0;
        }
}
} // namespace

void mh_build_system(mh_context *ctx, const mh_mesh *mesh, const mh_material &mat, mh_system *sys) {
    CubTemp tmp;
    hipStream_t st = ctx->stream;
    sys->ctx = ctx;
    sys->material = mat;
    const uint32_t npts = mesh->n_points, nt_in = mesh->n_tets;
    if (npts == 0 || nt_in == 0) mh_throw(MH_EEMPTY, "empty tet mesh");
    sys->n_points = npts;

    // --- FilterDegenerate
    DevArray<uint32_t> flag(ctx, nt_in), incl(ctx, nt_in);
    k_flag_tets<<<div_up(nt_in, TB), TB, 0, st>>>(mesh->points, mesh->tets, nt_in, flag);
    KERNEL_CHECK();
    inclusive_sum(ctx, tmp, flag, incl, nt_in);
    const uint32_t nt = read_u32(ctx, incl.get() + nt_in - 1);
    if (nt == 0) mh_throw(MH_EEMPTY, "every tet is degenerate");
    sys->kept_tets = nt;
    DevArray<uint32_t> tets(ctx, size_t(nt) * 4);
    k_compact_tets<<<div_up(nt_in, TB), TB, 0, st>>>(mesh->tets, flag, incl, nt_in, tets);
    KERNEL_CHECK();

    // --- BuildQuadMesh: unique edges ranked by first encounter
    const uint32_t nek = nt * 6;
    DevArray<uint64_t> ekeys(ctx, nek), ekeys_s(ctx, nek);
    DevArray<uint32_t> eidx(ctx, nek), eidx_s(ctx, nek), ehead(ctx, nek), eincl(ctx, nek);
    k_edge_keys<<<div_up(nek, TB), TB, 0, st>>>(tets, nt, ekeys, eidx);
    KERNEL_CHECK();
    sort_pairs(ctx, tmp, ekeys.get(), ekeys_s.get(), eidx.get(), eidx_s.get(), nek, 32 + bit_width64(npts));
    k_heads<<<div_up(nek, TB), TB, 0, st>>>(ekeys_s.get(), size_t(nek), ehead.get());
    KERNEL_CHECK();
    inclusive_sum(ctx, tmp, ehead, eincl, nek);
    const uint32_t ne = read_u32(ctx, eincl.get() + nek - 1);
    sys->n_edges = ne;
    const uint32_t nn = npts + ne;
    sys->n_nodes = nn;
    DevArray<uint32_t> first_idx(ctx, ne), first_s(ctx, ne), iota(ctx, ne), u_sorted(ctx, ne), rank_of(ctx, ne);
    DevArray<uint64_t> edge_key(ctx, ne), edge_key_by_rank(ctx, ne);
    k_unique_edges<<<div_up(nek, TB), TB, 0, st>>>(ekeys_s, eidx_s, ehead, eincl, nek, first_idx, edge_key, iota);
    KERNEL_CHECK();
    sort_pairs(ctx, tmp, first_idx.get(), first_s.get(), iota.get(), u_sorted.get(), ne, bit_width64(nek));
    k_scatter_rank<<<div_up(ne, TB), TB, 0, st>>>(u_sorted, ne, rank_of);
    KERNEL_CHECK();
    sys->elem_nodes_ref.reset(ctx, size_t(nt) * 10);
    k_assign_nodes<<<div_up(nek, TB), TB, 0, st>>>(tets, eidx_s, eincl, rank_of, nt, npts, sys->elem_nodes_ref);
    KERNEL_CHECK();
    k_edge_key_by_rank<<<div_up(ne, TB), TB, 0, st>>>(edge_key, rank_of, ne, edge_key_by_rank);
    KERNEL_CHECK();

    // --- internal Morton numbering
    std::vector<double> hp(size_t(npts) * 3);
    HIP_CHECK(hipMemcpyAsync(hp.data(), mesh->points.get(), hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    double lo[3] = {hp[0], hp[1], hp[2]}, hi[3] = {hp[0], hp[1], hp[2]};
    for (uint32_t i = 0; i < npts; ++i)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], hp[3 * size_t(i) + d]);
            hi[d] = std::max(hi[d], hp[3 * size_t(i) + d]);
        }
    const double extent = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300});
    DevArray<double> xyz_ref(ctx, size_t(nn) * 3);
    DevArray<uint64_t> mkey(ctx, nn), mkey_s(ctx, nn);
    DevArray<uint32_t> ids(ctx, nn);
    sys->perm.reset(ctx, nn);
    sys->inv_perm.reset(ctx, nn);
    k_node_xyz_keys<<<div_up(nn, TB), TB, 0, st>>>(mesh->points, npts, edge_key, rank_of, ne, double3{lo[0], lo[1], lo[2]}, 1.0 / extent, xyz_ref, mkey, ids);
    KERNEL_CHECK();
    sort_pairs(ctx, tmp, mkey.get(), mkey_s.get(), ids.get(), sys->perm.get(), nn, 63);
    DevArray<uint32_t> is_corner(ctx, nn), corner_incl(ctx, nn);
    k_invert_perm<<<div_up(nn, TB), TB, 0, st>>>(sys->perm, nn, npts, sys->inv_perm, is_corner);
    KERNEL_CHECK();
    inclusive_sum(ctx, tmp, is_corner, corner_incl, nn);
    sys->node_xyz.reset(ctx, size_t(nn) * 3);
    sys->parent_a.reset(ctx, nn);
    sys->parent_b.reset(ctx, nn);
    sys->p1_corner.reset(ctx, npts);
    sys->p1_xyz.reset(ctx, size_t(npts) * 3);
    k_internal_nodes<<<div_up(nn, TB), TB, 0, st>>>(sys->perm, sys->inv_perm, corner_incl, xyz_ref, edge_key_by_rank, nn, npts, sys->node_xyz,
                                                     sys->parent_a, sys->parent_b, sys->p1_corner, sys->p1_xyz);
    KERNEL_CHECK();
    sys->elem_nodes.reset(ctx, size_t(nt) * 10);
    DevArray<uint32_t> elem_p1(ctx, size_t(nt) * 4);
    k_renumber_elements<<<div_up(size_t(nt) * 10, TB), TB, 0, st>>>(sys->elem_nodes_ref, sys->inv_perm, corner_incl, nt, sys->elem_nodes, elem_p1);
    KERNEL_CHECK();
    // transposed interpolation lists
    {
        DevArray<uint32_t> k2(ctx, size_t(ne) * 2), v2(ctx, size_t(ne) * 2), k2s(ctx, size_t(ne) * 2);
        sys->p1_edge_mid.reset(ctx, size_t(ne) * 2);
        sys->p1_edge_ptr.reset(ctx, size_t(npts) + 1);
        k_edge_incidence<<<div_up(nn, TB), TB, 0, st>>>(sys->parent_a, sys->parent_b, sys->perm, nn, npts, k2, v2);
        KERNEL_CHECK();
        sort_pairs(ctx, tmp, k2.get(), k2s.get(), v2.get(), sys->p1_edge_mid.get(), size_t(ne) * 2, bit_width64(npts));
        k_segment_ptr<uint32_t><<<div_up(size_t(npts) + 1, TB), TB, 0, st>>>(k2s.get(), size_t(ne) * 2, npts, sys->p1_edge_ptr.get());
        KERNEL_CHECK();
    }

    // --- element bases, tables, patterns, assembly (P2 and its Galerkin P1 coarse operator)
    sys->elem_basis.reset(ctx, size_t(nt) * EB);
    k_element_basis<<<div_up(nt, TB), TB, 0, st>>>(mesh->points, tets, nt, sys->elem_basis);
    KERNEL_CHECK();
    std::vector<double> tq, tl;
    quad_tables(tq);
    linear_tables(tl);
    DevArray<double> tq_dev(ctx, tq.size()), tl_dev(ctx, tl.size());
    tq_dev.upload(tq.data(), tq.size());
    tl_dev.upload(tl.data(), tl.size());
    sys->L2.id = 2;
    sys->L1.id = 1;
    build_level<10>(ctx, tmp, sys->elem_nodes, nt, nn, sys->elem_basis, tq_dev, mat, sys->L2);
    build_level<4>(ctx, tmp, elem_p1, nt, npts, sys->elem_basis, tl_dev, mat, sys->L1);

    // --- rigid-body aggregates over runs of consecutive (Morton-ordered) P1 nodes
    if (const char *e = getenv("MH_AGG")) sys->agg_size = std::max(2, atoi(e));
    // The coarse operator is dense of order 6 n_agg and inverted explicitly (O(n0^2) memory, 2 n0^3 flops): with a fixed
    // aggregate size it would grow with the mesh (33 k at a million tets: 9 GB and 7e13 flops per set-up).  Aggregates
    // grow instead so that the order stays at or below MaxCoarseOrder (3 690 at the 100k-tet metric mesh is untouched);
    // larger aggregates make a weaker coarse correction (more iterations), never a failure.
    constexpr uint32_t MaxCoarseOrder = 6144;
    if (uint64_t(6) * (npts / sys->agg_size) > MaxCoarseOrder) sys->agg_size = uint32_t((uint64_t(6) * npts + MaxCoarseOrder - 1) / MaxCoarseOrder);
    sys->n_agg = std::max(1u, npts / sys->agg_size);
    sys->agg_t.reset(ctx, size_t(npts) * 18);
    k_aggregate_t<<<div_up(sys->n_agg, 64), 64, 0, st>>>(sys->p1_xyz, npts, sys->agg_size, sys->n_agg, sys->agg_t);
    KERNEL_CHECK();
    sys->points.reset(ctx, size_t(npts) * 3);
    HIP_CHECK(hipMemcpyAsync(sys->points.get(), mesh->points.get(), size_t(npts) * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st)); // host-side staging vectors (tables) must outlive their uploads
    sys->hierarchy_ready = false;
}
