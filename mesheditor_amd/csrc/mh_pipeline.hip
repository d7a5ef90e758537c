// Mesh -> matrices, entirely on the device (gfx950).
//
//   FilterDegenerate      reference src/audio/mesh2modes.cpp:42-60   -> k_flag_tets + scan + compaction
//   BuildQuadMesh         :246-264  midside ids in first-encounter order -> sort edge keys, rank by first occurrence
//   ComputeElementBases   :137-165  -> k_element_basis (volume + four gradients per tet, one 128-byte line)
//   AssembleQuadratic     :273-327  -> sorted (row node, col node) pair list = sparsity pattern + per-block
//                                      contributor lists; one lane evaluates each contribution, the contributions
//                                      of a block are staged in LDS and summed in list order by one thread
//                                      (no atomics, bit-reproducible); finished blocks leave as coalesced runs.
// Nodes are renumbered internally along a Morton curve so that the SpMM's gathers of x hit L2; results are mapped
// back to the reference's numbering at the boundary.  The P1 (corner-node) operator is the exact Galerkin coarse
// operator of the P2 one (P1 is a subspace of P2), so it is assembled directly by the same kernel with linear tables.
#include "mh_common.h"

#include <optional>

#include <hipcub/hipcub.hpp>

namespace {
constexpr int TB = 256;
constexpr int EB = 16; // doubles per element-basis row = one 128-byte line: four barycentric gradients as (x, y, z, volume) quadruples

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
                                uint32_t ne, double3 lo, double inv_extent, double3 origin, double *__restrict__ xyz_ref, uint64_t *__restrict__ mkey, uint32_t *__restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts + ne) return;
    double x, y, z;
    uint32_t node;
    // (origin: zero, or the bounding box's corner for a body far from the origin -- mh_build_system.  A point minus the corner is exact where the
    // two are within a factor of two, and rounds to eps of the DIFFERENCE otherwise: the node coordinates keep the digits the caller's own carry)
    if (i < npts) {
        node = i;
        x = pts[3 * size_t(i)] - origin.x; y = pts[3 * size_t(i) + 1] - origin.y; z = pts[3 * size_t(i) + 2] - origin.z;
    } else {
        const uint32_t u = i - npts;
        node = npts + rank_of[u];
        const uint32_t a = uint32_t(edge_key[u] >> 32), b = uint32_t(edge_key[u]);
        x = 0.5 * ((pts[3 * size_t(a)] - origin.x) + (pts[3 * size_t(b)] - origin.x));
        y = 0.5 * ((pts[3 * size_t(a) + 1] - origin.y) + (pts[3 * size_t(b) + 1] - origin.y));
        z = 0.5 * ((pts[3 * size_t(a) + 2] - origin.z) + (pts[3 * size_t(b) + 2] - origin.z));
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
__global__ void k_element_basis(const double *__restrict__ pts, const uint32_t *__restrict__ tets, uint32_t nt, double3 origin, double *__restrict__ basis) {
    const uint32_t el = blockIdx.x * blockDim.x + threadIdx.x;
    if (el >= nt) return;
    double v[4][3];
    // (origin: zero unless the body is far from the origin of its coordinates -- mh_build_system.  The cofactors below are sums of PRODUCTS of
    // coordinates, as the reference forms them: at 100 m they cancel eight digits and the gradients no longer sum to zero -- K then has no exact
    // rigid-body null space, which a factorisation does not notice and the block iteration's rigid-body columns do)
    const double o[3] = {origin.x, origin.y, origin.z};
    for (int a = 0; a < 4; ++a) {
        const uint32_t id = tets[4 * size_t(el) + a];
        for (int d = 0; d < 3; ++d) v[a][d] = pts[3 * size_t(id) + d] - o[d];
    }
    // det = dot(d - a, cross(b - a, c - a))  (GetTetDeterminant, mesh2modes.cpp:64-66)
    const double bx = v[1][0] - v[0][0], by = v[1][1] - v[0][1], bz = v[1][2] - v[0][2];
    const double cx = v[2][0] - v[0][0], cy = v[2][1] - v[0][1], cz = v[2][2] - v[0][2];
    const double dx = v[3][0] - v[0][0], dy = v[3][1] - v[0][1], dz = v[3][2] - v[0][2];
    const double det = dx * (by * cz - cy * bz) + dy * (bz * cx - cz * bx) + dz * (bx * cy - cx * by);
    double *out = basis + EB * size_t(el); // gradient i at out[4 i .. 4 i + 2], the volume beside each one (whichever gradient a lane loads brings it along)
    for (int i = 0; i < 4; ++i) out[4 * i + 3] = fabs(det / 6);
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
            out[4 * i + j] = sign * (crx + cry + crz) / det;
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

// The barycentric gradients a shape function's gradient is made of: a corner function's is a multiple of its own corner's,
// a midside function's a combination of its edge's two (edges in the reference's order 01 02 03 12 13 23).
__device__ __forceinline__ uint32_t support0(uint32_t a) { return uint32_t(0x2110003210ull >> (4 * a)) & 3u; }
__device__ __forceinline__ uint32_t support1(uint32_t a) { return uint32_t(0x3323213210ull >> (4 * a)) & 3u; }

// K/M assembly, one LANE PER CONTRIBUTION.  The contributor list (element, a, c) is sorted by node block, so a block's
// contributions are a contiguous run and a workgroup's TB blocks own one contiguous stretch of the list.
//   * Lanes walk that stretch TB contributions at a time: each evaluates ONE 3 x 3 stiffness contribution and one mass
//     value -- perfectly balanced, where one-thread-per-block leaves a wave waiting for the lane of a diagonal block
//     (6 .. 30 elements around its node, against one or two for most of its neighbours).
//   * A contribution needs only the two gradients in the support of a and the two in the support of c:
//         G = sum_{x, y in 0..1} W[a][c][x][y]  g_(s_a(x)) (x) g_(s_c(y)),   W = the integrals of the gradient coefficients,
//     four weights per (a, c) in LDS (the full table is 4 x 4 per pair with at most these four non-zero), the gradients as
//     four 32-byte reads from the element's 128-byte basis line.  ~70 fp64 instructions, no data-dependent branch.
//   * The TB contributions of a round go through LDS (ten doubles per entry, 16-byte accesses); the lane that owns a block adds
//     the entries of its run in list order -- every block is summed in one fixed order by one thread: no atomics,
//     bit-reproducible.  The next round's list entry is requested a round ahead.
//   * The finished blocks leave through LDS as whole runs of doubles (coalesced 2 KB stores per wave).
// tables: mass[NN][NN] then W[NN][NN][2][2].
template<int NN> __global__ void __launch_bounds__(TB) k_assemble_flat(const uint32_t *__restrict__ seg, const uint32_t *__restrict__ payload, uint32_t nblocks,
                                                                               const double *__restrict__ basis, const double *__restrict__ tables, double rho,
                                                                               double lambda, double mu, double *__restrict__ kval, double *__restrict__ mval) {
    constexpr int NT = NN * NN * 5;
    __shared__ double s_tab[NT];
    __shared__ __attribute__((aligned(16))) double s_c[10 * TB];
    for (int i = threadIdx.x; i < NT; i += TB) s_tab[i] = tables[i];
    const double *s_mass = s_tab;
    const double2 *s_w = reinterpret_cast<const double2 *>(s_tab + NN * NN);
    double2 *s_e = reinterpret_cast<double2 *>(s_c); // entry j = s_e[5 j .. 5 j + 4]
    const uint32_t tid = threadIdx.x, first = blockIdx.x * TB, here = min(uint32_t(TB), nblocks - first);
    const bool owner = tid < here;
    const uint32_t mine0 = owner ? seg[first + tid] : 0u, mine1 = owner ? seg[first + tid + 1] : 0u;
    const uint32_t q0 = seg[first], q1 = seg[first + here];
    uint32_t pl0 = q0 + tid < q1 ? payload[q0 + tid] : 0u; // lanes past the end of the stretch work on entry 0; nobody reads what they stage
    double k[9] = {}, m = 0;
    __syncthreads();
    for (uint32_t r = q0; r < q1; r += TB) {
        const uint32_t qn = r + TB + tid;
        const uint32_t pl1 = qn < q1 ? payload[qn] : 0u;
        {
            const uint32_t t = pl0 / (NN * NN), ac = pl0 % (NN * NN), a = ac / NN, c = ac % NN;
            const double2 *line = reinterpret_cast<const double2 *>(basis + EB * size_t(t));
            const uint32_t a0 = support0(a), a1 = support1(a), c0 = support0(c), c1 = support1(c);
            const double2 ga0 = line[2 * a0], ga0z = line[2 * a0 + 1], ga1 = line[2 * a1], ga1z = line[2 * a1 + 1];
            const double2 gc0 = line[2 * c0], gc0z = line[2 * c0 + 1], gc1 = line[2 * c1], gc1z = line[2 * c1 + 1];
            const double2 w0 = s_w[2 * ac], w1 = s_w[2 * ac + 1]; // W[0][0], W[0][1]; W[1][0], W[1][1]
            const double vol = ga0z.y;
            const double h0[3] = {w0.x * gc0.x + w0.y * gc1.x, w0.x * gc0.y + w0.y * gc1.y, w0.x * gc0z.x + w0.y * gc1z.x};
            const double h1[3] = {w1.x * gc0.x + w1.y * gc1.x, w1.x * gc0.y + w1.y * gc1.y, w1.x * gc0z.x + w1.y * gc1z.x};
            const double u0[3] = {ga0.x, ga0.y, ga0z.x}, u1[3] = {ga1.x, ga1.y, ga1z.x};
            double g[3][3], v[10];
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) g[pp][qq] = u0[pp] * h0[qq] + u1[pp] * h1[qq];
            const double trace = g[0][0] + g[1][1] + g[2][2];
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) v[3 * pp + qq] = vol * (lambda * g[pp][qq] + mu * g[qq][pp] + (pp == qq ? mu * trace : 0.0));
            v[9] = rho * vol * s_mass[ac];
#pragma unroll
            for (int e = 0; e < 5; ++e) s_e[5 * tid + e] = double2{v[2 * e], v[2 * e + 1]};
        }
        __syncthreads();
        const uint32_t lo = max(mine0, r), hi = min(mine1, r + TB);
        {
            auto add = [&](const double2 &e0, const double2 &e1, const double2 &e2, const double2 &e3, const double2 &e4) {
                k[0] += e0.x, k[1] += e0.y, k[2] += e1.x, k[3] += e1.y, k[4] += e2.x, k[5] += e2.y, k[6] += e3.x, k[7] += e3.y, k[8] += e4.x, m += e4.y;
            };
            // most runs have one or two entries: those are read together (the second read does not wait for the first sum),
            // the rest in a loop; always added in list order
            if (lo < hi) {
                const uint32_t j = lo - r, j2 = lo + 1 < hi ? j + 1 : j;
                const double2 e0 = s_e[5 * j], e1 = s_e[5 * j + 1], e2 = s_e[5 * j + 2], e3 = s_e[5 * j + 3], e4 = s_e[5 * j + 4];
                const double2 f0 = s_e[5 * j2], f1 = s_e[5 * j2 + 1], f2 = s_e[5 * j2 + 2], f3 = s_e[5 * j2 + 3], f4 = s_e[5 * j2 + 4];
                add(e0, e1, e2, e3, e4);
                if (lo + 1 < hi) add(f0, f1, f2, f3, f4);
            }
            for (uint32_t qq = lo + 2; qq < hi; ++qq) {
                const uint32_t j = qq - r;
                add(s_e[5 * j], s_e[5 * j + 1], s_e[5 * j + 2], s_e[5 * j + 3], s_e[5 * j + 4]);
            }
        }
        __syncthreads();
        pl0 = pl1;
    }
    // the finished blocks: stride 9 words of 8 bytes into LDS (conflict-free), out as whole runs
#pragma unroll
    for (int e = 0; e < 9; ++e) s_c[9 * tid + e] = k[e];
    if (owner) mval[first + tid] = m;
    __syncthreads();
    double *out = kval + 9 * size_t(first);
    for (uint32_t f = tid; f < 9 * here; f += TB) out[f] = s_c[f];
}

// The form it replaced, kept for the A/B record (MH_ASSEMBLE_BY_BLOCK=1): one thread per node block walks its own run of
// contributors and contracts the full 4 x 4 gradient table for each (skipping its zeros).
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
            const double2 *src = reinterpret_cast<const double2 *>(basis + EB * size_t(t));
#pragma unroll
            for (int i = 0; i < EB / 2; ++i) {
                const double2 v = src[i];
                eb[2 * i] = v.x, eb[2 * i + 1] = v.y;
            }
        }
        const double vol = eb[3];
        m += rho * vol * s_mass[a * NN + c];
        double g[3][3] = {};
        for (int kk = 0; kk < 4; ++kk) {
            for (int ll = 0; ll < 4; ++ll) {
                const double w = s_grad[((a * 4 + kk) * NN + c) * 4 + ll];
                if (w == 0) continue;
                for (int pp = 0; pp < 3; ++pp)
                    for (int qq = 0; qq < 3; ++qq) g[pp][qq] += w * (eb[4 * kk + pp] * eb[4 * ll + qq]);
            }
        }
        const double trace = g[0][0] + g[1][1] + g[2][2];
        for (int pp = 0; pp < 3; ++pp)
            for (int qq = 0; qq < 3; ++qq) k[pp][qq] += vol * (lambda * g[pp][qq] + mu * g[qq][pp] + (pp == qq ? mu * trace : 0.0));
    }
    for (int pp = 0; pp < 3; ++pp)
        for (int qq = 0; qq < 3; ++qq) s_out[9 * threadIdx.x + 3 * pp + qq] = k[pp][qq];
    if (valid) mval[b] = m;
    __syncthreads();
    const size_t first = size_t(blockIdx.x) * TB;
    const uint32_t count = uint32_t(min(size_t(TB), size_t(nblocks) - first)) * 9;
    for (uint32_t f = threadIdx.x; f < count; f += TB) kval[9 * first + f] = s_out[f];
}

// ---- level 0: rigid-body aggregates ------------------------------------------------------------------------
__global__ void k_aggregate_t(const double *__restrict__ p1_xyz, const uint32_t *__restrict__ agg_ptr, const uint32_t *__restrict__ agg_nodes, uint32_t nagg,
                              double *__restrict__ tmat) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nagg) return;
    const uint32_t l0 = agg_ptr[a], l1 = agg_ptr[a + 1];
    double c[3] = {0, 0, 0};
    for (uint32_t l = l0; l < l1; ++l)
        for (int d = 0; d < 3; ++d) c[d] += p1_xyz[3 * size_t(agg_nodes[l]) + d];
    const double cnt = double(l1 - l0);
    for (int d = 0; d < 3; ++d) c[d] /= cnt;
    double rn[3] = {0, 0, 0}; // squared norms of the three rotation columns
    for (uint32_t l = l0; l < l1; ++l) {
        const size_t i = agg_nodes[l];
        const double rx = p1_xyz[3 * i] - c[0], ry = p1_xyz[3 * i + 1] - c[1], rz = p1_xyz[3 * i + 2] - c[2];
        rn[0] += ry * ry + rz * rz; // |e_x x r|^2
        rn[1] += rx * rx + rz * rz;
        rn[2] += rx * rx + ry * ry;
    }
    const double st = 1.0 / sqrt(cnt);
    double sr[3];
    for (int q = 0; q < 3; ++q) sr[q] = rn[q] > 1e-300 ? 1.0 / sqrt(rn[q]) : 0.0;
    for (uint32_t l = l0; l < l1; ++l) {
        const size_t i = agg_nodes[l];
        const double rx = p1_xyz[3 * i] - c[0], ry = p1_xyz[3 * i + 1] - c[1], rz = p1_xyz[3 * i + 2] - c[2];
        double *t = tmat + 18 * i; // row-major 3 x 6
        for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) t[6 * p + q] = p == q ? st : 0.0;
        // columns 3..5: e_q x r
        t[6 * 0 + 3] = 0;            t[6 * 1 + 3] = -rz * sr[0]; t[6 * 2 + 3] = ry * sr[0];
        t[6 * 0 + 4] = rz * sr[1];   t[6 * 1 + 4] = 0;           t[6 * 2 + 4] = -rx * sr[1];
        t[6 * 0 + 5] = -ry * sr[2];  t[6 * 1 + 5] = rx * sr[2];  t[6 * 2 + 5] = 0;
    }
}

// Aggregates of P1 nodes for the rigid-body level, grown on the graph of the P1 operator (host index work at set-up, O(blocks)):
//   1. every node, in internal (Morton) order, whose neighbours are all still free founds an aggregate with them;
//   2. a leftover node joins the neighbouring aggregate it has the most connections to (as aggregates stood after step 1);
//   3. what is still free founds aggregates with its free neighbours; aggregates of fewer than four nodes (their rigid-body
//      columns would be dependent) are merged into their most connected neighbour;
//   4. pairwise merging (each aggregate with the free neighbour it shares the most node connections with) while the mean
//      size is well below `target` or the coarse order 6 n_agg exceeds `max_order`.
// Every aggregate is a CONNECTED node set.  Runs of consecutive Morton nodes (rounds 1-2) are compact on a structured grid, but
// on a scanned, thin-walled body they string together nodes from opposite faces and unrelated parts: the rigid-body modes of
// such a set are no coarse space (measured: the cycle's condition number 128 against 39 on an 8k-tet skillet scan, and no
// convergence in 300 iterations at 95k tets).  Deterministic: ties go to the lowest id.
uint32_t graph_aggregates(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col, uint32_t n, uint32_t target, uint32_t max_order,
                          std::vector<uint32_t> &agg_of) {
    constexpr uint32_t FREE = UINT32_MAX;
    agg_of.assign(n, FREE);
    uint32_t na = 0;
    for (uint32_t i = 0; i < n; ++i) { // 1
        if (agg_of[i] != FREE) continue;
        bool all_free = row_ptr[i + 1] > row_ptr[i];
        for (uint32_t p = row_ptr[i]; p < row_ptr[i + 1] && all_free; ++p) all_free = agg_of[col[p]] == FREE;
        if (!all_free) continue;
        for (uint32_t p = row_ptr[i]; p < row_ptr[i + 1]; ++p) agg_of[col[p]] = na; // (the row holds i itself: the diagonal block)
        agg_of[i] = na++;
    }
    { // 2
        const std::vector<uint32_t> founded = agg_of;
        std::vector<std::pair<uint32_t, uint32_t>> votes;
        for (uint32_t i = 0; i < n; ++i) {
            if (founded[i] != FREE) continue;
            votes.clear();
            for (uint32_t p = row_ptr[i]; p < row_ptr[i + 1]; ++p)
                if (founded[col[p]] != FREE) votes.emplace_back(founded[col[p]], 1u);
            if (votes.empty()) continue;
            std::sort(votes.begin(), votes.end());
            uint32_t best = votes[0].first, best_count = 0;
            for (size_t k = 0; k < votes.size();) {
                size_t e = k;
                while (e < votes.size() && votes[e].first == votes[k].first) ++e;
                if (uint32_t(e - k) > best_count) best_count = uint32_t(e - k), best = votes[k].first;
                k = e;
            }
            agg_of[i] = best;
        }
    }
    for (uint32_t i = 0; i < n; ++i) { // 3
        if (agg_of[i] != FREE) continue;
        for (uint32_t p = row_ptr[i]; p < row_ptr[i + 1]; ++p)
            if (agg_of[col[p]] == FREE) agg_of[col[p]] = na;
        agg_of[i] = na++;
    }
    if (na == 0) return 0;
    // aggregate graph: connection counts between aggregates
    std::vector<uint32_t> size;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> links; // per aggregate: (neighbour aggregate, node connections), ascending
    const auto rebuild = [&]() {
        size.assign(na, 0);
        for (uint32_t i = 0; i < n; ++i) ++size[agg_of[i]];
        std::vector<uint64_t> pairs;
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t p = row_ptr[i]; p < row_ptr[i + 1]; ++p)
                if (agg_of[col[p]] != agg_of[i]) pairs.push_back((uint64_t(agg_of[i]) << 32) | agg_of[col[p]]);
        std::sort(pairs.begin(), pairs.end());
        links.assign(na, {});
        for (size_t k = 0; k < pairs.size();) {
            size_t e = k;
            while (e < pairs.size() && pairs[e] == pairs[k]) ++e;
            links[pairs[k] >> 32].emplace_back(uint32_t(pairs[k]), uint32_t(e - k));
            k = e;
        }
    };
    const auto renumber = [&](std::vector<uint32_t> &into) { // into: aggregate -> merged aggregate (any labels) => dense ids in order of first use
        std::vector<uint32_t> fresh(na, FREE);
        uint32_t k = 0;
        for (uint32_t a = 0; a < na; ++a)
            if (fresh[into[a]] == FREE) fresh[into[a]] = k++;
        for (uint32_t i = 0; i < n; ++i) agg_of[i] = fresh[into[agg_of[i]]];
        na = k;
    };
    // (the aggregate graph costs a sort of every cross-aggregate connection: built only when something has to be merged)
    size.assign(na, 0);
    for (uint32_t i = 0; i < n; ++i) ++size[agg_of[i]];
    const bool any_tiny = std::any_of(size.begin(), size.end(), [](uint32_t s) { return s < 4; });
    const bool merging = uint64_t(n) * 3 < uint64_t(na) * target * 2 || uint64_t(6) * na > max_order;
    if (!any_tiny && !merging) return na;
    rebuild();
    // Small aggregates join their most connected neighbour, repeated until none with a neighbour is left: a one- or two-node
    // aggregate (or three collinear nodes) has linearly dependent rigid-body columns, T^T A T is then singular and the unpivoted
    // coarse elimination sees rounding-level pivots.  Merges are resolved through a union-find, so a chain of
    // tiny aggregates ends up under ONE label whatever order its links are visited in.
    for (int round = 0; round < 8 && std::any_of(size.begin(), size.end(), [](uint32_t s_) { return s_ < 4; }); ++round) {
        std::vector<uint32_t> parent(na);
        for (uint32_t a = 0; a < na; ++a) parent[a] = a;
        const auto find = [&](uint32_t a) {
            while (parent[a] != a) a = parent[a] = parent[parent[a]];
            return a;
        };
        bool any = false;
        for (uint32_t a = 0; a < na; ++a) {
            if (size[a] >= 4 || links[a].empty()) continue;
            uint32_t best = links[a][0].first, count = 0;
            for (const auto &[nb, c] : links[a])
                if (size[nb] >= 4 && c > count) count = c, best = nb;
            if (count == 0) // only tiny neighbours: the most connected of them
                for (const auto &[nb, c] : links[a])
                    if (c > count) count = c, best = nb;
            const uint32_t ra = find(a), rb = find(best);
            if (ra != rb) parent[ra] = rb, any = true;
        }
        if (!any) break;
        std::vector<uint32_t> into(na);
        for (uint32_t a = 0; a < na; ++a) into[a] = find(a);
        renumber(into);
        rebuild();
    }
    for (int round = 0; round < 12; ++round) { // 4
        const bool too_small = uint64_t(n) * 3 < uint64_t(na) * target * 2; // mean size < 2/3 target
        const bool too_many = uint64_t(6) * na > max_order;
        if (na <= 1 || !(too_small || too_many)) break;
        std::vector<uint32_t> into(na, FREE);
        for (uint32_t a = 0; a < na; ++a) {
            if (into[a] != FREE) continue;
            into[a] = a;
            uint32_t best = FREE, count = 0;
            for (const auto &[nb, c] : links[a])
                if (into[nb] == FREE && c > count) count = c, best = nb;
            if (best != FREE) into[best] = a;
        }
        const uint32_t before = na;
        renumber(into);
        if (na == before) break;
        rebuild();
    }
    return na;
}
} // namespace

// (for the lab library and its CPU test: the aggregation is host code, checkable without a device)
uint32_t mh_graph_aggregates(const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col, uint32_t n, uint32_t target, uint32_t max_order, std::vector<uint32_t> &agg_of) {
    return graph_aggregates(row_ptr, col, n, target, max_order, agg_of);
}

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
                                                     const double *__restrict__ tmat, const uint32_t *__restrict__ agg_of, const uint32_t *__restrict__ agg_ptr,
                                                     const uint32_t *__restrict__ agg_nodes, uint32_t nagg, double *__restrict__ a0) {
    const uint32_t ai = blockIdx.x, t = threadIdx.x;
    if (t >= 36) return;
    const uint32_t r = t / 6, c = t % 6;
    const size_t n0 = size_t(6) * nagg;
    for (uint32_t l = agg_ptr[ai]; l < agg_ptr[ai + 1]; ++l) {
        const uint32_t i = agg_nodes[l];
        const double *ti = tmat + 18 * size_t(i);
        const double t0 = ti[r], t1 = ti[6 + r], t2 = ti[12 + r];
        for (uint32_t p = row_ptr[i]; p < row_ptr[i + 1]; ++p) {
            const uint32_t j = col[p];
            const uint32_t aj = agg_of[j];
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
                 const double *support_tables_dev, const mh_material &mat, BsrLevel &lvl) {
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
        // One lane per contribution (k_assemble_flat) against one thread per node block (k_assemble), S100k, tools/ab_assembly.sh:
        // DESIGN.md section 5 has the measured pair.  A third form (one wave per node row, LDS-staged element data) was built in
        // between and measured slower than either (426 us; profiles/r02_ab_assembly.txt); it is gone.
        constexpr bool by_block = false; // round 1's kernel, one thread per node block: 573 us against 171 us
        if (by_block) k_assemble<NN><<<div_up(nb, TB), TB, 0, ctx->stream>>>(seg, pay_s, nb, basis, tables_dev, mat.density, lambda, mu, lvl.kval, lvl.mval);
        else k_assemble_flat<NN><<<div_up(nb, TB), TB, 0, ctx->stream>>>(seg, pay_s, nb, basis, support_tables_dev, mat.density, lambda, mu, lvl.kval, lvl.mval);
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
// mass[NN][NN] then W[NN][NN][2][2]: the entries of grad[a][k][c][l] with k, l in the supports of a and c (k_assemble_flat).
// Everything outside the supports must be an exact zero -- it is for Lagrange elements on barycentric coordinates.
std::vector<double> support_tables(const std::vector<double> &full, int nn) {
    static const int S0[10] = {0, 1, 2, 3, 0, 0, 0, 1, 1, 2}, S1[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
    std::vector<double> out(size_t(nn) * nn * 5, 0.0);
    std::copy(full.begin(), full.begin() + nn * nn, out.begin());
    for (int a = 0; a < nn; ++a)
        for (int c = 0; c < nn; ++c) {
            double *w = out.data() + nn * nn + (a * nn + c) * 4;
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < 4; ++l) {
                    const double v = full[nn * nn + ((a * 4 + k) * nn + c) * 4 + l];
                    const int x = k == S0[a] ? 0 : (k == S1[a] ? 1 : -1), y = l == S0[c] ? 0 : (l == S1[c] ? 1 : -1);
                    if (x < 0 || y < 0) {
                        if (v != 0) mh_throw(MH_EINVAL, "gradient table entry outside the shape functions' supports");
                        continue;
                    }
                    w[2 * x + y] = v;
                }
        }
    return out;
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
    // A body far from the origin (a scene's object at its world position: found by tools/probe/r06_odd_meshes_probe.py -- 100 m away the solve
    // ended on a failed self-check, 1 km away it returned nothing, the oracle's factorisation does not care): every use of the node coordinates
    // downstream is a DIFFERENCE (rigid-body vectors about a centroid, aggregates about theirs, the start block's box, the Morton keys), and a
    // coordinate of 1e3 carries an absolute rounding of 1e-13 into differences of 1e-2 -- a thousand times the floor the iteration's clauses
    // assume.  Farther than four extents from the origin the coordinates are kept relative to the bounding box's corner instead.  (Nearer: as
    // they come, bit for bit what rounds 1-5 computed.)  The element bases are formed from the same relative coordinates (k_element_basis).
    double far = 0;
    for (int d = 0; d < 3; ++d) far = std::max({far, std::fabs(lo[d]), std::fabs(hi[d])});
    const bool relative = far > 4.0 * extent;
    const double3 origin = relative ? double3{lo[0], lo[1], lo[2]} : double3{0, 0, 0};
    const double3 key_lo = relative ? double3{0, 0, 0} : double3{lo[0], lo[1], lo[2]};
    k_node_xyz_keys<<<div_up(nn, TB), TB, 0, st>>>(mesh->points, npts, edge_key, rank_of, ne, key_lo, 1.0 / extent, origin, xyz_ref, mkey, ids);
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
    sys->elem_p1.reset(ctx, size_t(nt) * 4);
    DevArray<uint32_t> &elem_p1 = sys->elem_p1;
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
    k_element_basis<<<div_up(nt, TB), TB, 0, st>>>(mesh->points, tets, nt, origin, sys->elem_basis);
    KERNEL_CHECK();
    std::vector<double> tq, tl;
    quad_tables(tq);
    linear_tables(tl);
    const std::vector<double> sq = support_tables(tq, 10), sl = support_tables(tl, 4);
    DevArray<double> tq_dev(ctx, tq.size()), tl_dev(ctx, tl.size()), sq_dev(ctx, sq.size()), sl_dev(ctx, sl.size());
    tq_dev.upload(tq.data(), tq.size());
    tl_dev.upload(tl.data(), tl.size());
    sq_dev.upload(sq.data(), sq.size());
    sl_dev.upload(sl.data(), sl.size());
    sys->L2.id = 2;
    sys->L1.id = 1;
    build_level<10>(ctx, tmp, sys->elem_nodes, nt, nn, sys->elem_basis, tq_dev, sq_dev, mat, sys->L2);
    build_level<4>(ctx, tmp, elem_p1, nt, npts, sys->elem_basis, tl_dev, sl_dev, mat, sys->L1);

    // --- rigid-body aggregates: connected node sets on the P1 operator's graph (graph_aggregates above)
    if (const char *e = getenv("MH_AGG")) sys->agg_target = uint32_t(std::max(2, atoi(e)));
    // The coarse operator is dense of order 6 n_agg and inverted explicitly (O(n0^2) memory, 2 n0^3 flops): aggregates are
    // merged until the order is at or below MaxCoarseOrder; larger aggregates make a weaker coarse correction (more
    // iterations), never a failure.
    constexpr uint32_t MaxCoarseOrder = 6144u;
    {
        std::vector<uint32_t> rp(size_t(npts) + 1), cl(sys->L1.n_blocks), agg_of;
        HIP_CHECK(hipMemcpyAsync(rp.data(), sys->L1.row_ptr.get(), rp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(cl.data(), sys->L1.col.get(), cl.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        sys->n_agg = graph_aggregates(rp, cl, npts, sys->agg_target, MaxCoarseOrder, agg_of);
        std::vector<uint32_t> ptr(size_t(sys->n_agg) + 1, 0), nodes(npts);
        for (uint32_t i = 0; i < npts; ++i) ++ptr[agg_of[i] + 1];
        for (uint32_t a = 0; a < sys->n_agg; ++a) ptr[a + 1] += ptr[a];
        std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
        for (uint32_t i = 0; i < npts; ++i) nodes[fill[agg_of[i]]++] = i;
        // connected bodies: components of the same graph (nodes no element touches stay with component 0)
        {
            std::vector<uint32_t> comp(npts, UINT32_MAX), stack;
            uint32_t nc = 0;
            for (uint32_t seed = 0; seed < npts; ++seed) {
                if (comp[seed] != UINT32_MAX || rp[seed + 1] == rp[seed]) continue;
                comp[seed] = nc;
                stack.assign(1, seed);
                while (!stack.empty()) {
                    const uint32_t v = stack.back();
                    stack.pop_back();
                    for (uint32_t p = rp[v]; p < rp[v + 1]; ++p)
                        if (comp[cl[p]] == UINT32_MAX) comp[cl[p]] = nc, stack.push_back(cl[p]);
                }
                ++nc;
            }
            sys->n_components = std::max(1u, nc);
            sys->unreferenced_points = 0;
            for (uint32_t i = 0; i < npts; ++i) sys->unreferenced_points += rp[i + 1] == rp[i] ? 1u : 0u;
            std::vector<uint32_t> pa(nn), node_comp(nn);
            std::vector<double> xyz(size_t(nn) * 3), cent(size_t(sys->n_components) * 3, 0.0), count(sys->n_components, 0.0);
            HIP_CHECK(hipMemcpyAsync(pa.data(), sys->parent_a.get(), size_t(nn) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(xyz.data(), sys->node_xyz.get(), xyz.size() * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            for (uint32_t i = 0; i < nn; ++i) {
                const uint32_t c = pa[i] < npts && comp[pa[i]] != UINT32_MAX ? comp[pa[i]] : 0u;
                node_comp[i] = c;
                count[c] += 1.0;
                for (int d = 0; d < 3; ++d) cent[size_t(3) * c + d] += xyz[size_t(3) * i + d];
            }
            for (uint32_t c = 0; c < sys->n_components; ++c)
                for (int d = 0; d < 3; ++d) cent[size_t(3) * c + d] /= std::max(count[c], 1.0);
            sys->node_component.reset(ctx, nn);
            sys->component_centroid.reset(ctx, cent.size());
            sys->node_component.upload(node_comp.data(), nn);
            sys->component_centroid.upload(cent.data(), cent.size());
            HIP_CHECK(hipStreamSynchronize(st));
        }
        sys->agg_of.reset(ctx, npts);
        sys->agg_ptr.reset(ctx, ptr.size());
        sys->agg_nodes.reset(ctx, npts);
        sys->agg_of.upload(agg_of.data(), npts);
        sys->agg_ptr.upload(ptr.data(), ptr.size());
        sys->agg_nodes.upload(nodes.data(), npts);
        HIP_CHECK(hipStreamSynchronize(st)); // (the staging vectors go out of scope)
    }
    sys->agg_t.reset(ctx, size_t(npts) * 18);
    k_aggregate_t<<<div_up(sys->n_agg, 64), 64, 0, st>>>(sys->p1_xyz, sys->agg_ptr, sys->agg_nodes, sys->n_agg, sys->agg_t);
    KERNEL_CHECK();
    sys->points.reset(ctx, size_t(npts) * 3);
    HIP_CHECK(hipMemcpyAsync(sys->points.get(), mesh->points.get(), size_t(npts) * 3 * sizeof(double), hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st)); // host-side staging vectors (tables) must outlive their uploads
    // --- sliver patches of the smoothers (mh_patch.hip): elements with a shape measure below 0.02 (a regular tetrahedron has 1,
    //     a Kuhn tetrahedron 0.66; MH_PATCH_Q sets the threshold, 0 disables)
    static const float patch_q = getenv("MH_PATCH_Q") ? float(atof(getenv("MH_PATCH_Q"))) : 0.02f;
    mh_select_patches(sys, patch_q);
    sys->hierarchy_ready = false;
}
