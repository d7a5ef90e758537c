// The quadratic operator applied element by element, without the assembled matrix: y = (K - sigma M) x over an n x w panel.
//
// Built to answer one question with a measurement: the BSR product gathers one x row-triple per stored
// node block (4.14 M of them at S100k, ~28 per node); per ELEMENT there are only ten (1.05 M), and the 2 964 bytes of block
// values a tet owns in BSR shrink to the 104 bytes of its volume and barycentric gradients.  The price is the scatter: the ten
// result rows of an element are partial sums that other elements add to.
//
//   * No element matrix is formed.  A quadratic tet's displacement gradient is LINEAR over the element, fixed by its values
//     H_v at the four corners; with g_k the (constant) barycentric gradients and u_a the nodal values,
//         H_v = sum_k T[k][v] (x) g_k,   T[k][v] = (4 [k = v] - 1) u_k + 4 [k != v] u_(kv)        ((kv) = midside node of edge k-v)
//     stress-like tensors s_v = lambda tr(H_v) I + mu (H_v + H_v^T), S_v = V/20 (s_v + sum_v' s_v')   (int l_v l_v' = V (1 + [v = v']) / 20)
//     and the result rows  y_k = 3 R[k][k] - sum_{v != k} R[k][v],  y_(kv) = 4 (R[k][v] + R[v][k]),  R[k][v] = S_v g_k.
//     About 700 flops per element and column -- the same count as the 39 stored 3 x 3 blocks a tet amounts to in BSR.  Checked
//     against the assembled K to 1e-15 (tests: mh_system_matvec which = 5 against which = 2).
//   * The consistent mass matrix of the quadratic tet has seven distinct entries (x V/420: 6, 1, -4, -6, 32, 16, 8); its product
//     is formed from the corner sum, the midside sum and each node's adjacent / opposite partners.
//   * Lanes are panel columns: the thirty gathers of an element are coalesced row segments, and so are its thirty scatters.
//   * The scatter uses hardware fp64 atomic adds.  The sum of an entry's 4 .. 30 contributions then depends on arrival order:
//     results differ in the last bits from run to run.  The eigensolver does NOT use this product (its solves are
//     bit-reproducible); it exists beside the BSR one for measurement.  A deterministic scatter needs element
//     colouring (tens of launches with no locality: every gather and read-modify-write goes to HBM, ~5 GB at w = 64) or
//     per-tile accumulators in LDS (a 1.5 KB row per node at w = 64: a hundred nodes per CU, i.e. no tile worth the name).
#include "../mh_common.h"

namespace {
// midside node (4 .. 9) of the edge between corners a and b, edges in the reference's order 01 02 03 12 13 23
__device__ __forceinline__ constexpr int mid_of(int a, int b) {
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return lo == 0 ? 3 + hi : (lo == 1 ? 5 + hi : 9); // 01->4 02->5 03->6 12->7 13->8 23->9
}

template<int CL, int MODE = 0> // lanes per element (16, 32 or 64): a wave works on 64 / CL elements.  MODE 1 / 2: timing experiments only (plain stores / one store)
__global__ void __launch_bounds__(64) k_elem_apply(const uint32_t *__restrict__ elem_nodes, const double *__restrict__ basis, uint32_t nt, double lambda, double mu,
                                                   double mass_scale /* -sigma rho / 420 */, const double *__restrict__ x, double *__restrict__ y, uint32_t w) {
    constexpr int EPW = 64 / CL, EBs = 16; // one basis line: gradient k at [4 k .. 4 k + 2], the volume at [3]
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const uint32_t el = wave * EPW + lane / CL;
    if (el >= nt) return;
    const uint32_t c_in = lane % CL;
    uint32_t node[10];
#pragma unroll
    for (int a = 0; a < 10; ++a) node[a] = elem_nodes[10 * size_t(el) + a];
    double g[4][3];
    const double vol = basis[EBs * size_t(el) + 3];
    const double ms = mass_scale * vol;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p) g[k][p] = basis[EBs * size_t(el) + 4 * k + p];
    for (uint32_t col = c_in; col < w; col += CL) { // panels wider than the lane group go round again
        double u[10][3];
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) u[a][p] = x[(size_t(3) * node[a] + p) * w + col];
        double out[10][3];
        // ---- mass part: mass_scale * (420 Mhat) u
        {
            double sc[3], sm[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                sc[p] = u[0][p] + u[1][p] + u[2][p] + u[3][p];
                sm[p] = u[4][p] + u[5][p] + u[6][p] + u[7][p] + u[8][p] + u[9][p];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    double adj = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (b != a) adj += u[mid_of(a, b)][p];
                    out[a][p] = ms * (5 * u[a][p] + sc[p] - 6 * sm[p] + 2 * adj);
                }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a + 1; b < 4; ++b) {
                    const int m = mid_of(a, b), opp = 13 - m; // the edge through the other two corners: 4<->9, 5<->8, 6<->7
#pragma unroll
                    for (int p = 0; p < 3; ++p) out[m][p] = ms * (16 * u[m][p] + 16 * sm[p] - 8 * u[opp][p] + 2 * (u[a][p] + u[b][p]) - 6 * sc[p]);
                }
        }
        // ---- stiffness part.  The stress is linear in H, so sum_v s_v = s(sum_v H_v), and sum_v T[k][v] = 4 (sum of the three
        // midside values around corner k): the total is known before the per-corner tensors are, which are then used and dropped
        // one at a time (registers: thirty inputs and thirty outputs per lane are already 120 of them).
        auto stress = [&](const double (&H)[3][3], double (&out6)[6]) {
            const double tr = lambda * (H[0][0] + H[1][1] + H[2][2]);
            out6[0] = tr + 2 * mu * H[0][0];
            out6[1] = tr + 2 * mu * H[1][1];
            out6[2] = tr + 2 * mu * H[2][2];
            out6[3] = mu * (H[0][1] + H[1][0]);
            out6[4] = mu * (H[0][2] + H[2][0]);
            out6[5] = mu * (H[1][2] + H[2][1]);
        };
        double tot[6];
        {
            double Hs[3][3] = {};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double A[3] = {0, 0, 0};
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b != k)
#pragma unroll
                        for (int p = 0; p < 3; ++p) A[p] += u[mid_of(k, b)][p];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) Hs[p][q] += 4 * A[p] * g[k][q];
            }
            stress(Hs, tot);
        }
        const double v20 = vol / 20;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double H[3][3] = {};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double T[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) T[p] = k == v ? 3 * u[k][p] : 4 * u[mid_of(k, v)][p] - u[k][p];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 3; ++q) H[p][q] += T[p] * g[k][q];
            }
            double S[6];
            stress(H, S);
#pragma unroll
            for (int i = 0; i < 6; ++i) S[i] = v20 * (S[i] + tot[i]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double R0 = S[0] * g[k][0] + S[3] * g[k][1] + S[4] * g[k][2];
                const double R1 = S[3] * g[k][0] + S[1] * g[k][1] + S[5] * g[k][2];
                const double R2 = S[4] * g[k][0] + S[5] * g[k][1] + S[2] * g[k][2];
                if (k == v) {
                    out[k][0] += 3 * R0, out[k][1] += 3 * R1, out[k][2] += 3 * R2;
                } else {
                    out[k][0] -= R0, out[k][1] -= R1, out[k][2] -= R2;
                    const int m = mid_of(k, v);
                    out[m][0] += 4 * R0, out[m][1] += 4 * R1, out[m][2] += 4 * R2;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 10; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if constexpr (MODE == 0) unsafeAtomicAdd(&y[(size_t(3) * node[a] + p) * w + col], out[a][p]);
                else if constexpr (MODE == 1) y[(size_t(3) * node[a] + p) * w + col] = out[a][p]; // (wrong sums: how much of the time is the atomic unit?)
                else if (a == 0 && p == 0) y[(size_t(3) * node[0]) * w + col] = out[0][0] + out[1][1] + out[2][2] + out[3][0] + out[4][1] + out[5][2] + out[6][0] + out[7][1] + out[8][2] + out[9][0];
            }
    }
}
} // namespace

// y = (K - sigma M) x, element by element (fp64 panel n x w, row-major).  y is cleared first.  Not bit-reproducible (atomics).
void mh_elementwise_apply(mh_context *ctx, const mh_system *sys, double sigma, const double *x, double *y, uint32_t w) {
    const size_t n = size_t(3) * sys->n_nodes;
    HIP_CHECK(hipMemsetAsync(y, 0, n * w * sizeof(double), ctx->stream));
    const mh_material &m = sys->material;
    const double lambda = (m.poisson_ratio * m.young_modulus) / ((1 + m.poisson_ratio) * (1 - 2 * m.poisson_ratio)), mu = m.young_modulus / (2 * (1 + m.poisson_ratio));
    const double mass_scale = -sigma * m.density / 420.0; // times the element volume inside the kernel
    const uint32_t nt = sys->kept_tets;
    auto go = [&](auto cl_tag) {
        constexpr int CL = decltype(cl_tag)::value;
        const uint32_t waves = div_up(nt, 64 / CL);
        static const int mode = getenv("MH_ELEM_MODE") ? atoi(getenv("MH_ELEM_MODE")) : 0; // 1, 2: timing experiments (results are wrong)
        if (mode == 1) k_elem_apply<CL, 1><<<waves, 64, 0, ctx->stream>>>(sys->elem_nodes, sys->elem_basis, nt, lambda, mu, mass_scale, x, y, w);
        else if (mode == 2) k_elem_apply<CL, 2><<<waves, 64, 0, ctx->stream>>>(sys->elem_nodes, sys->elem_basis, nt, lambda, mu, mass_scale, x, y, w);
        else k_elem_apply<CL><<<waves, 64, 0, ctx->stream>>>(sys->elem_nodes, sys->elem_basis, nt, lambda, mu, mass_scale, x, y, w);
    };
    if (w <= 16) go(std::integral_constant<int, 16>{});
    else if (w <= 32) go(std::integral_constant<int, 32>{});
    else go(std::integral_constant<int, 64>{});
    KERNEL_CHECK();
}
