// Sparse block kernels of the eigensolver (gfx950): BSR 3x3 fp64 SpMM over row-major n x w panels.
//
// Panels are row-major (one row of w contiguous doubles per DOF) so that a wavefront's 64 lanes map to panel
// columns: the gather of x for a node block is three coalesced w*8-byte segments and the 3x3 block values are
// wave-uniform (scalar loads).  HBM-bound: algorithmic bytes per launch = 76 B per node block (9 values + column
// index) + 4 B per row pointer + 2 * 8 * n * w for reading x and writing y once.
#include "mh_common.h"

namespace {
constexpr int TB = 256;

// CW = panel columns per row group (power of two <= 64); a wave covers 64/CW block rows.
// NC = columns per lane (stride CW) so that panels up to CW*NC wide read the matrix once.
template<typename T, int CW, int NC, bool WITH_M, bool WITH_A>
__global__ void __launch_bounds__(TB) k_spmm(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const T *__restrict__ vals9,
                                            const T *__restrict__ mscal, const T *__restrict__ x, T *__restrict__ y, T *__restrict__ y2,
                                            uint32_t nnodes, uint32_t w, int xcd_remap) {
    constexpr int RPW = 64 / CW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c0 = blockIdx.y * (CW * NC) + (lane % CW);
    // Workgroups are dealt round-robin over the 8 XCDs; remap so that each XCD walks one contiguous eighth of the
    // (Morton-ordered) rows and the gathered x rows are shared inside one L2.
    const uint32_t nb = gridDim.x, per = (nb + 7) / 8;
    const uint32_t bid = xcd_remap ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    uint32_t row = (bid * (TB / 64) + wave) * RPW + lane / CW;
    if (RPW == 1) row = __builtin_amdgcn_readfirstlane(row);
    if (row >= nnodes) return;
    bool active[NC];
    uint32_t cc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        active[k] = c0 + CW * k < w;
        cc[k] = active[k] ? c0 + CW * k : 0;
    }
    T a0[NC], a1[NC], a2[NC], m0[NC], m1[NC], m2[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) a0[k] = a1[k] = a2[k] = m0[k] = m1[k] = m2[k] = 0;
    const uint32_t p1 = row_ptr[row + 1];
    uint32_t p = row_ptr[row];
    // U node blocks per step: all column indices, then all x gathers, then all block values are issued before the
    // first FMA, so a wave keeps 3*U*NC gathers in flight instead of one dependent load chain per block.
    auto step = [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        uint32_t j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) j[u] = col[p + u];
        T xv[U][NC][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const T *xr = x + size_t(3) * j[u] * w;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                xv[u][k][0] = xr[cc[k]];
                xv[u][k][1] = xr[w + cc[k]];
                xv[u][k][2] = xr[2 * size_t(w) + cc[k]];
            }
        }
        T v[U][9], m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (WITH_A) {
#pragma unroll
                for (int e = 0; e < 9; ++e) v[u][e] = vals9[size_t(9) * (p + u) + e];
            }
            m[u] = WITH_M ? mscal[p + u] : T(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const T x0 = xv[u][k][0], x1 = xv[u][k][1], x2 = xv[u][k][2];
                if (WITH_A) {
                    a0[k] += v[u][0] * x0 + v[u][1] * x1 + v[u][2] * x2;
                    a1[k] += v[u][3] * x0 + v[u][4] * x1 + v[u][5] * x2;
                    a2[k] += v[u][6] * x0 + v[u][7] * x1 + v[u][8] * x2;
                }
                if (WITH_M) {
                    m0[k] += m[u] * x0;
                    m1[k] += m[u] * x1;
                    m2[k] += m[u] * x2;
                }
            }
        }
        p += U;
    };
    constexpr int UMAX = NC >= 3 ? 2 : (NC == 2 ? 4 : 8);
    while (p + UMAX <= p1) step(std::integral_constant<int, UMAX>{});
    if (UMAX >= 8 && p + 4 <= p1) step(std::integral_constant<int, 4>{});
    if (UMAX >= 4 && p + 2 <= p1) step(std::integral_constant<int, 2>{});
    while (p < p1) step(std::integral_constant<int, 1>{});
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        if (!active[k]) continue;
        const size_t o = size_t(3) * row * w + c0 + CW * k;
        if (WITH_A) { y[o] = a0[k]; y[o + w] = a1[k]; y[o + 2 * size_t(w)] = a2[k]; }
        if (WITH_M) { y2[o] = m0[k]; y2[o + w] = m1[k]; y2[o + 2 * size_t(w)] = m2[k]; }
    }
}

// ---- wide-load variant -----------------------------------------------------------------------------------------
// Measured on MI355X (S100k, tools/spmm_bench.py): the gather costs a fixed ~16-20 cycles of a CU's address path per
// wave-level load INSTRUCTION almost regardless of the bytes it moves (w = 16 and w = 64 columns take 390 us and
// 470 us for the same 12.6 M gathers), and the scalar path streams the block values at under 1.5 TB/s.  So this kernel
//   * gathers with 16-byte loads: a lane owns V = 2 doubles / 4 floats of a panel row, CL lanes cover the row, and
//     the wave's G = 64/CL lane groups work on G different node blocks of the row at once -- V*G times fewer gather
//     instructions than one 8-byte column per lane;
//   * brings the row's column indices and 9-value blocks in with coalesced vector loads (lane l takes elements
//     l, l+64, ... of the row's contiguous slice of the BSR arrays) into a wave-private LDS slice, from which every
//     lane group reads its own block's values;
//   * folds the G partial rows with a fixed shuffle tree at the end of the row (deterministic).
// The panel pitch w must be a multiple of V (16-byte rows).
template<typename T, int CL, bool WITH_M, bool WITH_A>
__global__ void __launch_bounds__(TB) k_spmm_wide(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const T *__restrict__ vals9,
                                                 const T *__restrict__ mscal, const T *__restrict__ x, T *__restrict__ y, T *__restrict__ y2, uint32_t nnodes,
                                                 uint32_t w, int xcd_remap) {
    constexpr int V = 16 / sizeof(T), G = 64 / CL, STRIP = 64, VP = sizeof(T) == 4 ? 12 : 10, U = 4;
    typedef T Vec __attribute__((ext_vector_type(V)));
    __shared__ __attribute__((aligned(16))) T sv[TB / 64][WITH_A ? STRIP * VP : 1];
    __shared__ T sm[TB / 64][WITH_M ? STRIP : 1];
    __shared__ uint32_t sc[TB / 64][STRIP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = lane % CL, g = lane / CL;
    const uint32_t nb_grid = gridDim.x, per = (nb_grid + 7) / 8;
    const uint32_t bid = xcd_remap ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    const uint32_t row = __builtin_amdgcn_readfirstlane(bid * (TB / 64) + wave);
    if (row >= nnodes) return;
    const bool act = V * c < w;
    const uint32_t coff = act ? V * c : 0;
    Vec acc[3], macc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = macc[i] = Vec(0);
    const uint32_t p0 = __builtin_amdgcn_readfirstlane(row_ptr[row]), p1 = __builtin_amdgcn_readfirstlane(row_ptr[row + 1]);
    T *svw = sv[wave];
    T *smw = sm[wave];
    uint32_t *scw = sc[wave];
    for (uint32_t base = p0; base < p1; base += STRIP) {
        const uint32_t nb = min(uint32_t(STRIP), p1 - base);
        if (base != p0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // previous strip fully consumed
        scw[lane] = uint32_t(lane) < nb ? col[base + lane] : 0u;
        if (WITH_A) {
            const T *src = vals9 + size_t(9) * base;
            T tmp[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const uint32_t f = lane + 64 * i;
                tmp[i] = f < 9 * nb ? src[f] : T(0);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const uint32_t f = lane + 64 * i;
                if (f < 9 * nb) svw[(f / 9) * VP + f % 9] = tmp[i];
            }
        }
        if (WITH_M) smw[lane] = uint32_t(lane) < nb ? mscal[base + lane] : T(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // round t: lane group g takes block t*G + g of the strip
        const uint32_t rounds = (nb + G - 1) / G;
        for (uint32_t t0 = 0; t0 < rounds; t0 += U) {
            Vec xv[U][3];
            uint32_t bi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u < rounds) { // wave-uniform
                    bi[u] = min((t0 + u) * G + g, nb - 1);
                    const T *xr = x + size_t(3) * scw[bi[u]] * w + coff;
                    xv[u][0] = *reinterpret_cast<const Vec *>(xr);
                    xv[u][1] = *reinterpret_cast<const Vec *>(xr + w);
                    xv[u][2] = *reinterpret_cast<const Vec *>(xr + 2 * size_t(w));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u < rounds) {
                    const bool ok = (t0 + u) * G + g < nb;
                    if (WITH_A) {
                        T v[9];
#pragma unroll
                        for (int e = 0; e < 9; ++e) v[e] = ok ? svw[bi[u] * VP + e] : T(0);
                        acc[0] += v[0] * xv[u][0] + v[1] * xv[u][1] + v[2] * xv[u][2];
                        acc[1] += v[3] * xv[u][0] + v[4] * xv[u][1] + v[5] * xv[u][2];
                        acc[2] += v[6] * xv[u][0] + v[7] * xv[u][1] + v[8] * xv[u][2];
                    }
                    if (WITH_M) {
                        const T m = ok ? smw[bi[u]] : T(0);
                        macc[0] += m * xv[u][0];
                        macc[1] += m * xv[u][1];
                        macc[2] += m * xv[u][2];
                    }
                }
            }
        }
    }
    // fold the G lane groups (fixed tree), group 0 stores
#pragma unroll
    for (int sft = G / 2; sft >= 1; sft >>= 1) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if (WITH_A) acc[i][e] += __shfl_down(acc[i][e], sft * CL, 64);
                if (WITH_M) macc[i][e] += __shfl_down(macc[i][e], sft * CL, 64);
            }
        }
    }
    if (g == 0 && act) {
        const size_t o = size_t(3) * row * w + coff;
        if (WITH_A) {
            *reinterpret_cast<Vec *>(y + o) = acc[0];
            *reinterpret_cast<Vec *>(y + o + w) = acc[1];
            *reinterpret_cast<Vec *>(y + o + 2 * size_t(w)) = acc[2];
        }
        if (WITH_M) {
            *reinterpret_cast<Vec *>(y2 + o) = macc[0];
            *reinterpret_cast<Vec *>(y2 + o + w) = macc[1];
            *reinterpret_cast<Vec *>(y2 + o + 2 * size_t(w)) = macc[2];
        }
    }
}

template<typename T, bool WITH_M, bool WITH_A>
bool launch_spmm_wide(mh_context *ctx, const BsrLevel &lvl, const T *vals9, const T *x, T *y, const T *mscal, T *y2, uint32_t w) {
    constexpr uint32_t V = 16 / sizeof(T);
    static const bool legacy = getenv("MH_SPMM_LEGACY") && atoi(getenv("MH_SPMM_LEGACY")) != 0;
    if (legacy || w % V != 0 || w > 64 * V) return false;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y2)) & 15) return false;
    static const int xcd = getenv("MH_SPMM_XCD") ? atoi(getenv("MH_SPMM_XCD")) : 1;
    const unsigned grid = (div_up(lvl.n_nodes, TB / 64) + 7) / 8 * 8;
    auto go = [&](auto cl_tag) {
        constexpr int CL = decltype(cl_tag)::value;
        k_spmm_wide<T, CL, WITH_M, WITH_A><<<grid, TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, vals9, mscal, x, y, y2, lvl.n_nodes, w, xcd);
    };
    const uint32_t lanes = div_up(w, V);
    if (lanes <= 8) go(std::integral_constant<int, 8>{});
    else if (lanes <= 16) go(std::integral_constant<int, 16>{});
    else if (lanes <= 32) go(std::integral_constant<int, 32>{});
    else go(std::integral_constant<int, 64>{});
    KERNEL_CHECK();
    return true;
}

template<typename T, bool WITH_M, bool WITH_A>
void launch_spmm(mh_context *ctx, const BsrLevel &lvl, const T *vals9, const T *x, T *y, const T *mscal, T *y2, uint32_t w) {
    if (launch_spmm_wide<T, WITH_M, WITH_A>(ctx, lvl, vals9, x, y, mscal, y2, w)) return;
    const uint32_t n = lvl.n_nodes;
    auto go = [&](auto cw_tag, auto nc_tag) {
        constexpr int CW = decltype(cw_tag)::value, NC = decltype(nc_tag)::value;
        constexpr int RPW = 64 / CW;
        static const int xcd = getenv("MH_SPMM_XCD") ? atoi(getenv("MH_SPMM_XCD")) : 1;
        // grid.x padded to a multiple of 8 so the XCD remap is a bijection onto [0, 8*per)
        dim3 grid((div_up(n, (TB / 64) * RPW) + 7) / 8 * 8, div_up(w, CW * NC));
        k_spmm<T, CW, NC, WITH_M, WITH_A><<<grid, TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, vals9, mscal, x, y, y2, n, w, xcd);
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    auto ic = [](auto v) { return v; };
    (void)ic;
#define MH_GO(CWV, NCV) go(std::integral_constant<int, CWV>{}, std::integral_constant<int, NCV>{})
    // One row per wave (wave-uniform block values through the scalar path) beats packing several rows into a wave
    // even when most lanes idle: measured at S100k, w = 32 takes 1013 us with two rows per wave and 525 us with one.
    if (w <= 8) MH_GO(8, 1);
    else if (w <= 64) MH_GO(64, 1);
    else if (w <= 128) MH_GO(64, 2);
    else if (w <= 192) MH_GO(64, 3);
    else MH_GO(64, 4);
#undef MH_GO
    KERNEL_CHECK();
}

// ---- LDS-staged variant ---------------------------------------------------------------------------------------
// One 1024-thread workgroup per tile of 64 consecutive rows; wave v owns rows 4v..4v+3 and keeps their accumulators in
// registers.  The tile's sorted unique column nodes are walked in segments of S nodes: the segment's x rows (3*w
// doubles each, contiguous) are staged in LDS once and every node block of the tile that points into the segment
// reads its x from there.  Each x row therefore leaves L2 once per tile instead of once per node block (~5x fewer
// gathered bytes on P2 tetrahedral meshes in Morton order).  The next segment's rows are fetched into registers
// before the current segment is consumed and written to LDS after it (issue-early / write-late), so the gather latency
// hides under the FMAs; block values stream through scalar loads exactly once.
template<int NC, int MAXR>
__global__ void __launch_bounds__(1024) k_spmm_tiled(const uint32_t *__restrict__ row_ptr, const uint16_t *__restrict__ local, const double *__restrict__ vals9,
                                                    const uint32_t *__restrict__ tile_uptr, const uint32_t *__restrict__ tile_ucols,
                                                    const double *__restrict__ x, double *__restrict__ y, uint32_t nnodes, uint32_t w, uint32_t seg_nodes,
                                                    uint32_t ntiles) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    constexpr int SI = 3 * NC; // 64-lane strips per staged x row (3*w <= 192*NC doubles)
    const uint32_t per = (gridDim.x + 7) / 8;
    const uint32_t tile = (blockIdx.x % 8) * per + blockIdx.x / 8; // XCD-contiguous tiles
    if (tile >= ntiles) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w3 = 3 * w;
    const uint32_t u0 = tile_uptr[tile], nu = tile_uptr[tile + 1] - u0;
    uint32_t cc[NC];
    bool active[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        active[k] = uint32_t(lane) + 64 * k < w;
        cc[k] = active[k] ? lane + 64 * k : 0;
    }
    uint32_t pstart[4], pend[4], lv[4], used[4];
    double acc[4][NC][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t r = tile * MH_TILE_ROWS + wave * 4 + q;
        pstart[q] = __builtin_amdgcn_readfirstlane(r < nnodes ? row_ptr[r] : 0);
        pend[q] = __builtin_amdgcn_readfirstlane(r < nnodes ? row_ptr[r + 1] : 0);
        lv[q] = pstart[q] + lane < pend[q] ? uint32_t(local[pstart[q] + lane]) : 0xffffu;
        used[q] = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[q][k][0] = acc[q][k][1] = acc[q][k][2] = 0;
    }
    double pre[MAXR][SI];
    auto fetch = [&](uint32_t base) { // this wave's rows of the segment starting at `base` -> registers
        const uint32_t cnt = min(seg_nodes, nu - base);
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const uint32_t k = wave + 16 * j;
            if (k < cnt) {
                const double *src = x + size_t(tile_ucols[u0 + base + k]) * w3;
#pragma unroll
                for (int i = 0; i < SI; ++i) {
                    const uint32_t off = lane + 64 * i;
                    pre[j][i] = off < w3 ? src[off] : 0.0;
                }
            }
        }
    };
    auto commit = [&](uint32_t base) { // registers -> LDS
        const uint32_t cnt = min(seg_nodes, nu - base);
#pragma unroll
        for (int j = 0; j < MAXR; ++j) {
            const uint32_t k = wave + 16 * j;
            if (k < cnt) {
#pragma unroll
                for (int i = 0; i < SI; ++i) {
                    const uint32_t off = lane + 64 * i;
                    if (off < w3) xs[size_t(k) * w3 + off] = pre[j][i];
                }
            }
        }
    };
    if (nu) fetch(0);
    if (nu) commit(0);
    __syncthreads();
    for (uint32_t base = 0; base < nu; base += seg_nodes) {
        const uint32_t cnt = min(seg_nodes, nu - base);
        const bool more = base + seg_nodes < nu;
        if (more) fetch(base + seg_nodes);
        const uint32_t lim = base + cnt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            for (;;) {
                const uint32_t avail = min(64u, pend[q] - pstart[q]);
                const uint32_t n = __popcll(__ballot(uint32_t(lane) >= used[q] && uint32_t(lane) < avail && lv[q] < lim));
                uint32_t p = pstart[q] + used[q];
                uint32_t first = used[q];
                auto step = [&](auto u_tag) {
                    constexpr int U = decltype(u_tag)::value;
                    double v[U][9];
                    uint32_t li[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        li[u] = __builtin_amdgcn_readlane(int(lv[q]), int(first) + u) - base;
#pragma unroll
                        for (int e = 0; e < 9; ++e) v[u][e] = vals9[size_t(9) * (p + u) + e];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double *xr = xs + size_t(li[u]) * w3;
#pragma unroll
                        for (int k = 0; k < NC; ++k) {
                            const double x0 = xr[cc[k]], x1 = xr[w + cc[k]], x2 = xr[2 * w + cc[k]];
                            acc[q][k][0] += v[u][0] * x0 + v[u][1] * x1 + v[u][2] * x2;
                            acc[q][k][1] += v[u][3] * x0 + v[u][4] * x1 + v[u][5] * x2;
                            acc[q][k][2] += v[u][6] * x0 + v[u][7] * x1 + v[u][8] * x2;
                        }
                    }
                    p += U;
                    first += U;
                };
                uint32_t done = 0;
                while (done + 4 <= n) { step(std::integral_constant<int, 4>{}); done += 4; }
                while (done < n) { step(std::integral_constant<int, 1>{}); done += 1; }
                used[q] += n;
                // a row longer than 64 node blocks: move the 64-block window once it is used up
                if (used[q] == 64 && pstart[q] + 64 < pend[q]) {
                    pstart[q] += 64;
                    lv[q] = pstart[q] + lane < pend[q] ? uint32_t(local[pstart[q] + lane]) : 0xffffu;
                    used[q] = 0;
                    continue;
                }
                break;
            }
        }
        __syncthreads();
        if (more) commit(base + seg_nodes);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t r = tile * MH_TILE_ROWS + wave * 4 + q;
        if (r >= nnodes) continue;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            if (!active[k]) continue;
            const size_t o = size_t(r) * w3 + lane + 64 * k;
            y[o] = acc[q][k][0];
            y[o + w] = acc[q][k][1];
            y[o + 2 * size_t(w)] = acc[q][k][2];
        }
    }
}

bool launch_spmm_tiled(mh_context *ctx, const BsrLevel &lvl, const double *vals9, const double *x, double *y, uint32_t w) {
    // Off by default: correct, but its barriers and per-row serial phases make it slower than the plain kernel on
    // MI355X (742 us vs 527 us at S100k, w = 64, even with the block values removed); kept for the next round.
    static const bool enabled = getenv("MH_SPMM_TILED") && atoi(getenv("MH_SPMM_TILED")) != 0;
    if (!enabled || !lvl.tiled || w < 24 || w > 256) return false; // narrow panels: the plain kernel packs several rows per wave
    static const int lds_kb = getenv("MH_SPMM_LDS_KB") ? atoi(getenv("MH_SPMM_LDS_KB")) : 150;
    const unsigned grid = (lvl.n_tiles + 7) / 8 * 8;
    auto go = [&](auto nc_tag, auto maxr_tag) {
        constexpr int NC = decltype(nc_tag)::value, MAXR = decltype(maxr_tag)::value;
        uint32_t seg = uint32_t(size_t(lds_kb) * 1024 / (size_t(24) * w));
        seg = std::max(4u, std::min(seg, uint32_t(16 * MAXR)));
        const size_t lds = size_t(seg) * 3 * w * sizeof(double);
        static bool attr_set = false;
        if (!attr_set) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spmm_tiled<NC, MAXR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        k_spmm_tiled<NC, MAXR><<<grid, 1024, lds, ctx->stream>>>(lvl.row_ptr, lvl.block_local, vals9, lvl.tile_uptr, lvl.tile_ucols, x, y, lvl.n_nodes, w, seg, lvl.n_tiles);
    };
    if (w <= 64) go(std::integral_constant<int, 1>{}, std::integral_constant<int, 6>{});
    else if (w <= 128) go(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
    else go(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
    KERNEL_CHECK();
    return true;
}
} // namespace

void mh_timer_flush(mh_context *ctx) {
    if (!ctx->timer_used) return;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->timer_used; ++i) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->timer_events[i].first, ctx->timer_events[i].second));
        ctx->spmm_ms += ms;
        ctx->spmm_bytes += ctx->timer_bytes[i];
        ++ctx->spmm_launches;
    }
    ctx->timer_used = 0;
}

namespace {
// HIP events around one launch on the context's stream; resolved lazily by mh_timer_flush.
struct TimedLaunch {
    mh_context *ctx;
    bool on;
    size_t slot{0};
    TimedLaunch(mh_context *c, bool enable, double algorithmic_bytes) : ctx(c), on(enable) {
        if (!on) return;
        if (ctx->timer_used == ctx->timer_events.size()) {
            hipEvent_t a, b;
            HIP_CHECK(hipEventCreate(&a));
            HIP_CHECK(hipEventCreate(&b));
            ctx->timer_events.emplace_back(a, b);
            ctx->timer_bytes.push_back(0);
        }
        slot = ctx->timer_used++;
        ctx->timer_bytes[slot] = algorithmic_bytes;
        HIP_CHECK(hipEventRecord(ctx->timer_events[slot].first, ctx->stream));
    }
    ~TimedLaunch() {
        if (on) (void)hipEventRecord(ctx->timer_events[slot].second, ctx->stream);
    }
};
// algorithmic bytes of one product: 9 values + column index per node block, row pointers, x read once, y written once
double spmm_bytes(const BsrLevel &lvl, uint32_t w, size_t scalar) {
    return (9.0 * scalar + 4.0) * double(lvl.n_blocks) + 4.0 * (double(lvl.n_nodes) + 1) + 2.0 * scalar * 3.0 * double(lvl.n_nodes) * w;
}
} // namespace

void mh_spmm(mh_context *ctx, const BsrLevel &lvl, const double *vals9, const double *x, double *y, const double *mscal, double *y2, uint32_t w) {
    if (w == 0) return;
    // Timed launches: products with the P2 operator's 3x3 blocks (A-values only).
    TimedLaunch timed(ctx, ctx->time_kernels && vals9 && !mscal && lvl.id == 2, spmm_bytes(lvl, w, sizeof(double)));
    if (vals9 && mscal) launch_spmm<double, true, true>(ctx, lvl, vals9, x, y, mscal, y2, w);
    else if (vals9) {
        if (!launch_spmm_tiled(ctx, lvl, vals9, x, y, w)) launch_spmm<double, false, true>(ctx, lvl, vals9, x, y, nullptr, nullptr, w);
    }
    else launch_spmm<double, true, false>(ctx, lvl, nullptr, x, nullptr, mscal, y2, w);
}

// fp32 product with the level's single-precision copy of A (preconditioner only).
void mh_spmm_f32(mh_context *ctx, const BsrLevel &lvl, const float *x, float *y, uint32_t w) {
    if (w == 0) return;
    TimedLaunch timed(ctx, ctx->time_kernels && lvl.id == 2, spmm_bytes(lvl, w, sizeof(float)));
    launch_spmm<float, false, true>(ctx, lvl, lvl.aval32.get(), x, y, static_cast<const float *>(nullptr), static_cast<float *>(nullptr), w);
}
