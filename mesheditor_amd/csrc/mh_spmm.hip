// Sparse block kernels of the eigensolver (gfx950): BSR 3x3 fp64 SpMM over row-major n x w panels.
//
// Panels are row-major (one row of w contiguous doubles per DOF) so that a wavefront's 64 lanes map to panel
// columns: the gather of x for a node block is three coalesced w*8-byte segments and the 3x3 block values are
// wave-uniform (scalar loads).  HBM-bound: algorithmic bytes per launch = 76 B per node block (9 values + column
// index) + 4 B per row pointer + 2 * 8 * n * w for reading x and writing y once.
#include "mh_common.h"

namespace {
#ifndef MH_SPMM_TB
#define MH_SPMM_TB 64
#endif
constexpr int TB = MH_SPMM_TB;
#ifndef MH_SPMM_U
#define MH_SPMM_U 1 // node-block rounds whose gathers are in flight together: 1 keeps the registers low (occupancy 7-8 waves per SIMD beats deeper unrolling)
#endif

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
// TV = matrix values, TX = panel read, TY = accumulators and panel written (TV = TY = double with TX = float gives the
// double-precision residual of a single-precision iterate at single-precision gather cost).
// MAPOUT: the results go to columns omap[c] (c < wreal) of panels of pitch ldy instead of a panel shaped like x -- the images
// A X, M X of the new Ritz vectors written straight into the active columns of the block.
// EPI = 1: instead of storing t = A d, the Chebyshev step that consumes it runs on the rows while they are in registers:
// r -= t, d' = c1 d + c2 D^-1 r, x += d'.  d' goes to a second buffer (other rows are still gathering d).  One launch and one
// pass over t less per smoothing step.
struct ChebStep {
    float *r = nullptr;
    const float *dinv = nullptr;
    float *d_out = nullptr;
    float *x = nullptr;
    float c1 = 0.f, c2 = 0.f;
};
// MAPOUT with a residual epilogue: the rows of A x and M x are in registers when they are stored, and theta is known, so the
// eigen-residual r = A x - theta M x of the row goes out beside them (compact panel of the launch's own pitch) together with
// the row's contributions to ||r||^2 and ||M x||^2 per column (optionally in the Jacobi scaling): the separate residual pass over
// A X, M X (three panel passes per iteration) and the column gather after it are not needed.  partial: [node][2][pitch].
struct ResidualEpilogue {
    const double *theta = nullptr; // per mapped column
    double *r_out = nullptr;       // n x pitch
    double *partial = nullptr;     // n_nodes x 2 x pitch
    const double *dinv = nullptr;  // 3 n_nodes weights, or null
};
#ifdef MH_SPMM_WAVES
#define MH_SPMM_OCC __attribute__((amdgpu_waves_per_eu(MH_SPMM_WAVES, 8)))
#else
#define MH_SPMM_OCC
#endif
// UR = gather rounds in flight per wave, PRE = the Chebyshev step's own rows requested before the products.  The big (P2) level
// runs UR = 1, PRE = false: occupancy hides its gathers best.  A level with fewer rows than the chip has wave slots (the P1
// operator: 20 k rows for 8 k slots) is a chain of dependent round trips per wave instead -- there every row's gathers go out
// together (UR = 4) and the epilogue's operands are prefetched.
template<typename TV, typename TX, typename TY, int V, int CL, bool WITH_M, bool WITH_A, bool MAPOUT = false, int EPI = 0, int UR = MH_SPMM_U, bool PRE = false>
__global__ void __launch_bounds__(TB) MH_SPMM_OCC k_spmm_wide(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const TV *__restrict__ vals9,
                                                 const TV *__restrict__ mscal, const TX *__restrict__ x, TY *__restrict__ y, TY *__restrict__ y2, uint32_t nnodes,
                                                 uint32_t w, int xcd_remap, uint32_t ldy = 0, uint32_t wreal = 0, const uint32_t *__restrict__ omap = nullptr, ChebStep epi = ChebStep{},
                                                 uint32_t xpitch = 0, ResidualEpilogue res = ResidualEpilogue{}) { // xpitch (MAPOUT): row pitch of x when the launch covers a column range of a wider panel
    constexpr int G = 64 / CL, STRIP = 64, VP = sizeof(TV) == 4 ? 12 : 10, U = UR; // V = panel entries per lane (16 bytes; 1 for odd pitches)
    typedef TX Vec __attribute__((ext_vector_type(V)));
    typedef TY Acc __attribute__((ext_vector_type(V)));
    // (one slot more than a strip holds: the slot after a strip's last block is a ZERO block on a valid column, and a lane group
    // without a block in the last round takes it -- no per-value selects in the loop, which the compiler had turned into nine
    // exec-masked LDS reads with their scalar bookkeeping, more instructions than the products themselves)
    __shared__ __attribute__((aligned(16))) TV sv[TB / 64][WITH_A ? (STRIP + 1) * VP : 1];
    __shared__ TV sm[TB / 64][WITH_M ? STRIP + 1 : 1];
    __shared__ uint32_t sc[TB / 64][STRIP + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = lane % CL, g = lane / CL; // CL need not divide 64 (20 lanes x 3 groups for 80 floats): lanes past G * CL idle
    const uint32_t nb_grid = gridDim.x, per = (nb_grid + 7) / 8;
    const uint32_t bid = xcd_remap ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    const uint32_t row = __builtin_amdgcn_readfirstlane(bid * (TB / 64) + wave);
    if (row >= nnodes) return;
    const bool act = V * c < w && g < G;
    const uint32_t coff = act ? V * c : 0;
    Acc acc[3], macc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = macc[i] = Acc(0);
    // EPI with PRE: the rows the step updates are requested now, so they arrive while the products are formed
    Acc er[PRE ? 3 : 1], ed[PRE ? 3 : 1], ex[PRE ? 3 : 1];
    TY edinv[PRE ? 3 : 1];
    if constexpr (EPI == 1 && PRE) {
        if (g == 0 && act) {
            const size_t o = size_t(3) * row * w + coff;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                er[i] = *reinterpret_cast<const Acc *>(epi.r + o + size_t(i) * w);
                ed[i] = __builtin_convertvector(*reinterpret_cast<const Vec *>(x + o + size_t(i) * w), Acc);
                ex[i] = *reinterpret_cast<const Acc *>(epi.x + o + size_t(i) * w);
                edinv[i] = epi.dinv[size_t(3) * row + i];
            }
        }
    }
    const uint32_t p0 = __builtin_amdgcn_readfirstlane(row_ptr[row]), p1 = __builtin_amdgcn_readfirstlane(row_ptr[row + 1]);
    TV *svw = sv[wave];
    TV *smw = sm[wave];
    uint32_t *scw = sc[wave];
    for (uint32_t base = p0; base < p1; base += STRIP) {
        const uint32_t nb = min(uint32_t(STRIP), p1 - base);
        if (base != p0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // previous strip fully consumed
        scw[lane] = uint32_t(lane) < nb ? col[base + lane] : 0u;
        if (WITH_A) {
            const TV *src = vals9 + size_t(9) * base;
            TV tmp[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const uint32_t f = lane + 64 * i;
                tmp[i] = f < 9 * nb ? src[f] : TV(0);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const uint32_t f = lane + 64 * i;
                if (f < 9 * nb) svw[(f / 9) * VP + f % 9] = tmp[i];
            }
        }
        if (WITH_M) smw[lane] = uint32_t(lane) < nb ? mscal[base + lane] : TV(0);
        if (lane == 0) { // the zero block
            scw[nb] = row;
            if (WITH_M) smw[nb] = TV(0);
        }
        if (WITH_A && lane < VP) svw[nb * VP + lane] = TV(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // round t: lane group g takes block t*G + g of the strip
        const uint32_t rounds = (nb + G - 1) / G;
        for (uint32_t t0 = 0; t0 < rounds; t0 += U) {
            Vec xv[U][3];
            uint32_t bi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u < rounds) { // wave-uniform
                    bi[u] = min((t0 + u) * G + g, nb);
                    const uint32_t xp = MAPOUT ? xpitch : w;
                    const TX *xr = x + size_t(3) * scw[bi[u]] * xp + coff;
                    xv[u][0] = *reinterpret_cast<const Vec *>(xr);
                    xv[u][1] = *reinterpret_cast<const Vec *>(xr + xp);
                    xv[u][2] = *reinterpret_cast<const Vec *>(xr + 2 * size_t(xp));
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u < rounds) {
                    const Acc x0 = __builtin_convertvector(xv[u][0], Acc), x1 = __builtin_convertvector(xv[u][1], Acc), x2 = __builtin_convertvector(xv[u][2], Acc);
                    if (WITH_A) {
                        TY v[9];
#pragma unroll
                        for (int e = 0; e < 9; ++e) v[e] = TY(svw[bi[u] * VP + e]);
                        // one fused multiply-add per term (a sum of three products first costs a fourth instruction)
                        acc[0] += v[0] * x0; acc[0] += v[1] * x1; acc[0] += v[2] * x2;
                        acc[1] += v[3] * x0; acc[1] += v[4] * x1; acc[1] += v[5] * x2;
                        acc[2] += v[6] * x0; acc[2] += v[7] * x1; acc[2] += v[8] * x2;
                    }
                    if (WITH_M) {
                        const TY m = TY(smw[bi[u]]);
                        macc[0] += m * x0;
                        macc[1] += m * x1;
                        macc[2] += m * x2;
                    }
                }
            }
        }
    }
    // fold the G lane groups (fixed tree; fixed chain when G is not a power of two), group 0 stores
    if constexpr ((G & (G - 1)) == 0) {
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
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                TY sa = acc[i][e], sm_ = macc[i][e];
#pragma unroll
                for (int j = 1; j < G; ++j) {
                    if (WITH_A) sa += __shfl_down(acc[i][e], j * CL, 64);
                    if (WITH_M) sm_ += __shfl_down(macc[i][e], j * CL, 64);
                }
                acc[i][e] = sa;
                macc[i][e] = sm_;
            }
        }
    }
    if constexpr (EPI == 1) {
        if (g == 0 && act) {
            const size_t o = size_t(3) * row * w + coff;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const size_t oi = o + size_t(i) * w;
                Acc rv, dv, xn;
                if constexpr (PRE) {
                    rv = er[i] - acc[i];
                    dv = epi.c1 * ed[i] + (epi.c2 * edinv[i]) * rv;
                    xn = ex[i] + dv;
                } else {
                    // loaded here rather than up front: nine more vector registers per lane would cost two waves of occupancy,
                    // and occupancy is what hides the gathers of the big level
                    rv = *reinterpret_cast<const Acc *>(epi.r + oi) - acc[i];
                    dv = epi.c1 * __builtin_convertvector(*reinterpret_cast<const Vec *>(x + oi), Acc) + (epi.c2 * epi.dinv[size_t(3) * row + i]) * rv;
                    xn = *reinterpret_cast<const Acc *>(epi.x + oi) + dv;
                }
                *reinterpret_cast<Acc *>(epi.r + oi) = rv;
                *reinterpret_cast<Acc *>(epi.d_out + oi) = dv;
                *reinterpret_cast<Acc *>(epi.x + oi) = xn;
            }
        }
        return;
    }
    if (MAPOUT) {
        if (g == 0 && act) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if (coff + e >= wreal) continue;
                const size_t o = size_t(3) * row * ldy + omap[coff + e];
                if (WITH_A) { y[o] = acc[0][e]; y[o + ldy] = acc[1][e]; y[o + 2 * size_t(ldy)] = acc[2][e]; }
                if (WITH_M) { y2[o] = macc[0][e]; y2[o + ldy] = macc[1][e]; y2[o + 2 * size_t(ldy)] = macc[2][e]; }
            }
            if constexpr (WITH_A && WITH_M && std::is_same<TY, double>::value && V == 2) {
                if (res.r_out) { // (launches that cover the whole panel in one column range: pitch w = xpitch)
                    Acc th, sr = Acc(0), sm = Acc(0);
                    th[0] = res.theta[coff];
                    th[1] = res.theta[coff + 1]; // (the pad column of an odd count reads one entry past the active ones: the array is longer)
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const Acc r = acc[i] - th * macc[i];
                        const TY wgt = res.dinv ? res.dinv[size_t(3) * row + i] : TY(1);
                        *reinterpret_cast<Acc *>(res.r_out + (size_t(3) * row + i) * w + coff) = r;
                        sr += wgt * (r * r);
                        sm += wgt * (macc[i] * macc[i]);
                    }
                    *reinterpret_cast<Acc *>(res.partial + (size_t(2) * row) * w + coff) = sr;
                    *reinterpret_cast<Acc *>(res.partial + (size_t(2) * row + 1) * w + coff) = sm;
                }
            }
        }
        return;
    }
    if (g == 0 && act) {
        const size_t o = size_t(3) * row * w + coff;
        if (WITH_A) {
            *reinterpret_cast<Acc *>(y + o) = acc[0];
            *reinterpret_cast<Acc *>(y + o + w) = acc[1];
            *reinterpret_cast<Acc *>(y + o + 2 * size_t(w)) = acc[2];
        }
        if (WITH_M) {
            *reinterpret_cast<Acc *>(y2 + o) = macc[0];
            *reinterpret_cast<Acc *>(y2 + o + w) = macc[1];
            *reinterpret_cast<Acc *>(y2 + o + 2 * size_t(w)) = macc[2];
        }
    }
}

// Fewer rows than ~4 waves per wave slot of the chip (256 CUs x 4 SIMDs x 8 waves): the launch is a latency chain per wave,
// not a throughput problem -- so the thinking went.  Measured (tools/ab_r02d.sh, S100k, P1 level = 19 683 rows): the
// unrolled / prefetching variant makes the average product launch SLOWER (104.0 us against 98.5 us over the 601 launches of
// a solve; the extra registers cost more occupancy than the overlapped round trips give back), the solve time is unchanged.
// Off by default; MH_SPMM_SMALL=1 selects it.
inline bool latency_bound_level(const BsrLevel &lvl) {
    constexpr bool on = false;
    return on && lvl.n_nodes < 32768;
}

template<typename TV, typename TX, typename TY, bool WITH_M, bool WITH_A>
bool launch_spmm_wide(mh_context *ctx, const BsrLevel &lvl, const TV *vals9, const TX *x, TY *y, const TV *mscal, TY *y2, uint32_t w) {
    constexpr uint32_t VFULL = 16 / sizeof(TX);
    constexpr bool legacy = false;
    constexpr bool mixed = !std::is_same<TV, TX>::value || !std::is_same<TX, TY>::value;
    if (legacy && !mixed) return false;
    constexpr int xcd = 1;
    const unsigned grid = (div_up(lvl.n_nodes, TB / 64) + 7) / 8 * 8;
    auto run = [&](auto v_tag) {
        constexpr int V = decltype(v_tag)::value;
        auto go = [&](auto cl_tag) {
            constexpr int CL = decltype(cl_tag)::value;
            if (latency_bound_level(lvl))
                k_spmm_wide<TV, TX, TY, V, CL, WITH_M, WITH_A, false, 0, 4><<<grid, TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, vals9, mscal, x, y, y2, lvl.n_nodes, w, xcd);
            else
                k_spmm_wide<TV, TX, TY, V, CL, WITH_M, WITH_A><<<grid, TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, vals9, mscal, x, y, y2, lvl.n_nodes, w, xcd);
        };
        const uint32_t lanes = div_up(w, uint32_t(V));
        if (lanes <= 8) go(std::integral_constant<int, 8>{});
        else if (lanes <= 16) go(std::integral_constant<int, 16>{});
        else if (lanes <= 20 && V == 4) go(std::integral_constant<int, 20>{}); // the 80-column block in single precision: 3 node blocks per round
        else if (lanes <= 32) go(std::integral_constant<int, 32>{});
        else go(std::integral_constant<int, 64>{});
    };
    const bool aligned16 = !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(y2)) & 15);
    if (w % VFULL == 0 && w <= 64 * VFULL && aligned16) run(std::integral_constant<int, int(VFULL)>{});
    else if (!mixed && w <= 16) run(std::integral_constant<int, 1>{}); // narrow odd pitch (SpMV: w = 1): one entry per lane, 4-8 node blocks per
                                                                       // wave at once: 184 us against 417 us at w = 1; above 16 columns k_spmm is faster
    else return false;
    KERNEL_CHECK();
    return true;
}

template<typename T, bool WITH_M, bool WITH_A>
void launch_spmm(mh_context *ctx, const BsrLevel &lvl, const T *vals9, const T *x, T *y, const T *mscal, T *y2, uint32_t w) {
    if (launch_spmm_wide<T, T, T, WITH_M, WITH_A>(ctx, lvl, vals9, x, y, mscal, y2, w)) return;
    const uint32_t n = lvl.n_nodes;
    auto go = [&](auto cw_tag, auto nc_tag) {
        constexpr int CW = decltype(cw_tag)::value, NC = decltype(nc_tag)::value;
        constexpr int RPW = 64 / CW;
        constexpr int xcd = 1;
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

} // namespace

void mh_timer_flush(mh_context *ctx) {
    if (!ctx->timer_used) return;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->timer_used; ++i) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, ctx->timer_events[i].first, ctx->timer_events[i].second));
        auto &t = ctx->totals[ctx->timer_class[i]];
        t.ms += ms;
        t.work += ctx->timer_work[i];
        ++t.launches;
    }
    ctx->timer_used = 0;
}

namespace {
// algorithmic bytes of one product: the block values read (9 per node block for A, 1 for M) + column index, the row
// pointers, x read once, every output panel written once
double spmm_bytes(const BsrLevel &lvl, uint32_t w, size_t value_bytes, size_t x_bytes, size_t y_bytes, bool with_a, bool with_m) {
    const double per_block = (with_a ? 9.0 : 0.0) * value_bytes + (with_m ? 1.0 : 0.0) * value_bytes + 4.0;
    const double panel = 3.0 * double(lvl.n_nodes) * w;
    return per_block * double(lvl.n_blocks) + 4.0 * (double(lvl.n_nodes) + 1) + panel * x_bytes + panel * y_bytes * ((with_a ? 1 : 0) + (with_m ? 1 : 0));
}
} // namespace

void mh_spmm(mh_context *ctx, const BsrLevel &lvl, const double *vals9, const double *x, double *y, const double *mscal, double *y2, uint32_t w) {
    if (w == 0) return;
    TimedLaunch timed(ctx, MH_KERNEL_SPMM, spmm_bytes(lvl, w, 8, 8, 8, vals9 != nullptr, mscal != nullptr));
    if (vals9 && mscal) launch_spmm<double, true, true>(ctx, lvl, vals9, x, y, mscal, y2, w);
    else if (vals9) {
        launch_spmm<double, false, true>(ctx, lvl, vals9, x, y, nullptr, nullptr, w);
    }
    else launch_spmm<double, true, false>(ctx, lvl, nullptr, x, nullptr, mscal, y2, w);
}

// A x and M x of an n x w panel (w even, 16-byte aligned) written to columns omap[c], c < wreal, of panels of pitch ldy.
void mh_spmm_mapped(mh_context *ctx, const BsrLevel &lvl, const double *vals9, const double *x, double *y, const double *mscal, double *y2, uint32_t w, uint32_t ldy,
                    uint32_t wreal, const uint32_t *omap, const double *res_theta, double *res_out, double *res_partial, const double *res_dinv) {
    if (w == 0 || wreal == 0) return;
    ResidualEpilogue res;
    if (res_out && w <= 128) { // one column range covers the panel: the epilogue's compact pitch is the panel's
        res.theta = res_theta;
        res.r_out = res_out;
        res.partial = res_partial;
        res.dinv = res_dinv;
    } else if (res_out) {
        mh_throw(MH_EINVAL, "residual epilogue needs a panel of at most 128 columns (got %u)", w);
    }
    if (w % 2 || (reinterpret_cast<uintptr_t>(x) & 15)) mh_throw(MH_EINVAL, "mapped product needs an even pitch and an aligned panel (got %u)", w);
    constexpr int xcd = 1;
    const unsigned grid = (div_up(lvl.n_nodes, TB / 64) + 7) / 8 * 8;
    // panels wider than the 128 columns one wave covers go in column ranges of the same pitch
    const uint32_t ranges = div_up(w, 128u), step = (div_up(w, ranges) + 1u) & ~1u;
    for (uint32_t c0 = 0; c0 < wreal; c0 += step) {
        const uint32_t wc = std::min(step, w - c0), wr = std::min(wc, wreal - c0);
        // (with the residual epilogue: one more panel written, and the two norm partials per node and column)
        TimedLaunch timed(ctx, MH_KERNEL_SPMM, spmm_bytes(lvl, wr, 8, 8, 8, true, true) + (res.r_out ? 8.0 * double(wr) * (3.0 + 2.0) * double(lvl.n_nodes) : 0.0));
        auto go = [&](auto cl_tag) {
            constexpr int CL = decltype(cl_tag)::value;
            k_spmm_wide<double, double, double, 2, CL, true, true, true><<<grid, TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, vals9, mscal, x + c0, y, y2, lvl.n_nodes, wc, xcd, ldy, wr, omap + c0,
                                                                                                  ChebStep{}, w, res);
        };
        const uint32_t lanes = wc / 2;
        if (lanes <= 8) go(std::integral_constant<int, 8>{});
        else if (lanes <= 16) go(std::integral_constant<int, 16>{});
        else if (lanes <= 32) go(std::integral_constant<int, 32>{});
        else go(std::integral_constant<int, 64>{});
        KERNEL_CHECK();
    }
}

// One Chebyshev-Jacobi step of the single-precision smoother with the product fused in (see ChebStep): t = A d never reaches
// memory.  false when the panel does not qualify for the wide-load kernel (the caller then runs product and step apart).
bool mh_spmm_f32_cheb_step(mh_context *ctx, const BsrLevel &lvl, const float *d_in, float *d_out, float *r, float *x, const float *dinv, float c1, float c2, uint32_t w) {
    if (w == 0) return true;
    const bool aligned16 = !((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(x)) & 15);
    if (w % 4 || w > 256 || !aligned16) return false;
    // algorithmic bytes: the product's (d read, no t written) plus the step's own passes: r and x read and written, d' written
    TimedLaunch timed(ctx, MH_KERNEL_SPMM, spmm_bytes(lvl, w, 4, 4, 0, true, false) + 5.0 * 4.0 * 3.0 * double(lvl.n_nodes) * w);
    constexpr int xcd = 1;
    const unsigned grid = (div_up(lvl.n_nodes, TB / 64) + 7) / 8 * 8;
    ChebStep epi;
    epi.r = r; epi.dinv = dinv; epi.d_out = d_out; epi.x = x; epi.c1 = c1; epi.c2 = c2;
    auto go = [&](auto cl_tag) {
        constexpr int CL = decltype(cl_tag)::value;
        if (latency_bound_level(lvl))
            k_spmm_wide<float, float, float, 4, CL, false, true, false, 1, 4, true><<<grid, TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval32.get(), static_cast<const float *>(nullptr), d_in,
                                                                                                                static_cast<float *>(nullptr), static_cast<float *>(nullptr), lvl.n_nodes, w, xcd, 0, 0, nullptr, epi);
        else
            k_spmm_wide<float, float, float, 4, CL, false, true, false, 1><<<grid, TB, 0, ctx->stream>>>(lvl.row_ptr, lvl.col, lvl.aval32.get(), static_cast<const float *>(nullptr), d_in,
                                                                                                       static_cast<float *>(nullptr), static_cast<float *>(nullptr), lvl.n_nodes, w, xcd, 0, 0, nullptr, epi);
    };
    const uint32_t lanes = w / 4;
    if (lanes <= 8) go(std::integral_constant<int, 8>{});
    else if (lanes <= 16) go(std::integral_constant<int, 16>{});
    else if (lanes <= 20) go(std::integral_constant<int, 20>{});
    else if (lanes <= 32) go(std::integral_constant<int, 32>{});
    else go(std::integral_constant<int, 64>{});
    KERNEL_CHECK();
    return true;
}

// fp32 product with the level's single-precision copy of A (preconditioner only).
void mh_spmm_f32(mh_context *ctx, const BsrLevel &lvl, const float *x, float *y, uint32_t w) {
    if (w == 0) return;
    TimedLaunch timed(ctx, MH_KERNEL_SPMM, spmm_bytes(lvl, w, 4, 4, 4, true, false));
    launch_spmm<float, false, true>(ctx, lvl, lvl.aval32.get(), x, y, static_cast<const float *>(nullptr), static_cast<float *>(nullptr), w);
}

// y (double) = A x for a single-precision panel x and the double-precision A: the residual path of the preconditioner.
// The pitch w must be a multiple of 4 (16-byte single-precision rows).
void mh_spmm_mixed(mh_context *ctx, const BsrLevel &lvl, const float *x, double *y, uint32_t w) {
    if (w == 0) return;
    TimedLaunch timed(ctx, MH_KERNEL_SPMM, spmm_bytes(lvl, w, 8, 4, 8, true, false));
    if (!launch_spmm_wide<double, float, double, false, true>(ctx, lvl, lvl.aval.get(), x, y, static_cast<const double *>(nullptr), static_cast<double *>(nullptr), w))
        mh_throw(MH_EINVAL, "mixed-precision product needs a 16-byte aligned panel of pitch %% 4 == 0 (got %u)", w);
}
