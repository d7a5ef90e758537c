// Tall-skinny dense block kernels of the eigensolver (gfx950), fp64 MFMA.
//
// gram:  G (wa x wb) = X^T Y for row-major panels X (n x wa), Y (n x wb), n ~ 10^5..10^6, w <= 256.
//   rocBLAS maps this shape (tiny m, n; huge k) onto a handful of workgroups; here the rows are split over ~2 per CU,
//   each workgroup stages KC rows of both panels in LDS once (every panel element leaves HBM exactly once) and its
//   waves accumulate 16x16 output tiles with v_mfma_f64_16x16x4_f64, operands read from LDS with ds_read_b64 at a row
//   pitch = 16 (mod 32) doubles (conflict free).  Partial Grams are summed in a fixed order by a second kernel, so the
//   result is bit-reproducible.  Roofline: max(8 n (wa + wb) bytes / HBM, 2 n wa wb flops / fp64 MFMA peak).
#include "mh_common.h"

#include <optional>

#include <map>

// hipFuncSetAttribute once per (kernel, device): function attributes are per device, and first calls may race between the host
// threads of concurrent solves -- a per-call-site table of once-flags indexed by the context's device.
struct PerDeviceOnce {
    static constexpr int MaxDevices = 64;
    std::once_flag flag[MaxDevices];
    template<typename F> void run(int device, F &&f) { std::call_once(flag[device >= 0 && device < MaxDevices ? device : 0], std::forward<F>(f)); }
};

// MH_TEST=own_gemm (A/B hook, round 5): the wide Gram blocks and basis updates of the 200-mode configuration through OUR kernels
// (k_gram_blocked cut into 160 x 80 blocks, k_combine in 256-column chunks) instead of the vendor's dgemm -- profiles/r05_config3_gemm_ab.txt
static bool mh_test_own_gemm() {
    static const bool on = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "own_gemm");
    return on;
}
namespace {
constexpr int KC = 32; // rows staged per step

typedef double double4_t __attribute__((ext_vector_type(4)));

__host__ __device__ inline int pad_pitch(int w) { // smallest p >= w with p = 16 (mod 32)
    int p = ((w + 15) / 16) * 16;
    if ((p % 32) != 16) p += 16;
    return p;
}

// Register-blocked Gram: every wave keeps a TI x TJ block of 16x16 output tiles in accumulators and, per 4-row k-step,
// reads TI + TJ operand fragments from LDS for TI*TJ MFMAs (0.4 LDS reads per MFMA at 5 x 5, against 2 when each tile
// fetches its own operands).  The workgroup's W = GI*GJ*KS waves tile the output GI x GJ ways and split the staged
// rows KS ways (wave ks takes the k-steps ks, ks+KS, ...); the KS partial blocks are folded through LDS in a fixed
// order at the end, so the result stays bit-reproducible.  Tile loops are unrolled at compile time and skipped with
// scalar branches on the kernel-argument widths: no exec-mask divergence around the MFMAs.
template<int TI, int TJ, int GI, int GJ, int KS, int OCC>
__global__ void __launch_bounds__(GI * GJ * KS * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) k_gram_blocked(const double *__restrict__ X, int ldx, int wa, const double *__restrict__ Y, int ldy, const uint32_t *__restrict__ ymap, int wb,
                                                                  size_t n, size_t rows_per_wg, double *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NTH = GI * GJ * KS * 64;
    constexpr int CA = TI * GI, CB = TJ * GJ; // 16-column strips of the two panels (all staged; missing ones as zeros)
    constexpr int PA = (CA * 16) % 32 == 16 ? CA * 16 : CA * 16 + 16, PB = (CB * 16) % 32 == 16 ? CB * 16 : CB * 16 + 16;
    double *Xs = smem, *Ys = smem + KC * PA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave % KS, gj = (wave / KS) % GJ, gi = wave / (KS * GJ);
    const int ti0 = gi * TI, tj0 = gj * TJ; // this wave's first tile row / column
    double4_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = double4_t{0, 0, 0, 0};
    const size_t r_begin = size_t(blockIdx.x) * rows_per_wg;
    const size_t r_end = min(n, r_begin + rows_per_wg);
    const int kk_lane = lane >> 4, c_lane = lane & 15;
    // staging map: 16 threads per row, NTH/16 rows per pass.  The hot loop carries no address arithmetic and no selects (beside
    // the fp64 matrix instructions of the CU's other workgroup every ordinary vector instruction waits for a gap between two of
    // them -- profiles/r03_combine_phases.txt): loads go to a scalar base that moves with the step plus per-thread byte offsets
    // fixed at kernel start, columns past the panels read a real column (their products land in outputs nobody stores), and only
    // a workgroup's last, partial step clamps its rows and writes zeros for the missing ones.
    constexpr int RPP = NTH / 16;
    static_assert(KC % RPP == 0 || RPP > KC, "staging passes");
    constexpr int PASSES = RPP >= KC ? 1 : KC / RPP;
    const int srow = tid >> 4, scol = tid & 15;
    const bool stager = srow < KC; // RPP > KC: the extra threads carry nothing
    const uint32_t xpitch = uint32_t(ldx) * 8u, ypitch = uint32_t(ldy) * 8u;
    uint32_t xoff[CA], yoff[CB]; // byte offset of this thread's column of each strip within a row
#pragma unroll
    for (int c = 0; c < CA; ++c) xoff[c] = uint32_t(min(scol + 16 * c, wa - 1)) * 8u;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int col = min(scol + 16 * c, wb - 1);
        yoff[c] = (ymap ? ymap[col] : uint32_t(col)) * 8u; // optional column map: Y's logical column -> physical column
    }
    double px[PASSES][CA], py[PASSES][CB];
    auto fetch = [&](size_t r0) {
        const char *xb = reinterpret_cast<const char *>(X + r0 * size_t(ldx)), *yb = reinterpret_cast<const char *>(Y + r0 * size_t(ldy)); // uniform
        if (r0 + KC <= r_end) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const uint32_t xr = uint32_t(stager ? ps * RPP + srow : 0) * xpitch, yr = uint32_t(stager ? ps * RPP + srow : 0) * ypitch;
#pragma unroll
                for (int c = 0; c < CA; ++c) px[ps][c] = *reinterpret_cast<const double *>(xb + size_t(xr + xoff[c]));
#pragma unroll
                for (int c = 0; c < CB; ++c) py[ps][c] = *reinterpret_cast<const double *>(yb + size_t(yr + yoff[c]));
            }
        } else { // rows past the end read the step's first row
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int k = ps * RPP + srow;
                const uint32_t row = (stager && r0 + k < r_end) ? uint32_t(k) : 0u;
#pragma unroll
                for (int c = 0; c < CA; ++c) px[ps][c] = *reinterpret_cast<const double *>(xb + size_t(row * xpitch + xoff[c]));
#pragma unroll
                for (int c = 0; c < CB; ++c) py[ps][c] = *reinterpret_cast<const double *>(yb + size_t(row * ypitch + yoff[c]));
            }
        }
    };
    auto commit = [&](size_t r0) { // registers -> LDS
        if (!stager) return;
        if (r0 + KC <= r_end) {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int k = ps * RPP + srow;
#pragma unroll
                for (int c = 0; c < CA; ++c) Xs[k * PA + scol + 16 * c] = px[ps][c];
#pragma unroll
                for (int c = 0; c < CB; ++c) Ys[k * PB + scol + 16 * c] = py[ps][c];
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int k = ps * RPP + srow;
                const bool rok = r0 + k < r_end;
#pragma unroll
                for (int c = 0; c < CA; ++c) Xs[k * PA + scol + 16 * c] = rok ? px[ps][c] : 0.0;
#pragma unroll
                for (int c = 0; c < CB; ++c) Ys[k * PB + scol + 16 * c] = rok ? py[ps][c] : 0.0;
            }
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (size_t r0 = r_begin; r0 < r_end; r0 += KC) {
        commit(r0);
        __syncthreads();
        if (r0 + KC < r_end) fetch(r0 + KC); // in flight while the MFMAs below run
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int step = 0; step < KC / (4 * KS); ++step) {
            const int kk = 4 * (ks + KS * step);
            const double *xr = Xs + (kk + kk_lane) * PA + ti0 * 16 + c_lane;
            const double *yr = Ys + (kk + kk_lane) * PB + tj0 * 16 + c_lane;
            double af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = xr[i * 16];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = yr[j * 16];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    // fold the KS row-splits of each output block through LDS (256 doubles per tile, lane-major), fixed order
    if (KS > 1) {
        double *fold = smem + size_t(gi * GJ + gj) * (TI * TJ * 256);
        for (int src = 1; src < KS; ++src) {
            if (ks == src) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) fold[((i * TJ + j) * 4 + reg) * 64 + lane] = acc[i][j][reg];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) acc[i][j][reg] += fold[((i * TJ + j) * 4 + reg) * 64 + lane];
            }
            __syncthreads();
        }
    }
    if (ks != 0) return;
    double *out = partial + size_t(blockIdx.x) * wa * wb;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = (tj0 + j) * 16 + c_lane;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = (ti0 + i) * 16 + kk_lane + 4 * reg;
                if (row < wa && col < wb) out[size_t(col) * wa + row] = acc[i][j][reg];
            }
        }
    }
}

// Sum of the workgroups' partial Grams, in a fixed order: a workgroup owns 64 consecutive entries (lanes: whole 512-byte
// runs of a partial), each of its 16 waves adds every 16th partial, then wave 0 adds the 16 sums in order.
__global__ void __launch_bounds__(1024) k_gram_reduce(const double *__restrict__ partial, int nwg, int wa, int wb, double *__restrict__ g, int ld) {
    __shared__ double s[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, idx = blockIdx.x * 64 + lane, total = wa * wb;
    double acc = 0;
    if (idx < total) {
#pragma unroll 8
        for (int w = wv; w < nwg; w += 16) acc += partial[size_t(w) * total + idx];
    }
    s[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && idx < total) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s[k][lane];
        g[size_t(idx / wa) * ld + idx % wa] = t;
    }
}

// ---- fused basis update ----------------------------------------------------------------------------------------
// Z = [X | W | P] * C for row-major panels X (n x wx), W (n x ww), P (n x wp) and a coefficient matrix C given
// row-major (k-major) as Ct[(wx+ww+wp)][nc]; the first n1 output columns go to out1 (n x n1), the rest to out2
// (n x (nc - n1)).  One launch replaces six rocBLAS dgemm calls of the LOBPCG update (new Ritz vectors and new search
// directions from the same basis) and reads the basis once.  A workgroup owns 64 rows (16 per wave) and all nc <= 256
// output columns; the basis rows and the matching coefficient rows are staged through LDS in K-chunks of 32 and
// multiplied with v_mfma_f64_16x16x4_f64.  Bound: 2 n m nc flops on fp64 MFMA vs 8 n (m + nc) bytes of HBM.
typedef __attribute__((address_space(1))) double GlobalDouble; // an address built from integers is a global one (a generic load would also count as an LDS access)
constexpr int CK = 16; // K chunk (32 measured slower: 475 vs 427 us at 75 + 75 -> 75)
// What shapes this kernel (in-kernel cycle stamps, profiles/r03_combine_phases.txt): beside the fp64 matrix instructions of the
// workgroup it shares the SIMDs with, a wave's ordinary vector instructions wait for a gap between two of them -- ~45 address /
// select instructions per chunk took 2 100 cycles, most of a 2 560-cycle matrix phase, and 120 of them around the output stores
// 11 000.  So the loop carries almost none: addresses come from a table in LDS (one multiply-add per staged basis element), from
// scalar bases with constant per-thread offsets (coefficients, outputs), and from compile-time LDS offsets (the chunk loop is
// unrolled over its two stage buffers).  Workgroups are persistent (row tiles dealt round-robin), so the table and the launch
// are paid once.
template<int NT, bool ACCUMULATE, bool MAPPED> // 16-column output tiles per wave (nc <= 16 * NT); ACCUMULATE: out += instead of out =; MAPPED: column maps on X / out1
__global__ void __launch_bounds__(256) k_combine(const double *__restrict__ X, int wx, int ldx, const uint32_t *__restrict__ xmap, const double *__restrict__ W, int ww,
                                                const double *__restrict__ P, int wp,
                                                const double *__restrict__ Ct, int ldc, int c0, int nc, size_t n, double *__restrict__ out1, int n1,
                                                double *__restrict__ out2, size_t split_stride, int ld1, const uint32_t *__restrict__ omap) {
    // X's logical column k lives at physical column xmap[k] of a panel of pitch ldx (xmap null: identity); out1's logical
    // column c goes to physical column omap[c] of a panel of pitch ld1 (in place over X is safe: a workgroup reads all of a
    // tile's 64 rows before it writes them)
    // this launch owns output columns c0 .. c0 + nc of the ldc the coefficient matrix has; with gridDim.y > 1 the K range
    // is cut into gridDim.y slices and slice s writes its partial product to out1 + s * split_stride
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = wx + ww + wp;
    const int kslice = (((m + int(gridDim.y) - 1) / int(gridDim.y)) + CK - 1) / CK * CK;
    const int kb = int(blockIdx.y) * kslice, ke = min(m, kb + kslice);
    if (gridDim.y > 1) out1 += size_t(blockIdx.y) * split_stride;
    constexpr int CP = (NT * 16) % 32 == 16 ? NT * 16 : NT * 16 + 16; // = 16 (mod 32): conflict-free B reads
    constexpr int SP = CK + 2; // A-tile pitch: rows 2 doubles apart mod 32 -> conflict-free ds_read_b64
    // two stage buffers: chunk c + 1 is written while chunk c is multiplied, one barrier per chunk
    constexpr int STAGE = 64 * SP + CK * CP; // 64 rows x SP of the basis, then CK x CP of the coefficients (all NT column strips)
    // Where basis column k lives: the address of its row-0 element and its panel's row pitch in bytes, for the columns of this
    // K slice (CK entries past the end repeat the last column).
    unsigned long long *kaddr = reinterpret_cast<unsigned long long *>(smem + 2 * STAGE);
    uint32_t *kpitch = reinterpret_cast<uint32_t *>(kaddr + (ke - kb + CK));
    for (int i = threadIdx.x; i < ke - kb + CK; i += 256) {
        const int k = min(kb + i, m - 1);
        const double *base = X;
        int col = MAPPED ? (k < wx ? int(xmap[k]) : 0) : k, pitch = ldx;
        if (k >= wx + ww) base = P, col = k - wx - ww, pitch = wp;
        else if (k >= wx) base = W, col = k - wx, pitch = ww;
        kaddr[i] = reinterpret_cast<unsigned long long>(base + col);
        kpitch[i] = uint32_t(pitch) * 8u;
    }
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // staging maps: S tile: thread -> (row = tid / 4, 4 consecutive k); C tile: thread -> (k row = tid / 16, NT columns 16 apart)
    static_assert(CK == 16, "one pass of 16 coefficient rows per chunk");
    constexpr int SJ = CK / 4;
    const int srow = tid >> 2, sk = (tid & 3) * SJ;
    const int ck = tid >> 4, cc = tid & 15;
    // Every load goes to a clamped, always-valid address and nothing is zeroed on the way to LDS except the basis columns past
    // the end of the K slice (last chunk only): rows past n and output columns past nc compute values that are never stored,
    // from finite data, and a zero basis entry silences whatever coefficient row it meets.
    // coefficient tile: a uniform base per chunk (scalar registers) plus per-thread byte offsets that never change; only the last
    // two column strips can pass nc, and only the last chunk can pass row m - 1
    const char *cbytes = reinterpret_cast<const char *>(Ct + c0);
    const uint32_t row_bytes = uint32_t(ldc) * 8u;
    const uint32_t coff = uint32_t(ck) * row_bytes + uint32_t(cc) * 8u;
    const uint32_t ctail0 = uint32_t(ck) * row_bytes + uint32_t(min(cc + 16 * (NT - 2), nc - 1)) * 8u, ctail1 = uint32_t(ck) * row_bytes + uint32_t(min(cc + 16 * (NT - 1), nc - 1)) * 8u;
    // outputs (C/D layout: column = lane & 15 of strip t, row = (lane >> 4) + 4 * reg): a strip lies in out1, in out2, or -- one at
    // most -- across the two; byte offsets of this lane's column within a row of either
    const int col_l = lane & 15, row_l = lane >> 4;
    const int pitch2 = ldc - n1;
    uint32_t ocol[MAPPED ? NT : 1]; // MAPPED: out1 columns through the map
    if (MAPPED) {
#pragma unroll
        for (int t = 0; t < NT; ++t) ocol[t] = omap[min(c0 + 16 * t + col_l, max(n1, 1) - 1)] * 8u;
    }
    double ps[SJ], pc[NT];
    const size_t tiles = (n + 63) / 64;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const size_t r0 = tile * 64;
        const uint32_t rs = uint32_t(min(r0 + srow, n - 1));
        double4_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = double4_t{0, 0, 0, 0};
        auto fetch = [&](int k0) {
#pragma unroll
            for (int j = 0; j < SJ; ++j) {
                const int i = k0 - kb + sk + j;
                ps[j] = *reinterpret_cast<const GlobalDouble *>(kaddr[i] + (unsigned long long)(rs) * kpitch[i]);
            }
            const char *cchunk = cbytes + size_t(k0) * row_bytes;
            uint32_t o = coff, o0 = ctail0, o1 = ctail1;
            if (k0 + CK > m) { // rows past m - 1 read row m - 1
                const uint32_t back = uint32_t(ck - min(ck, m - 1 - k0)) * row_bytes;
                o -= back, o0 -= back, o1 -= back;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) pc[t] = *reinterpret_cast<const double *>(cchunk + (t < NT - 2 ? size_t(o) + 128 * t : size_t(t == NT - 2 ? o0 : o1)));
        };
        auto commit = [&](int k0, double *stage) {
            double *Ss = stage + srow * SP + sk, *Cs = stage + 64 * SP + ck * CP + cc;
            if (k0 + CK <= ke) {
#pragma unroll
                for (int j = 0; j < SJ; ++j) Ss[j] = ps[j];
            } else {
#pragma unroll
                for (int j = 0; j < SJ; ++j) Ss[j] = k0 + sk + j < ke ? ps[j] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) Cs[16 * t] = pc[t];
        };
        // one chunk: the next one's loads go out first and land in the other buffer after this one's products
        auto chunk = [&](int k0, int B) {
            const bool more = k0 + CK < ke;
            if (more) fetch(k0 + CK);
            const double *Ss = smem + B * STAGE, *Cs = Ss + 64 * SP;
#pragma unroll
            for (int kk = 0; kk < CK; kk += 4) {
                const double a = Ss[(wave * 16 + col_l) * SP + kk + row_l];
                const double *brow = Cs + (kk + row_l) * CP + col_l;
                double bf[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bf[t] = brow[t * 16];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bf[t], acc[t], 0, 0, 0);
            }
            if (more) commit(k0 + CK, smem + (B ^ 1) * STAGE); // its readers passed the last barrier
            __syncthreads();
        };
        if (kb < ke) { // (an empty K slice -- more slices than chunks -- still stores its zeros)
            fetch(kb);
            commit(kb, smem); // (the previous tile's last chunk ended with a barrier)
        }
        __syncthreads();
        for (int k0 = kb, buf = 0; k0 < ke; k0 += CK, buf ^= 1) chunk(k0, buf);
        // stores: per accumulator row one 64-bit row address for each output (scalar base + row offset), strips at constant
        // byte offsets from it
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const size_t r = r0 + wave * 16 + row_l + 4 * reg;
            if (r >= n) continue;
            char *row1 = reinterpret_cast<char *>(out1 + r * size_t(ld1)) + (MAPPED ? 0 : (c0 + col_l) * 8);
            char *row2 = reinterpret_cast<char *>(out2 + r * size_t(pitch2)) + (c0 + col_l - n1) * 8;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t >= NT - 2 && t * 16 + col_l >= nc) continue; // only the last two strips can pass nc
                const int s0 = c0 + 16 * t; // first column of the strip (uniform)
                double *dst;
                if (s0 + 16 <= n1) dst = reinterpret_cast<double *>(row1 + (MAPPED ? size_t(ocol[t]) : size_t(128 * t)));
                else if (s0 >= n1) dst = reinterpret_cast<double *>(row2 + 128 * t);
                else dst = s0 + col_l < n1 ? reinterpret_cast<double *>(row1 + (MAPPED ? size_t(ocol[t]) : size_t(128 * t))) : reinterpret_cast<double *>(row2 + 128 * t);
                *dst = ACCUMULATE ? *dst + acc[t][reg] : acc[t][reg];
            }
        }
    }
}
__global__ void k_transpose_small(const double *__restrict__ c, int rows, int cols, int ld, double *__restrict__ ct, int col_off, int ct_cols, double scale = 1.0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i % rows, cc = i / rows;
    ct[size_t(r) * ct_cols + col_off + cc] = scale * c[size_t(cc) * ld + r];
}
} // namespace

namespace {
// One Gram block: G (wa x wb at leading dimension ld) = X[:, 0:wa]^T Y[:, 0:wb] for panels of row pitch ldx, ldy.
void gram_block(mh_context *ctx, size_t n, const double *x, uint32_t ldx, uint32_t wa, const double *y, uint32_t ldy, const uint32_t *ymap, uint32_t wb, double *g,
                uint32_t ld) {
    const int ti_n = int((wa + 15) / 16), tj_n = int((wb + 15) / 16);
    // ~2 workgroups per CU; each stages KC rows per step and owns a contiguous row range
    constexpr size_t nwg_cap = 512;
    int nwg = int(std::min<size_t>(nwg_cap, (n + KC - 1) / KC));
    size_t rows_per_wg = ((n + nwg - 1) / nwg + KC - 1) / KC * KC;
    nwg = int((n + rows_per_wg - 1) / rows_per_wg);
    const size_t need = size_t(nwg) * wa * wb * sizeof(double);
    if (ctx->gram_ws_bytes < need) {
        ctx->pool.release(ctx->gram_ws);
        ctx->gram_ws = ctx->pool.alloc(need + need / 4);
        ctx->gram_ws_bytes = need + need / 4;
    }
    double *workspace = static_cast<double *>(ctx->gram_ws);
    auto blocked = [&](auto ti, auto tj, auto gi, auto gj, auto ks, auto occ) {
        constexpr int TI = decltype(ti)::value, TJ = decltype(tj)::value, GI = decltype(gi)::value, GJ = decltype(gj)::value, KS = decltype(ks)::value, OCC = decltype(occ)::value;
        constexpr int CA = TI * GI, CB = TJ * GJ;
        constexpr int PA = (CA * 16) % 32 == 16 ? CA * 16 : CA * 16 + 16, PB = (CB * 16) % 32 == 16 ? CB * 16 : CB * 16 + 16;
        const size_t fold = KS > 1 ? size_t(GI * GJ) * TI * TJ * 256 * sizeof(double) : 0;
        const size_t bytes = std::max(size_t(KC) * (PA + PB) * sizeof(double), fold);
        static PerDeviceOnce attr;
        attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gram_blocked<TI, TJ, GI, GJ, KS, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
        k_gram_blocked<TI, TJ, GI, GJ, KS, OCC><<<nwg, GI * GJ * KS * 64, bytes, ctx->stream>>>(x, int(ldx), int(wa), y, int(ldy), ymap, int(wb), n, rows_per_wg, workspace);
    };
#define IC(v) std::integral_constant<int, v>{}
    if (ti_n <= 1 && tj_n <= 1) blocked(IC(1), IC(1), IC(1), IC(1), IC(4), IC(2));
    else if (ti_n <= 2 && tj_n <= 2) blocked(IC(2), IC(2), IC(1), IC(1), IC(4), IC(2));
    else if (ti_n <= 3 && tj_n <= 3) blocked(IC(3), IC(3), IC(1), IC(1), IC(4), IC(2));
    else if (ti_n <= 4 && tj_n <= 4) blocked(IC(4), IC(4), IC(1), IC(1), IC(4), IC(2));
    else if (ti_n <= 5 && tj_n <= 3) blocked(IC(5), IC(3), IC(1), IC(1), IC(4), IC(2));
    else if (ti_n <= 3 && tj_n <= 5) blocked(IC(3), IC(5), IC(1), IC(1), IC(4), IC(2));
    // wider blocks: at most 15 tiles per wave so that two waves share a SIMD (one hides the other's staging)
    else if (ti_n <= 5 && tj_n <= 6) blocked(IC(5), IC(3), IC(1), IC(2), IC(2), IC(2));
    else if (ti_n <= 6 && tj_n <= 5) blocked(IC(3), IC(5), IC(2), IC(1), IC(2), IC(2));
    else if (ti_n <= 10 && tj_n <= 6) blocked(IC(5), IC(3), IC(2), IC(2), IC(1), IC(2));
    else if (ti_n <= 6 && tj_n <= 10) blocked(IC(3), IC(5), IC(2), IC(2), IC(1), IC(2));
    // (an eight-wave 160 x 192 form -- same waves per CU, the X columns read once for twice the outputs -- measured slower in round 4:
    // 2.50 against 2.10 ms on 240 x 240; what a launch costs is tile slots, 15 per wave whether used or not: see mh_gram's cuts)
    // (a nine-wave form holding a whole 240 x 240 block -- 25 tiles per wave, every panel element read once -- was built in round 4
    // and ran 11.6 ms against the 2.07 ms of six launches of this form: 3 waves on a SIMD leave 170 registers for 200 of accumulators)
    else mh_throw(MH_EINVAL, "gram block %u x %u too wide", wa, wb);
#undef IC
    KERNEL_CHECK();
    k_gram_reduce<<<div_up(size_t(wa) * wb, 64), 1024, 0, ctx->stream>>>(workspace, nwg, int(wa), int(wb), g, int(ld));
    KERNEL_CHECK();
}
} // namespace

// G (wa x wb, column-major, leading dimension ld) = X^T Y.  Blocks wider than 160 x 96 columns are cut into a grid
// of column blocks (each a launch of the register-blocked kernel over the same rows).
void mh_gram(mh_context *ctx, size_t n, const double *x, uint32_t wa, const double *y, uint32_t wb, double *g, uint32_t ld, uint32_t ldy, const uint32_t *ymap) {
    if (!wa || !wb) return;
    if (!ldy) ldy = wb;
    // Wide blocks (both sides >= 128 columns: the 200-mode configuration) through the vendor's batched dgemm, one batch member per row
    // slab (split-K by hand: the output alone is four macro tiles), partials added in a fixed order by k_gram_reduce as for our own
    // kernel.  Row-major panels are column-major transposes: G = (X^T)(Y^T)^T = dgemm(N, T) on the stored arrays.
    // (below 128 columns a side the library loses to our kernel: 257 against 174 us on 80 x 80, 544 against 333 on 160 x 80 at 447 k rows)
    if (wa >= 128 && wb >= 128 && n >= 65536 && !ymap && !mh_test_own_gemm()) {
        const uint32_t slabs = 128; // (32 ... 256 slabs: 1.41 ... 1.45 ms on 240 x 240 at 542 k rows; 512: 1.54)
        const size_t rows = n / slabs, rest = n - rows * slabs; // the first `slabs` members take `rows` rows each, one more call the rest
        const size_t members = slabs + (rest ? 1 : 0), need = members * size_t(wa) * wb * sizeof(double);
        if (ctx->gram_ws_bytes < need) {
            ctx->pool.release(ctx->gram_ws);
            ctx->gram_ws = ctx->pool.alloc(need + need / 4);
            ctx->gram_ws_bytes = need + need / 4;
        }
        double *ws = static_cast<double *>(ctx->gram_ws);
        const double one = 1, zero = 0;
        ROCBLAS_CHECK(rocblas_dgemm_strided_batched(ctx->blas, rocblas_operation_none, rocblas_operation_transpose, rocblas_int(wa), rocblas_int(wb), rocblas_int(rows), &one, x,
                                                    rocblas_int(wa), rocblas_stride(rows * wa), y, rocblas_int(ldy), rocblas_stride(rows * ldy), &zero, ws, rocblas_int(wa),
                                                    rocblas_stride(size_t(wa) * wb), rocblas_int(slabs)));
        if (rest)
            ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_transpose, rocblas_int(wa), rocblas_int(wb), rocblas_int(rest), &one, x + rows * slabs * wa,
                                        rocblas_int(wa), y + rows * slabs * ldy, rocblas_int(ldy), &zero, ws + size_t(slabs) * wa * wb, rocblas_int(wa)));
        k_gram_reduce<<<div_up(size_t(wa) * wb, 64), 1024, 0, ctx->stream>>>(ws, int(members), int(wa), int(wb), g, int(ld));
        KERNEL_CHECK();
        return;
    }
    const bool a_long = wa >= wb;
    // A launch costs its tile SLOTS (15 per wave, 2 x 2 waves at the widest: 160 x 96), used or not: 240 x 240 cut evenly into 2 x 3
    // launches of 128 x 80 fills 40 of 60 slots each (2.10 ms at 542 k rows); cut as 160 + 80 by 3 x 80 it fills 50 of 60 and 25 of 30
    // (1.6 ms).  So: the long side greedily in 160s, the short side evenly in pieces of at most 96 columns.
    const uint32_t cap_long = 160, cap_short = 96;
    const uint32_t w_long = a_long ? wa : wb, w_short = a_long ? wb : wa;
    const uint32_t n_short = div_up(w_short, cap_short), step_short = (div_up(w_short, n_short) + 15) / 16 * 16;
    // (a remainder of under 32 columns would be a launch of mostly empty slots over all the rows: cut evenly then)
    const uint32_t rest = w_long % cap_long, n_long = div_up(w_long, cap_long);
    const uint32_t step_long = rest && rest < 32 ? (div_up(w_long, n_long) + 15) / 16 * 16 : cap_long;
    for (uint32_t l0 = 0; l0 < w_long; l0 += step_long)
        for (uint32_t s0 = 0; s0 < w_short; s0 += step_short) {
            const uint32_t i0 = a_long ? l0 : s0, j0 = a_long ? s0 : l0;
            const uint32_t ca = std::min(a_long ? step_long : step_short, wa - i0), cb = std::min(a_long ? step_short : step_long, wb - j0);
            gram_block(ctx, n, x + i0, wa, ca, ymap ? y : y + j0, ldy, ymap ? ymap + j0 : nullptr, cb, g + size_t(j0) * ld + i0, ld);
        }
}

// Row-major (k-major) packing of two column-major coefficient blocks side by side: ct[m][n1 + n2]
void mh_pack_coefficients(mh_context *ctx, const double *c1, uint32_t n1, const double *c2, uint32_t n2, uint32_t m, uint32_t ld, double *ct) {
    if (n1) {
        k_transpose_small<<<div_up(size_t(m) * n1, 256), 256, 0, ctx->stream>>>(c1, int(m), int(n1), int(ld), ct, 0, int(n1 + n2));
        KERNEL_CHECK();
    }
    if (n2) {
        k_transpose_small<<<div_up(size_t(m) * n2, 256), 256, 0, ctx->stream>>>(c2, int(m), int(n2), int(ld), ct, int(n1), int(n1 + n2));
        KERNEL_CHECK();
    }
}

namespace {
// full[xmap[k]][:] = ct[k][:] for the wx mapped rows (k-major coefficient matrices of pitch nc)
__global__ void k_spread_rows(const double *__restrict__ ct, const uint32_t *__restrict__ xmap, uint32_t wx, uint32_t nc, double *__restrict__ full) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= size_t(wx) * nc) return;
    full[size_t(xmap[i / nc]) * nc + i % nc] = ct[i];
}
__global__ void k_iota(uint32_t *p, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
} // namespace

namespace {
// Workgroups of an instantiation that fit a CU with `lds` bytes of dynamic LDS each (its two stage buffers may exceed the 64 KB
// a launch may ask for by default: the limit is raised once per device); asked once per (device, 4 KB LDS class).
template<int NT, bool ACCUMULATE, bool MAPPED> int combine_residency(mh_context *ctx, size_t lds) {
    static PerDeviceOnce attr;
    attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_combine<NT, ACCUMULATE, MAPPED>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
    static std::mutex guard;
    static std::map<std::pair<int, size_t>, int> known;
    std::lock_guard<std::mutex> lock(guard);
    auto [it, fresh] = known.try_emplace({ctx->device, lds >> 12}, 1);
    if (fresh) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_combine<NT, ACCUMULATE, MAPPED>), 256, ((lds >> 12) + 1) << 12) == hipSuccess && nb > 0) it->second = std::min(nb, 8);
    }
    return it->second;
}
} // namespace

// out1 (n x n1), out2 (n x (nc - n1)) = [X | W | P] * Ct  (+= when accumulate)
const uint32_t *mh_identity_map(mh_context *ctx) {
    if (!ctx->iota) {
        ctx->iota = static_cast<uint32_t *>(ctx->pool.alloc(1024 * sizeof(uint32_t)));
        k_iota<<<4, 256, 0, ctx->stream>>>(ctx->iota, 1024);
        KERNEL_CHECK();
    }
    return ctx->iota;
}

void mh_combine(mh_context *ctx, size_t n, const double *x, uint32_t wx, const double *w, uint32_t ww, const double *p, uint32_t wp, const double *ct, uint32_t nc,
                double *out1, uint32_t n1, double *out2, bool accumulate, uint32_t ldx, const uint32_t *xmap, uint32_t ld1, const uint32_t *omap, uint32_t col_begin,
                uint32_t col_count) {
    if (!nc) return;
    if (!col_count) { col_begin = 0; col_count = nc; }
    const bool caller_omap = omap != nullptr;
    if (col_begin + col_count > nc) mh_throw(MH_EINVAL, "combine: columns %u + %u exceed %u", col_begin, col_count, nc);
    const bool mapped = xmap != nullptr;
    if (mapped && accumulate) mh_throw(MH_EINVAL, "combine: column maps are not supported with accumulate");
    if (mapped && !omap) { // the mapped kernel writes out1 through a map: identity when the caller has none
        if (n1 > 1024) mh_throw(MH_EINVAL, "combine: %u mapped output columns exceed 1024", n1);
        omap = mh_identity_map(ctx);
    }
    // Wide blocks (the 200-mode configuration: several hundred basis and output columns): the vendor dgemm reaches 50-65 TF/s
    // on these tall-skinny shapes where our 256-column chunks stay at 36 (tools/probe/gemm_probe.py).  Row-major panels are
    // column-major transposes, so out^T (cols x n) = Ct^T-block (cols x k) * panel^T (k x n), one call per panel and output;
    // a column map on X becomes zero coefficient rows for the columns left out.
    constexpr bool wide_blas = true;
    const uint32_t m_total = wx + ww + wp;
    // (bench.py's roofline_combine: one timed span and one work figure per call, whichever path it takes)
    TimedLaunch timed(ctx, MH_KERNEL_COMBINE, 2.0 * double(n) * double(m_total) * double(col_count));
    std::optional<TimedLaunch> timed_full; // (the same span once more for the class of full-size updates: X and P of an iteration together)
    if (m_total >= 200 && col_count >= 128) timed_full.emplace(ctx, MH_KERNEL_COMBINE_FULL, 2.0 * double(n) * double(m_total) * double(col_count));
    if (ctx->time_kernels) ctx->totals[MH_KERNEL_COMBINE_BYTES].work += 8.0 * double(n) * (double(m_total) + double(col_count)), ctx->totals[MH_KERNEL_COMBINE_BYTES].launches += 1;
    if (wide_blas && !mh_test_own_gemm() && m_total >= 400 && col_count >= 128 && !caller_omap && out1 != x && out2 != x && n >= 65536) {
        const uint32_t px = ldx ? ldx : wx; // physical columns of the X panel
        const double *cx = ct; // coefficient rows of the X part, k-major with pitch nc
        DevArray<double> ct_full;
        if (mapped) { // rows of Ct spread to the mapped columns of X, zeros elsewhere
            ct_full.reset(ctx, size_t(px) * nc);
            ct_full.zero();
            k_spread_rows<<<div_up(size_t(wx) * nc, 256), 256, 0, ctx->stream>>>(ct, xmap, wx, nc, ct_full.get());
            KERNEL_CHECK();
            cx = ct_full.get();
        }
        const double one = 1, zero = 0;
        struct Part { const double *panel; uint32_t width, pitch; const double *coeff; };
        const Part parts[3] = {{x, mapped ? px : wx, px, cx}, {w, ww, ww, ct + size_t(wx) * nc}, {p, wp, wp, ct + size_t(wx + ww) * nc}};
        auto emit = [&](double *out, uint32_t pitch, uint32_t c_begin, uint32_t c_count) {
            if (!c_count) return;
            bool first = true;
            for (const Part &part : parts) {
                if (!part.panel || !part.width) continue;
                ROCBLAS_CHECK(rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, rocblas_int(c_count), rocblas_int(n), rocblas_int(part.width), &one,
                                            part.coeff + c_begin, rocblas_int(nc), part.panel, rocblas_int(part.pitch), (first && !accumulate) ? &zero : &one, out, rocblas_int(pitch)));
                first = false;
            }
        };
        const uint32_t c_end = col_begin + col_count;
        if (col_begin < n1) emit(out1 + col_begin, ld1 ? ld1 : n1, col_begin, std::min(c_end, n1) - col_begin);
        if (c_end > n1) emit(out2 + (std::max(col_begin, n1) - n1), nc - n1, std::max(col_begin, n1), c_end - std::max(col_begin, n1));
        if (mapped) HIP_CHECK(hipStreamSynchronize(ctx->stream)); // ct_full returns to the pool
        return;
    }
    // more than 256 output columns: column chunks, each a launch over the same basis
    const uint32_t chunks = div_up(col_count, 256), step = (div_up(col_count, chunks) + 15) / 16 * 16;
    for (uint32_t c0 = col_begin; c0 < col_begin + col_count; c0 += step) {
        const uint32_t ncc = std::min(step, col_begin + col_count - c0);
        auto go = [&](auto nt_tag) {
            constexpr int NT = decltype(nt_tag)::value;
            constexpr int CP = (NT * 16) % 32 == 16 ? NT * 16 : NT * 16 + 16;
            const size_t lds = 2 * (size_t(64) * (CK + 2) + size_t(CK) * CP) * sizeof(double) + size_t(m_total + CK) * 12;
            const size_t lds_m = lds;
            const int per_cu = accumulate ? combine_residency<NT, true, false>(ctx, lds) : mapped ? combine_residency<NT, false, true>(ctx, lds) : combine_residency<NT, false, false>(ctx, lds);
            const unsigned grid = unsigned(std::min<size_t>(div_up(n, 64), size_t(per_cu) * ctx->cu_count));
#define MH_COMBINE_ARGS x, int(wx), int(ldx ? ldx : wx), xmap, w, int(ww), p, int(wp), ct, int(nc), int(c0), int(ncc), n, out1, int(n1), out2, 0, int(ld1 ? ld1 : n1), omap
            if (accumulate) k_combine<NT, true, false><<<grid, 256, lds, ctx->stream>>>(MH_COMBINE_ARGS);
            else if (mapped) k_combine<NT, false, true><<<grid, 256, lds_m, ctx->stream>>>(MH_COMBINE_ARGS);
            else k_combine<NT, false, false><<<grid, 256, lds, ctx->stream>>>(MH_COMBINE_ARGS);
#undef MH_COMBINE_ARGS
        };
        const int ntile = int((ncc + 15) / 16);
        switch ((ntile + 1) / 2) { // the kernel computes all NT column strips: pick the smallest even NT that covers the chunk
            case 1: go(std::integral_constant<int, 2>{}); break;
            case 2: go(std::integral_constant<int, 4>{}); break;
            case 3: go(std::integral_constant<int, 6>{}); break;
            case 4: go(std::integral_constant<int, 8>{}); break;
            case 5: go(std::integral_constant<int, 10>{}); break;
            case 6: go(std::integral_constant<int, 12>{}); break;
            case 7: go(std::integral_constant<int, 14>{}); break;
            default: go(std::integral_constant<int, 16>{}); break;
        }
        KERNEL_CHECK();
    }
}

namespace {
__global__ void k_sum_slices(const double *__restrict__ partial, int slices, size_t count, double *__restrict__ out) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = 0;
    for (int q = 0; q < slices; ++q) s += partial[size_t(q) * count + i]; // fixed order
    out[i] = s;
}
} // namespace

// out (n x nc, row-major) = A (n x m, row-major) * Ct (m x nc, row-major) for a SHORT A (n of a few thousand rows, m
// comparable): the rows alone give too few workgroups, so the K range is cut into `slices` and the partial products
// are added in a fixed order.  Used for the coarse solve x0 = A0^-1 r0 of the preconditioner (n = m = 6 * aggregates).
void mh_short_product(mh_context *ctx, size_t n, const double *a, uint32_t m, const double *ct, uint32_t nc, double *out, double *partial, uint32_t slices) {
    if (!nc || !n) return;
    if (nc > 256) mh_throw(MH_EINVAL, "short product: %u columns exceed 256", nc);
    const size_t stride = n * nc;
    auto go = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr int CP = (NT * 16) % 32 == 16 ? NT * 16 : NT * 16 + 16;
        const uint32_t kslice = (div_up(m, slices) + CK - 1) / CK * CK; // as the kernel cuts the K range
        const size_t lds = 2 * (size_t(64) * (CK + 2) + size_t(CK) * CP) * sizeof(double) + size_t(std::min(kslice, m) + CK) * 12;
        if (lds > 160 * 1024) mh_throw(MH_EINVAL, "short product: %u basis columns per slice exceed the staging table", kslice);
        (void)combine_residency<NT, false, false>(ctx, lds); // (raises the kernel's LDS limit on first use)
        const dim3 grid(div_up(n, 64), slices); // one 64-row tile per workgroup: a few hundred workgroups in all, about one round of the device
        k_combine<NT, false, false><<<grid, 256, lds, ctx->stream>>>(a, int(m), int(m), nullptr, nullptr, 0, nullptr, 0, ct, int(nc), 0, int(nc), n, slices > 1 ? partial : out, int(nc), nullptr, stride,
                                                             int(nc), nullptr);
    };
    const int ntile = int((nc + 15) / 16);
    switch ((ntile + 1) / 2) {
        case 1: go(std::integral_constant<int, 2>{}); break;
        case 2: go(std::integral_constant<int, 4>{}); break;
        case 3: go(std::integral_constant<int, 6>{}); break;
        case 4: go(std::integral_constant<int, 8>{}); break;
        case 5: go(std::integral_constant<int, 10>{}); break;
        case 6: go(std::integral_constant<int, 12>{}); break;
        case 7: go(std::integral_constant<int, 14>{}); break;
        default: go(std::integral_constant<int, 16>{}); break;
    }
    KERNEL_CHECK();
    if (slices > 1) {
        k_sum_slices<<<div_up(stride, 256), 256, 0, ctx->stream>>>(partial, int(slices), stride, out);
        KERNEL_CHECK();
    }
}

// Row-major (k-major) packing of two column-major blocks stacked vertically, scaled: ct[(r1 + r2)][cols]
void mh_pack_stacked(mh_context *ctx, const double *c1, uint32_t r1, const double *c2, uint32_t r2, uint32_t cols, double scale, double *ct) {
    if (r1) {
        k_transpose_small<<<div_up(size_t(r1) * cols, 256), 256, 0, ctx->stream>>>(c1, int(r1), int(cols), int(r1), ct, 0, int(cols), scale);
        KERNEL_CHECK();
    }
    if (r2) {
        k_transpose_small<<<div_up(size_t(r2) * cols, 256), 256, 0, ctx->stream>>>(c2, int(r2), int(cols), int(r2), ct + size_t(r1) * cols, 0, int(cols), scale);
        KERNEL_CHECK();
    }
}

// ---- small symmetric tridiagonalisation ----------------------------------------------------------------------------
// Householder reduction A = Q T Q^T of a symmetric matrix of order m <= 256 in ONE workgroup (LAPACK dsytd2, lower
// storage convention on output: D, E, tau and the reflector tails below the subdiagonal of A, as rocsolver_dormtr
// expects).  rocSOLVER's sytrd spends 3.5 ms of a 5.0 ms syevd at m = 222 in ~170 tiny launches; here the 400 KB matrix
// stays in L2 and the 1 024 threads of one workgroup sweep it three times per column (matvec, rank-2 update), with the
// small vectors in LDS.  Bound by the barriers and the dependent sweeps of each of the m steps (running the last 128
// columns out of an LDS image changed nothing): the reflector scalars and the two dot products are therefore computed
// redundantly by every wave from LDS -- six barriers per step, no serial section.  The full symmetric matrix is kept up to date (both triangles), so every pass is column-major
// coalesced.  All reductions run in a fixed order: bit-reproducible.
namespace {
// Sum of buf[0 .. count) computed by every wave for itself (count <= 256, fixed order): no barrier, no broadcast.
__device__ inline double wave_sum_lds(const double *buf, int count, int lane) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        s += i < count ? buf[i] : 0.0;
    }
    return mh_wave_sum(s); // (DPP, not ds_bpermute: see mh_common.h)
}

__global__ void __launch_bounds__(1024) k_sytrd_small(double *__restrict__ A, int m, double *__restrict__ D, double *__restrict__ E, double *__restrict__ TAU) {
    // per step: x (raw column), v (reflector), w, squares / products for the wave-local reductions, matvec partials
    __shared__ double xs[256], v[256], wv[256], sq[256], part[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int k = 0; k + 1 < m; ++k) {
        const int l = m - k - 1; // order of the trailing block, rows/cols k+1 .. m-1
        // thread map of the two sweeps: RB row slots (power of two >= l) x NG column groups, all 1 024 threads busy
        int rb = 32;
        while (rb < l) rb <<= 1;
        const int ng = 1024 / rb, rr = tid & (rb - 1), cq = tid / rb;
        double *col = A + size_t(k) * m + (k + 1);
        if (tid < l) {
            const double x = col[tid];
            xs[tid] = x;
            sq[tid] = tid >= 1 ? x * x : 0.0;
        }
        __syncthreads(); // (1) column published
        // every thread derives the reflector scalars itself: no serial section, no broadcast
        const double xnorm2 = wave_sum_lds(sq, l, lane);
        const double alpha = xs[0];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xnorm2 > 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (tid == 0) {
            D[k] = A[size_t(k) * m + k];
            E[k] = beta;
            TAU[k] = tau;
        }
        if (tau == 0.0) { // H = I (uniform: every thread computed the same bits)
            __syncthreads();
            continue;
        }
        if (tid < l) {
            const double vi = tid == 0 ? 1.0 : xs[tid] * scale;
            v[tid] = vi;
            col[tid] = tid == 0 ? beta : vi; // subdiagonal entry, then the reflector tail
        }
        __syncthreads(); // (2) reflector published
        // p = tau * A22 v: each column group sums its columns, the groups are added in order
        double *a22 = A + size_t(k + 1) * m + (k + 1);
        double acc = 0.0;
        if (rr < l)
            for (int c = cq; c < l; c += ng) acc += a22[size_t(c) * m + rr] * v[c];
        part[cq * rb + rr] = acc;
        __syncthreads(); // (3) partial products published
        double p = 0.0;
        if (tid < l) {
            double sum = 0.0;
            for (int g = 0; g < ng; ++g) sum += part[g * rb + tid];
            p = tau * sum;
            sq[tid] = p * v[tid];
        }
        __syncthreads(); // (4) p . v terms published
        const double pv = wave_sum_lds(sq, l, lane);
        if (tid < l) wv[tid] = p - 0.5 * tau * pv * v[tid];
        __syncthreads(); // (5) w published
        // A22 -= v w^T + w v^T
        if (rr < l) {
            const double vr = v[rr], wr = wv[rr];
            double *arow = a22 + rr;
            for (int c = cq; c < l; c += ng) arow[size_t(c) * m] -= vr * wv[c] + wr * v[c];
        }
        __syncthreads(); // (6) trailing block updated
    }
    if (tid == 0) {
        D[m - 1] = A[size_t(m - 1) * m + (m - 1)];
        TAU[m - 1] = 0.0;
    }
}
} // namespace

// The same reduction with the rank-2 update of step k - 1 deferred into the product pass of step k: the trailing block is
// read, updated, written back and multiplied by the new reflector in ONE sweep (two sweeps per column instead of three, one
// barrier fewer).  Column k of the logically updated matrix is formed first, from the stored column and the pending pair.
#ifndef MH_SYTRD_THREADS
#define MH_SYTRD_THREADS 1024
#endif
__global__ void __launch_bounds__(MH_SYTRD_THREADS) k_sytrd_small_fused(double *__restrict__ A, int m, double *__restrict__ D, double *__restrict__ E, double *__restrict__ TAU) {
    __shared__ double xs[256], v[256], vp[256], wp[256], sq[256], xnext[256], part[MH_SYTRD_THREADS];
    const int tid = threadIdx.x, lane = tid & 63;
    bool pending = false; // (vp, wp): reflector and w of the previous step, indexed over ITS trailing block (this step's index + 1)
    for (int k = 0; k + 1 < m; ++k) {
        const int l = m - k - 1; // order of the trailing block, rows/cols k+1 .. m-1
        int rb = 32;
        while (rb < l) rb <<= 1;
        const int ng = max(1, MH_SYTRD_THREADS / rb), rr = tid & (rb - 1), cq = tid / rb;
        double *col = A + size_t(k) * m + (k + 1);
        // column k of the updated matrix: index 0 of the previous trailing block is this column, index i + 1 is row k + 1 + i
        // (after the first step the stored values come from xnext, where the previous sweep left the first column of its block:
        // no global round trip at the head of the step)
        if (tid < l) {
            double x = pending ? xnext[tid + 1] : col[tid];
            if (pending) x -= vp[tid + 1] * wp[0] + wp[tid + 1] * vp[0];
            xs[tid] = x;
            sq[tid] = tid >= 1 ? x * x : 0.0;
        }
        double dk = pending ? xnext[0] : A[size_t(k) * m + k];
        if (pending) dk -= 2.0 * vp[0] * wp[0];
        __syncthreads(); // (1) column published
        const double xnorm2 = wave_sum_lds(sq, l, lane);
        const double alpha = xs[0];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xnorm2 > 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (tid == 0) {
            D[k] = dk;
            E[k] = beta;
            TAU[k] = tau;
        }
        if (tid < l) {
            const double vi = tau == 0.0 ? 0.0 : (tid == 0 ? 1.0 : xs[tid] * scale);
            v[tid] = vi;
            col[tid] = tid == 0 ? beta : (tau == 0.0 ? xs[tid] : vi); // subdiagonal entry, then the reflector tail (H = I: the column itself, all zeros below)
        }
        __syncthreads(); // (2) reflector published
        // one sweep over the trailing block: apply the pending pair, write back, multiply by v
        double *a22 = A + size_t(k + 1) * m + (k + 1);
        double acc = 0.0;
        if (rr < l) {
            const double vpr = pending ? vp[rr + 1] : 0.0, wpr = pending ? wp[rr + 1] : 0.0;
            for (int c = cq; c < l; c += ng) {
                double a = a22[size_t(c) * m + rr];
                if (pending) {
                    a -= vpr * wp[c + 1] + wpr * vp[c + 1];
                    a22[size_t(c) * m + rr] = a;
                }
                if (c == 0) xnext[rr] = a; // first column of this block = column k + 1 (and its diagonal entry) for the next step
                acc += a * v[c];
            }
        }
        part[cq * rb + rr] = acc;
        __syncthreads(); // (3) partial products published
        double p = 0.0;
        if (tid < l) {
            double sum = 0.0;
            for (int g = 0; g < ng; ++g) sum += part[g * rb + tid];
            p = tau * sum;
            sq[tid] = p * v[tid];
        }
        __syncthreads(); // (4) p . v terms published
        const double pv = wave_sum_lds(sq, l, lane);
        if (tid < l) {
            vp[tid] = v[tid]; // becomes the pending pair of the next step (tau = 0: v = 0, w = 0, a no-op)
            wp[tid] = p - 0.5 * tau * pv * v[tid];
        }
        pending = true;
        __syncthreads(); // (5) pending pair published
    }
    if (tid == 0) {
        double dl = A[size_t(m - 1) * m + (m - 1)];
        if (pending && m >= 2) dl -= 2.0 * vp[0] * wp[0]; // the last pending pair lives on the 1 x 1 trailing block
        D[m - 1] = dl;
        TAU[m - 1] = 0.0;
    }
}

// ---- the same reduction on ONE CU with the matrix in REGISTERS (round 5) ---------------------------------------------------------
// The lower triangle of a symmetric matrix of order <= 256 is 263 KB: it fits the register file of one CU (eight waves x 256 registers).
// Thread (ti, tj) = (tid & 31, tid >> 5) of 512 holds the entries (ti + 32 a, tj + 16 b), b <= 2 a + 1, of a 256 x 256 frame with the
// matrix at its END (row r at position r + 256 - m, so that what is left of the matrix always ends with the last block); the 32 x 32
// blocks on the diagonal (b = 2 a, 2 a + 1) are held in full, both triangles.  Nothing of the matrix moves during the reduction.  Per
// column k, three barriers:
//   (P) every thread multiplies its entries by v both ways (an entry (r, c) of a block below the diagonal gives a_rc v_c to p_r and
//       a_rc v_r to p_c; a diagonal block, held in full, only the former); the parts go to LDS by tj / by ti;
//   (R) two threads per row add them in that order: p = tau A v;
//   (U) every wave forms p . v, every thread w = p - (tau / 2)(p . v) v at its rows and columns and applies A -= v w^T + w v^T to its
//       entries; the half-wave that holds column k + 1 first works out that column alone, its reflector (norm by DPP sums, one sqrt, two
//       divisions) and publishes v for the next column while the other waves are still updating.
// No exchange between workgroups, no co-residency assumption, sums in a fixed order (bit-reproducible).  The sixteen columns a column
// lies in are a template parameter: "has the reduction left this block behind" is decided by the compiler.  The barriers fence LDS ONLY:
// the stores to global memory (d, e, tau, the reflector columns -- nothing in the kernel reads them back) are never waited for.
namespace {
constexpr int REGS_RA = 8, REGS_CB = 16; // row blocks of 32, column blocks of 16
__device__ __forceinline__ void regs_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    mh_lds_writes_landed(); // (the compiler has been seen to drop this wait at a loop header: see mh_common.h)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#ifdef MH_REGS_STAMPS
__device__ unsigned long long g_regs_stamps[8];
#define REGS_STAMP(i) do { const unsigned long long now_ = clock64(); acc_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define REGS_STAMP(i) do { } while (0)
#endif
struct RegsShared {
    double vs[2][256], part[16][258], partc[32][257], ps[256], scal[2]; // (row lengths chosen against LDS bank conflicts: partc is written down its columns)
};
struct RegsOut {
    double *__restrict__ A, *__restrict__ D, *__restrict__ E, *__restrict__ TAU;
    int m, off;
};
using RegsMatrix = double[REGS_RA][REGS_CB];
// The half-wave tj == kn % 16 that holds column kn (column block BN): the column as the update of column kn - 1 leaves it (UPDATE; the
// registers are updated with everybody else's afterwards, to the same bits or not: the column is not read from them again), its
// reflector, and v, tau for the next column into the other LDS buffer.
template <int BN, bool UPDATE>
__device__ __forceinline__ void regs_next_reflector(const RegsMatrix &val, RegsShared &sh, const RegsOut &out, int kn, int ti, int tj, const double (&vr)[REGS_RA],
                                                    const double (&vc)[REGS_CB], double half) {
    constexpr int AN = BN / 2;
    const int k2 = kn + 1;
    __builtin_amdgcn_s_setprio(3); // (the one serial stretch of a column: ahead of the wave it shares its SIMD with)
    double x[REGS_RA], sq = 0.0, wc = 0.0;
    if (UPDATE) wc = sh.ps[tj + 16 * BN] - half * vc[BN];
#pragma unroll
    for (int a = AN; a < REGS_RA; ++a) {
        const int r = ti + 32 * a;
        double t = val[a][BN];
        if (UPDATE) {
            const double wr = sh.ps[r] - half * vr[a];
            t -= vr[a] * wc;
            t -= wr * vc[BN];
        }
        x[a] = t;
        sq += r > k2 ? t * t : 0.0;
        if (r == kn) out.D[kn - out.off] = t;
    }
    const int base = __builtin_amdgcn_readfirstlane(threadIdx.x & 32);
    sq = mh_row_sum(sq);
    sq = mh_lane_value(sq, base) + mh_lane_value(sq, base + 16);
    double alpha = x[AN]; // row k2 = kn + 1 lies in row block AN or the next one, at lane k2 % 32 of this half-wave
    if constexpr (AN + 1 < REGS_RA) alpha = (k2 >> 5) == AN ? x[AN] : x[AN + 1];
    alpha = mh_lane_value(alpha, base + (k2 & 31));
    double tau = 0.0, beta = alpha, scale = 0.0;
    if (sq > 0.0) {
        double norm, inverse_norm;
        mh_fast_sqrt_rsqrt(fma(alpha, alpha, sq), norm, inverse_norm);
        beta = -copysign(norm, alpha);
        tau = fma(alpha, copysign(inverse_norm, alpha), 1.0); // (beta - alpha) / beta = 1 - alpha / beta = 1 + |alpha| / norm
        scale = mh_fast_rcp(alpha - beta);
    }
    double *vnext = sh.vs[k2 & 1];
    const size_t column = size_t(kn - out.off) * out.m;
#pragma unroll
    for (int a = AN & ~1; a < REGS_RA; ++a) { // (from an even block: the dot product p . v reads whole pairs of blocks)
        const int r = ti + 32 * a;
        double v = 0.0;
        if (a >= AN) {
            v = r == k2 ? 1.0 : (r > k2 ? x[a] * scale : 0.0);
            if (r > kn) out.A[column + (r - out.off)] = r == k2 ? beta : v; // the subdiagonal entry, then the reflector tail (LAPACK's lower storage)
        }
        vnext[r] = v;
    }
    if (ti == 0) sh.scal[k2 & 1] = tau, out.E[kn - out.off] = beta, out.TAU[kn - out.off] = tau;
    __builtin_amdgcn_s_setprio(0);
}
// The columns at the positions 16 B ... 16 B + 15.
template <int B>
__device__ __forceinline__ void regs_block_columns(RegsMatrix &val, RegsShared &sh, const RegsOut &out, int tid, int ti, int tj, int lane
#ifdef MH_REGS_STAMPS
                                                   , unsigned long long (&acc_)[7]
#endif
) {
    constexpr int BA = B / 2; // the first live row block
    const int k_begin = max(16 * B, out.off), k_end = min(16 * B + 16, 255);
#ifdef MH_REGS_STAMPS
    unsigned long long last_ = clock64();
#endif
    for (int k = k_begin; k < k_end; ++k) {
        const int k1 = k + 1;
        const double *v = sh.vs[k1 & 1];
        const double tau = sh.scal[k1 & 1];
        // (P) p = A22 v
        double vr[REGS_RA], vc[REGS_CB];
#pragma unroll
        for (int a = BA; a < REGS_RA; ++a) vr[a] = v[ti + 32 * a];
#pragma unroll
        for (int b = B; b < REGS_CB; ++b) vc[b] = v[tj + 16 * b];
#pragma unroll
        for (int a = BA; a < REGS_RA; ++a) {
            double pr = 0.0;
#pragma unroll
            for (int b = B; b <= 2 * a + 1; ++b) pr += val[a][b] * vc[b];
            sh.part[tj][ti + 32 * a] = pr;
        }
#pragma unroll
        for (int b = B; b < REGS_CB - 2; ++b) {
            double pc = 0.0;
#pragma unroll
            for (int a = b / 2 + 1; a < REGS_RA; ++a) pc += val[a][b] * vr[a];
            sh.partc[ti][tj + 16 * b] = pc;
        }
        REGS_STAMP(0);
        regs_barrier();
        REGS_STAMP(1);
        // (R) p_r = tau (the row parts by tj, then the column parts by ti, in order): two threads per row, eight + sixteen parts each
        {
            const int r = tid >> 1, q = tid & 1;
            if (r >= 16 * B) { // (live rows; the column parts of a column block left behind are not written any more)
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) t += sh.part[8 * q + j][r];
                if (r < 32 * (REGS_RA - 1)) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) t += sh.partc[16 * q + j][r];
                }
                t += mh_dpp_move<MH_DPP_QUAD_XOR1>(t);
                if (q == 0) sh.ps[r] = tau * t;
            }
        }
        REGS_STAMP(2);
        regs_barrier();
        REGS_STAMP(3);
        // (U) w = p - (tau / 2) (p . v) v at this thread's rows and columns; A22 -= v w^T + w v^T on its entries, one term per sweep
        double pv = 0.0;
#pragma unroll
        for (int q = BA / 2; q < 4; ++q) pv += sh.ps[lane + 64 * q] * v[lane + 64 * q]; // (rows 64 (BA / 2) ... 32 BA - 1, if any: ps from an earlier column, times v = 0)
        pv = mh_wave_sum(pv);
        const double half = 0.5 * tau * pv;
        if (tj == (k1 & 15) && k1 < 255) { // column k + 1 first, by its holders: the next reflector is under way while the others update
            if ((k & 15) < 15) regs_next_reflector<B, true>(val, sh, out, k1, ti, tj, vr, vc, half);
            else if constexpr (B + 1 < REGS_CB) regs_next_reflector<B + 1, true>(val, sh, out, k1, ti, tj, vr, vc, half);
        }
        REGS_STAMP(4);
#pragma unroll
        for (int b = B; b < REGS_CB; ++b) {
            const double wc = sh.ps[tj + 16 * b] - half * vc[b];
#pragma unroll
            for (int a = b / 2; a < REGS_RA; ++a) val[a][b] -= vr[a] * wc;
        }
#pragma unroll
        for (int a = BA; a < REGS_RA; ++a) {
            const double wr = sh.ps[ti + 32 * a] - half * vr[a];
#pragma unroll
            for (int b = B; b <= 2 * a + 1; ++b) val[a][b] -= wr * vc[b];
        }
        REGS_STAMP(5);
        regs_barrier();
        REGS_STAMP(6);
    }
}
#ifdef MH_REGS_STAMPS
#define REGS_ACC , acc_
#else
#define REGS_ACC
#endif
template <int B> __device__ __forceinline__ void regs_from_block(RegsMatrix &val, RegsShared &sh, const RegsOut &out, int tid, int ti, int tj, int lane
#ifdef MH_REGS_STAMPS
                                                                 , unsigned long long (&acc_)[7]
#endif
) {
    if (out.off < 16 * B + 16) regs_block_columns<B>(val, sh, out, tid, ti, tj, lane REGS_ACC);
    if constexpr (B + 1 < REGS_CB) regs_from_block<B + 1>(val, sh, out, tid, ti, tj, lane REGS_ACC);
}
template <int B> __device__ __forceinline__ void regs_first_reflector(const RegsMatrix &val, RegsShared &sh, const RegsOut &out, int ti, int tj) {
    const double none_r[REGS_RA] = {}, none_c[REGS_CB] = {};
    if ((out.off >> 4) == B) regs_next_reflector<B, false>(val, sh, out, out.off, ti, tj, none_r, none_c, 0.0);
    else if constexpr (B + 1 < REGS_CB) regs_first_reflector<B + 1>(val, sh, out, ti, tj);
}
__global__ void __launch_bounds__(512) k_sytrd_regs(double *__restrict__ A, int m, double *__restrict__ D, double *__restrict__ E, double *__restrict__ TAU) {
    __shared__ RegsShared sh;
    const int tid = threadIdx.x, ti = tid & 31, tj = tid >> 5, lane = tid & 63, off = 256 - m;
    const RegsOut out{A, D, E, TAU, m, off};
#ifdef MH_REGS_STAMPS
    unsigned long long acc_[7] = {};
#endif
    RegsMatrix val; // val[a][b], b <= 2 a + 1: position (ti + 32 a, tj + 16 b) of the frame
#pragma unroll
    for (int a = 0; a < REGS_RA; ++a)
#pragma unroll
        for (int b = 0; b < REGS_CB; ++b)
            if (b <= 2 * a + 1) {
                const int r = ti + 32 * a - off, c = tj + 16 * b - off;
                val[a][b] = (r >= 0 && c >= 0) ? (r >= c ? A[size_t(c) * m + r] : A[size_t(r) * m + c]) : 0.0; // (the lower triangle is the input; zero outside the matrix: no guards in the sweeps)
            }
    if (tid < 256) sh.ps[tid] = 0.0, sh.vs[0][tid] = 0.0, sh.vs[1][tid] = 0.0;
    regs_barrier();
    if (tj == (off & 15) && off < 255) regs_first_reflector<0>(val, sh, out, ti, tj); // from the matrix as it is
    regs_barrier();
    regs_from_block<0>(val, sh, out, tid, ti, tj, lane REGS_ACC);
    if (tid == 511) D[m - 1] = val[REGS_RA - 1][REGS_CB - 1], TAU[m - 1] = 0.0; // the last diagonal entry: position (255, 255)
#ifdef MH_REGS_STAMPS
    if (tid == 0)
        for (int i = 0; i < 7; ++i) g_regs_stamps[i] += acc_[i];
#endif
}
#ifdef MH_REGS_STAMPS
} // namespace
extern "C" int mh_debug_regs_stamps(unsigned long long *out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_regs_stamps), sizeof(g_regs_stamps));
    if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_regs_stamps), z, sizeof(z)); }
    return 0;
}
namespace {
#endif
} // namespace

// ---- the same reduction on SEVERAL CUs ---------------------------------------------------------------------------
// The one-workgroup kernel streams the trailing block from L2 twice per column through one CU's memory path (~60 GB/s:
// 4 us per column, 0.9 ms at m = 230 -- the largest single kernel of a solve).  Here G workgroups hold the matrix in
// their LDS, column c with workgroup c mod G, kept fully symmetric so that (A v)_c is the dot product of column c with v:
//   per column k, every workgroup
//     1. knows the previous reflector pair (v, w) and column k as its owner last stored it; forms the up-to-date column,
//        its reflector v_k and tau_k for itself (redundantly: no broadcast hop);
//     2. sweeps its own columns c > k once: applies the previous pair, takes the dot product with v_k, and PUBLISHES
//        p_c -- and, if it owns column k + 1, that column as it now stands;
//     3. collects all of p and column k + 1 from the others and forms w_k for itself.
// One exchange per column, and it is not a barrier: every published double travels as two self-tagged 8-byte granules
// (32 payload bits + a 32-bit tag = launch epoch and step), written by a write-through (sc0 sc1) store and polled with
// cache-bypassing loads until both tags match -- no fence, no flag, no ordering needed (an aligned 8-byte granule is
// written whole).  Slots alternate by step parity: a workgroup can only write step k + 2 after it has read
// all of step k + 1, which exists only after every workgroup that publishes at step k + 1 has finished reading step k; a
// workgroup that leaves (no column left) publishes an acknowledgement at its last step instead, which the others collect.
// Only workgroups with blockIdx % 8 == 0 take part (blocks b and b + 8 share an XCD and its L2 as the dispatcher is
// observed to deal them; correctness does not depend on it).  Sums run in a fixed order for fixed G: bit-reproducible.
// Every poll is bounded: a workgroup that gives up raises *gave_up and the result is garbage the caller must not use.
#ifndef MH_SYTRD_GROUPS
#define MH_SYTRD_GROUPS 16
#endif
namespace {
constexpr size_t SYTRD_XCH_WORDS = 2 * 2 * 256 * 2; // value slots [parity][kind][index][2 granules]; then 2 x 2 acknowledgement words, then the give-up flag
constexpr size_t SYTRD_ACK_WORDS = 2 * 2;
constexpr int SYTRD_LD = 272; // LDS column stride in doubles: the four 16-lane column groups of a wave start 32 banks apart
typedef unsigned granule_pair __attribute__((ext_vector_type(4))); // {payload low, tag, payload high, tag}: two self-tagged 8-byte granules
// One 16-byte write-through store per value (a scalar-sized sc1 store is a fabric write of its own: half as many this way).
// If the store were ever torn, it would tear between the two granules, each of which carries its own tag.
__device__ __forceinline__ void publish_tagged(unsigned long long *slot, double value, unsigned tag) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(value);
    const granule_pair g = {unsigned(bits), tag, unsigned(bits >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot), "v"(g) : "memory");
}
// two values at once: both loads of a poll round are in flight together (one memory round trip per round)
__device__ __forceinline__ bool collect_tagged2(const unsigned long long *slot_a, const unsigned long long *slot_b, unsigned tag, double &a, double &b) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
        granule_pair ga, gb;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(ga), "=&v"(gb) : "v"(slot_a), "v"(slot_b) : "memory");
        if (ga.y == tag && ga.w == tag && gb.y == tag && gb.w == tag) {
            a = __longlong_as_double((long long)((unsigned long long)ga.x | ((unsigned long long)ga.z << 32)));
            b = __longlong_as_double((long long)((unsigned long long)gb.x | ((unsigned long long)gb.z << 32)));
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template<int G> __global__ void __launch_bounds__(256) k_sytrd_multi(double *__restrict__ A, int m, double *__restrict__ D, double *__restrict__ E, double *__restrict__ TAU,
                                                                   unsigned long long *__restrict__ xch, unsigned epoch, int *__restrict__ gave_up) {
    if (blockIdx.x % 8) return;
    const int g = blockIdx.x / 8, tid = threadIdx.x, lane = tid & 63;
    __shared__ double a[16 * SYTRD_LD];
    // xs: the up-to-date column of the step from its diagonal entry down; sq: its squares; pq: p . v terms; prs: p
    __shared__ double v[256], vp[256], wp[256], xs[264], sq[264], pq[256], prs[256];
    __shared__ int s_fail;
    // exchange slots: [parity][kind: 0 = p, 1 = column][index][2 granules]
    auto slot = [&](int parity, int kind, int index) { return xch + ((size_t(parity) * 2 + kind) * 256 + index) * 2; };
    auto ack_slot = [&](int parity) { return xch + SYTRD_XCH_WORDS + size_t(parity) * 2; }; // behind the value slots: one leaver per step
    const int jl = tid >> 4, t16 = tid & 15, cl = g + jl * G; // this thread's local column in the sweep
    const int last_col = g + G * ((m - 1 - g) / G);           // the last column this workgroup owns
    for (int j = 0; j < 16; ++j) {
        const int c = g + j * G;
        if (c < m && tid < m) a[j * SYTRD_LD + tid] = A[size_t(c) * m + tid];
    }
    if (tid < m) {
        const double x = A[tid]; // column 0
        xs[tid] = x;
        sq[tid] = tid >= 2 ? x * x : 0.0;
    }
    v[tid] = 0.0, vp[tid] = 0.0, wp[tid] = 0.0;
    if (tid == 0) s_fail = 0;
    for (int k = 0; k + 1 < m; ++k) {
        const int l = m - k - 1; // order of the trailing block, rows/cols k+1 .. m-1
        const unsigned tag = (epoch << 9) | unsigned(k + 1);
        const int parity = k & 1;
        const bool mine = g == k % G; // column k is this workgroup's: it reports the step's scalars and stores the reflector
        mh_lds_writes_landed(); // (hipcc drops the barrier's own LDS wait at this loop header: see the function)
        __syncthreads(); // (1) xs, sq and the pending pair (vp, wp) are in place
        const double xnorm2 = wave_sum_lds(sq, l + 1, lane);
        const double alpha = xs[1];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xnorm2 > 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (mine && tid == 0) {
            D[k] = xs[0];
            E[k] = beta;
            TAU[k] = tau;
        }
        double vi = 0.0;
        if (tid < l) {
            vi = tau == 0.0 ? 0.0 : (tid == 0 ? 1.0 : xs[tid + 1] * scale);
            v[k + 1 + tid] = vi;
            // the subdiagonal entry, then the reflector tail (tau = 0: the column is zero below it), for the back-transformation.
            // Step 0's store waits until this step's values have been collected: every workgroup reads column 0 of A at its
            // start, and only the collection proves that all of them are past that.
            if (mine && k > 0) A[size_t(k) * m + k + 1 + tid] = tid == 0 ? beta : vi;
        }
        // A workgroup without a column beyond k is done -- but its departure must be SEEN.  Every workgroup that stays
        // publishes at every step, so collecting a step's values proves that those have finished reading the step before,
        // which is what allows the slots to alternate.  The owner of column k, when that is its last one (k >= m - G: exactly
        // one workgroup leaves per step from there on), has nothing to publish at this step: it was still polling step
        // k - 1's slots a moment ago, and nothing the others collect at step k would show that it has stopped, so they could
        // reach step k + 1 and overwrite those slots under it.  It therefore publishes an acknowledgement (after its own
        // collection of step k - 1, in program order behind barrier (1)), and the workgroups that stay collect it below.
        const bool someone_leaves = k >= m - G;
        if (last_col <= k) {
            if (last_col == k && tid == 0) publish_tagged(ack_slot(parity), 1.0, tag);
            return;
        }
        __syncthreads(); // (2) reflector published inside the workgroup
        if (cl > k && cl < m) {
            const double vpc = vp[cl], wpc = wp[cl];
            double *col = a + jl * SYTRD_LD;
            double acc = 0.0;
            for (int r = k + 1 + t16; r < m; r += 16) {
                const double aa = col[r] - (vp[r] * wpc + wp[r] * vpc);
                col[r] = aa;
                acc += aa * v[r];
                if (cl == k + 1) publish_tagged(slot(parity, 1, r), aa, tag);
            }
            acc = mh_row_sum(acc);
            if (t16 == 0) publish_tagged(slot(parity, 0, cl), acc, tag);
        }
        double pr = 0.0, xn = 0.0;
        bool ok = true;
        if (tid < l) {
            ok = collect_tagged2(slot(parity, 0, k + 1 + tid), slot(parity, 1, k + 1 + tid), tag, pr, xn);
            pr *= tau;
            prs[tid] = pr;
            pq[tid] = pr * vi;
        } else if (someone_leaves && tid == 255) { // (l <= 255: this thread never collects a value) the leaver's acknowledgement
            double a0, a1;
            ok = collect_tagged2(ack_slot(parity), ack_slot(parity), tag, a0, a1);
        }
        if (!ok) s_fail = 1;
        __syncthreads(); // (3) p and its products with v published inside the workgroup
        if (s_fail) { // (uniform) give up: the others will time out in their own polls
            if (tid == 0) *gave_up = 1;
            return;
        }
        const double pv = wave_sum_lds(pq, l, lane);
        if (tid < l) {
            const double wi = pr - 0.5 * tau * pv * vi;
            const double v0 = v[k + 1], w0 = prs[0] - 0.5 * tau * pv * v0;
            const double xc = xn - (vi * w0 + wi * v0); // column k + 1 brought up to date with this step's pair
            if (mine && k == 0) A[size_t(k) * m + k + 1 + tid] = tid == 0 ? beta : vi;
            vp[k + 1 + tid] = vi;
            wp[k + 1 + tid] = wi;
            xs[tid] = xc;
            sq[tid] = tid >= 2 ? xc * xc : 0.0;
        }
    }
    __syncthreads();
    if (tid == 0) { // (the owner of column m - 1 is the one workgroup that gets here)
        D[m - 1] = xs[0];
        TAU[m - 1] = 0.0;
    }
}
} // namespace

// ---- the same for orders up to 768 (the Rayleigh-Ritz problem of a 215-pair solve: 3 x 240 columns) ---------------------------
// k_sytrd_multi keeps one matrix row per thread (256 threads) and sixteen columns per workgroup in LDS: order <= 256.  Here a workgroup
// has 1 024 threads -- still one row per thread, and one WAVE per local column in the sweep -- and still holds sixteen columns (column c with workgroup c mod G, G = 48: 100 KB of
// LDS for the columns, 43 KB for the step's vectors), and the workgroups sit on all eight XCDs -- 48 of them with 143 KB of LDS each
// do not fit the 32 CUs of one.  The exchange is the same: self-tagged granules, one round per column, bounded polls, the leaver's
// acknowledgement; the tag takes ten bits for the step.  rocSOLVER's sytrd, which this replaces above order 256, issues ~110 launches
// per column block (latrd: four kernels per column): 8.3 ms at order 720.
namespace {
constexpr int WIDE_MAXM = 768, WIDE_LD = 784, WIDE_COLS = 16, WIDE_G = 48;
constexpr size_t WIDE_XCH_WORDS = 2 * 2 * size_t(WIDE_MAXM) * 2;
__global__ void __launch_bounds__(1024) k_sytrd_wide(double *__restrict__ A, int m, double *__restrict__ D, double *__restrict__ E, double *__restrict__ TAU,
                                                    unsigned long long *__restrict__ xch, unsigned epoch, int *__restrict__ gave_up) {
    constexpr int G = WIDE_G;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *a = lds;                           // WIDE_COLS x WIDE_LD
    double *v = a + WIDE_COLS * WIDE_LD;       // reflector of the step, by global row
    double *vp = v + WIDE_MAXM, *wp = vp + WIDE_MAXM; // the pending pair, by global row
    double *xs = wp + WIDE_MAXM;               // the up-to-date column of the step from its diagonal entry down (l + 1 entries)
    double *sq = xs + WIDE_MAXM + 8, *pq = sq + WIDE_MAXM + 8, *prs = pq + WIDE_MAXM;
    __shared__ int s_fail;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6; // one thread per matrix row, one wave per local column
    auto slot = [&](int parity, int kind, int index) { return xch + ((size_t(parity) * 2 + kind) * WIDE_MAXM + index) * 2; };
    auto ack_slot = [&](int parity) { return xch + WIDE_XCH_WORDS + size_t(parity) * 2; };
    auto wave_sum = [&](const double *buf, int count) { // every wave for itself, fixed order
        double s_ = 0.0;
        for (int i = lane; i < count; i += 64) s_ += buf[i];
        s_ = mh_wave_sum(s_);
        return s_;
    };
    const int cl = g + wave * G; // this wave's column in the sweep
    if (g >= m) return;          // (fewer columns than workgroups: this one owns none)
    const int last_col = g + G * ((m - 1 - g) / G);
    for (int j = 0; j < WIDE_COLS; ++j) {
        const int c = g + j * G;
        if (c < m && tid < m) a[j * WIDE_LD + tid] = A[size_t(c) * m + tid];
    }
    if (tid < m) {
        const double x = A[tid]; // column 0
        xs[tid] = x;
        sq[tid] = tid >= 2 ? x * x : 0.0;
    }
    if (tid < WIDE_MAXM) v[tid] = 0.0, vp[tid] = 0.0, wp[tid] = 0.0;
    if (tid == 0) s_fail = 0;
    const int workgroups = m < G ? m : G; // the ones that own a column
    for (int k = 0; k + 1 < m; ++k) {
        const int l = m - k - 1;
        const unsigned tag = (epoch << 10) | unsigned(k + 1);
        const int parity = k & 1;
        const bool mine = g == k % G;
        mh_lds_writes_landed(); // (as in k_sytrd_multi)
        __syncthreads(); // (1) xs, sq and the pending pair are in place
        const double xnorm2 = wave_sum(sq, l + 1);
        const double alpha = xs[1];
        double tau = 0.0, beta = alpha, scale = 0.0;
        if (xnorm2 > 0.0) {
            beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (mine && tid == 0) {
            D[k] = xs[0];
            E[k] = beta;
            TAU[k] = tau;
        }
        double vi = 0.0;
        if (tid < l) {
            vi = tau == 0.0 ? 0.0 : (tid == 0 ? 1.0 : xs[tid + 1] * scale);
            v[k + 1 + tid] = vi;
            if (mine && k > 0) A[size_t(k) * m + k + 1 + tid] = tid == 0 ? beta : vi; // (step 0's store waits for the collection: see k_sytrd_multi)
        }
        const bool someone_leaves = k >= m - workgroups;
        if (last_col <= k) {
            if (last_col == k && tid == 0) publish_tagged(ack_slot(parity), 1.0, tag);
            return;
        }
        __syncthreads(); // (2) reflector published inside the workgroup
        if (cl > k && cl < m) {
            const double vpc = vp[cl], wpc = wp[cl];
            double *col = a + wave * WIDE_LD;
            double acc = 0.0;
            for (int r = k + 1 + lane; r < m; r += 64) {
                const double aa = col[r] - (vp[r] * wpc + wp[r] * vpc);
                col[r] = aa;
                acc += aa * v[r];
                if (cl == k + 1) publish_tagged(slot(parity, 1, r), aa, tag);
            }
            acc = mh_wave_sum(acc);
            if (lane == 0) publish_tagged(slot(parity, 0, cl), acc, tag);
        }
        double pr = 0.0, xn = 0.0;
        bool ok = true;
        if (tid < l) {
            ok = collect_tagged2(slot(parity, 0, k + 1 + tid), slot(parity, 1, k + 1 + tid), tag, pr, xn);
            pr *= tau;
            prs[tid] = pr;
            pq[tid] = pr * vi;
        } else if (someone_leaves && tid == 1023) { // (l <= 767: this thread never collects a value) the leaver's acknowledgement
            double a0, a1;
            ok = collect_tagged2(ack_slot(parity), ack_slot(parity), tag, a0, a1);
        }
        if (!ok) s_fail = 1;
        __syncthreads(); // (3) p and its products with v published inside the workgroup
        if (s_fail) {
            if (tid == 0) *gave_up = 1;
            return;
        }
        const double pv = wave_sum(pq, l);
        if (tid < l) {
            const double wi = pr - 0.5 * tau * pv * vi;
            const double v0 = v[k + 1], w0 = prs[0] - 0.5 * tau * pv * v0;
            const double xc = xn - (vi * w0 + wi * v0); // column k + 1 brought up to date with this step's pair
            if (mine && k == 0) A[size_t(k) * m + k + 1 + tid] = tid == 0 ? beta : vi;
            vp[k + 1 + tid] = vi;
            wp[k + 1 + tid] = wi;
            xs[tid] = xc;
            sq[tid] = tid >= 2 ? xc * xc : 0.0;
        }
    }
    __syncthreads();
    if (tid == 0) { // (the owner of column m - 1 is the one workgroup that gets here)
        D[m - 1] = xs[0];
        TAU[m - 1] = 0.0;
    }
}
} // namespace

// Orders 257 .. 768.  The exchange area is the context's (grown on first use); a give-up is reported through mh_sytrd_gave_up.
void mh_sytrd_wide(mh_context *ctx, double *a, uint32_t m, double *d, double *e, double *tau) {
    if (m < 2 || m > uint32_t(WIDE_MAXM)) mh_throw(MH_EINVAL, "sytrd_wide: order %u outside 2..%d", m, WIDE_MAXM);
    constexpr size_t words = WIDE_XCH_WORDS + SYTRD_ACK_WORDS;
    if (!ctx->sytrd_xch_wide) {
        ctx->sytrd_xch_wide = static_cast<unsigned long long *>(ctx->pool.alloc(words * sizeof(unsigned long long) + 64));
        HIP_CHECK(hipMemsetAsync(ctx->sytrd_xch_wide, 0, words * sizeof(unsigned long long) + 64, ctx->stream));
    }
    if ((++ctx->sytrd_epoch_wide & 0x3fffffu) == 0) { // the tag's epoch field wraps: clear the slots so that no old tag can match
        ctx->sytrd_epoch_wide = 1;
        HIP_CHECK(hipMemsetAsync(ctx->sytrd_xch_wide, 0, words * sizeof(unsigned long long) + 64, ctx->stream));
    }
    ctx->sytrd_flag = reinterpret_cast<int *>(ctx->sytrd_xch_wide + words);
    HIP_CHECK(hipMemsetAsync(ctx->sytrd_flag, 0, sizeof(int), ctx->stream));
    constexpr size_t lds = (size_t(WIDE_COLS) * WIDE_LD + 3 * size_t(WIDE_MAXM) + 2 * (size_t(WIDE_MAXM) + 8) + 2 * size_t(WIDE_MAXM)) * sizeof(double);
    static PerDeviceOnce attr;
    attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sytrd_wide), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds))); }); // (the kernel has four static bytes too: not the full 160 KB)
    k_sytrd_wide<<<WIDE_G, 1024, lds, ctx->stream>>>(a, int(m), d, e, tau, ctx->sytrd_xch_wide, ctx->sytrd_epoch_wide, ctx->sytrd_flag);
    KERNEL_CHECK();
}

void mh_sytrd_small(mh_context *ctx, double *a, uint32_t m, double *d, double *e, double *tau, int variant) {
    if (m < 1 || m > 256) mh_throw(MH_EINVAL, "sytrd_small: order %u outside 1..256", m);
    constexpr bool fused = true;
    // Default since round 5: the register-resident one-CU kernel (variant 3) at every order -- 43 / 95 / 407 / 506 us at 32 / 64 / 222 / 256
    // against 94 / 172 / 741 / 890 us of the better of the other two (one workgroup with the matrix in LDS below 64, sixteen workgroups
    // with a tagged exchange per column from there: profiles/r05_sytrd_regs.txt).  MH_TEST=sytrd_multi: the round-4 choice, for A/B runs.
    static const bool old_default = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "sytrd_multi");
    static const bool one_group = getenv("MH_TEST") && strstr(getenv("MH_TEST"), "sytrd_fused");
    if (variant < 0 && one_group) variant = 0;
    if (variant < 0 && !old_default) variant = 3;
    const bool multi = variant < 0 ? m >= 64 : variant == 1;
    // the give-up flag belongs to THIS call: a timeout of an earlier launch (co-resident work stalling a workgroup) must not
    // condemn every later reduction on the context (the one-workgroup kernels never raise it)
    if (ctx->sytrd_flag) HIP_CHECK(hipMemsetAsync(ctx->sytrd_flag, 0, sizeof(int), ctx->stream));
    if (multi && m >= 32) {
        constexpr int G = MH_SYTRD_GROUPS;
        constexpr size_t words = SYTRD_XCH_WORDS + SYTRD_ACK_WORDS;
        if (!ctx->sytrd_xch) {
            ctx->sytrd_xch = static_cast<unsigned long long *>(ctx->pool.alloc(words * sizeof(unsigned long long) + 64));
            HIP_CHECK(hipMemsetAsync(ctx->sytrd_xch, 0, words * sizeof(unsigned long long) + 64, ctx->stream));
        }
        if ((++ctx->sytrd_epoch & 0x7fffffu) == 0) { // the tag's epoch field wraps: clear the slots so that no old tag can match
            ctx->sytrd_epoch = 1;
            HIP_CHECK(hipMemsetAsync(ctx->sytrd_xch, 0, words * sizeof(unsigned long long) + 64, ctx->stream));
        }
        ctx->sytrd_flag = reinterpret_cast<int *>(ctx->sytrd_xch + words);
        HIP_CHECK(hipMemsetAsync(ctx->sytrd_flag, 0, sizeof(int), ctx->stream));
        k_sytrd_multi<G><<<8 * G, 256, 0, ctx->stream>>>(a, int(m), d, e, tau, ctx->sytrd_xch, ctx->sytrd_epoch, ctx->sytrd_flag);
    } else if (variant == 3) k_sytrd_regs<<<1, 512, 0, ctx->stream>>>(a, int(m), d, e, tau);
    else if (fused) k_sytrd_small_fused<<<1, MH_SYTRD_THREADS, 0, ctx->stream>>>(a, int(m), d, e, tau);
    else k_sytrd_small<<<1, 1024, 0, ctx->stream>>>(a, int(m), d, e, tau);
    KERNEL_CHECK();
}

// Whether a workgroup of the multi-workgroup reduction ever gave up waiting for the others on this context (their values never
// arrived within ~4 s): its output is then garbage.  Reads four bytes back; call it where the stream is synchronised anyway.
bool mh_sytrd_gave_up(mh_context *ctx) {
    if (!ctx->sytrd_flag) return false;
    int flag = 0;
    HIP_CHECK(hipMemcpyAsync(&flag, ctx->sytrd_flag, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return flag != 0;
}

// Z <- Q Z for the orthogonal factor of mh_sytrd_small (LAPACK's lower storage: reflector k has v[k+1] = 1 and its tail in
// A(k+2:, k)), one wave per column of Z: the column lives in registers (4 entries per lane, m <= 256), each reflector costs one
// wave reduction, the next reflector's entries are requested before the current one is applied (orders up to 768 after
// mh_sytrd_wide: 8 or 12 entries per lane).  Replaces rocSOLVER's blocked
// ormtr (~35 launches of larft/larfb pieces per call at these orders) by one launch.
namespace {
template<int Q, int PF> // Q entries of the column per lane (order <= 64 Q), PF reflectors in flight
__global__ void __launch_bounds__(64) k_apply_q(const double *__restrict__ A, const double *__restrict__ tau, int m, double *__restrict__ Z, int ldz) {
    const int lane = threadIdx.x;
    double *zc = Z + size_t(blockIdx.x) * ldz;
    double z[Q], v[PF][Q], tk[PF];
#pragma unroll
    for (int q = 0; q < Q; ++q) z[q] = lane + 64 * q < m ? zc[lane + 64 * q] : 0.0;
    auto load = [&](int k, double (&dst)[Q], double &t) { // reflector k on rows k + 1 .. m - 1 (k < 0: nothing, tau 0)
        t = k >= 0 ? tau[k] : 0.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = lane + 64 * q;
            dst[q] = (k >= 0 && i < m && i > k + 1) ? A[size_t(k) * m + i] : (i == k + 1 ? 1.0 : 0.0);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) load(m - 2 - u, v[u], tk[u]);
    for (int k0 = m - 2; k0 >= 0; k0 -= PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) { // reflector k0 - u sits in slot u; its slot is refilled with reflector k0 - u - PF once applied
            double dot = 0;
#pragma unroll
            for (int q = 0; q < Q; ++q) dot += v[u][q] * z[q];
            dot = mh_wave_sum(dot);
            const double t = tk[u] * dot;
#pragma unroll
            for (int q = 0; q < Q; ++q) z[q] -= t * v[u][q];
            load(k0 - u - PF, v[u], tk[u]);
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (lane + 64 * q < m) zc[lane + 64 * q] = z[q];
}
} // namespace

void mh_apply_q(mh_context *ctx, const double *a, const double *tau, uint32_t m, double *z, uint32_t ldz, uint32_t ncols) {
    if (m > 768) mh_throw(MH_EINVAL, "apply_q: order %u above 768", m);
    if (!ncols || m < 2) return;
    if (m <= 256) k_apply_q<4, 4><<<ncols, 64, 0, ctx->stream>>>(a, tau, int(m), z, int(ldz));
    else if (m <= 512) k_apply_q<8, 4><<<ncols, 64, 0, ctx->stream>>>(a, tau, int(m), z, int(ldz));
    else k_apply_q<12, 4><<<ncols, 64, 0, ctx->stream>>>(a, tau, int(m), z, int(ldz));
    KERNEL_CHECK();
}

// ---- lowest eigenpairs of a symmetric tridiagonal matrix, one workgroup ---------------------------------------------------
// The Rayleigh-Ritz step needs only the k lowest of the m eigenpairs (the active Ritz vectors), and it needs them as an
// orthonormal basis of the right invariant subspace, not as exactly diagonalising vectors (a rotation inside the wanted
// subspace of size 1e-10 changes nothing the iteration can see).  rocSOLVER's divide and conquer computes all m pairs through
// ~20 small launches (1.1 ms at m = 222).  Here, in one launch:
//   1. multisection (Sturm counts, LAPACK dlaebz pivot handling) for the k lowest eigenvalues, all in parallel, 1024 / k points
//      per eigenvalue and step;
//   2. two steps of inverse iteration per eigenvalue from a pseudo-random start, each by Gaussian elimination with partial
//      pivoting on T - lambda I (LAPACK dlagtf / dlagts), one thread per eigenvalue, vectors in LDS ([row][vector]: conflict
//      free), the U factor in a global scratch of the same layout;
//   3. classical Gram-Schmidt with reorthogonalisation over the k vectors (exact multiplets give independent vectors of the same
//      eigenspace from their different starts; close pairs come out of step 2 orthogonal to eps ||T|| / gap, which this repairs);
//   4. the residual max_j ||T x_j - lambda_j x_j||_inf / ||T|| goes to *quality for the caller to judge (fallback: divide and conquer).
// z (m x k, column-major, leading dimension ldz) receives the vectors, w the eigenvalues in ascending order.
#ifndef MH_TRIDIAG_ROUNDS
#define MH_TRIDIAG_ROUNDS 1 // one step from a random start leaves neighbours at eps ||T|| / gap ~ 1e-10, which is all the Rayleigh-Ritz step needs
#endif
namespace {
// Eigenvalue j of the tridiagonal matrix by 255-way multisection, one workgroup per eigenvalue (the counts are independent, and
// one CU alone is throughput-bound on the k * m * evaluations divisions): 8 steps instead of the 17 thirteen-way steps the
// single-workgroup kernel needs.  lam[j] = j-th smallest eigenvalue.
template<int R> // order <= 256 R
__global__ void __launch_bounds__(256) k_tridiag_values(const double *__restrict__ D, const double *__restrict__ E, int m, double *__restrict__ lam) {
    __shared__ double d[256 * R], e2[256 * R], sabs[256 * R], bnd[2];
    __shared__ int cnt[256];
    const int tid = threadIdx.x, j = blockIdx.x;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int i = tid + 256 * q;
        const double ea = i + 1 < m ? E[i] : 0.0;
        d[i] = i < m ? D[i] : 0.0;
        e2[i] = ea * ea;
        sabs[i] = fabs(ea);
    }
    __syncthreads();
    double gl = 1.7976931348623157e308, gh = -1.7976931348623157e308, emax = 0.0;
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int i = tid + 256 * q;
        emax = fmax(emax, e2[i]);
        if (i < m) {
            const double r = (i ? sabs[i - 1] : 0.0) + sabs[i];
            gl = fmin(gl, d[i] - r);
            gh = fmax(gh, d[i] + r);
        }
    }
    __shared__ double rl[256], rh[256], rm[256];
    rl[tid] = gl; rh[tid] = gh; rm[tid] = emax;
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
        if (tid < half) { rl[tid] = fmin(rl[tid], rl[tid + half]); rh[tid] = fmax(rh[tid], rh[tid + half]); rm[tid] = fmax(rm[tid], rm[tid + half]); }
        __syncthreads();
    }
    const double eps = 2.220446049250313e-16;
    const double tnorm = fmax(fabs(rl[0]), fabs(rh[0]));
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, rm[0]);
    if (tid == 0) {
        bnd[0] = rl[0] - (2.0 * tnorm * eps * m + 2.0 * pivmin);
        bnd[1] = rh[0] + (2.0 * tnorm * eps * m + 2.0 * pivmin);
    }
    __syncthreads();
    for (int s = 0; s < 8; ++s) {
        const double a = bnd[0], b = bnd[1];
        const double x = a + (b - a) * (double(tid + 1) / 256.0); // tid = 255 evaluates b itself: count(b) >= j + 1 by construction
        int c = 0;
        double q = d[0] - x;
        if (q <= pivmin) { ++c; q = fmin(q, -pivmin); }
        for (int i = 1; i < m; ++i) {
            double r = __builtin_amdgcn_rcp(q);
            r = fma(fma(-q, r, 1.0), r, r);
            q = fma(-e2[i - 1], r, d[i]) - x;
            if (q <= pivmin) { ++c; q = fmin(q, -pivmin); }
        }
        cnt[tid] = c;
        __syncthreads();
        // the first point whose count reaches j + 1 bounds the eigenvalue from above, its predecessor from below
        if (c >= j + 1 && (tid == 0 || cnt[tid - 1] < j + 1)) {
            bnd[1] = x;
            bnd[0] = tid == 0 ? a : a + (b - a) * (double(tid) / 256.0);
        }
        __syncthreads();
    }
    if (tid == 0) lam[j] = 0.5 * (bnd[0] + bnd[1]);
}

// ---- the same for orders 257 .. 768 (the 200-mode configuration), where the k vectors no longer fit one workgroup's LDS ----------
//   1. k_tridiag_values<3>: one workgroup per eigenvalue, as above;
//   2. k_tridiag_invit: one thread per eigenvalue (one wave per workgroup), the vector and the U factor in global memory
//      ([row][vector]: a wave's accesses are one 512-byte run), the start vector generated on the fly;
//   3. Cholesky-QR twice on the k vectors (Gram matrix and updates by mh_small_gemm, the order-k factorisation and triangular
//      inverse by the library): exact multiplets and close pairs come out of step 2 as independent but not orthogonal vectors;
//   4. k_tridiag_residual: max_j ||T z_j - lambda_j z_j||_inf / ||T|| for the caller to judge, as in the one-workgroup kernel.
constexpr int WIDE_T = 768;
__device__ inline void tridiag_scale(const double *d, const double *e, int m, int tid, int nthreads, double *red, double &tnorm, double &pivmin) {
    // Gershgorin bound of |T| and the pivot floor, by one wave or workgroup (red: 2 * 16 doubles of LDS); all threads return the same bits
    double gh = 0.0, em = 0.0;
    for (int i = tid; i < m; i += nthreads) {
        const double el = i ? fabs(e[i - 1]) : 0.0, er = i + 1 < m ? fabs(e[i]) : 0.0;
        gh = fmax(gh, fabs(d[i]) + el + er);
        em = fmax(em, er * er);
    }
    for (int off = 32; off > 0; off >>= 1) {
        gh = fmax(gh, __shfl_xor(gh, off, 64));
        em = fmax(em, __shfl_xor(em, off, 64));
    }
    if ((tid & 63) == 0) red[tid >> 6] = gh, red[16 + (tid >> 6)] = em;
    __syncthreads();
    tnorm = 0.0;
    double e2max = 0.0;
    for (int q = 0; q < (nthreads + 63) / 64; ++q) tnorm = fmax(tnorm, red[q]), e2max = fmax(e2max, red[16 + q]);
    pivmin = 2.2250738585072014e-308 * fmax(1.0, e2max);
    __syncthreads();
}
__global__ void __launch_bounds__(64) k_tridiag_invit(const double *__restrict__ D, const double *__restrict__ E, int m, int k, const double *__restrict__ lam, double *__restrict__ X,
                                                     double *__restrict__ ufac) {
    __shared__ double d[WIDE_T], e[WIDE_T], red[32];
    const int tid = threadIdx.x, j = blockIdx.x * 64 + tid;
    for (int i = tid; i < m; i += 64) d[i] = D[i], e[i] = i + 1 < m ? E[i] : 0.0;
    __syncthreads();
    double tnorm, pivmin;
    tridiag_scale(d, e, m, tid, 64, red, tnorm, pivmin);
    if (j >= k) return;
    const double tiny = fmax(2.220446049250313e-16 * tnorm, pivmin);
    const double lj = lam[j];
    unsigned long long st = 0x9e3779b97f4a7c15ull * (unsigned long long)(j + 1);
    auto next_start = [&]() { // pseudo-random start, different per vector (splitmix-style), entries in (-1, 1)
        st += 0x9e3779b97f4a7c15ull;
        unsigned long long zz = st;
        zz = (zz ^ (zz >> 30)) * 0xbf58476d1ce4e5b9ull;
        zz = (zz ^ (zz >> 27)) * 0x94d049bb133111ebull;
        zz ^= zz >> 31;
        return double(zz >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    };
    // forward elimination with row interchanges (dlagtf), applied to the right-hand side on the way
    double cd = d[0] - lj, cs = m > 1 ? e[0] : 0.0;
    double ri = next_start();
    for (int i = 0; i + 1 < m; ++i) {
        const double sb = e[i], nd = d[i + 1] - lj, ns = i + 2 < m ? e[i + 1] : 0.0;
        double rn = next_start();
        double u0, u1, u2, keep;
        if (fabs(cd) >= fabs(sb)) {
            if (fabs(cd) < tiny) cd = copysign(tiny, cd);
            const double mult = sb / cd;
            u0 = cd; u1 = cs; u2 = 0.0;
            cd = nd - mult * cs;
            cs = ns;
            rn -= mult * ri;
            keep = ri;
        } else {
            const double mult = cd / sb;
            u0 = sb; u1 = nd; u2 = ns;
            cd = cs - mult * nd;
            cs = -mult * ns;
            keep = rn;
            rn = ri - mult * rn;
        }
        X[size_t(i) * k + j] = keep;
        ufac[(size_t(3) * i) * k + j] = u0;
        ufac[(size_t(3) * i + 1) * k + j] = u1;
        ufac[(size_t(3) * i + 2) * k + j] = u2;
        ri = rn;
    }
    if (fabs(cd) < tiny) cd = copysign(tiny, cd);
    // back substitution (dlagts), overwriting the right-hand side; the rows come back eight at a time
    double y1 = ri / cd, y2 = 0.0, big = fabs(y1);
    X[size_t(m - 1) * k + j] = y1;
    for (int i0 = m - 2; i0 >= 0; i0 -= 8) {
        double u[8][3], x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = max(i0 - q, 0);
            u[q][0] = ufac[(size_t(3) * i) * k + j];
            u[q][1] = ufac[(size_t(3) * i + 1) * k + j];
            u[q][2] = ufac[(size_t(3) * i + 2) * k + j];
            x[q] = X[size_t(i) * k + j];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 - q;
            if (i < 0) break;
            const double y = (x[q] - u[q][1] * y1 - u[q][2] * y2) / u[q][0];
            X[size_t(i) * k + j] = y;
            big = fmax(big, fabs(y));
            y2 = y1;
            y1 = y;
        }
    }
    // unit 2-norm, scaled by the largest entry first (the solution of a nearly singular system is huge)
    const double ib = 1.0 / fmax(big, 2.2250738585072014e-308);
    double nrm = 0.0;
    for (int i = 0; i < m; ++i) {
        const double v = X[size_t(i) * k + j] * ib;
        nrm += v * v;
    }
    const double sc = ib / sqrt(nrm);
    for (int i = 0; i < m; ++i) X[size_t(i) * k + j] *= sc;
}
// quality[0] = max over the k columns of ||T z - lambda z||_inf / ||T|| (bits of a non-negative double: atomicMax on the word; NaN
// compares above everything), one workgroup per column of Z (m x k column-major).  quality[0] is zeroed by the caller.
__global__ void __launch_bounds__(256) k_tridiag_residual(const double *__restrict__ D, const double *__restrict__ E, int m, const double *__restrict__ lam, const double *__restrict__ Z, int ldz,
                                                         unsigned long long *__restrict__ quality) {
    __shared__ double d[WIDE_T], e[WIDE_T], red[32], worst_of[4];
    const int tid = threadIdx.x, j = blockIdx.x;
    for (int i = tid; i < m; i += 256) d[i] = D[i], e[i] = i + 1 < m ? E[i] : 0.0;
    __syncthreads();
    double tnorm, pivmin;
    tridiag_scale(d, e, m, tid, 256, red, tnorm, pivmin);
    const double *z = Z + size_t(j) * ldz, lj = lam[j];
    double worst = 0.0;
    bool bad = false;
    for (int i = tid; i < m; i += 256) {
        const double r = (d[i] - lj) * z[i] + (i ? e[i - 1] * z[i - 1] : 0.0) + (i + 1 < m ? e[i] * z[i + 1] : 0.0);
        bad |= !(fabs(r) <= 1.7976931348623157e308);
        worst = fmax(worst, fabs(r));
    }
    if (bad) worst = __longlong_as_double(0x7ff8000000000000ll);
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(worst, off, 64);
        worst = (o != o || worst != worst) ? __longlong_as_double(0x7ff8000000000000ll) : fmax(worst, o);
    }
    if ((tid & 63) == 0) worst_of[tid >> 6] = worst;
    __syncthreads();
    if (tid == 0) {
        double q = 0.0;
        bool nan = false;
        for (int a = 0; a < 4; ++a) nan |= worst_of[a] != worst_of[a], q = fmax(q, worst_of[a]);
        q = nan ? __longlong_as_double(0x7ff8000000000000ll) : q / fmax(tnorm, 2.2250738585072014e-308);
        atomicMax(quality, static_cast<unsigned long long>(__double_as_longlong(q)));
    }
}
__global__ void __launch_bounds__(1024) k_tridiag_lowest(const double *__restrict__ D, const double *__restrict__ E, int m, int k, double *__restrict__ w,
                                                        double *__restrict__ z, int ldz, double *__restrict__ ufac, double *__restrict__ quality, const double *__restrict__ lam_in) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *d = sm, *e = d + 256, *e2 = e + 256, *lam = e2 + 256, *lo = lam + 256, *hi = lo + 256, *coef = hi + 256, *red = coef + 256; // red: 1024
    int *cnt = reinterpret_cast<int *>(red + 1024); // 1024 ints
    double *X = red + 1024 + 512; // m x k, [row][vector]
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 256) {
        d[tid] = tid < m ? D[tid] : 0.0;
        const double ev = tid + 1 < m ? E[tid] : 0.0;
        e[tid] = ev;
        e2[tid] = ev * ev;
    }
    __syncthreads();
    // Gershgorin interval and the scale of T
    {
        double gl = 1.7976931348623157e308, gh = -1.7976931348623157e308;
        for (int i = tid; i < m; i += 1024) {
            const double r = (i ? fabs(e[i - 1]) : 0.0) + fabs(e[i]);
            gl = fmin(gl, d[i] - r);
            gh = fmax(gh, d[i] + r);
        }
        for (int off = 32; off > 0; off >>= 1) {
            gl = fmin(gl, __shfl_xor(gl, off, 64));
            gh = fmax(gh, __shfl_xor(gh, off, 64));
        }
        if (lane == 0) { red[tid >> 6] = gl; red[16 + (tid >> 6)] = gh; }
    }
    __syncthreads();
    double glo = red[0], ghi = red[16];
    for (int q = 1; q < 16; ++q) { glo = fmin(glo, red[q]); ghi = fmax(ghi, red[16 + q]); }
    const double tnorm = fmax(fabs(glo), fabs(ghi));
    const double eps = 2.220446049250313e-16;
    double e2max = 0;
    for (int i = 0; i < m; ++i) e2max = fmax(e2max, e2[i]); // uniform, 222 LDS reads
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, e2max);
    glo -= 2.0 * tnorm * eps * m + 2.0 * pivmin;
    ghi += 2.0 * tnorm * eps * m + 2.0 * pivmin;
    __syncthreads();
    // number of eigenvalues <= x (dlaebz)
    auto count = [&](double x) {
        int c = 0;
        double q = d[0] - x;
        if (q <= pivmin) { ++c; q = fmin(q, -pivmin); }
        for (int i = 1; i < m; ++i) {
            // reciprocal by the hardware estimate and one Newton step (the IEEE division sequence is 3x the instructions, and
            // this loop is what the phase costs; a count only needs the sign of q, and |q| is kept away from 0 by pivmin)
            double r = __builtin_amdgcn_rcp(q);
            r = fma(fma(-q, r, 1.0), r, r);
            q = fma(-e2[i - 1], r, d[i]) - x;
            if (q <= pivmin) { ++c; q = fmin(q, -pivmin); }
        }
        return c;
    };
    const long long t_start = wall_clock64();
    // 1. multisection: eigenvalue j (0-based) is the smallest x with count(x) >= j + 1
    const int tpe = min(16, 1024 / k); // points per eigenvalue and step
    const int grp = tid / tpe, sub = tid % tpe;
    if (tid < k) { lo[tid] = lam_in ? lam_in[tid] : glo; hi[tid] = lam_in ? lam_in[tid] : ghi; }
    __syncthreads();
    const int steps = lam_in ? 0 : int(ceil(58.0 / log2(double(tpe + 1)))) + 1; // eigenvalues given: k_tridiag_values ran before
    for (int s = 0; s < steps; ++s) {
        double a = 0, b = 0;
        if (grp < k) {
            a = lo[grp];
            b = hi[grp];
            const double x = a + (b - a) * (double(sub + 1) / double(tpe + 1));
            cnt[tid] = count(x);
        }
        __syncthreads();
        if (grp < k && sub == 0) {
            // first point whose count reaches j + 1 bounds the eigenvalue from above, its predecessor from below
            double na = a, nb = b;
            for (int t = 0; t < tpe; ++t) {
                const double x = a + (b - a) * (double(t + 1) / double(tpe + 1));
                if (cnt[grp * tpe + t] >= grp + 1) { nb = x; break; }
                na = x;
            }
            lo[grp] = na;
            hi[grp] = nb;
        }
        __syncthreads();
    }
    if (tid < k) lam[tid] = 0.5 * (lo[tid] + hi[tid]);
    __syncthreads();
    const long long t_bisect = wall_clock64();
    // 2. inverse iteration, one thread per eigenvalue; U factor rows in ufac[(3 i + q) * k + j]
    const double tiny = fmax(eps * tnorm, pivmin);
    if (tid < k) {
        const int j = tid;
        const double lj = lam[j];
        // pseudo-random start, different per vector (splitmix-style), entries in (-1, 1)
        unsigned long long st = 0x9e3779b97f4a7c15ull * (unsigned long long)(j + 1);
        for (int i = 0; i < m; ++i) {
            st += 0x9e3779b97f4a7c15ull;
            unsigned long long zz = st;
            zz = (zz ^ (zz >> 30)) * 0xbf58476d1ce4e5b9ull;
            zz = (zz ^ (zz >> 27)) * 0x94d049bb133111ebull;
            zz ^= zz >> 31;
            X[i * k + j] = double(zz >> 11) * (2.0 / 9007199254740992.0) - 1.0;
        }
        for (int round = 0; round < MH_TRIDIAG_ROUNDS; ++round) {
            // forward elimination with row interchanges, applied to the right-hand side on the way
            double cd = d[0] - lj, cs = m > 1 ? e[0] : 0.0; // current row: diagonal, superdiagonal (second superdiagonal is 0 before a swap)
            double ri = X[j];
            for (int i = 0; i + 1 < m; ++i) {
                const double sb = e[i], nd = d[i + 1] - lj, ns = i + 2 < m ? e[i + 1] : 0.0;
                double rn = X[(i + 1) * k + j];
                double u0, u1, u2;
                // (u0 holds the RECIPROCAL of the pivot: the elimination needs it anyway, and the back substitution then has no division in its
                // chain; the reciprocal by the hardware estimate and Newton steps -- one per row, each waiting for the previous row's)
                if (fabs(cd) >= fabs(sb)) {
                    if (fabs(cd) < tiny) cd = copysign(tiny, cd);
                    u0 = mh_fast_rcp(cd);
                    const double mult = sb * u0;
                    u1 = cs; u2 = 0.0;
                    cd = nd - mult * cs;
                    cs = ns;
                    rn -= mult * ri;
                    X[i * k + j] = ri;
                } else {
                    u0 = mh_fast_rcp(sb);
                    const double mult = cd * u0;
                    u1 = nd; u2 = ns;
                    cd = cs - mult * nd;
                    cs = -mult * ns;
                    const double t = ri;
                    X[i * k + j] = rn;
                    rn = t - mult * rn;
                }
                ufac[(size_t(3) * i) * k + j] = u0;
                ufac[(size_t(3) * i + 1) * k + j] = u1;
                ufac[(size_t(3) * i + 2) * k + j] = u2;
                ri = rn;
            }
            if (fabs(cd) < tiny) cd = copysign(tiny, cd);
            // back substitution, overwriting the right-hand side; then normalise
            double y1 = ri / cd, y2 = 0.0, nrm = y1 * y1;
            X[(m - 1) * k + j] = y1;
            // the factor rows come back from global memory eight at a time (one L2 round trip per eight dependent steps)
            for (int i0 = m - 2; i0 >= 0; i0 -= 8) {
                double u[8][3];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = max(i0 - q, 0);
                    u[q][0] = ufac[(size_t(3) * i) * k + j];
                    u[q][1] = ufac[(size_t(3) * i + 1) * k + j];
                    u[q][2] = ufac[(size_t(3) * i + 2) * k + j];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 - q;
                    if (i < 0) break;
                    const double y = (X[i * k + j] - u[q][1] * y1 - u[q][2] * y2) * u[q][0];
                    X[i * k + j] = y;
                    nrm += y * y;
                    y2 = y1;
                    y1 = y;
                }
            }
            // guard against overflow of the squared norm: rescale by the largest entry first when needed
            if (!(nrm < 1e300)) {
                double big = 0;
                for (int i = 0; i < m; ++i) big = fmax(big, fabs(X[i * k + j]));
                nrm = 0;
                for (int i = 0; i < m; ++i) { const double v = X[i * k + j] / big; X[i * k + j] = v; nrm += v * v; }
            }
            const double inv = 1.0 / sqrt(nrm);
            for (int i = 0; i < m; ++i) X[i * k + j] *= inv;
        }
    }
    __syncthreads();
    const long long t_invit = wall_clock64();
    // 3. classical Gram-Schmidt, left-looking over the vectors; a second pass only when the first cancelled more than half of
    //    the (unit) vector ("twice is enough").  Two barriers per pass: the coefficients come out of lane shuffles (eight
    //    adjacent lanes per earlier vector), finished vectors stay UNNORMALISED until the end (the coefficient of vector a
    //    carries 1 / |x_a|^2, which the eight threads that own a remember), and the squared norm is read by every wave for
    //    itself.  (A panel-wise form -- block projection twice, then Cholesky-QR of eight vectors twice -- was built and
    //    measured slower: 308 against 217 us at m = 222, k = 74; its tiny in-register factorisations are dependent chains of
    //    square roots and divisions.)
    double my_inv2 = 1.0; // 1 / |x_a|^2 of the earlier vector this thread takes coefficients for (a = tid >> 3)
    for (int j = 0; j < k; ++j) {
        double left = 1.0;
        for (int pass = 0; pass < 2; ++pass) {
            // coefficient of vector a in vector j: eight threads per a, each a strided eighth of the rows, added in a fixed order
            {
                const int a = tid >> 3, part = tid & 7;
                double s = 0;
                if (a < j) {
#pragma unroll 8
                    for (int i = part; i < m; i += 8) s += X[i * k + a] * X[i * k + j];
                }
                s = mh_quad_sum(s);
                s += mh_dpp_move<MH_DPP_ROW_HALF_MIRROR>(s); // (eight lanes)
                if (a < j && part == 0) coef[a] = s * my_inv2;
            }
            __syncthreads();
            // x_j -= X[:, :j] c: four threads per row; the squares of the new entries go out for the norm
            {
                const int i = tid >> 2, part = tid & 3;
                double s = 0;
                if (i < m) {
#pragma unroll 8
                    for (int a = part; a < j; a += 4) s += coef[a] * X[i * k + a];
                }
                s = mh_quad_sum(s);
                if (part == 0) {
                    double v = 0.0;
                    if (i < m) {
                        v = X[i * k + j] - s;
                        X[i * k + j] = v;
                    }
                    red[i] = v * v; // i < 256
                }
            }
            __syncthreads();
            left = wave_sum_lds(red, m, lane); // every wave the same bits; red is next written behind the next barrier
            if (left > 0.5 || j == 0) break; // uniform
        }
        if ((tid >> 3) == j) my_inv2 = 1.0 / left;
        if (tid == 0) lo[j] = 1.0 / sqrt(left); // (lo / hi are free after the multisection) the vector's final scale
    }
    __syncthreads();
    for (int idx = tid; idx < m * k; idx += 1024) X[idx] *= lo[idx % k];
    __syncthreads();
    const long long t_gs = wall_clock64();
    // 4. residual of the finished pairs, and output
    double worst = 0;
    for (int idx = tid; idx < m * k; idx += 1024) {
        const int i = idx / k, j = idx % k;
        const double x = X[idx];
        const double r = (d[i] - lam[j]) * x + (i ? e[i - 1] * X[idx - k] : 0.0) + (i + 1 < m ? e[i] * X[idx + k] : 0.0);
        worst = fmax(worst, fabs(r));
        z[size_t(j) * ldz + i] = x;
    }
    for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off, 64));
    __syncthreads();
    if (lane == 0) red[tid >> 6] = worst;
    __syncthreads();
    if (tid == 0) {
        double q = 0;
        for (int a = 0; a < 16; ++a) q = fmax(q, red[a]);
        *quality = q / fmax(tnorm, 2.2250738585072014e-308);
        quality[1] = double(t_bisect - t_start); // phase durations in 100 MHz ticks (diagnostics, MH_VERBOSE)
        quality[2] = double(t_invit - t_bisect);
        quality[3] = double(t_gs - t_invit);
        quality[4] = double(wall_clock64() - t_gs);
    }
    if (tid < k) w[tid] = lam[tid];
}
} // namespace

// false when the problem does not fit the one-workgroup kernel (the vectors must fit in LDS)
bool mh_tridiag_lowest(mh_context *ctx, const double *d, const double *e, uint32_t m, uint32_t k, double *w, double *z, uint32_t ldz, double *ufac, double *quality,
                       double *lam_scratch) {
    if (m < 2 || m > 256 || k < 1 || k > m || k > 128) return false;
    const size_t lds = (size_t(7) * 256 + 1024 + 512 + size_t(m) * k) * sizeof(double);
    if (lds > 158 * 1024) return false;
    static PerDeviceOnce attr;
    attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tridiag_lowest), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
    constexpr bool spread = true;
    if (spread) {
        k_tridiag_values<1><<<k, 256, 0, ctx->stream>>>(d, e, int(m), lam_scratch);
        KERNEL_CHECK();
    }
    k_tridiag_lowest<<<1, 1024, lds, ctx->stream>>>(d, e, int(m), int(k), w, z, int(ldz), ufac, quality, spread ? lam_scratch : nullptr);
    KERNEL_CHECK();
    return true;
}

// Orders 257 .. 768, k <= 256: see the comment above k_tridiag_invit.  work: 2 k m + 3 k m + 2 k k + k doubles; z (m x k, ldz) receives the
// orthonormal vectors, w the eigenvalues; *quality_host the residual measure (NaN when a factorisation failed).  Synchronises the stream.
bool mh_tridiag_lowest_wide(mh_context *ctx, const double *d, const double *e, uint32_t m, uint32_t k, double *w, double *z, uint32_t ldz, double *work, int *info2, double *quality_host) {
    if (m <= 256 || m > WIDE_T || k < 1 || k > 256 || k > m) return false;
    const size_t km = size_t(k) * m;
    double *xt = work, *xt2 = xt + km, *ufac = xt2 + km, *g = ufac + 3 * km, *linv = g + size_t(k) * k, *qual = linv + size_t(k) * k;
    k_tridiag_values<3><<<k, 256, 0, ctx->stream>>>(d, e, int(m), w);
    KERNEL_CHECK();
    k_tridiag_invit<<<div_up(k, 64), 64, 0, ctx->stream>>>(d, e, int(m), int(k), w, xt, ufac);
    KERNEL_CHECK();
    DevArray<int> info4_array(ctx, 4);
    int *info4 = info4_array.get();
    HIP_CHECK(hipMemsetAsync(info4, 0, 4 * sizeof(int), ctx->stream));
    // xt is X^T (k x m column-major).  Pass 1: xt2 = L^-1 xt; pass 2: z = xt2^T L^-T
    for (int pass = 0; pass < 2; ++pass) {
        const double *src = pass == 0 ? xt : xt2;
        mh_small_gemm(ctx, false, true, k, k, m, 1.0, src, k, src, k, 0.0, g, k);
        mh_potrf(ctx, g, k, k, info4 + 2 * pass);
        HIP_CHECK(hipMemsetAsync(linv, 0, size_t(k) * k * sizeof(double), ctx->stream));
        ROCBLAS_CHECK(rocblas_dtrtri(ctx->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, rocblas_int(k), g, rocblas_int(k), linv, rocblas_int(k)));
        if (pass == 0) mh_small_gemm(ctx, false, false, k, m, k, 1.0, linv, k, src, k, 0.0, xt2, k);
        else mh_small_gemm(ctx, true, true, m, k, k, 1.0, src, k, linv, k, 0.0, z, ldz);
    }
    HIP_CHECK(hipMemsetAsync(qual, 0, sizeof(double), ctx->stream));
    k_tridiag_residual<<<k, 256, 0, ctx->stream>>>(d, e, int(m), w, z, int(ldz), reinterpret_cast<unsigned long long *>(qual));
    KERNEL_CHECK();
    int hinfo[4] = {0, 0, 0, 0};
    HIP_CHECK(hipMemcpyAsync(quality_host, qual, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(hinfo, info4, sizeof(hinfo), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (hinfo[0] != 0 || hinfo[2] != 0) *quality_host = std::numeric_limits<double>::quiet_NaN();
    return true;
}

// ---- small Cholesky --------------------------------------------------------------------------------------------------
// Lower Cholesky factor of a symmetric positive definite matrix of order w <= 128 (column-major, ld w) by ONE workgroup
// with the matrix in LDS: the Gram matrices of the Cholesky-QR steps.  info = 0, or k + 1 when the pivot of column k is not
// positive (as LAPACK's potrf).  rocSOLVER's potf2 takes ~100 us at these orders; this runs ~3 barriers per column.
namespace {
__global__ void __launch_bounds__(256) k_potrf_small(double *__restrict__ A, int w, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double L[]; // w x w, column-major, pitch w + 1 (bank spread)
    __shared__ int s_fail;
    const int tid = threadIdx.x, pitch = w + 1;
    for (int idx = tid; idx < w * w; idx += 256) L[(idx / w) * pitch + idx % w] = A[idx];
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int k = 0; k < w; ++k) {
        const double akk = L[k * pitch + k];
        if (!(akk > 0.0)) { // uniform: every thread reads the same value
            if (tid == 0) s_fail = k + 1;
            break;
        }
        const double d = sqrt(akk), inv = 1.0 / d;
        __syncthreads(); // everyone has read the pivot
        for (int i = k + tid; i < w; i += 256) L[k * pitch + i] = i == k ? d : L[k * pitch + i] * inv;
        __syncthreads();
        // trailing update of the lower triangle: columns j > k, rows i >= j
        const int rem = w - k - 1;
        for (int idx = tid; idx < rem * rem; idx += 256) {
            const int j = k + 1 + idx / rem, i = k + 1 + idx % rem;
            if (i >= j) L[j * pitch + i] -= L[k * pitch + i] * L[k * pitch + j];
        }
        __syncthreads();
    }
    __syncthreads();
    for (int idx = tid; idx < w * w; idx += 256) {
        const int j = idx / w, i = idx % w;
        if (i >= j) A[idx] = L[j * pitch + i]; // the strictly upper part keeps the input, as potrf
    }
    if (tid == 0) {
        info[0] = s_fail;
        // info[1]: 16 * log2(largest / smallest diagonal entry of L) -- for a Gram matrix scaled to unit diagonal the square of
        // that ratio estimates its condition number, which is what the caller's Cholesky-QR loses in orthogonality
        double lo = 1.7976931348623157e308, hi = 0;
        for (int k = 0; k < w; ++k) { const double v = L[k * pitch + k]; lo = fmin(lo, v); hi = fmax(hi, v); }
        info[1] = s_fail || !(lo > 0) ? 1 << 20 : int(16.0 * log2(hi / lo));
    }
}
} // namespace

// The same factorisation eight columns at a time: two barriers per PANEL instead of three per column.  Every thread factors
// the 8 x 8 diagonal block for itself in registers (uniform work, no broadcast), a thread per row solves the panel below
// it, then the trailing triangle takes the panel's eight rank-1 updates one after the other.  Every entry goes through
// exactly the operations of the column-by-column kernel in the same order: the two produce the same bits.
namespace {
// With linv: the Cholesky-QR step of the search directions in ONE launch -- factor, L <- diag(1 / dscale) L (k_unscale_chol's rule), and
// linv = L^-1 (full w x w, zeros above the diagonal) by a blocked in-place inversion in LDS, eight columns per step from the last block to
// the first: the 8 x 8 diagonal block is inverted in registers by every thread, the panel below becomes -T (P D^-1) with T the inverse
// of the trailing block already in place (two threads per row, alternate columns).  Replaces the chain potrf, unscale, memset, the
// library's trtri (three to five launches of its own): ~130 us of launches and gaps per Cholesky-QR, two per iteration.
__global__ void __launch_bounds__(256) k_potrf_panels(double *__restrict__ A, int w, int *__restrict__ info, const double *__restrict__ dscale = nullptr,
                                                      double *__restrict__ linv = nullptr) {
    constexpr int NB = 8;
    extern __shared__ __attribute__((aligned(16))) double L[]; // w x w, column-major, pitch w + 1 (bank spread)
    const int tid = threadIdx.x, pitch = w + 1;
    for (int idx = tid; idx < w * w; idx += 256) L[(idx / w) * pitch + idx % w] = A[idx];
    __syncthreads();
    int fail = 0; // uniform
    for (int k0 = 0; k0 < w && !fail; k0 += NB) {
        const int nb = min(NB, w - k0);
        double blk[NB][NB], inv[NB]; // lower triangle of the diagonal block, column q in blk[.][q]
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int p = q; p < NB; ++p) blk[p][q] = (p < nb) ? L[(k0 + q) * pitch + k0 + p] : (p == q ? 1.0 : 0.0); // identity padding: inert
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const double akk = blk[q][q];
            if (!(akk > 0.0) && !fail) fail = k0 + q + 1;
            double d = 1.0;
            inv[q] = 1.0;
            if (akk > 0.0) mh_fast_sqrt_rsqrt(akk, d, inv[q]); // (eight of these in a row, every one waiting for the previous column's update: the panel's time)
            blk[q][q] = d;
#pragma unroll
            for (int p = q + 1; p < NB; ++p) blk[p][q] *= inv[q];
#pragma unroll
            for (int j = q + 1; j < NB; ++j)
#pragma unroll
                for (int p = j; p < NB; ++p) blk[p][j] -= blk[p][q] * blk[j][q];
        }
        if (fail) break;
        __syncthreads(); // everyone has read the block before anyone overwrites it
        if (tid < NB * NB) {
            const int q = tid / NB, pr = tid % NB;
            if (pr >= q && pr < nb) {
                double val = 0.0;
#pragma unroll
                for (int qq = 0; qq < NB; ++qq)
#pragma unroll
                    for (int pp = qq; pp < NB; ++pp)
                        if (qq == q && pp == pr) val = blk[pp][qq]; // (register arrays cannot be indexed dynamically)
                L[(k0 + q) * pitch + k0 + pr] = val;
            }
        }
        for (int i = k0 + nb + tid; i < w; i += 256) { // the panel's rows below the block: x L_kk^T = a
            double x[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) x[q] = q < nb ? L[(k0 + q) * pitch + i] : 0.0;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
#pragma unroll
                for (int qq = 0; qq < q; ++qq) x[q] -= x[qq] * blk[q][qq];
                x[q] *= inv[q];
            }
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (q < nb) L[(k0 + q) * pitch + i] = x[q];
        }
        __syncthreads();
        // trailing lower triangle: columns j beyond the panel, rows i >= j
        const int rem = w - k0 - nb;
        for (int idx = tid; idx < rem * rem; idx += 256) {
            const int j = k0 + nb + idx / rem, i = k0 + nb + idx % rem;
            if (i >= j) {
                double a = L[j * pitch + i];
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    if (q < nb) a -= L[(k0 + q) * pitch + i] * L[(k0 + q) * pitch + j];
                L[j * pitch + i] = a;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        info[0] = fail;
        double lo = 1.7976931348623157e308, hi = 0;
        for (int k = 0; k < w; ++k) { const double v = L[k * pitch + k]; lo = fmin(lo, v); hi = fmax(hi, v); }
        info[1] = fail || !(lo > 0) ? 1 << 20 : int(16.0 * log2(hi / lo)); // as k_potrf_small
    }
    if (!linv) {
        for (int idx = tid; idx < w * w; idx += 256) {
            const int j = idx / w, i = idx % w;
            if (i >= j) A[idx] = L[j * pitch + i]; // the strictly upper part keeps the input, as potrf
        }
        return;
    }
    __syncthreads(); // (thread 0 has read the diagonal)
    // L <- diag(1 / dscale) L, out to A with zeros above the diagonal (k_unscale_chol)
    for (int idx = tid; idx < w * w; idx += 256) {
        const int j = idx / w, i = idx % w;
        double v = 0.0;
        if (i >= j) {
            const double d = dscale[i];
            v = d > 0 ? L[j * pitch + i] / d : (i == j ? 1.0 : 0.0);
            L[j * pitch + i] = v;
        }
        A[idx] = v;
    }
    __syncthreads();
    if (fail) return; // uniform
    double *scr = L + size_t(w) * pitch; // 8 values per panel row: the second half of a row's sum
    for (int kb = (w + NB - 1) / NB - 1; kb >= 0; --kb) {
        const int k0 = kb * NB, nb = min(NB, w - k0), r0 = k0 + nb, rem = w - r0;
        // inverse of the diagonal block, by every thread (lower triangle, identity padding)
        double dm[NB][NB], di[NB][NB];
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int pr = q; pr < NB; ++pr) dm[pr][q] = (pr < nb) ? L[(k0 + q) * pitch + k0 + pr] : (pr == q ? 1.0 : 0.0);
#pragma unroll
        for (int q = 0; q < NB; ++q) di[q][q] = mh_fast_rcp(dm[q][q]);
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int pr = q + 1; pr < NB; ++pr) {
                double t = 0.0;
#pragma unroll
                for (int u = q; u < pr; ++u) t += dm[pr][u] * di[u][q];
                di[pr][q] = -di[pr][pr] * t;
            }
        // Y = P D^-1, a row per thread, in place
        if (tid < rem) {
            const int i = r0 + tid;
            double pv[NB], y[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) pv[q] = q < nb ? L[(k0 + q) * pitch + i] : 0.0;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                y[q] = 0.0;
#pragma unroll
                for (int pr = q; pr < NB; ++pr) y[q] += pv[pr] * di[pr][q];
            }
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (q < nb) L[(k0 + q) * pitch + i] = y[q];
        }
        __syncthreads();
        // X = -T Y: two threads per row, alternate columns of T
        const int row = tid & 127, half = tid >> 7;
        double acc[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) acc[q] = 0.0;
        if (row < rem) {
            const int i = r0 + row;
            for (int k = r0 + half; k <= i; k += 2) {
                const double t = L[k * pitch + i];
#pragma unroll
                for (int q = 0; q < NB; ++q) acc[q] += t * L[(k0 + q) * pitch + k]; // (columns beyond nb of a last, short block: the next block's, times a result never stored)
            }
            if (half) {
#pragma unroll
                for (int q = 0; q < NB; ++q) scr[row * NB + q] = acc[q];
            }
        }
        __syncthreads();
        if (row < rem && !half) {
            const int i = r0 + row;
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (q < nb) L[(k0 + q) * pitch + i] = -(acc[q] + scr[row * NB + q]);
        }
        if (tid < NB * NB) {
            const int q = tid / NB, pr = tid % NB;
            if (pr >= q && pr < nb) {
                double val = 0.0;
#pragma unroll
                for (int qq = 0; qq < NB; ++qq)
#pragma unroll
                    for (int pp = qq; pp < NB; ++pp)
                        if (qq == q && pp == pr) val = di[pp][qq];
                L[(k0 + q) * pitch + k0 + pr] = val;
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < w * w; idx += 256) {
        const int j = idx / w, i = idx % w;
        linv[idx] = i >= j ? L[j * pitch + i] : 0.0;
    }
}
} // namespace

// Inverse of a symmetric positive definite block of order w <= 128 (column-major, leading dimension lda) into out (w x w,
// leading dimension ldo): in-place Gauss-Jordan without pivoting in registers, one workgroup.  Every pivot of an SPD matrix is
// positive; a pivot that is not sets *info = its index + 1 (info is only ever raised: clear it before a sequence of calls).
namespace {
// Explicit inverse of an SPD block of order <= 128 by in-place Gauss-Jordan elimination, one workgroup of 1 024 threads, the
// (padded) 128 x 128 iterate in registers for the whole elimination: thread (tr, tc) = (tid & 31, tid >> 5) owns the 4 x 4 entries
// (tr + 32 a, tc + 32 b).  Step p needs row p and column p of the current matrix.  Between a processed and an unprocessed index
// the iterate is antisymmetric (rows were scaled by +1/pivot, columns by -1/pivot) and symmetric otherwise, so column p is row p
// with the sign of the processed entries flipped: only the 32 owners of row p publish it (LDS, two alternating buffers), one
// barrier per step.  A 4 x 4 tile reads 4 + 4 + 1 published values per step for its 16 updates; the 1 x 16 strips this kernel
// used first read 18, and the LDS pipe -- 16 waves x 18 reads x 4 cycles against 512 cycles of fp64 arithmetic per step --
// was what bound it: 179 -> ~75 us at order 128 with the same operations per entry, bit for bit.
__global__ void __launch_bounds__(1024) k_spd_inverse_small(const double *__restrict__ A, int lda, int w, double *__restrict__ out, int ldo, int *__restrict__ info) {
    __shared__ double rowbuf[2][128 + 1]; // row p, and 1 / pivot (one division, by the pivot's owner, instead of 1 024 of them per step)
    const int tid = threadIdx.x, tr = tid & 31, tc = tid >> 5;
    double val[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = tr + 32 * a, j = tc + 32 * b;
            val[a][b] = (i < w && j < w) ? A[size_t(j) * lda + i] : (i == j ? 1.0 : 0.0); // identity padding: inert
        }
    int bad = 0; // index + 1 of the first pivot that was not positive (the elimination runs on: its output is then garbage, and flagged)
    for (int p = 0; p < w; ++p) {
        double *rb = rowbuf[p & 1];
        const int ap = p >> 5, lp = p & 31; // row / column p inside its owners' tiles (uniform)
        if (tr == lp) {
            __builtin_amdgcn_s_setprio(3); // (the serial stretch of a step: the row's owners ahead of the waves they share SIMDs with)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a == ap) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        rb[tc + 32 * b] = val[a][b];
                        if (tc == lp && b == ap) { // 1 / pivot: the hardware's reciprocal and two Newton steps (a full division is ~3x the dependent instructions, on every step's critical path)
                            const double d = val[a][b];
                            double r = __builtin_amdgcn_rcp(d);
                            r = fma(fma(-d, r, 1.0), r, r);
                            r = fma(fma(-d, r, 1.0), r, r);
                            rb[128] = r;
                        }
                    }
                }
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        mh_lds_writes_landed();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const double piv = rb[p], inv = rb[128];
        if (!(piv > 0.0) && !bad) bad = p + 1; // uniform
        double f[4], fi[4], r[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = tr + 32 * a;
            f[a] = i < p ? -rb[i] : rb[i]; // element (i, p)
            fi[a] = f[a] * inv;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) r[b] = rb[tc + 32 * b];
        // generic entries first, without selects (the fp64 ALU work of the step is what should bound it) ...
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) val[a][b] = fma(-fi[a], r[b], val[a][b]);
        // ... then row p (its owners only) and column p (one register per tile row of its owners), picked by uniform compares
        if (tr == lp) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a == ap) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) val[a][b] = r[b] * inv;
                }
        }
        if (tc == lp) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b == ap) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) val[a][b] = (tr + 32 * a == p) ? inv : -fi[a]; // element (i, p) of the result
                }
        }
    }
    if (bad && tid == 0) atomicMax(info, bad);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = tr + 32 * a, j = tc + 32 * b;
            if (i < w && j < w) out[size_t(j) * ldo + i] = val[a][b];
        }
}
} // namespace

void mh_spd_inverse_small(mh_context *ctx, const double *a, uint32_t lda, uint32_t w, double *out, uint32_t ldo, int *info) {
    if (w < 1 || w > 128) mh_throw(MH_EINVAL, "spd_inverse_small: order %u outside 1..128", w);
    // (Round 5, docs/LAB_NOTEBOOK.md section 12: inside the coarse set-up the elimination loop runs 115-136 us instead of 87 because waves of
    // the main stream's device-filling products share its SIMDs -- they need no LDS, so reserving LDS kept nobody out; claiming the whole
    // register file (128 instead of 96 registers) does, but the workgroup then waits as long for an EMPTY CU as it gained: 160 us per call in
    // the trace either way, and the set-up's step is bound by the rank-128 update beside it, not by this kernel.)
    k_spd_inverse_small<<<1, 1024, 0, ctx->stream>>>(a, int(lda), int(w), out, int(ldo), info);
    KERNEL_CHECK();
}

// Factor, unscale and invert in one launch (k_potrf_panels with linv): a <- diag(1 / dscale) chol(a) (zeros above the diagonal), linv <- its inverse.
void mh_potrf_small_inverse(mh_context *ctx, double *a, uint32_t w, int *info, const double *dscale, double *linv) {
    if (w < 1 || w > 128) mh_throw(MH_EINVAL, "potrf_small_inverse: order %u outside 1..128", w);
    static PerDeviceOnce attr;
    attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_potrf_panels), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); });
    k_potrf_panels<<<1, 256, (size_t(w) * (w + 1) + size_t(8) * w) * sizeof(double), ctx->stream>>>(a, int(w), info, dscale, linv);
    KERNEL_CHECK();
}

void mh_potrf_small(mh_context *ctx, double *a, uint32_t w, int *info) {
    if (w < 1 || w > 128) mh_throw(MH_EINVAL, "potrf_small: order %u outside 1..128", w);
    static PerDeviceOnce attr;
    attr.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_potrf_small), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); });
    constexpr bool panels = true;
    if (panels) {
        static PerDeviceOnce attr2;
        attr2.run(ctx->device, [] { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_potrf_panels), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); });
        k_potrf_panels<<<1, 256, size_t(w) * (w + 1) * sizeof(double), ctx->stream>>>(a, int(w), info);
    } else k_potrf_small<<<1, 256, size_t(w) * (w + 1) * sizeof(double), ctx->stream>>>(a, int(w), info);
    KERNEL_CHECK();
}

// Lower Cholesky factor of any order in 128-column blocks, every piece a kernel of ours or a rocBLAS level-3 routine: the diagonal block
// by k_potrf_panels, its triangular inverse by rocblas_dtrtri, the panel below and the trailing update by k_small_gemm.  It exists because
// rocsolver_dpotrf is disturbed by work running beside it on the device (mh_eigs.hip, "Concurrency contract": wrong factors in ~10 % of
// the calls while another stream runs the Gram kernel) -- three host threads solving 140 or 215 pairs each on their own contexts lost rank
// or converged to perturbed eigenvalues in a third of the solves (tools/concurrent_solves.py 3 18 14 215) while blocks wider than 128
// columns still went through it.  info[0] = 0 or the 1-based column of the first non-positive pivot; info[1] = the diagonal spread report
// of k_potrf_panels for orders <= 128, 1 << 20 (unknown) above.  Same time as the library's at order 240 (~140 us).
namespace {
__global__ void k_potrf_merge_info(const int *__restrict__ block_info, int k0, int *__restrict__ info) {
    if (info[0] == 0 && block_info[0] != 0) info[0] = k0 + block_info[0];
    info[1] = 1 << 20;
}
} // namespace
void mh_potrf(mh_context *ctx, double *a, uint32_t ld, uint32_t w, int *info) {
    if (!w) return;
    if (w <= 128 && ld == w) return mh_potrf_small(ctx, a, w, info);
    constexpr uint32_t nb = 128;
    DevArray<double> blk(ctx, size_t(nb) * nb), linv(ctx, size_t(nb) * nb), pan(ctx, size_t(w) * nb);
    DevArray<int> binfo(ctx, 2);
    HIP_CHECK(hipMemsetAsync(info, 0, 2 * sizeof(int), ctx->stream));
    for (uint32_t k0 = 0; k0 < w; k0 += nb) {
        const uint32_t wk = std::min(nb, w - k0), r = w - k0 - wk;
        double *akk = a + size_t(k0) * ld + k0;
        HIP_CHECK(hipMemcpy2DAsync(blk.get(), size_t(wk) * sizeof(double), akk, size_t(ld) * sizeof(double), size_t(wk) * sizeof(double), wk, hipMemcpyDeviceToDevice, ctx->stream));
        mh_potrf_small(ctx, blk, wk, binfo);
        k_potrf_merge_info<<<1, 1, 0, ctx->stream>>>(binfo, int(k0), info);
        KERNEL_CHECK();
        HIP_CHECK(hipMemcpy2DAsync(akk, size_t(ld) * sizeof(double), blk.get(), size_t(wk) * sizeof(double), size_t(wk) * sizeof(double), wk, hipMemcpyDeviceToDevice, ctx->stream));
        if (!r) break;
        HIP_CHECK(hipMemsetAsync(linv, 0, size_t(wk) * wk * sizeof(double), ctx->stream));
        ROCBLAS_CHECK(rocblas_dtrtri(ctx->blas, rocblas_fill_lower, rocblas_diagonal_non_unit, rocblas_int(wk), blk, rocblas_int(wk), linv, rocblas_int(wk)));
        double *a21 = akk + wk, *a22 = a + size_t(k0 + wk) * ld + k0 + wk;
        mh_small_gemm(ctx, false, true, r, wk, wk, 1.0, a21, ld, linv, wk, 0.0, pan, r); // L21 = A21 L11^-T
        HIP_CHECK(hipMemcpy2DAsync(a21, size_t(ld) * sizeof(double), pan.get(), size_t(r) * sizeof(double), size_t(r) * sizeof(double), wk, hipMemcpyDeviceToDevice, ctx->stream));
        mh_small_gemm(ctx, false, true, r, r, wk, -1.0, pan, r, pan, r, 1.0, a22, ld); // A22 -= L21 L21^T
    }
    // (no synchronisation: the workspaces return to the context's pool, whose blocks are only ever used on this same stream)
}

// ---- small dense products --------------------------------------------------------------------------------------
// C (M x N) = alpha op(A) op(B) + beta C, column-major, for the ORDER-m matrices of the Rayleigh-Ritz step (m = 3 x block
// width: 240 .. 720).  The library's dgemm picks a 128 x 128 macro tile at these sizes: 4 workgroups at order 240 (131 us a
// call), 36 at order 720 (487 us) on a 256-CU device.  Here a workgroup of four waves owns a 32 x 32 tile of C (64 / 529
// workgroups), a wave one 16 x 16 tile accumulated with v_mfma_f64_16x16x4_f64 straight from global memory (the operands are
// L2 resident: at most 3 x 4.4 MB), eight k-steps of loads in flight per wave.  Strides carry the transpositions.
namespace {
__global__ void __launch_bounds__(256) k_small_gemm(int M, int N, int K, double alpha, const double *__restrict__ A, long sai, long sak, const double *__restrict__ B, long sbk, long sbj,
                                                   double beta, double *__restrict__ C, int ldc) {
    typedef double double4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c_lane = lane & 15, k_lane = lane >> 4;
    const int i0 = blockIdx.x * 32 + (wave & 1) * 16, j0 = blockIdx.y * 32 + (wave >> 1) * 16;
    if (i0 >= M || j0 >= N) return;
    const int ia = i0 + c_lane, jb = j0 + c_lane;
    const bool a_ok = ia < M, b_ok = jb < N;
    const double *ap = A + (a_ok ? ia : 0) * sai + k_lane * sak, *bp = B + (b_ok ? jb : 0) * sbj + k_lane * sbk;
    double4_t acc = {0, 0, 0, 0};
    int k = 0;
    for (; k + 32 <= K; k += 32) {
        double af[8], bf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) af[u] = ap[(k + 4 * u) * sak], bf[u] = bp[(k + 4 * u) * sbk];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_ok ? af[u] : 0.0, b_ok ? bf[u] : 0.0, acc, 0, 0, 0);
    }
    for (; k < K; k += 4) {
        const bool k_ok = k + k_lane < K;
        const double av = a_ok && k_ok ? ap[k * sak] : 0.0, bv = b_ok && k_ok ? bp[k * sbk] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    if (!b_ok) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = i0 + k_lane + 4 * reg;
        if (row >= M) continue;
        double *c = C + size_t(jb) * ldc + row;
        *c = beta == 0.0 ? alpha * acc[reg] : alpha * acc[reg] + beta * *c;
    }
}
} // namespace

void mh_small_gemm(mh_context *ctx, bool ta, bool tb, uint32_t M, uint32_t N, uint32_t K, double alpha, const double *a, uint32_t lda, const double *b, uint32_t ldb, double beta, double *c,
                   uint32_t ldc) {
    if (!M || !N) return;
    const dim3 grid(div_up(M, 32), div_up(N, 32));
    k_small_gemm<<<grid, 256, 0, ctx->stream>>>(int(M), int(N), int(K), alpha, a, ta ? long(lda) : 1L, ta ? 1L : long(lda), b, tb ? long(ldb) : 1L, tb ? 1L : long(ldb), beta, c, int(ldc));
    KERNEL_CHECK();
}
