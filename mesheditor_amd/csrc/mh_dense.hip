// Tall-skinny dense block kernels of the eigensolver (gfx950), fp64 MFMA.
//
// gram:  G (wa x wb) = X^T Y for row-major panels X (n x wa), Y (n x wb), n ~ 10^5..10^6, w <= 256.
//   rocBLAS maps this shape (tiny m, n; huge k) onto a handful of workgroups; here the rows are split over ~2 per CU,
//   each workgroup stages KC rows of both panels in LDS once (every panel element leaves HBM exactly once) and its
//   waves accumulate 16x16 output tiles with v_mfma_f64_16x16x4_f64, operands read from LDS with ds_read_b64 at a row
//   pitch = 16 (mod 32) doubles (conflict free).  Partial Grams are summed in a fixed order by a second kernel, so the
//   result is bit-reproducible.  Roofline: max(8 n (wa + wb) bytes / HBM, 2 n wa wb flops / fp64 MFMA peak).
#include "mh_common.h"

namespace {
constexpr int KC = 16; // rows staged per step

typedef double double4_t __attribute__((ext_vector_type(4)));

__host__ __device__ inline int pad_pitch(int w) { // smallest p >= w with p = 16 (mod 32)
    int p = ((w + 15) / 16) * 16;
    if ((p % 32) != 16) p += 16;
    return p;
}

template<int WAVES, int MAXT>
__global__ void __launch_bounds__(WAVES * 64) k_gram(const double *__restrict__ X, int wa, const double *__restrict__ Y, int wb, size_t n, size_t rows_per_wg,
                                                    double *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int pa = pad_pitch(wa), pb = pad_pitch(wb);
    double *Xs = smem, *Ys = smem + KC * pa;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ti_n = (wa + 15) / 16, tj_n = (wb + 15) / 16, ntiles = ti_n * tj_n;
    double4_t acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = double4_t{0, 0, 0, 0};
    const size_t r_begin = size_t(blockIdx.x) * rows_per_wg;
    const size_t r_end = min(n, r_begin + rows_per_wg);
    const int kk_lane = lane >> 4, c_lane = lane & 15;
    for (size_t r0 = r_begin; r0 < r_end; r0 += KC) {
        // stage KC rows (zero beyond the panel / the row range)
        for (int i = tid; i < KC * pa; i += WAVES * 64) {
            const int k = i / pa, c = i % pa;
            const size_t r = r0 + k;
            Xs[i] = (c < wa && r < r_end) ? X[r * wa + c] : 0.0;
        }
        for (int i = tid; i < KC * pb; i += WAVES * 64) {
            const int k = i / pb, c = i % pb;
            const size_t r = r0 + k;
            Ys[i] = (c < wb && r < r_end) ? Y[r * wb + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const double *xr = Xs + (kk + kk_lane) * pa + c_lane;
            const double *yr = Ys + (kk + kk_lane) * pb + c_lane;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int tile = wave + t * WAVES;
                if (tile < ntiles) {
                    const int ti = tile / tj_n, tj = tile % tj_n;
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[ti * 16], yr[tj * 16], acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    double *out = partial + size_t(blockIdx.x) * wa * wb;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int tile = wave + t * WAVES;
        if (tile >= ntiles) continue;
        const int ti = tile / tj_n, tj = tile % tj_n;
        const int j = tj * 16 + c_lane;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = ti * 16 + kk_lane + 4 * reg;
            if (i < wa && j < wb) out[size_t(j) * wa + i] = acc[t][reg];
        }
    }
}

// 8 threads per output entry: each sums a strided eighth of the partial Grams, then a fixed-order combine
__global__ void k_gram_reduce(const double *__restrict__ partial, int nwg, int wa, int wb, double *__restrict__ g, int ld) {
    __shared__ double s[256];
    const int idx = blockIdx.x * 32 + (threadIdx.x >> 3), part = threadIdx.x & 7;
    double acc = 0;
    if (idx < wa * wb)
        for (int w = part; w < nwg; w += 8) acc += partial[size_t(w) * wa * wb + idx];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (part == 0 && idx < wa * wb) {
        double t = 0;
        for (int k = 0; k < 8; ++k) t += s[threadIdx.x + k];
        g[size_t(idx / wa) * ld + idx % wa] = t;
    }
}
} // namespace

// G (wa x wb, column-major, leading dimension ld) = X^T Y
void mh_gram(mh_context *ctx, size_t n, const double *x, uint32_t wa, const double *y, uint32_t wb, double *g, uint32_t ld) {
    if (!wa || !wb) return;
    const int ntiles = int((wa + 15) / 16) * int((wb + 15) / 16);
    if (ntiles > 256) mh_throw(MH_EINVAL, "gram: block width %u x %u exceeds 256 x 256", wa, wb);
    int nwg = int(std::min<size_t>(256, (n + KC - 1) / KC));
    size_t rows_per_wg = ((n + nwg - 1) / nwg + KC - 1) / KC * KC;
    nwg = int((n + rows_per_wg - 1) / rows_per_wg);
    const size_t need = size_t(nwg) * wa * wb * sizeof(double);
    if (ctx->gram_ws_bytes < need) {
        ctx->pool.release(ctx->gram_ws);
        ctx->gram_ws = ctx->pool.alloc(need + need / 4);
        ctx->gram_ws_bytes = need + need / 4;
    }
    double *workspace = static_cast<double *>(ctx->gram_ws);
    const size_t lds = size_t(KC) * (pad_pitch(int(wa)) + pad_pitch(int(wb))) * sizeof(double);
    if (ntiles <= 32) {
        k_gram<4, 8><<<nwg, 256, lds, ctx->stream>>>(x, int(wa), y, int(wb), n, rows_per_wg, workspace);
    } else if (ntiles <= 64) {
        k_gram<8, 8><<<nwg, 512, lds, ctx->stream>>>(x, int(wa), y, int(wb), n, rows_per_wg, workspace);
    } else {
        k_gram<16, 16><<<nwg, 1024, lds, ctx->stream>>>(x, int(wa), y, int(wb), n, rows_per_wg, workspace);
    }
    KERNEL_CHECK();
    k_gram_reduce<<<div_up(size_t(wa) * wb, 32), 256, 0, ctx->stream>>>(workspace, nwg, int(wa), int(wb), g, int(ld));
    KERNEL_CHECK();
}
