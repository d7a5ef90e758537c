/* Lab bench of libmodalhip: measurement and experiment entry points OUTSIDE the path's ABI (libmodalhip_lab.so).
 * Nothing here replaces a reference interface; include/modalhip.h is the drop-in boundary. */
#pragma once
#include "../../../include/modalhip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Average device time of `reps` back-to-back K x products over a resident n x width panel, and the algorithmic bytes of one
 * launch (76 B per node block + 4 B per row pointer + 16 B per panel entry). */
int mhl_system_bench_spmm(mh_system *, uint32_t width, uint32_t reps, double *avg_ms, double *algorithmic_bytes);
/* The matrix-free element-by-element operator: one product in the reference's DOF order, and its timing loop. */
int mhl_system_elementwise_matvec(mh_system *, const double *x, double *y, uint32_t width);
int mhl_system_bench_elementwise(mh_system *, uint32_t width, uint32_t reps, double *avg_ms);
/* The fp32 smoother's fused product-and-Chebyshev step on level 2 / 1 over an n x width panel cut into `slabs` contiguous panels. */
int mhl_system_bench_cheb_step(mh_system *, int level, uint32_t width, uint32_t slabs, uint32_t reps, double *avg_ms);
/* kind 0 = Gram G = X^T Y (X n x wa, Y n x wb), kind 1 = basis update Z = [X | W] C: average device time of `reps` launches. */
int mhl_context_bench_dense(mh_context *, int kind, uint64_t n, uint32_t wa, uint32_t wb, uint32_t reps, double *avg_ms);
/* The Rayleigh-Ritz step's Householder tridiagonalisation called directly: a (m x m, symmetric, both triangles, m <= 256) ->
 * d[m], e[m - 1]; variant 0 = one workgroup, 1 = several workgroups exchanging tagged values. */
int mhl_context_tridiagonalize(mh_context *, int variant, uint32_t m, const double *a, double *d, double *e, uint32_t reps, double *avg_ms);
/* the same with the reflectors (m x m, LAPACK's lower storage) and tau returned; variant 2 = the wide kernel, orders up to 768 */
int mhl_context_tridiagonalize_full(mh_context *, int variant, uint32_t m, const double *a, double *d, double *e, double *reflectors, double *tau, uint32_t reps, double *avg_ms);
/* C (M x N, ldc) = alpha op(A) op(B) + beta C through the Rayleigh-Ritz step's small-product kernel; column-major host arrays, c in and out */
int mhl_context_potrf_inverse(mh_context *, uint32_t w, const double *a, const double *dscale, double *l, double *linv, int *info2);
int mhl_context_spd_inverse(mh_context *, uint32_t w, const double *a, double *out, uint32_t reps, double *avg_ms);
int mhl_context_small_gemm(mh_context *, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, double alpha, const double *a, uint32_t lda, const double *b, uint32_t ldb, double beta, double *c,
                           uint32_t ldc, uint32_t reps, double *avg_ms);
/* G (wa x wb, column-major) = X^T Y for row-major host panels (n x wa), (n x wb): the solver's Gram kernel */
int mhl_context_gram(mh_context *, uint64_t n, const double *x, uint32_t wa, const double *y, uint32_t wb, double *g);
/* measured HBM ceilings of the device: a streaming copy (read + written bytes per second) and a streaming read, in GB/s */
int mhl_context_bench_stream(mh_context *, uint64_t bytes, uint32_t reps, double *copy_gbs, double *read_gbs);
/* the context's device pool: bytes held from the device, bytes of them idle in the cache, the cap on the idle part */
int mhl_context_pool_stats(mh_context *, uint64_t *reserved, uint64_t *idle, uint64_t *cap);
/* The rigid-body level's graph aggregation (host code): CSR node graph in (diagonal entries included), aggregate per node out. */
uint32_t mhl_graph_aggregates(const uint32_t *row_ptr, const uint32_t *col, uint32_t n, uint32_t target, uint32_t max_order, uint32_t *agg_of);
/* Soak of the multi-workgroup tridiagonalisation's tagged exchange with its transport as a parameter (lab/mh_soak.hip): `launches` runs of
 * one fixed matrix, every collected value compared with a recorded undisturbed run; out receives the deviation records. */
int mhl_sytrd_soak(mh_context *, uint32_t transport, uint32_t m, uint32_t launches, volatile int *recorded, volatile int *go, void *out, uint64_t out_bytes);
uint64_t mhl_sytrd_soak_bytes(void);
/* one kind of a wide solve's work, alone, over and over on the context's stream until *stop (lab/mh_soak.hip lists the kinds) */
int mhl_soak_aggressor(mh_context *, int kind, volatile int *stop, uint64_t *count);
#ifdef __cplusplus
}
#endif
