// fp64 MFMA issue-rate probe (gfx950): N back-to-back v_mfma_f64_16x16x4_f64 on T independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template<int T> __global__ void k(double *out, int iters, double a0, double b0) {
    double4_t acc[T];
    for (int t = 0; t < T; ++t) acc[t] = double4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
    for (int t = 0; t < T; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int T> void run(int waves_per_cu, int iters) {
    double *out;
    const int threads = 256, wgs = 256 * waves_per_cu / 4;
    hipMalloc(&out, sizeof(double) * threads * wgs);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<T><<<wgs, threads>>>(out, 10, 1.0, 2.0);
    hipEventRecord(e0);
    k<T><<<wgs, threads>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = double(wgs) * 4 * iters * T, flops = n_mfma * 2048;
    // cycles per MFMA per SIMD at 2.4 GHz: SIMDs = 1024
    printf("T=%2d waves/CU=%2d  %.3f ms  %.1f TF/s  %.1f cycles per MFMA per SIMD (2.4 GHz)\n", T, waves_per_cu, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_mfma / 1024));
    hipFree(out);
}
int main() {
    run<1>(4, 20000); run<4>(4, 5000); run<8>(4, 2500); run<8>(8, 2500); run<8>(16, 2500); run<25>(4, 1000);
    return 0;
}
