// Micro-benchmark of rocSOLVER's small symmetric eigensolver variants at the Rayleigh-Ritz sizes of the block solver.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
int main() {
    rocblas_handle h;
    rocblas_create_handle(&h);
    for (int n : {96, 150, 222, 240}) {
        std::vector<double> A(size_t(n) * n);
        std::mt19937_64 rng(1);
        std::normal_distribution<double> g;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i <= j; ++i) A[size_t(j) * n + i] = A[size_t(i) * n + j] = (i == j ? 10.0 * (i + 1) : 0.0) + g(rng);
        double *dA, *dA0, *dD, *dE, *dT, *dZ;
        int *info, *nev;
        hipMalloc(&dA, n * n * 8); hipMalloc(&dA0, n * n * 8); hipMalloc(&dD, n * 8); hipMalloc(&dE, n * 8); hipMalloc(&dT, n * 8); hipMalloc(&dZ, n * n * 8);
        hipMalloc(&info, 16); hipMalloc(&nev, 16);
        hipMemcpy(dA0, A.data(), n * n * 8, hipMemcpyHostToDevice);
        auto timeit = [&](const char *name, auto fn) {
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemcpy(dA, dA0, n * n * 8, hipMemcpyDeviceToDevice);
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                fn();
                hipDeviceSynchronize();
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("n=%4d %-34s %8.3f ms\n", n, name, best * 1e3);
        };
        timeit("syevd", [&] { rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dD, dE, info); });
        timeit("syevdj", [&] { rocsolver_dsyevdj(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dD, info); });
        timeit("syevdx lowest third", [&] { rocsolver_dsyevdx(h, rocblas_evect_original, rocblas_erange_index, rocblas_fill_lower, n, dA, n, 0, 0, 1, n / 3, nev, dD, dZ, n, info); });
        timeit("sytrd", [&] { rocsolver_dsytrd(h, rocblas_fill_lower, n, dA, n, dD, dE, dT); });
        timeit("sytrd+orgtr", [&] { rocsolver_dsytrd(h, rocblas_fill_lower, n, dA, n, dD, dE, dT); rocsolver_dorgtr(h, rocblas_fill_lower, n, dA, n, dT); });
        timeit("sytrd+orgtr+stedc", [&] { rocsolver_dsytrd(h, rocblas_fill_lower, n, dA, n, dD, dE, dT); rocsolver_dorgtr(h, rocblas_fill_lower, n, dA, n, dT); rocsolver_dstedc(h, rocblas_evect_original, n, dD, dE, dA, n, info); });
        timeit("syevd values only", [&] { rocsolver_dsyevd(h, rocblas_evect_none, rocblas_fill_lower, n, dA, n, dD, dE, info); });
    }
    return 0;
}
