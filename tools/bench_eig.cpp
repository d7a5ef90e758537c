// Micro-benchmark: rocSOLVER small symmetric eigensolvers (sizes of the LOBPCG Rayleigh-Ritz problem).
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("fail %s -> %d\n", #x, int(e_)); return 1; } } while (0)
int main() {
    rocblas_handle h; CK(rocblas_create_handle(&h));
    for (int n : {150, 225, 450, 690}) {
        std::vector<double> A(n * n), B(n * n);
        std::mt19937_64 rng(1); std::normal_distribution<double> g;
        for (int j = 0; j < n; ++j) for (int i = 0; i <= j; ++i) { double v = g(rng); A[j * n + i] = A[i * n + j] = v; double w = 0.01 * g(rng); B[j * n + i] = B[i * n + j] = w + (i == j ? 1.0 : 0.0); }
        double *dA, *dB, *dA0, *dB0, *dD, *dE; int *info;
        CK(hipMalloc(&dA, n * n * 8)); CK(hipMalloc(&dB, n * n * 8)); CK(hipMalloc(&dA0, n * n * 8)); CK(hipMalloc(&dB0, n * n * 8)); CK(hipMalloc(&dD, n * 8)); CK(hipMalloc(&dE, n * 8)); CK(hipMalloc(&info, 16));
        CK(hipMemcpy(dA0, A.data(), n * n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB0, B.data(), n * n * 8, hipMemcpyHostToDevice));
        auto timeit = [&](const char *name, auto fn) {
            double best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemcpy(dA, dA0, n * n * 8, hipMemcpyDeviceToDevice); hipMemcpy(dB, dB0, n * n * 8, hipMemcpyDeviceToDevice); hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now(); fn(); hipDeviceSynchronize();
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("n=%4d %-28s %8.3f ms\n", n, name, best * 1e3);
        };
        timeit("sygvd", [&] { rocsolver_dsygvd(h, rocblas_eform_ax, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dB, n, dD, dE, info); });
        timeit("syevd", [&] { rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dD, dE, info); });
        int *nsweeps; double *resid; hipMalloc(&nsweeps, 16); hipMalloc(&resid, 16);
        timeit("syevj", [&] { rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_lower, n, dA, n, 1e-14, resid, 20, nsweeps, dD, info); });
        timeit("potrf+2trsm+syevd+trsm", [&] {
            const double one = 1;
            rocsolver_dpotrf(h, rocblas_fill_lower, n, dB, n, info);
            rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n, n, &one, dB, n, dA, n);
            rocblas_dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, n, &one, dB, n, dA, n);
            rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, dA, n, dD, dE, info);
            rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, n, n, &one, dB, n, dA, n);
        });
        timeit("potrf", [&] { rocsolver_dpotrf(h, rocblas_fill_lower, n, dB, n, info); });
    }
    // large dense: potrf / potri / trsm at coarse-level sizes
    for (int n : {7400, 14700}) {
        double *dA; int *info; CK(hipMalloc(&dA, size_t(n) * n * 8)); CK(hipMalloc(&info, 16));
        std::vector<double> A(size_t(n) * n, 0.0);
        for (int i = 0; i < n; ++i) { A[size_t(i) * n + i] = 4; if (i + 1 < n) A[size_t(i) * n + i + 1] = A[size_t(i + 1) * n + i] = -1; }
        for (const char *what : {"potrf", "potri"}) {
            hipDeviceSynchronize(); auto t0 = std::chrono::steady_clock::now();
            if (what[3] == 'r' && what[4] == 'f') { hipMemcpy(dA, A.data(), size_t(n) * n * 8, hipMemcpyHostToDevice); hipDeviceSynchronize(); t0 = std::chrono::steady_clock::now(); rocsolver_dpotrf(h, rocblas_fill_lower, n, dA, n, info); }
            else rocsolver_dpotri(h, rocblas_fill_lower, n, dA, n, info);
            hipDeviceSynchronize();
            printf("n=%5d %-8s %8.2f ms\n", n, what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
        }
        double *dX; int w = 75; CK(hipMalloc(&dX, size_t(n) * w * 8)); hipMemset(dX, 0, size_t(n) * w * 8);
        const double one = 1, zero = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize(); auto t0 = std::chrono::steady_clock::now();
            rocblas_dtrsm(h, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit, w, n, &one, dA, n, dX, w);
            hipDeviceSynchronize();
            printf("n=%5d trsm w=75 %8.2f ms\n", n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
        }
        double *dY; CK(hipMalloc(&dY, size_t(n) * w * 8));
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize(); auto t0 = std::chrono::steady_clock::now();
            rocblas_dsymm(h, rocblas_side_right, rocblas_fill_lower, w, n, &one, dA, n, dX, w, &zero, dY, w);
            hipDeviceSynchronize();
            printf("n=%5d symm w=75 %8.2f ms\n", n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
            t0 = std::chrono::steady_clock::now();
            rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, w, n, n, &one, dX, w, dA, n, &zero, dY, w);
            hipDeviceSynchronize();
            printf("n=%5d gemm w=75 %8.2f ms\n", n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
        }
        hipFree(dA); hipFree(dX); hipFree(dY);
    }
    return 0;
}
