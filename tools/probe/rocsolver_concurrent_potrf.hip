// Reproducer attempt for the failure that made libmodalhip stop calling rocSOLVER factorisations beside other streams
// ("Concurrency"): several host threads, each with its own stream and rocBLAS handle, factorise
// private copies of ONE symmetric positive definite matrix with rocsolver_dpotrf while another stream keeps the
// device busy with a register- and LDS-heavy kernel.  Every call must return info == 0 and the same factor.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/rocsolver_concurrent_potrf.hip -o /tmp/potrf_probe -lrocblas -lrocsolver && /tmp/potrf_probe [n] [threads] [reps]
// Prints the number of calls with info != 0 and of factors that differ from the serial one.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void __launch_bounds__(256) k_busy(double *out, int rounds) { // 160 KB of LDS per workgroup, long register chains
    extern __shared__ double tile[];
    double acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = threadIdx.x + i;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < 20000; i += 256) tile[i] = acc[i % 24] * 1.0000001 + r;
        __syncthreads();
        for (int i = 0; i < 24; ++i) acc[i] = acc[i] * 0.999 + tile[(threadIdx.x * 37 + i * 101 + r) % 20000];
        __syncthreads();
    }
    double s = 0;
    for (int i = 0; i < 24; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 3690, threads = argc > 2 ? atoi(argv[2]) : 3, reps = argc > 3 ? atoi(argv[3]) : 20;
    std::vector<double> a(size_t(n) * n);
    srand(7);
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) a[size_t(j) * n + i] = a[size_t(i) * n + j] = (i == j ? n : 0.0) + (rand() % 2001 - 1000) * 1e-3;
    auto factor = [&](hipStream_t st, rocblas_handle h, double *d, int *dinfo, std::vector<double> &l) {
        hipMemcpyAsync(d, a.data(), a.size() * 8, hipMemcpyHostToDevice, st);
        rocsolver_dpotrf(h, rocblas_fill_lower, n, d, n, dinfo);
        int info = -1;
        hipMemcpyAsync(&info, dinfo, 4, hipMemcpyDeviceToHost, st);
        l.resize(a.size());
        hipMemcpyAsync(l.data(), d, a.size() * 8, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        return info;
    };
    std::vector<double> serial;
    {
        hipStream_t st;
        hipStreamCreate(&st);
        rocblas_handle h;
        rocblas_create_handle(&h);
        rocblas_set_stream(h, st);
        double *d;
        int *di;
        hipMalloc(&d, a.size() * 8);
        hipMalloc(&di, 4);
        printf("serial info %d\n", factor(st, h, d, di, serial));
    }
    std::atomic<int> bad_info{0}, bad_factor{0}, stop{0};
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_busy), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::thread noise([&] {
        hipStream_t st;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        double *out;
        hipMalloc(&out, 1024 * 256 * 8);
        while (!stop.load()) {
            k_busy<<<1024, 256, 160 * 1024, st>>>(out, 50);
            hipStreamSynchronize(st);
        }
    });
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            hipStream_t st;
            hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            rocblas_handle h;
            rocblas_create_handle(&h);
            rocblas_set_stream(h, st);
            double *d;
            int *di;
            hipMalloc(&d, a.size() * 8);
            hipMalloc(&di, 4);
            std::vector<double> l;
            for (int r = 0; r < reps; ++r) {
                if (factor(st, h, d, di, l) != 0) ++bad_info;
                else {
                    bool same = true;
                    for (int j = 0; j < n && same; ++j)
                        for (int i = j; i < n; ++i)
                            if (l[size_t(j) * n + i] != serial[size_t(j) * n + i]) { same = false; break; }
                    if (!same) ++bad_factor;
                }
            }
        });
    for (auto &t : pool) t.join();
    stop = 1;
    noise.join();
    printf("n %d, %d threads x %d factorisations beside a busy stream: info != 0 in %d calls, factor differs from the serial one in %d calls\n", n, threads, reps,
           bad_info.load(), bad_factor.load());
    return 0;
}
