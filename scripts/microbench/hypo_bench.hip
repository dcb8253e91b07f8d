// Stage clocks of the device hypothesis solvers (csrc/hypo_solvers.hpp) -- diagnostic, not part of the product.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -o scripts/microbench/hypo_bench.bin scripts/microbench/hypo_bench.hip && scripts/microbench/hypo_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <random>
#include <vector>

__device__ unsigned long long g_stamps[8 * 64];
#define HYPO_STAMP(k)                                                                  \
    do {                                                                               \
        if (threadIdx.x == 0) g_stamps[blockIdx.x * 8 + (k)] = wall_clock64();          \
    } while (0)
#include "../../rd_vio_amd/csrc/hypo_solvers.hpp"

__global__ __launch_bounds__(64) void epnp_kernel(const double *X, const double *u, double *models) {
    __shared__ hypo::EpnpWork work;
    const hypo::WaveExec x{(int)threadIdx.x};
    hypo::epnp6(x, &work, X + 18 * blockIdx.x, u + 12 * blockIdx.x, models + 12 * blockIdx.x);
}

__global__ __launch_bounds__(64) void ess_kernel(const double *p1, const double *p2, double *models, int *counts) {
    __shared__ hypo::Ess5Work work;
    __shared__ int n_found;
    const hypo::WaveExec x{(int)threadIdx.x};
    hypo::essential5(x, &work, p1 + 10 * blockIdx.x, p2 + 10 * blockIdx.x, models + 90 * blockIdx.x, &n_found);
    if (threadIdx.x == 0) counts[blockIdx.x] = n_found;
}

int main() {
    const int n = 32;
    std::mt19937 rng(3);
    std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> X(18 * n), u(12 * n);
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < 6; ++i) {
            const double P[3] = {2 * U(rng), 1.5 * U(rng), 4 + U(rng)};
            for (int k = 0; k < 3; ++k) X[18 * b + 3 * i + k] = P[k];
            u[12 * b + 2 * i] = (P[0] + 0.1) / (P[2] + 0.05) + 1e-3 * U(rng);
            u[12 * b + 2 * i + 1] = (P[1] - 0.05) / (P[2] + 0.05) + 1e-3 * U(rng);
        }
    double *dX, *du, *dm;
    (void)hipMalloc(&dX, X.size() * 8); (void)hipMalloc(&du, u.size() * 8); (void)hipMalloc(&dm, 12 * n * 8);
    (void)hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(epnp_kernel, dim3(n), dim3(64), 0, 0, dX, du, dm);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> s(8 * 64);
        (void)hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(g_stamps), s.size() * 8);
        std::printf("epnp6 x %d: kernel %.1f us; block 0 stages (x 10 ns):", n, 1e3 * ms);
        const char *names[6] = {"control points", "MtM", "jacobi12", "L", "candidates", "select+rodrigues"};
        for (int k = 0; k < 6; ++k) std::printf(" %s %llu", names[k], s[k + 1] - s[k]);
        std::printf(" total %llu\n", s[6] - s[0]);
    }
    // ---- five-point essential solver
    {
        std::vector<double> a(10 * n), b(10 * n);
        for (int k = 0; k < n; ++k)
            for (int i = 0; i < 5; ++i) {
                const double P[3] = {2 * U(rng), 1.5 * U(rng), 4 + U(rng)};
                a[10 * k + 2 * i] = P[0] / P[2]; a[10 * k + 2 * i + 1] = P[1] / P[2];
                b[10 * k + 2 * i] = (P[0] + 0.1) / (P[2] + 0.03); b[10 * k + 2 * i + 1] = (P[1] - 0.05) / (P[2] + 0.03);
            }
        double *da, *db, *dmm; int *dc;
        (void)hipMalloc(&da, a.size() * 8); (void)hipMalloc(&db, b.size() * 8); (void)hipMalloc(&dmm, 90 * n * 8); (void)hipMalloc(&dc, n * 4);
        (void)hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(ess_kernel, dim3(n), dim3(64), 0, 0, da, db, dmm, dc);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> s(8 * 64);
            (void)hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(g_stamps), s.size() * 8);
            std::printf("essential5 x %d: kernel %.1f us; block 0 stages (x 10 ns):", n, 1e3 * ms);
            const char *names[5] = {"A, AtA", "jacobi9", "polynomials", "gauss-jordan + hqr", "eigenvectors"};
            for (int k = 0; k < 5; ++k) std::printf(" %s %llu", names[k], s[k + 1] - s[k]);
            std::printf(" total %llu\n", s[5] - s[0]);
        }
    }
    return 0;
}
