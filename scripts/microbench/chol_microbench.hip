// Developer microbenchmark (GPU box only, not part of the product): where does the LDS Cholesky spend its time?
// Round-2 finding (N = 135): 49.7 us per factorisation; nine diagonal blocks alone 18.3 us (2.0 us = 4900 cycles per 15 pivots:
// a lone wavefront retires about one instruction per 8 cycles, the block is instruction-issue bound), a 15-row panel 0.9 us, one
// MFMA trailing tile 0.63 us in isolation, an LDS flag round trip between two wavefronts 0.12 us, a workgroup barrier 6 ns.
// A pipelined variant (pivot chain on its own wavefront, LDS counters instead of barriers) measured 54 - 87 us: slower.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I rd_vio_amd/csrc scripts/microbench/chol_microbench.hip -o /tmp/chol_mb && /tmp/chol_mb
#include <hip/hip_runtime.h>

#include <cstdio>
#include <random>
#include <vector>

#include "block_linalg.hpp"

constexpr int T = 512;
constexpr int NMAX = 165;
constexpr size_t CAP = (NMAX + 1) * (NMAX + 2) / 2 + 225 * 11;

__global__ __launch_bounds__(T) void bench(const double *A, int N, int reps, unsigned long long *out, double *chk) {
    __shared__ BlockShared<T> sh_store;
    LdsShared<T> &sh = *(LdsShared<T> *)&sh_store;
    __shared__ __attribute__((aligned(16))) double buf[CAP];
    lds_double *Lp = RDVIO_LDS(buf), *Dinv = Lp + (N + 1) * (N + 2) / 2;
    const int t = threadIdx.x, ntri = (N + 1) * (N + 2) / 2;
    auto reload = [&]() {
        for (int i = t; i < ntri; i += T) Lp[i] = A[i];
        __syncthreads();
    };
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // 0: the product's LDS Cholesky; 1: the same without the block inverses
    for (int v = 0; v < 2; ++v) {
        for (int r = 0; r < reps; ++r) {
            reload();
            const unsigned long long t0 = wall_clock64();
            const int ok = cholesky_lds<T>(sh, Lp, Dinv, N, 0.0, v == 0);
            const unsigned long long t1 = wall_clock64();
            acc[v] += t1 - t0;
            if (t == 0 && r == 0) chk[v] = ok ? Lp[tri(N) + N - 1] : -1.0;
        }
    }
    // 2: nine diagonal blocks back to back on wave 0 (fresh data each time)
    for (int r = 0; r < reps; ++r) {
        reload();
        const unsigned long long t0 = wall_clock64();
        if (t < 64)
            for (int kb = 0; kb < N / 15; ++kb) cholesky_diag_block<T>(sh, Lp, 15 * kb, 0.0);
        const unsigned long long t1 = wall_clock64();
        acc[2] += t1 - t0;
        __syncthreads();
    }
    // 3: nine 15-row panel solves back to back on wave 0
    for (int r = 0; r < reps; ++r) {
        reload();
        if (t < 16) sh.vec[t] = 0.5;
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        if (t < 64)
            for (int kb = 0; kb + 1 < N / 15; ++kb)
                if (t < 15) {  // the panel loop body of cholesky_lds for one row
                    const int k0 = 15 * kb;
                    lds_double *row = Lp + tri(k0 + 15 + t) + k0;
                    double x[15];
#pragma unroll
                    for (int c = 0; c < 15; ++c) x[c] = row[c];
#pragma unroll
                    for (int c = 0; c < 15; ++c) {
                        const lds_double *Lc = Lp + tri(k0 + c) + k0;
                        double sacc = x[c];
#pragma unroll
                        for (int q = 0; q < 15; ++q)
                            if (q < c) sacc = __builtin_fma(-x[q], Lc[q], sacc);
                        x[c] = sacc * sh.vec[c];
                    }
#pragma unroll
                    for (int c = 0; c < 15; ++c) row[c] = x[c];
                }
        const unsigned long long t1 = wall_clock64();
        acc[3] += t1 - t0;
        __syncthreads();
    }
    // 4: nine trailing 15x15 blocks back to back on wave 0
    for (int r = 0; r < reps; ++r) {
        reload();
        const unsigned long long t0 = wall_clock64();
        if (t < 64)
            for (int kb = 0; kb + 1 < N / 15; ++kb) cholesky_trailing_tile(Lp, 15 * kb, N + 1, 0, 0);
        const unsigned long long t1 = wall_clock64();
        acc[4] += t1 - t0;
        __syncthreads();
    }
    // 5: publish / wait ping-pong between wave 0 and wave 1, 100 round trips
    {
        if (t == 0) sh.fcol[0] = sh.fcol[1] = 0;
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        typedef __attribute__((address_space(3))) int lds_int;
        lds_int *a = (lds_int *)&sh.fcol[0], *b = (lds_int *)&sh.fcol[1];
        auto publish = [](lds_int *p, int v) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto wait_ge = [](lds_int *p, int v) {
            while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        if (t < 64)
            for (int k = 1; k <= 100; ++k) { publish(a, k); wait_ge(b, k); }
        else if (t < 128)
            for (int k = 1; k <= 100; ++k) { wait_ge(a, k); publish(b, k); }
        const unsigned long long t1 = wall_clock64();
        acc[5] = t1 - t0;
        __syncthreads();
    }
    // 6: backward substitution (cholesky_solve_lds) after a serial factorisation
    for (int r = 0; r < reps; ++r) {
        reload();
        cholesky_lds<T>(sh, Lp, Dinv, N);
        const unsigned long long t0 = wall_clock64();
        cholesky_solve_lds<T>(sh, Lp, Dinv, N, chk + 8);
        const unsigned long long t1 = wall_clock64();
        acc[6] += t1 - t0;
    }
    // 7: 100 workgroup barriers
    {
        const unsigned long long t0 = wall_clock64();
        for (int k = 0; k < 100; ++k) __syncthreads();
        acc[7] = wall_clock64() - t0;
    }
    if (t == 0)
        for (int i = 0; i < 8; ++i) out[i] = acc[i];
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 135, reps = 50;
    std::mt19937 rng(1);
    std::normal_distribution<double> g(0.0, 1.0);
    std::vector<double> M((size_t)N * N), A((size_t)(N + 1) * (N + 2) / 2, 0.0);
    for (double &m : M) m = g(rng);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = i == j ? N : 0.0;
            for (int k = 0; k < N; ++k) s += M[(size_t)i * N + k] * M[(size_t)j * N + k];
            A[(size_t)i * (i + 1) / 2 + j] = s;
        }
    for (int j = 0; j < N; ++j) A[(size_t)N * (N + 1) / 2 + j] = g(rng);
    double *dA, *dchk;
    unsigned long long *dout;
    hipMalloc(&dA, A.size() * 8);
    hipMalloc(&dchk, 1024 * 8);
    hipMalloc(&dout, 64);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bench, dim3(1), dim3(T), 0, 0, dA, N, reps, dout, dchk);
    hipDeviceSynchronize();
    unsigned long long out[8];
    double chk[8];
    hipMemcpy(out, dout, 64, hipMemcpyDeviceToHost);
    hipMemcpy(chk, dchk, 64, hipMemcpyDeviceToHost);
    const char *names[8] = {"cholesky_lds (with block inverses)", "cholesky_lds (factor only)", "9 diag blocks (wave 0)", "8 x 15 panel rows (wave 0)",
                            "8 trailing 15x15 blocks (wave 0)", "100 publish/wait round trips", "backward substitution", "100 workgroup barriers"};
    for (int i = 0; i < 8; ++i) {
        const double us = out[i] * 0.01 / ((i == 5 || i == 7) ? 1 : reps);
        std::printf("%-36s %9.2f us\n", names[i], us);
    }
    std::printf("check (last entry of L^-1 b): %.12g %.12g\n", chk[0], chk[1]);
    return 0;
}
