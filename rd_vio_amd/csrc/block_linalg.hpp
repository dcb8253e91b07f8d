// Workgroup-level building blocks shared by the solver and the marginalisation kernels: fixed-order block
// reductions and the blocked (15-wide) Cholesky factorisation / solve.  T = threads per workgroup.
#pragma once
#include "dmath.hpp"

template <int T>
struct BlockShared {
    double red[T];
    double blk[15 * 16];
    double vec[16];
    int flag;
};

template <int T>
DM double block_sum(BlockShared<T> &sh, double v) {
    const int t = threadIdx.x;
    sh.red[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = T / 2; s > 0; s >>= 1) {
        if (t < s) sh.red[t] += sh.red[t + s];
        __syncthreads();
    }
    const double r = sh.red[0];
    __syncthreads();
    return r;
}
template <int T>
DM double block_max(BlockShared<T> &sh, double v) {
    const int t = threadIdx.x;
    sh.red[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = T / 2; s > 0; s >>= 1) {
        if (t < s) sh.red[t] = fmax(sh.red[t], sh.red[t + s]);
        __syncthreads();
    }
    const double r = sh.red[0];
    __syncthreads();
    return r;
}
// blocked (15-wide) in-place Cholesky of the N x N matrix M (lower triangle), N a multiple of 15.
// Returns 0 on a non-positive / non-finite pivot.
template <int T>
DM int cholesky_blocked(BlockShared<T> &sh, double *M, int N) {
    const int t = threadIdx.x;
    const int nb = N / 15;
    if (t == 0) sh.flag = 1;
    __syncthreads();
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = 15 * kb;
        // (1) diagonal block: factor in LDS by the first wave
        for (int i = t; i < 225; i += T) sh.blk[(i / 15) * 16 + (i % 15)] = M[(size_t)(k0 + i / 15) * N + k0 + (i % 15)];
        __syncthreads();
        if (t < 64) {
            for (int j = 0; j < 15; ++j) {
                double d = sh.blk[j * 16 + j];
                if (!(d > 0.0) || !isfinite(d)) {
                    if (t == 0) sh.flag = 0;
                    d = 1.0;
                }
                d = sqrt(d);
                __builtin_amdgcn_wave_barrier();
                if (t == j) sh.blk[j * 16 + j] = d;
                if (t > j && t < 15) sh.blk[t * 16 + j] /= d;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // trailing update inside the block: entries (r, c), j < c <= r < 15
                for (int e = t; e < 225; e += 64) {
                    const int r = e / 15, c = e - 15 * r;
                    if (c > j && r >= c) sh.blk[r * 16 + c] -= sh.blk[r * 16 + j] * sh.blk[c * 16 + j];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int i = t; i < 225; i += T) {
            const int r = i / 15, c = i - 15 * r;
            if (c <= r) M[(size_t)(k0 + r) * N + k0 + c] = sh.blk[r * 16 + c];
        }
        // (2) panel below: row i solves x L_kk^T = M[i, k0:k0+15]
        for (int i = k0 + 15 + t; i < N; i += T) {
            double x[15];
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                double s = M[(size_t)i * N + k0 + c];
                for (int q = 0; q < c; ++q) s -= x[q] * sh.blk[c * 16 + q];
                x[c] = s / sh.blk[c * 16 + c];
            }
#pragma unroll
            for (int c = 0; c < 15; ++c) M[(size_t)i * N + k0 + c] = x[c];
        }
        __syncthreads();
        // (3) trailing update of the lower triangle
        const int rem = N - (k0 + 15);
        for (int e = t; e < rem * rem; e += T) {
            const int r = e / rem, c = e - r * rem;
            if (c > r) continue;
            const double *Lr = M + (size_t)(k0 + 15 + r) * N + k0, *Lc = M + (size_t)(k0 + 15 + c) * N + k0;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 15; ++q) acc += Lr[q] * Lc[q];
            M[(size_t)(k0 + 15 + r) * N + k0 + 15 + c] -= acc;
        }
        __syncthreads();
    }
    return sh.flag;
}

// solve L L^T y = b in place (y overwrites b), blocked like the factorisation; the 15x15 diagonal block is
// staged in LDS so the sequential triangular solve never waits on global memory
template <int T>
DM void cholesky_solve(BlockShared<T> &sh, const double *M, int N, double *b) {
    const int t = threadIdx.x;
    const int nb = N / 15;
    for (int kb = 0; kb < nb; ++kb) {  // forward
        const int k0 = 15 * kb;
        for (int i = t; i < 225; i += T) sh.blk[(i / 15) * 16 + (i % 15)] = M[(size_t)(k0 + i / 15) * N + k0 + (i % 15)];
        if (t < 15) sh.vec[t] = b[k0 + t];
        __syncthreads();
        if (t == 0) {
            for (int c = 0; c < 15; ++c) {
                double s = sh.vec[c];
                for (int q = 0; q < c; ++q) s -= sh.blk[c * 16 + q] * sh.vec[q];
                sh.vec[c] = s / sh.blk[c * 16 + c];
            }
        }
        __syncthreads();
        if (t < 15) b[k0 + t] = sh.vec[t];
        for (int i = k0 + 15 + t; i < N; i += T) {
            double s = b[i];
#pragma unroll
            for (int q = 0; q < 15; ++q) s -= M[(size_t)i * N + k0 + q] * sh.vec[q];
            b[i] = s;
        }
        __syncthreads();
    }
    for (int kb = nb - 1; kb >= 0; --kb) {  // backward
        const int k0 = 15 * kb;
        for (int i = t; i < 225; i += T) sh.blk[(i / 15) * 16 + (i % 15)] = M[(size_t)(k0 + i / 15) * N + k0 + (i % 15)];
        if (t < 15) sh.vec[t] = b[k0 + t];
        __syncthreads();
        if (t == 0) {
            for (int c = 14; c >= 0; --c) {
                double s = sh.vec[c];
                for (int q = c + 1; q < 15; ++q) s -= sh.blk[q * 16 + c] * sh.vec[q];
                sh.vec[c] = s / sh.blk[c * 16 + c];
            }
        }
        __syncthreads();
        if (t < 15) b[k0 + t] = sh.vec[t];
        for (int i = t; i < k0; i += T) {
            double s = b[i];
#pragma unroll
            for (int q = 0; q < 15; ++q) s -= M[(size_t)(k0 + q) * N + i] * sh.vec[q];
            b[i] = s;
        }
        __syncthreads();
    }
}

