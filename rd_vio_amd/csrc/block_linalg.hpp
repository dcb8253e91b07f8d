// Workgroup-level FP64 building blocks shared by the solver and the marginalisation kernels (gfx950):
//  * fixed-order block reductions (wave xor-shuffle + one LDS stage: 1 barrier per reduction, bitwise reproducible)
//  * v_mfma_f64_16x16x4_f64 tile products for the three GEMM-shaped pieces (S^T S, A^T W A, Cholesky trailing update)
//  * wave-per-row mat-vec with coalesced row reads
//  * blocked (15-wide) Cholesky factorisation / solve with the diagonal block staged in LDS
// T = threads per workgroup (multiple of 64).
#pragma once
#include "dmath.hpp"

typedef double double4_t __attribute__((ext_vector_type(4)));
#ifdef RDVIO_PROF_CHOL
// diagnostic build: ticks (100 MHz) of thread 0 inside cholesky_lds -- [0] panel, [1] tile (0,0), [2] diagonal block, [3] waiting
// at the barriers (= the other wavefronts' trailing tiles), [4] diagonal inverses; copied to summary[72..76] at the end of a solve
__device__ unsigned long long rdvio_chol_prof[8];
#endif

// Pointers that keep the LDS address space across (noinline) function boundaries: a generic `double *` to LDS makes the
// compiler emit FLAT loads / stores, which resolve the aperture first and cost about twice the latency of ds_read / ds_write.
typedef __attribute__((address_space(3))) double lds_double;
// Global-memory operands of the hot phases.  The descriptor's pointers are generic and come out of LDS as per-lane values: an
// access through them is a FLAT instruction whose 64-bit address every lane computes with VALU arithmetic.  RDVIO_UG(p) makes
// the pointer wave-uniform (two v_readfirstlane) and global-typed: `global_load` off a scalar base with a 32-bit index.
typedef __attribute__((address_space(1))) double gdouble;
typedef const __attribute__((address_space(1))) double cgdouble;
DM unsigned long long rdvio_uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// ONLY for operands that always live in global memory: the solver redirects its small vectors (g, yp, sig / diag / grad / gn, the
// landmark scalars, x / xd and their candidates, user, lfree) into LDS when there is room -- those stay generic pointers.
#ifndef RDVIO_CHECK_UG
#define RDVIO_UG(p) ((cgdouble *)rdvio_uniform64((unsigned long long)(p)))
#define RDVIO_UGW(p) ((gdouble *)rdvio_uniform64((unsigned long long)(p)))
#else
// Checking build (RDVIO_CHECK_UG=1 python rd_vio_amd/build.py --force; never a timed build): a generic pointer into LDS turned
// global-typed reads the LDS aperture's address as a global one -- round 2's HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION.  Here
// such a pointer is REFUSED instead: the violation is counted in a device word the host reads at the next fetch
// (rdvio_hip_ba_fetch / rdvio_hip_marginalize_fetch return RDVIO_ERR_HIP) and the access goes to a scratch line.
__device__ unsigned g_rdvio_ug_violations;
__device__ double g_rdvio_ug_sink[64];
DM unsigned long long rdvio_ug_checked(const void *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool in_lds = __builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)p);
#else
    const bool in_lds = false;
#endif
    if (in_lds) {
        atomicAdd(&g_rdvio_ug_violations, 1u);
        return (unsigned long long)g_rdvio_ug_sink;
    }
    return rdvio_uniform64((unsigned long long)p);
}
#define RDVIO_UG(p) ((cgdouble *)rdvio_ug_checked((const void *)(p)))
#define RDVIO_UGW(p) ((gdouble *)rdvio_ug_checked((const void *)(p)))
#endif
// Lane-masked load WITHOUT an exec-masked block: `ok ? p[i] : 0.0` compiles to v_cmp / s_and_saveexec / s_cbranch_execz / load /
// s_or exec -- two scalar mask operations and a branch per load, ~60-90 cycles of a lone wavefront's issue (measured: the
// group products spent 3.8 of 8.4 us issuing 96 such loads).  Here every lane loads (a masked lane reads entry `safe`, which
// the caller guarantees to be valid) and the value is selected afterwards: two v_cndmask.  Only where (nearly) all lanes are
// active anyway: a vector load occupies the CU's memory pipeline in proportion to its active lanes, so turning sparsely
// masked loads into full ones costs more than the branches (measured in ne_h_blocks).
DM double rdvio_ldm(cgdouble *p, int off, bool ok, int safe = 0) {
    const double v = p[ok ? off : safe];
    return ok ? v : 0.0;
}
#define RDVIO_LDS(p) ((lds_double *)(p))

template <int T>
struct BlockShared {
    double red[2][16][T / 64];  // double-buffered partials for up to 16 simultaneous reductions
    double blk[15 * 16];
    double vec[16];
    double spec[16];  // solver: the current linearisation's step-selection scalars (|gn|, |g|, alpha, g.gn, q_uu q_uv q_vv l_u l_v)
    int flag;
    int lost;  // set when a helper workgroup did not answer in time (solver)
    int seq;  // command sequence number of the helper-workgroup protocol (solver)
    // small per-problem index tables (solver): frame -> free column block, column block -> prior frame, preintegration sources
    // per frame (up to RDVIO_SOLVER_MAX_FRAMES = 64, most of them constant anchors) / per free column block (<= 32)
    int fcol[64], pcol[32], band_src[32 * 6], g_src[32 * 2];
    int pfix[64], pfixc[32];  // pose-constant flag per frame / per free column block
    double Jri[32 * 9];  // Jr^-1(e_theta) of the prior frames at the current linearisation
    double xv[512];      // staged vector operand (pose step / prior error); = RDVIO_SOLVER_XV, bounds checked in rdvio_ba_prepare
    double st[64 * 16];  // frame states being evaluated (x or the candidate)
    double ub[64 * 6];   // user-state biases (bias linearisation of the preintegration factors)
    double ext[18];      // extrinsics (14) + sqrt_inv_cov (4)
    double cam[64 * 12]; // per frame: camera-to-world rotation (row-major 3x3) and camera centre of the states being evaluated
};
// the kernels hand the block around as an LDS-typed reference, so that accesses compile to ds_read / ds_write in
// noinline callees too (a generic reference makes them FLAT)
template <int T>
using LdsShared = __attribute__((address_space(3))) BlockShared<T>;
// a member array handed to an (inlined) routine that takes plain pointers: the cast folds away after inlining
#define RDVIO_GEN(a) ((double *)(a))

// dst[0..n) (LDS) = src[0..n) (global), all T threads, U loads per thread in flight.  The plain loop
// `for (i = t; i < n; i += T) dst[i] = src[i]` compiles to load -> wait -> store per trip: one L2 round trip (~0.2 us) per
// element and thread; batching makes a 16-element-per-thread copy two round trips instead of sixteen.
template <int T, int U = 8>
DM void stage_to_lds(lds_double *dst, const double *__restrict__ src, int n) {
    const int t = threadIdx.x;
    for (int o = t; o < n; o += T * U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (o + u * T < n) ? src[o + u * T] : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (o + u * T < n) dst[o + u * T] = v[u];
    }
}

// sum_i a[i * sa] * x[i * sx] with the loads of U iterations issued together (memory-level parallelism: a single
// workgroup has little other latency hiding).  Summation order is fixed.
template <int U = 8>
DM double dot_strided(const double *__restrict__ a, long sa, const double *__restrict__ x, long sx, int n) {
    double acc = 0.0;
    int i = 0;
    for (; i + U <= n; i += U) {
        double av[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            av[u] = a[(long)(i + u) * sa];
            xv[u] = x[(long)(i + u) * sx];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += av[u] * xv[u];
    }
    if (i < n) {  // (remainder as one masked batch: a one-element loop is a round trip per element)
        double av[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            av[u] = (i + u < n) ? a[(long)(i + u) * sa] : 0.0;
            xv[u] = (i + u < n) ? x[(long)(i + u) * sx] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u < n) acc += av[u] * xv[u];
    }
    return acc;
}

// y[r] = sum_c M[c * ld + r] * x[c] for the row r = t / 4 owned by lanes 4r .. 4r+3 (c = part, part + 4, ...), combined
// with two xor-shuffles: every lane of the quad returns the row's sum.  Call with r < R for all four lanes or none.
DM double quad_col_dot(const double *__restrict__ M, long ld, const double *__restrict__ x, int C, int r, int part) {
    double acc = 0.0;
    int c = part;
    for (; c + 60 < C; c += 64) {  // 16 loads in flight per lane: two L2 round trips cover 128 columns
        double mv[16], xv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            mv[u] = M[(long)(c + 4 * u) * ld + r];
            xv[u] = x[c + 4 * u];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += mv[u] * xv[u];
    }
    for (; c + 28 < C; c += 32) {
        double mv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            mv[u] = M[(long)(c + 4 * u) * ld + r];
            xv[u] = x[c + 4 * u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += mv[u] * xv[u];
    }
    if (c < C) {  // the remainder (at most seven columns of this part) as ONE masked batch: a one-column loop is a round trip per column
        double mv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = c + 4 * u < C;
            mv[u] = ok ? M[(long)(c + 4 * u) * ld + r] : 0.0;
            xv[u] = ok ? x[c + 4 * u] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c + 4 * u < C) acc += mv[u] * xv[u];
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    return acc;
}

// two right-hand sides at once: (M x)[r] and (M y)[r] from one pass over the matrix (same summation order as quad_col_dot)
DM void quad_col_dot2(const double *__restrict__ M, long ld, const double *__restrict__ x, const double *__restrict__ y, int C,
                      int r, int part, double &ox, double &oy) {
    double ax = 0.0, ay = 0.0;
    int c = part;
    for (; c + 60 < C; c += 64) {
        double mv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) mv[u] = M[(long)(c + 4 * u) * ld + r];
#pragma unroll
        for (int u = 0; u < 16; ++u) { ax += mv[u] * x[c + 4 * u]; ay += mv[u] * y[c + 4 * u]; }
    }
    for (; c + 28 < C; c += 32) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = M[(long)(c + 4 * u) * ld + r];
#pragma unroll
        for (int u = 0; u < 8; ++u) { ax += mv[u] * x[c + 4 * u]; ay += mv[u] * y[c + 4 * u]; }
    }
    if (c < C) {  // (remainder as one masked batch, as in quad_col_dot)
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = (c + 4 * u < C) ? M[(long)(c + 4 * u) * ld + r] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c + 4 * u < C) { ax += mv[u] * x[c + 4 * u]; ay += mv[u] * y[c + 4 * u]; }
    }
    ax += __shfl_xor(ax, 1);
    ax += __shfl_xor(ax, 2);
    ay += __shfl_xor(ay, 1);
    ay += __shfl_xor(ay, 2);
    ox = ax;
    oy = ay;
}

// quad_col_dot2 with the matrix behind a wave-uniform global-typed pointer and the vectors behind LDS-typed pointers: every load is
// `scalar base + per-lane offset (row r, part) + compile-time column step`, the vector reads are ds_read with immediate
// offsets -- the generic-pointer form spends two to four VALU instructions on each 64-bit address.  Same summation order.
DM void quad_col_dot2_u(cgdouble *M, int ld, const lds_double *x, const lds_double *y, int C, int r, int part, double &ox, double &oy) {
    double ax = 0.0, ay = 0.0;
    const int lane_off = part * ld + r;
    const lds_double *xp = x + part, *yp = y + part;
    int c = 0;   // (column = part + c)
    for (; part + c + 60 < C; c += 64) {
        double mv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) mv[u] = M[(c + 4 * u) * ld + lane_off];
#pragma unroll
        for (int u = 0; u < 16; ++u) { ax += mv[u] * xp[c + 4 * u]; ay += mv[u] * yp[c + 4 * u]; }
    }
    for (; part + c + 28 < C; c += 32) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = M[(c + 4 * u) * ld + lane_off];
#pragma unroll
        for (int u = 0; u < 8; ++u) { ax += mv[u] * xp[c + 4 * u]; ay += mv[u] * yp[c + 4 * u]; }
    }
    if (part + c < C) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = (part + c + 4 * u < C) ? M[(c + 4 * u) * ld + lane_off] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (part + c + 4 * u < C) { ax += mv[u] * xp[c + 4 * u]; ay += mv[u] * yp[c + 4 * u]; }
    }
    ax += __shfl_xor(ax, 1);
    ax += __shfl_xor(ax, 2);
    ay += __shfl_xor(ay, 1);
    ay += __shfl_xor(ay, 2);
    ox = ax;
    oy = ay;
}

// typed-pointer forms of quad_col_dot / quad_col_dotk (see quad_col_dot2_u): M behind a wave-uniform global-typed pointer, the
// vector(s) in LDS.  Same summation order as the generic forms.
template <int KN>
DM void quad_col_dotk_u(cgdouble *M, int ld, const lds_double *x, int xs, int K, int C, int r, int part, double (&out)[KN]) {
    double acc[KN];
#pragma unroll
    for (int k = 0; k < KN; ++k) acc[k] = 0.0;
    const int lane_off = part * ld + r;
    const lds_double *xp = x + part;
    int c = 0;   // (column = part + c)
    for (; part + c + 60 < C; c += 64) {
        double mv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) mv[u] = M[(c + 4 * u) * ld + lane_off];
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[k] += mv[u] * xp[k * xs + c + 4 * u];
    }
    for (; part + c + 28 < C; c += 32) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = M[(c + 4 * u) * ld + lane_off];
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[k] += mv[u] * xp[k * xs + c + 4 * u];
    }
    if (part + c < C) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = (part + c + 4 * u < C) ? M[(c + 4 * u) * ld + lane_off] : 0.0;
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (part + c + 4 * u < C) acc[k] += mv[u] * xp[k * xs + c + 4 * u];
    }
#pragma unroll
    for (int k = 0; k < KN; ++k) {
        acc[k] += __shfl_xor(acc[k], 1);
        acc[k] += __shfl_xor(acc[k], 2);
        out[k] = acc[k];
    }
}
DM double quad_col_dot_u(cgdouble *M, int ld, const lds_double *x, int C, int r, int part) {
    double o[1];
    quad_col_dotk_u<1>(M, ld, x, 0, 1, C, r, part, o);
    return o[0];
}

// up to four right-hand sides x_k = x + k * xs at once, each summed exactly like quad_col_dot sums it
DM void quad_col_dot4(const double *__restrict__ M, long ld, const double *__restrict__ x, int xs, int K, int C, int r, int part,
                      double (&out)[4]) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int c = part;
    for (; c + 60 < C; c += 64) {
        double mv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) mv[u] = M[(long)(c + 4 * u) * ld + r];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[k] += mv[u] * x[k * xs + c + 4 * u];
    }
    for (; c + 28 < C; c += 32) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = M[(long)(c + 4 * u) * ld + r];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[k] += mv[u] * x[k * xs + c + 4 * u];
    }
    if (c < C) {  // (remainder as one masked batch, as in quad_col_dot)
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = (c + 4 * u < C) ? M[(long)(c + 4 * u) * ld + r] : 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + 4 * u < C) acc[k] += mv[u] * x[k * xs + c + 4 * u];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc[k] += __shfl_xor(acc[k], 1);
        acc[k] += __shfl_xor(acc[k], 2);
        out[k] = acc[k];
    }
}

// the same for up to KN right-hand sides (the speculative trial steps)
template <int KN>
DM void quad_col_dotk(const double *__restrict__ M, long ld, const double *__restrict__ x, int xs, int K, int C, int r, int part,
                      double (&out)[KN]) {
    double acc[KN];
#pragma unroll
    for (int k = 0; k < KN; ++k) acc[k] = 0.0;
    int c = part;
    for (; c + 60 < C; c += 64) {
        double mv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) mv[u] = M[(long)(c + 4 * u) * ld + r];
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[k] += mv[u] * x[k * xs + c + 4 * u];
    }
    for (; c + 28 < C; c += 32) {
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = M[(long)(c + 4 * u) * ld + r];
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[k] += mv[u] * x[k * xs + c + 4 * u];
    }
    if (c < C) {  // (remainder as one masked batch, as in quad_col_dot)
        double mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = (c + 4 * u < C) ? M[(long)(c + 4 * u) * ld + r] : 0.0;
#pragma unroll
        for (int k = 0; k < KN; ++k)
            if (k < K)
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c + 4 * u < C) acc[k] += mv[u] * x[k * xs + c + 4 * u];
    }
#pragma unroll
    for (int k = 0; k < KN; ++k) {
        acc[k] += __shfl_xor(acc[k], 1);
        acc[k] += __shfl_xor(acc[k], 2);
        out[k] = acc[k];
    }
}

DM double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
DM double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// Sum NV values over the workgroup; every thread receives the totals.  One __syncthreads per call: the partial
// buffers alternate between two banks, so the next call cannot overwrite values another wave is still reading.
template <int T, int NV>
DM void block_sum_n(LdsShared<T> &sh, double (&v)[NV], int &phase) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0) sh.red[phase][i][wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < T / 64; ++q) s += sh.red[phase][i][q];
        v[i] = s;
    }
    phase ^= 1;
}
template <int T>
DM double block_sum(LdsShared<T> &sh, double v, int &phase) {
    double a[1] = {v};
    block_sum_n<T, 1>(sh, a, phase);
    return a[0];
}
template <int T>
DM double block_max(LdsShared<T> &sh, double v, int &phase) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const double s = wave_max(v);
    if (lane == 0) sh.red[phase][0][wave] = s;
    __syncthreads();
    double m = sh.red[phase][0][0];
#pragma unroll
    for (int q = 1; q < T / 64; ++q) m = fmax(m, sh.red[phase][0][q]);
    phase ^= 1;
    return m;
}

// Operands of one 15 x 15 x 15 product  C[m][n] = sum_q a(q, m) b(q, n)  for v_mfma_f64_16x16x4_f64 (a(q, m) =
// Ap[q * sak + m * sam], b likewise): the eight loads of a lane are independent, so one memory round trip feeds the
// whole tile; mfma_run15 then returns C[row = (lane >> 4) + 4 r][col = lane & 15] in register r.
DM void mfma_load15(const double *__restrict__ Ap, long sak, long sam, const double *__restrict__ Bp, long sbk, long sbn,
                    double (&a)[4], double (&b)[4]) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = 4 * u + kk;
        const bool ok = i < 15 && q < 15;
        a[u] = ok ? Ap[q * sak + i * sam] : 0.0;
        b[u] = ok ? Bp[q * sbk + i * sbn] : 0.0;
    }
}
DM double4_t mfma_run15(const double (&a)[4], const double (&b)[4]) {
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    return acc;
}

// One 16x16 output tile  C[m0.., n0..] = sum_k a(k, m) * w(k) * b(k, n)  on one wavefront with
// v_mfma_f64_16x16x4_f64.  a(k, m) = Ap[k * sak + m * sam], b(k, n) = Bp[k * sbk + n * sbn]; w may be null.
// Lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; result register r of lane l is
// C[row = (l >> 4) + 4 r][col = l & 15].
// P: pointer type of the operands (plain `const double *` for global memory, `const lds_double *` for operands staged
// in LDS -- a generic pointer to LDS would compile to FLAT loads); HW: whether the weight vector w is present.
template <class P, bool HW>
DM double4_t mfma_tile_g(P Ap, long sak, long sam, P Bp, long sbk, long sbn, P w, int K, int m0, int n0, int M, int N) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const bool am = (m0 + i) < M, bn = (n0 + i) < N;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    P ap = Ap + (long)(m0 + i) * sam, bp = Bp + (long)(n0 + i) * sbn;
    int k0 = 0;
    for (; k0 + 16 <= K; k0 += 16) {  // 4 MFMAs per trip: 8 independent loads in flight
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + kk;
            a[u] = am ? ap[(long)k * sak] : 0.0;
            b[u] = bn ? bp[(long)k * sbk] : 0.0;
            if (HW) a[u] *= w[k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k0 < K; k0 += 4) {
        const int k = k0 + kk;
        double a = 0.0, b = 0.0;
        if (k < K) {
            a = am ? ap[(long)k * sak] : 0.0;
            b = bn ? bp[(long)k * sbk] : 0.0;
            if (HW) a *= w[k];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
DM double4_t mfma_tile_f64(const double *__restrict__ Ap, long sak, long sam, const double *__restrict__ Bp, long sbk,
                           long sbn, const double *__restrict__ w, int K, int m0, int n0, int M, int N) {
    return w ? mfma_tile_g<const double *, true>(Ap, sak, sam, Bp, sbk, sbn, w, K, m0, n0, M, N)
             : mfma_tile_g<const double *, false>(Ap, sak, sam, Bp, sbk, sbn, Ap, K, m0, n0, M, N);
}

// C (M x N, row-major, ld = ldc) = A^T diag(w) B with A: K x M, B: K x N row-major; tiles spread over the waves.
// If `lower_only`, tiles strictly above the diagonal are skipped (symmetric result, M == N).
// MIRROR (square, lower_only): an off-diagonal tile is stored a second time transposed, so that C is complete.
template <int T, class P, bool HW, bool MIRROR = false>
DM void block_gemm_tn_g(double *__restrict__ C, int ldc, P A, int lda, P B, int ldb, P w, int M, int N, int K, bool lower_only) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = T / 64;
    const int tm = (M + 15) / 16, tn = (N + 15) / 16;
    // a square lower-only product walks the list of lower tiles, so that every wavefront gets the same number of them
    // (tile = wave, wave + nw, ... over the full grid would hand wavefront 0 a whole tile column and the last one a tile)
    const bool tri_walk = lower_only && M == N;
    const int ntile = tri_walk ? tm * (tm + 1) / 2 : tm * tn;
    for (int tile = wave; tile < ntile; tile += nw) {
        int bi, bj;
        if (tri_walk) {
            bi = 0;
            while ((bi + 1) * (bi + 2) / 2 <= tile) ++bi;
            bj = tile - bi * (bi + 1) / 2;
        } else {
            bi = tile / tn;
            bj = tile - bi * tn;
            if (lower_only && bj > bi && 16 * bj + 16 < N) continue;  // (a trailing extra column, N = M + 1, is always computed)
        }
        const double4_t acc = mfma_tile_g<P, HW>(A, lda, 1, B, ldb, 1, w, K, 16 * bi, 16 * bj, M, N);
        const int col = 16 * bj + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * bi + (lane >> 4) + 4 * r;
            if (row < M && col < N) {
                C[(long)row * ldc + col] = acc[r];
                if (MIRROR && bi != bj) C[(long)col * ldc + row] = acc[r];
            }
        }
    }
}
template <int T>
DM void block_gemm_tn(double *__restrict__ C, int ldc, const double *__restrict__ A, int lda, const double *__restrict__ B,
                      int ldb, const double *__restrict__ w, int M, int N, int K, bool lower_only) {
    if (w) block_gemm_tn_g<T, const double *, true>(C, ldc, A, lda, B, ldb, w, M, N, K, lower_only);
    else block_gemm_tn_g<T, const double *, false>(C, ldc, A, lda, B, ldb, A, M, N, K, lower_only);
}
// the same with the operands (and weights, if any) staged in LDS
template <int T>
DM void block_gemm_tn_lds(double *__restrict__ C, int ldc, const lds_double *A, int lda, const lds_double *B, int ldb,
                          const lds_double *w, bool has_w, int M, int N, int K, bool lower_only) {
    if (has_w) block_gemm_tn_g<T, const lds_double *, true>(C, ldc, A, lda, B, ldb, w, M, N, K, lower_only);
    else block_gemm_tn_g<T, const lds_double *, false>(C, ldc, A, lda, B, ldb, A, M, N, K, lower_only);
}

// C = A^T diag(w) A (lower tiles + a trailing extra column, like block_gemm_tn with lower_only) for an operand too large
// for LDS as a whole: K is walked in chunks of rows that DO fit (A is row-major K x lda, so a chunk is one contiguous run of
// memory: every thread keeps tens of coalesced loads in flight instead of one dependent L2 round trip per four rows of K),
// every wavefront keeps the accumulators of its tiles in registers across the chunks.  Same MFMA sequence per tile as the
// unstaged product (K ascending, four rows per instruction; chunks are multiples of four rows): bit-identical results.
// lds: scratch of lds_cap doubles; M = rows of C, N = columns (M or M + 1), w = per-row weight (global, K entries).
template <int T, int MAXT = 12>
__device__ __attribute__((noinline)) bool block_gemm_tn_chunked(double *__restrict__ C, int ldc, const double *__restrict__ A, int lda, const double *__restrict__ wgt, int M, int N,
                              int K, lds_double *lds, size_t lds_cap) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, nw = T / 64;
    const int tm = (M + 15) / 16, tn = (N + 15) / 16;
    // this wavefront's tiles: walk the tile grid with the same skip rule as block_gemm_tn_g (strictly-upper tiles are
    // skipped unless they belong to the trailing extra column)
    int tbi[MAXT], tbj[MAXT], nt = 0, seen = 0;
    {   // workgroup-uniform capacity check BEFORE anything else: the chunk loop below contains barriers
        int total = 0;
        for (int tile = 0; tile < tm * tn; ++tile) {
            const int bi = tile / tn, bj = tile - bi * tn;
            if (!(bj > bi && 16 * bj + 16 < N)) ++total;
        }
        if ((total + nw - 1) / nw > MAXT) return false;
    }
    for (int tile = 0; tile < tm * tn; ++tile) {
        const int bi = tile / tn, bj = tile - bi * tn;
        if (bj > bi && 16 * bj + 16 < N) continue;
        if (seen++ % nw != wave) continue;
#pragma unroll
        for (int q = 0; q < MAXT; ++q)
            if (q == nt) { tbi[q] = bi; tbj[q] = bj; }
        ++nt;
    }
    double4_t acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) acc[q] = double4_t{0.0, 0.0, 0.0, 0.0};
    int kc = (int)(lds_cap / (size_t)(lda + 1));
    kc &= ~3;
    if (kc < 4) return false;
    lds_double *As = lds, *ws = lds + (size_t)kc * lda;
    const int i = lane & 15, kk = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += kc) {
        const int kn = (K - k0 < kc) ? K - k0 : kc;
        const double *src = A + (size_t)k0 * lda;
        stage_to_lds<T, 16>(As, src, kn * lda);
        stage_to_lds<T, 1>(ws, wgt + k0, kn);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXT; ++q) {
            if (q < nt) {
                const int m0 = 16 * tbi[q], n0 = 16 * tbj[q];
                const bool am = (m0 + i) < M, bn = (n0 + i) < N;
                const lds_double *ap = As + m0 + i, *bp = As + n0 + i;
                double4_t a4 = acc[q];
                int kq = 0;
                for (; kq + 16 <= kn; kq += 16) {
                    double av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kq + 4 * u + kk;
                        av[u] = am ? ap[(size_t)k * lda] : 0.0;
                        bv[u] = bn ? bp[(size_t)k * lda] : 0.0;
                        av[u] *= ws[k];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], a4, 0, 0, 0);
                }
                for (; kq < kn; kq += 4) {
                    const int k = kq + kk;
                    double av = 0.0, bv = 0.0;
                    if (k < kn) {
                        av = am ? ap[(size_t)k * lda] : 0.0;
                        bv = bn ? bp[(size_t)k * lda] : 0.0;
                        av *= ws[k];
                    }
                    a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, a4, 0, 0, 0);
                }
                acc[q] = a4;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < MAXT; ++q)
        if (q < nt) {
            const int col = 16 * tbj[q] + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * tbi[q] + (lane >> 4) + 4 * r;
                if (row < M && col < N) C[(long)row * ldc + col] = acc[q][r];
            }
        }
    return true;
}

// y[row] = sum_c Mx[row * ld + c] * x[c]  (+ add[row]) for row in [0, R): one wave per row, coalesced row reads.
// Calls emit(row, value) on lane 0 of the owning wave.
template <int T, class Emit>
DM void block_matvec_rows(const double *__restrict__ Mx, int ld, int R, int C, const double *__restrict__ x, Emit emit) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = T / 64;
    for (int row = wave; row < R; row += nw) {
        double acc = 0.0;
        for (int c = lane; c < C; c += 64) acc += Mx[(long)row * ld + c] * x[c];
        acc = wave_sum(acc);
        if (lane == 0) emit(row, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Global-memory blocked Cholesky (windows whose packed triangle does not fit LDS: more than RDVIO_LDS_CHOL_MAX_FRAMES free
// frames, i.e. BASELINE config 5).  Every access to M is an L2 round trip (~0.2 us), so the routine is organised around
// the number of DEPENDENT round trips, not around flops:
//   * the diagonal block is staged into LDS (packed) and factored by the register routine of the LDS Cholesky
//     (cholesky_diag_block: v_readlane pivots, no memory on the pivot chain);
//   * the panel is one thread per row with its fifteen loads issued together;
//   * the trailing update walks TWO 16 x 16 tiles per trip with all operand and target loads (24-32 per lane) in flight
//     before the first MFMA -- one round trip per pair of tiles instead of five per tile;
//   * optionally (NR = N + 1) the right-hand side rides along as row N of the matrix (ld stays N) and leaves the
//     factorisation as L^-1 b: the forward substitution is free, as in the LDS version.
// Returns 0 on a non-positive / non-finite pivot.
// ---------------------------------------------------------------------------------------------------------
DM int tri(int r) { return r * (r + 1) / 2; }


// broadcast a double from a (wave-uniform) lane through SGPRs
DM double readlane_d(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
// 1/sqrt(x) to full double precision: hardware estimate + two Newton steps (no FP64 divide / sqrt sequence)
DM double rsqrt_nr(double x) {
    // (explicit FMAs: the factorisation is not compared bit for bit with anything, and on the pivot chain -- which is
    // instruction-issue bound -- a fused step is one instruction instead of two)
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = __builtin_fma(y, __builtin_fma(-(h * y), y, 0.5), y);
    y = __builtin_fma(y, __builtin_fma(-(h * y), y, 0.5), y);
    return y;
}

template <int T> DM void cholesky_diag_block(LdsShared<T> &sh, lds_double *Lp, int k0, double tol);

DM void cholesky_trailing_pair_global(double *__restrict__ M, int N, int NR, int k0, int bi0, int bj0, int bi1, int bj1, bool two) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const int base = k0 + 15;
    const int ra[2] = {base + 16 * bi0 + i, base + 16 * bi1 + i}, rb[2] = {base + 16 * bj0 + i, base + 16 * bj1 + i};
    double a[2][4], b[2][4], c[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bool on = p == 0 || two;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = 4 * u + kk;
            a[p][u] = (on && ra[p] < NR && q < 15) ? M[(size_t)ra[p] * N + k0 + q] : 0.0;
            b[p][u] = (on && rb[p] < NR && q < 15) ? M[(size_t)rb[p] * N + k0 + q] : 0.0;
        }
        const int bi = p == 0 ? bi0 : bi1, bj = p == 0 ? bj0 : bj1;
        const int col = base + 16 * bj + (lane & 15);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int row = base + 16 * bi + (lane >> 4) + 4 * r4;
            c[p][r4] = (on && row < NR && col <= row && col < N) ? M[(size_t)row * N + col] : 0.0;
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p][u], b[p][u], acc, 0, 0, 0);
        const bool on = p == 0 || two;
        const int bi = p == 0 ? bi0 : bi1, bj = p == 0 ? bj0 : bj1;
        const int col = base + 16 * bj + (lane & 15);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int row = base + 16 * bi + (lane >> 4) + 4 * r4;
            if (on && row < NR && col <= row && col < N) M[(size_t)row * N + col] = c[p][r4] - acc[r4];
        }
    }
}

template <int T>
__device__ __attribute__((noinline)) int cholesky_blocked(LdsShared<T> &sh, double *M, int N, int NR = -1) {
    const int t = threadIdx.x, wave = t >> 6, nw = T / 64;
    const int nb = N / 15;
    if (NR < 0) NR = N;
    lds_double *Pk = RDVIO_LDS(sh.xv);   // the diagonal block, packed (entry (r, c) at tri(r) + c); xv is idle during a factorisation
    if (t == 0) sh.flag = 1;
    __syncthreads();
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = 15 * kb;
        if (t < 225) {
            const int r = t / 15, c = t - 15 * r;
            if (c <= r) Pk[r * (r + 1) / 2 + c] = M[(size_t)(k0 + r) * N + k0 + c];
        }
        __syncthreads();
        if (wave == 0) cholesky_diag_block<T>(sh, Pk, 0, 0.0);   // factor -> Pk, reciprocal pivots -> sh.vec, sh.flag on a bad pivot
        __syncthreads();
        if (t < 225) {
            const int r = t / 15, c = t - 15 * r;
            if (c <= r) M[(size_t)(k0 + r) * N + k0 + c] = Pk[r * (r + 1) / 2 + c];
        }
        // panel below (and the right-hand-side row): row i solves x L_kk^T = M[i, k0:k0+15]
        for (int i = k0 + 15 + t; i < NR; i += T) {
            double x[15];
#pragma unroll
            for (int c = 0; c < 15; ++c) x[c] = M[(size_t)i * N + k0 + c];
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                double s = x[c];
#pragma unroll
                for (int q = 0; q < 15; ++q)
                    if (q < c) s = __builtin_fma(-x[q], Pk[c * (c + 1) / 2 + q], s);
                x[c] = s * sh.vec[c];
            }
#pragma unroll
            for (int c = 0; c < 15; ++c) M[(size_t)i * N + k0 + c] = x[c];
        }
        __syncthreads();
        // trailing update of the lower triangle (and of the right-hand-side row): C[r][c] -= sum_q P[r][q] P[c][q]
        const int rem = NR - (k0 + 15);
        if (rem > 0 && k0 + 15 < N) {
            const int tn = (rem + 15) / 16, ntile = tn * (tn + 1) / 2;
            for (int p = 2 * wave; p < ntile; p += 2 * nw) {
                int bi0 = 0;
                while ((bi0 + 1) * (bi0 + 2) / 2 <= p) ++bi0;
                const int bj0 = p - bi0 * (bi0 + 1) / 2;
                const bool two = p + 1 < ntile;
                int bi1 = bi0, bj1 = bj0 + 1;
                if (bj1 > bi1) {
                    ++bi1;
                    bj1 = 0;
                }
                cholesky_trailing_pair_global(M, N, NR, k0, bi0, bj0, two ? bi1 : bi0, two ? bj1 : bj0, two);
            }
        }
        __syncthreads();
    }
    return sh.flag;
}

// L_kk y = z (forward) or L_kk^T y = z (backward) for one 15 x 15 diagonal block staged in sh.blk (row r at 16 r), z in
// sh.vec[0..14], on ONE wavefront with every operand in registers: lane r keeps row r (forward) or column r (backward) of the
// block, the solved component travels through v_readlane -- fifteen short steps instead of a 105-term chain on one thread.
template <int T>
DM void block_triangular_solve(LdsShared<T> &sh, bool forward) {
    const int lane = threadIdx.x & 63, r = lane < 15 ? lane : 14;
    double l[15], z = sh.vec[r];
#pragma unroll
    for (int c = 0; c < 15; ++c) l[c] = forward ? sh.blk[r * 16 + c] : sh.blk[c * 16 + r];   // row r / column r
    double y = 0.0;
    if (forward) {
#pragma unroll
        for (int c = 0; c < 15; ++c) {
            const double yc = readlane_d(z, c) / readlane_d(l[c], c);   // z_c / L_cc (uniform)
            if (lane == c) y = yc;
            if (lane > c) z = __builtin_fma(-l[c], yc, z);
        }
    } else {
#pragma unroll
        for (int c = 14; c >= 0; --c) {
            const double yc = readlane_d(z, c) / readlane_d(l[c], c);   // lane c's l[c] = L_cc in both layouts
            if (lane == c) y = yc;
            if (lane < c) z = __builtin_fma(-l[c], yc, z);             // column r, entry c: L[c][r]
        }
    }
    if (lane < 15) sh.vec[lane] = y;
}

// solve L L^T y = b in place (y overwrites b), blocked like the factorisation; M's row stride is N.
// rhs_row: b is first taken from row N of M, where cholesky_blocked(.., NR = N + 1) left L^-1 b -- then forward = false and
// only the backward substitution runs.
template <int T>
__device__ __attribute__((noinline)) void cholesky_solve(LdsShared<T> &sh, const double *M, int N, double *b, bool forward = true, bool backward = true,
                                                         bool rhs_row = false) {
    const int t = threadIdx.x;
    const int nb = N / 15;
    if (rhs_row) {
        for (int i = t; i < N; i += T) b[i] = M[(size_t)N * N + i];
        __syncthreads();
    }
    if (forward)
        for (int kb = 0; kb < nb; ++kb) {
            const int k0 = 15 * kb;
            for (int i = t; i < 225; i += T) sh.blk[(i / 15) * 16 + (i % 15)] = M[(size_t)(k0 + i / 15) * N + k0 + (i % 15)];
            if (t < 15) sh.vec[t] = b[k0 + t];
            __syncthreads();
            if (t < 64) block_triangular_solve<T>(sh, true);
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
    if (backward)
        for (int kb = nb - 1; kb >= 0; --kb) {
            const int k0 = 15 * kb;
            for (int i = t; i < 225; i += T) sh.blk[(i / 15) * 16 + (i % 15)] = M[(size_t)(k0 + i / 15) * N + k0 + (i % 15)];
            if (t < 15) sh.vec[t] = b[k0 + t];
            __syncthreads();
            if (t < 64) block_triangular_solve<T>(sh, false);
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

// ---------------------------------------------------------------------------------------------------------
// LDS-resident Cholesky.  The lower triangle of the N x N matrix lives in LDS packed row-major:
// entry (r, c), c <= r, at Lp[r (r + 1) / 2 + c].
// Factorisation: un-normalised outer-product form -- for column j every remaining entry does
//     a_rc -= a_rj a_cj / a_jj          (j < c <= r)
// with ONE barrier per column (no separate "scale the column" phase; the 1/sqrt(a_jj) scaling is applied to all
// columns in one pass at the end).  The chain of N dependent steps is what bounds a Cholesky at this size; in LDS a
// step is ~500 cycles (LDS bandwidth: N^3/6 entry updates x 4 LDS operations) instead of several L2 round trips.
// Dinv receives the inverses of the 15 x 15 diagonal blocks of L so the triangular solves are small mat-vecs.
// ---------------------------------------------------------------------------------------------------------
// Blocked (15-wide) right-looking Cholesky on the LDS-resident packed matrix.  Per block column:
//   (1) the 15 x 15 diagonal block is factored by ONE wavefront entirely in registers: lane r holds row r, pivots and
//       multipliers travel through v_readlane (no LDS round trips in the 15-step dependent chain);
//   (2) the panel below is solved one thread per row against the block's factor (reciprocal pivots, no divides);
//   (3) the trailing matrix is updated with v_mfma_f64_16x16x4 tiles (K = 15): ~0.2 instructions per entry instead of
//       ~30 for scalar indexed updates -- with one workgroup the factorisation is instruction-issue bound.
// 15 x 15 diagonal block at k0 factored in the registers of one wavefront (lane r = row r); rinv[c] = 1 / L_cc goes
// to sh.vec for the panel.  The next pivot is finished first in every step, so that its reciprocal square root (a
// ~200-cycle dependent chain) runs in the shadow of the remaining column updates.
template <int T>
DM void cholesky_diag_block(LdsShared<T> &sh, lds_double *Lp, int k0, double tol) {
    const int lane = threadIdx.x & 63;
    const int r = lane < 15 ? lane : 14;
    const lds_double *row = Lp + tri(k0 + r) + k0;
    double a[15];
#pragma unroll
    for (int c = 0; c < 15; ++c) a[c] = (lane < 15 && c <= lane) ? row[c] : 0.0;
    bool bad = false;
    double piv = readlane_d(a[0], 0);
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        if (!(piv > tol) || !isfinite(piv)) bad = true;
        const double rs = rsqrt_nr(piv);   // 1 / L_jj
        const double lj = (lane == j) ? piv * rs : a[j] * rs;
        a[j] = lj;
        if (lane == j) sh.vec[j] = rs;
        if (j + 1 < 15) {
            a[j + 1] = __builtin_fma(-lj, readlane_d(lj, j + 1), a[j + 1]);
            piv = readlane_d(a[j + 1], j + 1);
        }
        // the remaining multipliers L[c][j], c >= j + 2, are off the pivot chain: the column goes through LDS once (one
        // write, then broadcast reads -- one instruction per multiplier instead of two v_readlane; the step is issue-bound)
        if (j + 2 < 15) {
            if (lane < 16) sh.blk[16 * j + lane] = lj;  // (16 slots per column; nobody else touches sh.blk during the factorisation)
#pragma unroll
            for (int c = j + 2; c < 15; ++c) a[c] = __builtin_fma(-lj, sh.blk[16 * j + c], a[c]);
        }
    }
    if (bad && lane == 0) sh.flag = 0;
    if (lane < 15) {
        lds_double *wrow = Lp + tri(k0 + lane) + k0;
#pragma unroll
        for (int c = 0; c < 15; ++c)
            if (c <= lane) wrow[c] = a[c];
    }
}

// one lower 16 x 16 tile (bi, bj) of the trailing update C -= P P^T behind block column k0, on the matrix cores
DM void cholesky_trailing_tile(lds_double *Lp, int k0, int N, int bi, int bj) {  // N: rows of the packed triangle
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, kk = lane >> 4;
    const int ra = k0 + 15 + 16 * bi + i, rb = k0 + 15 + 16 * bj + i;
    const lds_double *pa = Lp + tri(ra < N ? ra : N - 1) + k0, *pb = Lp + tri(rb < N ? rb : N - 1) + k0;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = 4 * u + kk;
        const double av = (ra < N && q < 15) ? pa[q] : 0.0, bv = (rb < N && q < 15) ? pb[q] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    const int col = k0 + 15 + 16 * bj + (lane & 15);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int rowi = k0 + 15 + 16 * bi + (lane >> 4) + 4 * r4;
        if (rowi < N && col <= rowi) Lp[tri(rowi) + col] -= acc[r4];
    }
}

// Two barriers per block column, with one block of lookahead: while wavefronts 1.. apply the trailing update of block
// column k, wavefront 0 applies only the tile that holds the next diagonal block and factors it straight away (the
// 15 dependent pivots of a diagonal block are the longest serial piece of a column).
// Returns 0 when a pivot is not above tol (or not finite).
// The packed triangle carries one extra row, N, holding the right-hand side b: the panel and trailing steps treat it
// like any other row, and it leaves the factorisation as L^-1 b -- the forward substitution of the solve, for free.
template <int T>
__device__ __attribute__((noinline)) int cholesky_lds(LdsShared<T> &sh, lds_double *Lp, lds_double *Dinv, int N, double tol = 0.0, bool want_dinv = true) {
    const int t = threadIdx.x, wave = t >> 6, nw = T / 64;
    const int nb = N / 15, NR = N + 1;
#ifdef RDVIO_PROF_CHOL
    unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pc4 = 0, pt;
#define CH_T(acc) do { const unsigned long long n__ = wall_clock64(); acc += n__ - pt; pt = n__; } while (0)
    pt = wall_clock64();
#else
#define CH_T(acc) do {} while (0)
#endif
    if (t == 0) sh.flag = 1;
    __syncthreads();
    if (wave == 0 && nb > 0) cholesky_diag_block<T>(sh, Lp, 0, tol);
    CH_T(pc2);
    __syncthreads();
    CH_T(pc3);
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = 15 * kb;
        // ---- panel: row i solves x L_kk^T = row  (forward substitution with reciprocal pivots)
        for (int i = k0 + 15 + t; i < NR; i += T) {
            lds_double *row = Lp + tri(i) + k0;
            double x[15];
#pragma unroll
            for (int c = 0; c < 15; ++c) x[c] = row[c];
#pragma unroll
            for (int c = 0; c < 15; ++c) {
                const lds_double *Lc = Lp + tri(k0 + c) + k0;
                double s = x[c];
#pragma unroll
                for (int q = 0; q < 15; ++q)
                    if (q < c) s = __builtin_fma(-x[q], Lc[q], s);
                x[c] = s * sh.vec[c];
            }
#pragma unroll
            for (int c = 0; c < 15; ++c) row[c] = x[c];
        }
        CH_T(pc0);
        if (kb + 1 == nb) break;
        __syncthreads();
        CH_T(pc3);
        // ---- trailing update C -= P P^T, lower 16 x 16 tiles; tile (0, 0) and the next diagonal block on wavefront 0
        const int rem = NR - (k0 + 15), tn = (rem + 15) / 16;
        if (wave == 0) {
            cholesky_trailing_tile(Lp, k0, NR, 0, 0);
            CH_T(pc1);
            cholesky_diag_block<T>(sh, Lp, k0 + 15, tol);
            CH_T(pc2);
        } else {
            for (int tile = wave; tile < tn * tn; tile += nw - 1) {  // tiles 1.. (tile 0 is (0, 0))
                const int bi = tile / tn, bj = tile - bi * tn;
                if (bj > bi) continue;
                cholesky_trailing_tile(Lp, k0, NR, bi, bj);
            }
        }
        __syncthreads();
        CH_T(pc3);
    }
    __syncthreads();
    CH_T(pc3);
    // inverses of the diagonal blocks: column c of L_kk^-1 by forward substitution, one thread per (block, column)
    for (int o = t; want_dinv && o < nb * 15; o += T) {
        const int kb = o / 15, c = o - 15 * kb;
        double x[15];
#pragma unroll
        for (int r = 0; r < 15; ++r) {
            const lds_double *Lr = Lp + tri(15 * kb + r) + 15 * kb;
            double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < 15; ++q)
                if (q < r) s = __builtin_fma(-Lr[q], x[q], s);
            x[r] = (r >= c) ? s / Lr[r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 15; ++r) Dinv[225 * kb + 15 * r + c] = x[r];
    }
    __syncthreads();
    CH_T(pc4);
#ifdef RDVIO_PROF_CHOL
    if (t == 0) { rdvio_chol_prof[0] += pc0; rdvio_chol_prof[1] += pc1; rdvio_chol_prof[2] += pc2; rdvio_chol_prof[3] += pc3; rdvio_chol_prof[4] += pc4; }
#endif
    return sh.flag;
}

// solve L L^T y = b: L packed in LDS with row N = L^-1 b left there by cholesky_lds, Dinv = inverses of the 15 x 15
// diagonal blocks; only the backward substitution remains.  y goes to out.
template <int T>
__device__ __attribute__((noinline)) void cholesky_solve_lds(LdsShared<T> &sh, const lds_double *Lp, const lds_double *Dinv, int N, double *b) {
    const int t = threadIdx.x, nb = N / 15;
    double *y = RDVIO_GEN(sh.xv);  // N <= 512
    for (int i = t; i < N; i += T) y[i] = Lp[tri(N) + i];
    __syncthreads();
    // backward: y_k = Dinv_k^T y_k ; earlier rows -= L_ki^T y_k.  One wavefront runs the whole chain: its LDS operations
    // execute in program order, so the nb dependent steps need no workgroup barrier (three per step otherwise).
    if (t < 64) {
        for (int kb = nb - 1; kb >= 0; --kb) {
            double v = 0.0;
            if (t < 15) {
#pragma unroll
                for (int q = 0; q < 15; ++q) v = __builtin_fma(Dinv[225 * kb + 15 * q + t], y[15 * kb + q], v);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (t < 15) y[15 * kb + t] = v;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int i = t; i < 15 * kb; i += 64) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < 15; ++q) s = __builtin_fma(Lp[tri(15 * kb + q) + i], y[15 * kb + q], s);
                y[i] -= s;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (int i = t; i < N; i += T) b[i] = y[i];
    __syncthreads();
}
