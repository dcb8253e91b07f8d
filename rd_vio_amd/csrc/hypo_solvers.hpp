// Minimal-sample hypothesis solvers of RD-VIO's RANSAC / PARSAC loops, written ONCE for the host and for gfx950:
//
//   epnp6        solve_pnp_6pt            /root/reference/src/rdvio_geometry/include/rdvio/geometry/pnp.h:11-48
//                (cv::solvePnP(..., SOLVEPNP_EPNP) on float32 copies of six correspondences, identity camera matrix, the pose
//                round-tripped through a float32 Rodrigues vector; EPnP restated from Lepetit, Moreno-Noguer, Fua 2009:
//                control points, null space of M^T M, three beta initialisations + Gauss-Newton, Arun alignment, the
//                lowest reprojection error wins -- OpenCV is not in this image, PARITY UNPINNED)
//   essential5   solve_essential_5pt      /root/reference/src/rdvio_geometry/src/essential.cpp:8-299
//                (null-space basis, ten cubic constraints in GRevLex order, Gauss-Jordan to the action matrix of
//                multiplication by x, real eigenvectors -> essential matrices; Eigen::JacobiSVD / EigenSolver replaced by a
//                Jacobi eigensolver on A^T A and Hessenberg-QR + inverse iteration)
//   rotation2    solve_rotation_2pt       /root/reference/src/rdvio_geometry/include/rdvio/geometry/wahba.h:8-26
//
// The same source runs in two shapes.  Every algorithm is a sequence of STEPS; a step is either a set of independent work
// items (`each`: every item writes locations no other item of the step touches) or a single-lane section (`one`).  The host
// instantiates the steps with SerialExec (plain loops: the host road of host/pipeline/parsac.hpp and geom.hpp, i.e. the CPU
// path), the device with WaveExec (one 64-lane workgroup per hypothesis, work items over the lanes, the scratch in LDS,
// parsac_kernels.hip / gate_kernels.hip).  An item computes the same expression in the same order in both shapes and both
// compilers run without FMA contraction and with IEEE division / square root, so host and device results are bit-identical
// -- which is what lets the GPU path and the CPU path of the pipeline keep identical inlier masks and feature indices.
// Transcendentals (the Rodrigues round trip needs atan, sin, cos) are the fdlibm kernels below, not the platform libm.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HYPO_HD __host__ __device__ inline
#define HYPO_UNROLL _Pragma("unroll")
#else
#define HYPO_HD inline
#define HYPO_UNROLL _Pragma("GCC unroll 16")
#endif
// (fixed trip counts + full unrolling keep the small arrays of the one-lane sections in registers on the device: a dynamically
// indexed private array lives in scratch memory, a round trip to HBM per access)

// diagnostic hook (scripts/microbench/hypo_bench.hip defines it to record a clock per stage); nothing in product builds
#ifndef HYPO_STAMP
#define HYPO_STAMP(k)
#endif

namespace hypo {

// ------------------------------------------------------------------------------------------------------------ executors
struct SerialExec {
    static constexpr bool lanes = false;  // items run one after the other: a value several items need is computed once
    template <class F>
    void each(int n, F f) const {
        for (int i = 0; i < n; ++i) f(i);
    }
    template <class F>
    void one(F f) const {
        f();
    }
};
#if defined(__HIPCC__)
struct WaveExec {  // one workgroup of 64 lanes; every array a step touches lives in LDS
    static constexpr bool lanes = true;   // items run side by side: recomputing a shared value per item costs nothing, a step does
    int lane;
    template <class F>
    __device__ void each(int n, F f) const {
        for (int i = lane; i < n; i += 64) f(i);
        __syncthreads();
    }
    template <class F>
    __device__ void one(F f) const {
        if (lane == 0) f();
        __syncthreads();
    }
};
#endif

// ------------------------------------------------------------------------------------------------------------ scalar math
HYPO_HD double dabs(double x) { return x < 0 ? -x : x; }
HYPO_HD double dmax(double a, double b) { return a > b ? a : b; }
HYPO_HD bool dfinite(double x) { return (x - x) == 0.0; }

/* The three kernels below follow fdlibm (k_sin.c, k_cos.c, s_atan.c):
 * Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.  Developed at SunPro, a Sun Microsystems, Inc. business.
 * Permission to use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved. */
HYPO_HD double ksin(double x) {  // |x| <= pi/4
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
HYPO_HD double kcos(double x) {  // |x| <= pi/4
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return 1.0 - (0.5 * z - z * r);
}
// sine and cosine of h in [0, pi/2 + small]
HYPO_HD void sincos_quadrant(double h, double &s, double &c) {
    const double pio4 = 7.85398163397448278999e-01, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    if (h <= pio4) {
        s = ksin(h);
        c = kcos(h);
    } else {
        const double y = (pio2_hi - h) + pio2_lo;
        s = kcos(y);
        c = ksin(y);
    }
}
HYPO_HD double atan_pos(double x) {  // x >= 0 (inf allowed)
    const double hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01, -1.11111104054623557880e-01,
                           9.09088713343650656196e-02,  -7.69187620504482999495e-02, 6.66107313738753120669e-02, -5.83357013379057348645e-02,
                           4.97687799461593236017e-02,  -3.65315727442169155270e-02, 1.62858201153657823623e-02};
    int id;
    if (x < 0.4375) {
        id = -1;
    } else if (x < 1.1875) {
        if (x < 0.6875) {
            id = 0;
            x = (2.0 * x - 1.0) / (2.0 + x);
        } else {
            id = 1;
            x = (x - 1.0) / (x + 1.0);
        }
    } else if (x < 2.4375) {
        id = 2;
        x = (x - 1.5) / (1.0 + 1.5 * x);
    } else {
        id = 3;
        x = -1.0 / x;
    }
    const double z = x * x, w = z * z;
    const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    return hi[id] - ((x * (s1 + s2) - lo[id]) - x);
}

// Jacobi rotation (c, s) that annihilates a_pq of the symmetric 2 x 2 block [[app, apq], [apq, aqq]]: tan = sgn(theta) /
// (|theta| + sqrt(theta^2 + 1)) with theta = (aqq - app) / (2 apq), written with h = |d| + sqrt(d^2 + 4 apq^2) so that it costs
// two square roots and one division in a row instead of five long operations
HYPO_HD void jacobi_cs(double app, double aqq, double apq, double &c, double &s) {
    c = 1.0;
    s = 0.0;
    if (apq == 0.0) return;
    const double d = aqq - app, a2 = 2.0 * apq;
    const double h = dabs(d) + sqrt(d * d + a2 * a2);
    const double inv = 1.0 / sqrt(h * h + a2 * a2);
    c = h * inv;
    s = (d >= 0 ? a2 : -a2) * inv;
}

// ------------------------------------------------------------------------------------------------------------ small dense pieces (one lane)
// cyclic Jacobi on a symmetric n x n matrix (row-major, destroyed), n <= 5; V columns = eigenvectors
template <int n>
HYPO_HD void jacobi_small(double *A, double *V, double *lam) {
    HYPO_UNROLL
    for (int i = 0; i < n * n; ++i) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    double prev = 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dg = 0;
        HYPO_UNROLL
        for (int i = 0; i < n; ++i) {
            HYPO_UNROLL
            for (int j = 0; j < n; ++j) {
                const double a = A[i * n + j];
                if (i == j) dg += a * a;
                else off += a * a;
            }
        }
        // converged, or stagnating at the round-off floor (a rank-deficient matrix never gets below it)
        if (off <= 1e-300 || off <= 1e-28 * dg || (sweep > 0 && off <= 1e-24 * dg && off >= 0.25 * prev)) break;
        prev = off;
        HYPO_UNROLL
        for (int p = 0; p < n - 1; ++p) {
            HYPO_UNROLL
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double c, s;
                jacobi_cs(A[p * n + p], A[q * n + q], apq, c, s);
                HYPO_UNROLL
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                HYPO_UNROLL
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                HYPO_UNROLL
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    HYPO_UNROLL
    for (int i = 0; i < n; ++i) lam[i] = A[i * n + i];
}

// indices 0..2 of lam in ascending order (stable), without a dynamically indexed write
HYPO_HD void ascending3(const double *lam, int *ord) {
    const bool b10 = lam[1] < lam[0], b20 = lam[2] < lam[0], b21 = lam[2] < lam[1];
    const int r0 = (b10 ? 1 : 0) + (b20 ? 1 : 0), r1 = (b10 ? 0 : 1) + (b21 ? 1 : 0);   // ranks of elements 0 and 1 (element 2 has the third)
    ord[0] = r0 == 0 ? 0 : (r1 == 0 ? 1 : 2);
    ord[1] = r0 == 1 ? 0 : (r1 == 1 ? 1 : 2);
    ord[2] = r0 == 2 ? 0 : (r1 == 2 ? 1 : 2);
}

// indices of lam[0..n) in ascending order, stable (insertion sort)
HYPO_HD void ascending(int n, const double *lam, int *ord) {
    for (int i = 0; i < n; ++i) {
        int j = i;
        while (j > 0 && lam[i] < lam[ord[j - 1]]) {
            ord[j] = ord[j - 1];
            --j;
        }
        ord[j] = i;
    }
}

HYPO_HD double det3(const double *A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
HYPO_HD void inverse3(const double *A, double *I) {
    const double d = det3(A);
    I[0] = (A[4] * A[8] - A[5] * A[7]) / d; I[1] = (A[2] * A[7] - A[1] * A[8]) / d; I[2] = (A[1] * A[5] - A[2] * A[4]) / d;
    I[3] = (A[5] * A[6] - A[3] * A[8]) / d; I[4] = (A[0] * A[8] - A[2] * A[6]) / d; I[5] = (A[2] * A[3] - A[0] * A[5]) / d;
    I[6] = (A[3] * A[7] - A[4] * A[6]) / d; I[7] = (A[1] * A[6] - A[0] * A[7]) / d; I[8] = (A[0] * A[4] - A[1] * A[3]) / d;
}
HYPO_HD void cross3(const double *a, const double *b, double *c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
HYPO_HD double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
HYPO_HD void matvec3(const double *A, const double *v, double *o) {
    for (int r = 0; r < 3; ++r) o[r] = A[3 * r] * v[0] + A[3 * r + 1] * v[1] + A[3 * r + 2] * v[2];
}
HYPO_HD void normalize3(double *v) {
    const double n = sqrt(dot3(v, v));
    v[0] /= n; v[1] /= n; v[2] /= n;
}

// SVD of a 3 x 3 matrix A = U diag(s) V^T (row-major), s descending; U, V orthogonal (completed by cross products when A is
// rank deficient): eigen-decomposition of A^T A, then u_c = A v_c / s_c with Gram-Schmidt
HYPO_HD void svd3(const double *A, double *U, double *s, double *V) {
    double AtA[9], Vv[9], lam[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) AtA[3 * i + j] = A[i] * A[j] + A[3 + i] * A[3 + j] + A[6 + i] * A[6 + j];
    jacobi_small<3>(AtA, Vv, lam);
    int asc[3], ord[3];
    ascending3(lam, asc);
    ord[0] = asc[2]; ord[1] = asc[1]; ord[2] = asc[0];
    double v[3][3], u[3][3];
    for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < 3; ++r) v[c][r] = Vv[3 * r + ord[c]];
        s[c] = sqrt(dmax(lam[ord[c]], 0.0));
    }
    cross3(v[0], v[1], v[2]);  // right-handed completion (also fixes a degenerate third vector)
    const double tol = 1e-12 * dmax(s[0], 1e-300);
    if (s[0] > tol) {
        matvec3(A, v[0], u[0]);
        for (int r = 0; r < 3; ++r) u[0][r] /= s[0];
    } else {
        u[0][0] = 1; u[0][1] = 0; u[0][2] = 0;
    }
    if (s[1] > tol) {
        matvec3(A, v[1], u[1]);
        for (int r = 0; r < 3; ++r) u[1][r] /= s[1];
        const double d = dot3(u[1], u[0]);
        for (int r = 0; r < 3; ++r) u[1][r] -= d * u[0][r];
        normalize3(u[1]);
    } else {
        const double a[3] = {dabs(u[0][0]) < 0.9 ? 1.0 : 0.0, dabs(u[0][0]) < 0.9 ? 0.0 : 1.0, 0.0};
        cross3(u[0], a, u[1]);
        normalize3(u[1]);
    }
    if (s[2] > tol) {
        matvec3(A, v[2], u[2]);
        for (int r = 0; r < 3; ++r) u[2][r] /= s[2];
        const double d0 = dot3(u[2], u[0]);
        for (int r = 0; r < 3; ++r) u[2][r] -= d0 * u[0][r];
        const double d1 = dot3(u[2], u[1]);
        for (int r = 0; r < 3; ++r) u[2][r] -= d1 * u[1][r];
        normalize3(u[2]);
    } else {
        cross3(u[0], u[1], u[2]);
    }
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            U[3 * r + c] = u[c][r];
            V[3 * r + c] = v[c][r];
        }
}

// least squares x = argmin |A x - b| (ROWS x NC row-major, NC <= 5) through the normal equations, Gaussian elimination with
// row exchanges that bring the larger pivot up (compare-and-exchange: no dynamically indexed row); a singular system yields
// non-finite values (the callers test for them)
template <int ROWS, int NC>
HYPO_HD void ls_solve(const double *A, const double *b, double *x) {
    double G[NC][NC + 1];
    HYPO_UNROLL
    for (int i = 0; i < NC; ++i) {
        HYPO_UNROLL
        for (int j = 0; j < NC; ++j) {
            double s = 0;
            HYPO_UNROLL
            for (int r = 0; r < ROWS; ++r) s += A[r * NC + i] * A[r * NC + j];
            G[i][j] = s;
        }
        double s = 0;
        HYPO_UNROLL
        for (int r = 0; r < ROWS; ++r) s += A[r * NC + i] * b[r];
        G[i][NC] = s;
    }
    HYPO_UNROLL
    for (int c = 0; c < NC; ++c) {
        HYPO_UNROLL
        for (int r = c + 1; r < NC; ++r) {
            const bool up = dabs(G[r][c]) > dabs(G[c][c]);
            HYPO_UNROLL
            for (int j = 0; j <= NC; ++j) {
                const double gc = G[c][j], gr = G[r][j];
                G[c][j] = up ? gr : gc;
                G[r][j] = up ? gc : gr;
            }
        }
        HYPO_UNROLL
        for (int r = c + 1; r < NC; ++r) {
            const double f = G[r][c] / G[c][c];
            HYPO_UNROLL
            for (int j = c; j <= NC; ++j) G[r][j] -= f * G[c][j];
        }
    }
    HYPO_UNROLL
    for (int c = NC - 1; c >= 0; --c) {
        double s = G[c][NC];
        HYPO_UNROLL
        for (int j = c + 1; j < NC; ++j) s -= G[c][j] * x[j];
        x[c] = s / G[c][c];
    }
}

// quaternion (x, y, z, w) of a rotation matrix (Eigen's Quaternion(Matrix3) recipe)
HYPO_HD void quat_from_mat(const double *R, double *q) {
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        double s = sqrt(t + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
}
HYPO_HD void mat_from_quat(const double *q, double *R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// pnp.h:38-46: rotation -> Rodrigues vector -> float32 -> rotation -> float32 entries
HYPO_HD void rodrigues_float_round_trip(const double *Rin, double *Rout) {
    double q[4], rv[3] = {0, 0, 0};
    quat_from_mat(Rin, q);
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n != 0.0) {  // lie_algebra.h:18-21 (Eigen::AngleAxis): angle = 2 atan2(|v|, |w|)
        const double angle = 2.0 * atan_pos(n / dabs(q[3]));
        if (q[3] < 0) n = -n;
        for (int k = 0; k < 3; ++k) rv[k] = angle * q[k] / n;
    }
    for (int k = 0; k < 3; ++k) rv[k] = (double)(float)rv[k];
    const double th = sqrt(dot3(rv, rv));
    double qo[4] = {0, 0, 0, 1};
    if (th != 0.0) {
        double s, c;
        sincos_quadrant(0.5 * th, s, c);
        s = s / th;
        qo[0] = s * rv[0]; qo[1] = s * rv[1]; qo[2] = s * rv[2]; qo[3] = c;
    }
    mat_from_quat(qo, Rout);
    for (int k = 0; k < 9; ++k) Rout[k] = (double)(float)Rout[k];
}

// wahba.h:8-26 (solve_rotation_2pt): the rotation R with p2_k ~ R p1_k for two bearing pairs -- Kabsch on the 3 x 3 correlation,
// R = V diag(1, 1, det(V U^T)) U^T; one lane
HYPO_HD void rotation2(const double *p1 /* 2 x 3 */, const double *p2 /* 2 x 3 */, double *R /* 9 */) {
    double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, U[9], V[9], s[3], VUt[9];
    HYPO_UNROLL
    for (int k = 0; k < 2; ++k) {
        HYPO_UNROLL
        for (int i = 0; i < 3; ++i) {
            HYPO_UNROLL
            for (int j = 0; j < 3; ++j) cov[3 * i + j] += p1[3 * k + i] * p2[3 * k + j];
        }
    }
    HYPO_UNROLL
    for (int i = 0; i < 9; ++i) cov[i] *= 0.5;
    svd3(cov, U, s, V);
    HYPO_UNROLL
    for (int r = 0; r < 3; ++r) {
        HYPO_UNROLL
        for (int c = 0; c < 3; ++c) VUt[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
    }
    const double e = det3(VUt) >= 0.0 ? 1.0 : -1.0;
    HYPO_UNROLL
    for (int r = 0; r < 3; ++r) {
        HYPO_UNROLL
        for (int c = 0; c < 3; ++c) R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + (V[3 * r + 2] * e) * U[3 * c + 2];
    }
}

// the inlier test of find_rotation_matrix (stereo.cpp:82-84): acos((R p1) . p2) <= threshold, decided without the arc cosine:
// acos(d) is defined for |d| <= 1 (NaN beyond: not an inlier) and decreasing, so the test is cos(threshold) <= d <= 1.  One
// predicate for host and device (cos_threshold comes from the host's cos()): both roads decide every point alike.
HYPO_HD bool rotation_inlier(const double *R, const double *p1, const double *p2, double cos_threshold) {
    double q[3];
    matvec3(R, p1, q);
    const double d = dot3(q, p2);
    return d <= 1.0 && d >= cos_threshold;
}

// ------------------------------------------------------------------------------------------------------------ wave-parallel Jacobi
// Round-robin ("chess tournament") ordering: n even, round r in [0, n - 1), pair k in [0, n / 2) -> (p < q), every pair once a sweep
template <int n>
HYPO_HD void rr_pair(int r, int k, int &p, int &q) {
    constexpr int m = n - 1;
    const int a = k == 0 ? m : (r + k) % m, b = k == 0 ? r : (r - k + m) % m;
    p = a < b ? a : b;
    q = a < b ? b : a;
}

// Eigen-decomposition of a symmetric n x n matrix (row-major, destroyed; n <= 12, a template parameter so that the index
// arithmetic of the work items is by constants) by Jacobi rotations in round-robin order: the n/2 rotations of a round act on
// disjoint index pairs, so a round is two steps of independent items --
//   rows:    B = J^T A, one item per (pair, column); the item computes its pair's rotation itself (from A, which this step only
//            reads) -- the same value on every item of the pair -- and the column-0 item files it for the next step;
//   columns: A = B J and V = V J, one item per (pair, row).
// Convergence: off-diagonal mass <= 1e-28 of the diagonal's (norm ratio 1e-14; the sweep that gets there squares it once more).
// B: n x n scratch, cs: 2 * 6 doubles, red: 2 * 12 doubles, flag: 1 int of scratch.
template <int n, class X>
HYPO_HD void jacobi_rr(const X &x, double *A, double *B, double *V, double *lam, double *cs, double *red, int *flag) {
    constexpr int ne = (n + 1) & ~1, half = ne / 2, rounds = ne - 1, cstep = (n + 1) / 2;
    x.each(n * n, [=](int i) { V[i] = (i / n == i % n) ? 1.0 : 0.0; });
    for (int sweep = 0; sweep < 40; ++sweep) {
        x.each(n, [=](int j) {   // column sums of squares, then their total in column order
            double off = 0, dg = 0;
            for (int i = 0; i < n; ++i) {
                const double a = A[i * n + j];
                if (i == j) dg += a * a;
                else off += a * a;
            }
            red[2 * j] = off;
            red[2 * j + 1] = dg;
        });
        x.one([=]() {
            double off = 0, dg = 0;
            for (int j = 0; j < n; ++j) {
                off += red[2 * j];
                dg += red[2 * j + 1];
            }
            // converged, or stagnating at the round-off floor (a rank-deficient matrix never gets below it); lam[0] carries
            // the previous sweep's off-diagonal mass until the eigenvalues are written
            *flag = (off <= 1e-300 || off <= 1e-28 * dg || (sweep > 0 && off <= 1e-24 * dg && off >= 0.25 * lam[0])) ? 1 : 0;
            lam[0] = off;
        });
        if (*flag) break;
        for (int r = 0; r < rounds; ++r) {
            if constexpr (!X::lanes)   // (serial executor: the rotations once per pair instead of once per item -- same values)
                x.each(half, [=](int k) {
                    int p, q;
                    rr_pair<ne>(r, k, p, q);
                    if (q < n) jacobi_cs(A[p * n + p], A[q * n + q], A[p * n + q], cs[2 * k], cs[2 * k + 1]);
                });
            // an item carries two columns (rows) so that a 64-lane wavefront covers a step in ONE pass, with independent
            // rotations in flight per lane
            x.each(half * cstep, [=](int idx) {   // B = J^T A: item = (pair, columns jj and jj + cstep)
                const int k = idx / cstep, jj = idx % cstep;
                int p, q;
                rr_pair<ne>(r, k, p, q);
                if (q >= n) {   // the pair of the padding index (odd n): row p passes through
                    B[p * n + jj] = A[p * n + jj];
                    if (jj + cstep < n) B[p * n + jj + cstep] = A[p * n + jj + cstep];
                    return;
                }
                double c, s;
                if constexpr (X::lanes) {
                    jacobi_cs(A[p * n + p], A[q * n + q], A[p * n + q], c, s);
                    if (jj == 0) {
                        cs[2 * k] = c;
                        cs[2 * k + 1] = s;
                    }
                } else {
                    c = cs[2 * k];
                    s = cs[2 * k + 1];
                }
                const double ap0 = A[p * n + jj], aq0 = A[q * n + jj];
                B[p * n + jj] = c * ap0 - s * aq0;
                B[q * n + jj] = s * ap0 + c * aq0;
                if (jj + cstep < n) {
                    const double ap1 = A[p * n + jj + cstep], aq1 = A[q * n + jj + cstep];
                    B[p * n + jj + cstep] = c * ap1 - s * aq1;
                    B[q * n + jj + cstep] = s * ap1 + c * aq1;
                }
            });
            x.each(half * cstep, [=](int idx) {   // A = B J and V = V J: item = (pair, rows ii and ii + cstep)
                const int k = idx / cstep, ii = idx % cstep;
                int p, q;
                rr_pair<ne>(r, k, p, q);
                if (q >= n) {
                    A[ii * n + p] = B[ii * n + p];
                    if (ii + cstep < n) A[(ii + cstep) * n + p] = B[(ii + cstep) * n + p];
                    return;
                }
                const double c = cs[2 * k], s = cs[2 * k + 1];
                {
                    const double bp = B[ii * n + p], bq = B[ii * n + q], vp = V[ii * n + p], vq = V[ii * n + q];
                    A[ii * n + p] = c * bp - s * bq;
                    A[ii * n + q] = s * bp + c * bq;
                    V[ii * n + p] = c * vp - s * vq;
                    V[ii * n + q] = s * vp + c * vq;
                }
                if (ii + cstep < n) {
                    const int i1 = ii + cstep;
                    const double bp = B[i1 * n + p], bq = B[i1 * n + q], vp = V[i1 * n + p], vq = V[i1 * n + q];
                    A[i1 * n + p] = c * bp - s * bq;
                    A[i1 * n + q] = s * bp + c * bq;
                    V[i1 * n + p] = c * vp - s * vq;
                    V[i1 * n + q] = s * vp + c * vq;
                }
            });
        }
    }
    x.each(n, [=](int i) { lam[i] = A[i * n + i]; });
}

// ------------------------------------------------------------------------------------------------------------ EPnP, six points
struct EpnpWork {  // scratch of one hypothesis (LDS on the device)
    double pw[6][3], us[6][2];  // float32-rounded copies of the sample
    double cws[4][3], alphas[6][4];
    double MtM[144], Vv[144], Bw[144], lam[12], cs[12], red[24];
    double v[4][12], L[6][10], rho[6];
    double cand[3][13];  // per beta initialisation: R (9), t (3), mean reprojection error
    int ord[12], flag, ok;
};

// pose [R | t] (12 doubles: R row-major, then t) from six 3-D points X (world) and their normalised image points u.  A
// degenerate sample ends in the identity pose (cv::solvePnP would leave an unusable pose; the inlier test rejects either).
template <class X>
HYPO_HD void epnp6(const X &x, EpnpWork *w, const double *Xs /* 6 x 3 */, const double *xs /* 6 x 2 */, double *model /* 12 */) {
    HYPO_STAMP(0);
    x.each(6, [=](int i) {
        for (int k = 0; k < 3; ++k) w->pw[i][k] = (double)(float)Xs[3 * i + k];
        for (int k = 0; k < 2; ++k) w->us[i][k] = (double)(float)xs[2 * i + k];
    });
    // control points: centroid + principal directions of the sample; barycentric coordinates
    x.one([=]() {
        double c0[3] = {0, 0, 0};
        for (int i = 0; i < 6; ++i)
            for (int k = 0; k < 3; ++k) c0[k] = c0[k] + w->pw[i][k];
        for (int k = 0; k < 3; ++k) c0[k] = c0[k] / 6.0;
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Ve[9], le[3];
        for (int i = 0; i < 6; ++i) {
            const double d[3] = {w->pw[i][0] - c0[0], w->pw[i][1] - c0[1], w->pw[i][2] - c0[2]};
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) C[3 * a + b] += d[a] * d[b];
        }
        jacobi_small<3>(C, Ve, le);
        int asc[3];
        ascending3(le, asc);
        for (int k = 0; k < 3; ++k) w->cws[0][k] = c0[k];
        for (int i = 1; i < 4; ++i) {
            const int c = asc[3 - i];  // descending eigenvalues
            const double kk = sqrt(dmax(le[c], 0.0) / 6.0);
            for (int k = 0; k < 3; ++k) w->cws[i][k] = c0[k] + kk * Ve[3 * k + c];
        }
        double CC[9], Ci[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 1; j < 4; ++j) CC[3 * i + j - 1] = w->cws[j][i] - w->cws[0][i];
        w->ok = dabs(det3(CC)) > 0.0 ? 1 : 0;
        if (w->ok) {
            inverse3(CC, Ci);
            for (int i = 0; i < 6; ++i) {
                const double d[3] = {w->pw[i][0] - c0[0], w->pw[i][1] - c0[1], w->pw[i][2] - c0[2]};
                double a[3];
                matvec3(Ci, d, a);
                w->alphas[i][1] = a[0]; w->alphas[i][2] = a[1]; w->alphas[i][3] = a[2];
                w->alphas[i][0] = 1.0 - a[0] - a[1] - a[2];
            }
        }
    });
    HYPO_STAMP(1);
    if (w->ok) {
        // M^T M of the 12 x 12 system (two rows per point)
        x.each(144, [=](int e) {
            const int p = e / 12, q = e % 12;
            double s = 0.0;
            for (int i = 0; i < 6; ++i) {
                const double ap = w->alphas[i][p / 3], aq = w->alphas[i][q / 3];
                const double r1p = p % 3 == 0 ? ap : (p % 3 == 1 ? 0.0 : ap * (0.0 - w->us[i][0]));
                const double r1q = q % 3 == 0 ? aq : (q % 3 == 1 ? 0.0 : aq * (0.0 - w->us[i][0]));
                const double r2p = p % 3 == 0 ? 0.0 : (p % 3 == 1 ? ap : ap * (0.0 - w->us[i][1]));
                const double r2q = q % 3 == 0 ? 0.0 : (q % 3 == 1 ? aq : aq * (0.0 - w->us[i][1]));
                s += r1p * r1q + r2p * r2q;
            }
            w->MtM[e] = s;
        });
        HYPO_STAMP(2);
        jacobi_rr<12>(x, w->MtM, w->Bw, w->Vv, w->lam, w->cs, w->red, &w->flag);
        HYPO_STAMP(3);
        x.one([=]() { ascending(12, w->lam, w->ord); });
        x.each(48, [=](int e) {  // v[0] = the smallest eigenvalue's vector ... v[3] = the fourth smallest
            const int k = e / 12, i = e % 12;
            w->v[k][i] = w->Vv[12 * i + w->ord[k]];
        });
        x.each(66, [=](int e) {  // the 6 x 10 matrix L of the distance constraints and their right-hand side rho
            const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
            if (e >= 60) {
                const int j = e - 60;
                double d2 = 0;
                for (int k = 0; k < 3; ++k) {
                    const double d = w->cws[pa[j]][k] - w->cws[pb[j]][k];
                    d2 += d * d;
                }
                w->rho[j] = d2;
                return;
            }
            const int j = e / 10, col = e % 10;
            const int ca[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3}, cb[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
            const int a = ca[col], b = cb[col];
            double d = 0;
            for (int k = 0; k < 3; ++k) d += (w->v[a][3 * pa[j] + k] - w->v[a][3 * pb[j] + k]) * (w->v[b][3 * pa[j] + k] - w->v[b][3 * pb[j] + k]);
            w->L[j][col] = a == b ? d : 2 * d;
        });
        HYPO_STAMP(4);
        // the three beta initialisations, each refined by Gauss-Newton and turned into a pose (one item each)
        x.each(3, [=](int c) {
            double b[4] = {0, 0, 0, 0};
            if (c == 0) {  // N = 4: [B11 B12 B13 B14] from columns 0 1 3 6 of L
                double A[24], sol[4];
                HYPO_UNROLL
                for (int j = 0; j < 6; ++j) {
                    A[4 * j] = w->L[j][0]; A[4 * j + 1] = w->L[j][1]; A[4 * j + 2] = w->L[j][3]; A[4 * j + 3] = w->L[j][6];
                }
                ls_solve<6, 4>(A, w->rho, sol);
                if (sol[0] < 0) { b[0] = sqrt(-sol[0]); b[1] = -sol[1] / b[0]; b[2] = -sol[2] / b[0]; b[3] = -sol[3] / b[0]; }
                else { b[0] = sqrt(sol[0]); b[1] = sol[1] / b[0]; b[2] = sol[2] / b[0]; b[3] = sol[3] / b[0]; }
            } else {       // N = 2: [B11 B12 B22] from columns 0 1 2; N = 3: [B11 B12 B22 B13 B23] from columns 0..4
                double sol[5] = {0, 0, 0, 0, 0};
                if (c == 1) {
                    double A[18];
                    HYPO_UNROLL
                    for (int j = 0; j < 6; ++j) {
                        A[3 * j] = w->L[j][0]; A[3 * j + 1] = w->L[j][1]; A[3 * j + 2] = w->L[j][2];
                    }
                    ls_solve<6, 3>(A, w->rho, sol);
                } else {
                    double A[30];
                    HYPO_UNROLL
                    for (int j = 0; j < 6; ++j) {
                        HYPO_UNROLL
                        for (int k = 0; k < 5; ++k) A[5 * j + k] = w->L[j][k];
                    }
                    ls_solve<6, 5>(A, w->rho, sol);
                }
                if (sol[0] < 0) { b[0] = sqrt(-sol[0]); b[1] = (sol[2] < 0) ? sqrt(-sol[2]) : 0.0; }
                else { b[0] = sqrt(sol[0]); b[1] = (sol[2] > 0) ? sqrt(sol[2]) : 0.0; }
                if (sol[1] < 0) b[0] = -b[0];
                b[2] = c == 2 ? sol[3] / b[0] : 0.0;
                b[3] = 0.0;
            }
            double *out = w->cand[c];
            out[12] = -1.0;  // unusable until proven otherwise
            if (!(dfinite(b[0]) && dfinite(b[1]) && dfinite(b[2]) && dfinite(b[3]))) return;
            for (int it = 0; it < 5; ++it) {  // Gauss-Newton on the six distance constraints
                double A[24], r[6], dx[4];
                HYPO_UNROLL
                for (int i = 0; i < 6; ++i) {
                    const double *l = w->L[i];
                    A[4 * i + 0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                    A[4 * i + 1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                    A[4 * i + 2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
                    A[4 * i + 3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
                    r[i] = w->rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                                        l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
                }
                ls_solve<6, 4>(A, r, dx);
                HYPO_UNROLL
                for (int k = 0; k < 4; ++k) b[k] += dx[k];
            }
            // camera-frame points from the betas, sign fix, Arun alignment, mean reprojection error
            double ccs[4][3], pc[6][3];
            HYPO_UNROLL
            for (int i = 0; i < 4; ++i) {
                HYPO_UNROLL
                for (int k = 0; k < 3; ++k) {
                    double s = 0.0;
                    HYPO_UNROLL
                    for (int m = 0; m < 4; ++m) s = s + b[m] * w->v[m][3 * i + k];
                    ccs[i][k] = s;
                }
            }
            HYPO_UNROLL
            for (int i = 0; i < 6; ++i) {
                HYPO_UNROLL
                for (int k = 0; k < 3; ++k) {
                    double s = 0.0;
                    HYPO_UNROLL
                    for (int j = 0; j < 4; ++j) s = s + w->alphas[i][j] * ccs[j][k];
                    pc[i][k] = s;
                }
            }
            const bool flip = pc[0][2] < 0.0;
            HYPO_UNROLL
            for (int i = 0; i < 6; ++i) {
                HYPO_UNROLL
                for (int k = 0; k < 3; ++k) pc[i][k] = flip ? -pc[i][k] : pc[i][k];
            }
            double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
            HYPO_UNROLL
            for (int i = 0; i < 6; ++i) {
                HYPO_UNROLL
                for (int k = 0; k < 3; ++k) {
                    pc0[k] = pc0[k] + pc[i][k];
                    pw0[k] = pw0[k] + w->pw[i][k];
                }
            }
            HYPO_UNROLL
            for (int k = 0; k < 3; ++k) {
                pc0[k] = pc0[k] / 6.0;
                pw0[k] = pw0[k] / 6.0;
            }
            double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            HYPO_UNROLL
            for (int i = 0; i < 6; ++i) {
                HYPO_UNROLL
                for (int r = 0; r < 3; ++r) {
                    HYPO_UNROLL
                    for (int q = 0; q < 3; ++q) ABt[3 * r + q] += (pc[i][r] - pc0[r]) * (w->pw[i][q] - pw0[q]);
                }
            }
            double U[9], V[9], sv[3], R[9];
            svd3(ABt, U, sv, V);
            HYPO_UNROLL
            for (int r = 0; r < 3; ++r) {
                HYPO_UNROLL
                for (int q = 0; q < 3; ++q) R[3 * r + q] = U[3 * r] * V[3 * q] + U[3 * r + 1] * V[3 * q + 1] + U[3 * r + 2] * V[3 * q + 2];
            }
            if (det3(R) < 0) {
                R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8];
            }
            double Rp[3];
            matvec3(R, pw0, Rp);
            const double t[3] = {pc0[0] - Rp[0], pc0[1] - Rp[1], pc0[2] - Rp[2]};
            double err = 0.0;
            for (int i = 0; i < 6; ++i) {
                double q[3];
                matvec3(R, w->pw[i], q);
                for (int k = 0; k < 3; ++k) q[k] = q[k] + t[k];
                const double du = w->us[i][0] - q[0] / q[2], dv = w->us[i][1] - q[1] / q[2];
                err += sqrt(du * du + dv * dv);
            }
            for (int k = 0; k < 9; ++k) out[k] = R[k];
            for (int k = 0; k < 3; ++k) out[9 + k] = t[k];
            const double mean = err / 6.0;
            out[12] = dfinite(mean) ? mean : -1.0;
        });
    }
    HYPO_STAMP(5);
    x.one([=]() {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
        if (w->ok) {
            double best = 1.79769313486231570815e+308;
            for (int c = 0; c < 3; ++c) {
                const double e = w->cand[c][12];
                if (e >= 0.0 && e < best) {
                    best = e;
                    for (int k = 0; k < 9; ++k) R[k] = w->cand[c][k];
                    for (int k = 0; k < 3; ++k) t[k] = w->cand[c][9 + k];
                }
            }
        }
        rodrigues_float_round_trip(R, model);
        for (int k = 0; k < 3; ++k) model[9 + k] = (double)(float)t[k];
    });
    HYPO_STAMP(6);
}

// ------------------------------------------------------------------------------------------------------------ five-point essential
enum Mono { XXX = 0, XXY, XYY, YYY, XXZ, XYZ, YYZ, XZZ, YZZ, ZZZ, XX, XY, YY, XZ, YZ, ZZ, MX, MY, MZ, MI };

HYPO_HD void poly_zero(double *r) {
    for (int i = 0; i < 20; ++i) r[i] = 0.0;
}
HYPO_HD void poly_add(const double *a, const double *b, double *r) {
    for (int i = 0; i < 20; ++i) r[i] = a[i] + b[i];
}
HYPO_HD void poly_sub(const double *a, const double *b, double *r) {
    for (int i = 0; i < 20; ++i) r[i] = a[i] - b[i];
}
// product truncated at degree 3, term by term as essential.cpp:50-104 (r must not alias a or c)
HYPO_HD void poly_mul(const double *a, const double *c, double *r) {
    r[MI] = a[MI] * c[MI];
    r[MZ] = a[MI] * c[MZ] + a[MZ] * c[MI];
    r[MY] = a[MI] * c[MY] + a[MY] * c[MI];
    r[MX] = a[MI] * c[MX] + a[MX] * c[MI];
    r[ZZ] = a[MI] * c[ZZ] + a[MZ] * c[MZ] + a[ZZ] * c[MI];
    r[YZ] = a[MI] * c[YZ] + a[MZ] * c[MY] + a[MY] * c[MZ] + a[YZ] * c[MI];
    r[XZ] = a[MI] * c[XZ] + a[MZ] * c[MX] + a[MX] * c[MZ] + a[XZ] * c[MI];
    r[YY] = a[MI] * c[YY] + a[MY] * c[MY] + a[YY] * c[MI];
    r[XY] = a[MI] * c[XY] + a[MY] * c[MX] + a[MX] * c[MY] + a[XY] * c[MI];
    r[XX] = a[MI] * c[XX] + a[MX] * c[MX] + a[XX] * c[MI];
    r[ZZZ] = a[MI] * c[ZZZ] + a[MZ] * c[ZZ] + a[ZZ] * c[MZ] + a[ZZZ] * c[MI];
    r[YZZ] = a[MI] * c[YZZ] + a[MZ] * c[YZ] + a[MY] * c[ZZ] + a[ZZ] * c[MY] + a[YZ] * c[MZ] + a[YZZ] * c[MI];
    r[XZZ] = a[MI] * c[XZZ] + a[MZ] * c[XZ] + a[MX] * c[ZZ] + a[ZZ] * c[MX] + a[XZ] * c[MZ] + a[XZZ] * c[MI];
    r[YYZ] = a[MI] * c[YYZ] + a[MZ] * c[YY] + a[MY] * c[YZ] + a[YZ] * c[MY] + a[YY] * c[MZ] + a[YYZ] * c[MI];
    r[XYZ] = a[MI] * c[XYZ] + a[MZ] * c[XY] + a[MY] * c[XZ] + a[MX] * c[YZ] + a[YZ] * c[MX] + a[XZ] * c[MY] + a[XY] * c[MZ] + a[XYZ] * c[MI];
    r[XXZ] = a[MI] * c[XXZ] + a[MZ] * c[XX] + a[MX] * c[XZ] + a[XZ] * c[MX] + a[XX] * c[MZ] + a[XXZ] * c[MI];
    r[YYY] = a[MI] * c[YYY] + a[MY] * c[YY] + a[YY] * c[MY] + a[YYY] * c[MI];
    r[XYY] = a[MI] * c[XYY] + a[MY] * c[XY] + a[MX] * c[YY] + a[YY] * c[MX] + a[XY] * c[MY] + a[XYY] * c[MI];
    r[XXY] = a[MI] * c[XXY] + a[MY] * c[XX] + a[MX] * c[XY] + a[XY] * c[MX] + a[XX] * c[MY] + a[XXY] * c[MI];
    r[XXX] = a[MI] * c[XXX] + a[MX] * c[XX] + a[XX] * c[MX] + a[XXX] * c[MI];
}

// Eigenvalues of a general real 10 x 10 matrix (row-major, destroyed): elimination to Hessenberg form + QR with implicit double
// shifts (the classic EISPACK elmhes / hqr recipe), laid out in steps: every decision (pivot, deflation, shift, the reflector of a
// QR step) is taken in a one-lane section and filed in the shared state, the row and column updates it implies are items over the
// columns / rows.  ok = 0: no convergence.
struct Hqr10 {
    double a[100], wr[10], wi[10], mult[10];
    double t, p, q, r, x, y, z, w, anorm;
    int nn, l, m, k, its, mode, apply, ok, piv;
};

template <class X>
HYPO_HD void real_eigenvalues10(const X &xq, Hqr10 *h) {
    constexpr int n = 10;
#define HA(i, j) h->a[(i) * n + (j)]
    // ---- elmhes: for every column m - 1 the pivot row comes up, the multipliers y_i = a(i, m-1) / pivot are filed, then
    // A <- L^-1 A (rows i > m, an item per entry) and A <- A L (column m, an item per row), L = I + sum_i y_i e_i e_m^T
    for (int m = 1; m < n - 1; ++m) {
        xq.one([=]() {
            double x = 0.0;
            int i = m;
            for (int j = m; j < n; ++j)
                if (dabs(HA(j, m - 1)) > dabs(x)) {
                    x = HA(j, m - 1);
                    i = j;
                }
            if (i != m) {
                for (int j = m - 1; j < n; ++j) {
                    const double t = HA(i, j);
                    HA(i, j) = HA(m, j);
                    HA(m, j) = t;
                }
                for (int j = 0; j < n; ++j) {
                    const double t = HA(j, i);
                    HA(j, i) = HA(j, m);
                    HA(j, m) = t;
                }
            }
            for (int r = m + 1; r < n; ++r) {
                double y = 0.0;
                if (x != 0.0 && HA(r, m - 1) != 0.0) {
                    y = HA(r, m - 1) / x;
                    HA(r, m - 1) = y;
                }
                h->mult[r] = y;
            }
        });
        xq.each((n - m - 1) * (n - m), [=](int e) {   // rows i = m+1 .. n-1, columns j = m .. n-1
            const int i = m + 1 + e / (n - m), j = m + e % (n - m);
            HA(i, j) -= h->mult[i] * HA(m, j);
        });
        xq.each(n, [=](int j) {   // column m of every row
            double s = HA(j, m);
            for (int i = m + 1; i < n; ++i) s += h->mult[i] * HA(j, i);
            HA(j, m) = s;
        });
    }
    xq.one([=]() {
        for (int i = 2; i < n; ++i)
            for (int j = 0; j < i - 1; ++j) HA(i, j) = 0.0;
        for (int i = 0; i < n; ++i) h->wr[i] = h->wi[i] = 0.0;
        double anorm = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = (i - 1 > 0 ? i - 1 : 0); j < n; ++j) anorm += dabs(HA(i, j));
        h->anorm = anorm;
        h->nn = n - 1;
        h->t = 0.0;
        h->its = 0;
        h->ok = 1;
        h->mode = 0;
    });
    // ---- hqr.  mode after the deflation check: 1 one root found, 2 two roots found, 3 a QR step over k = m .. nn-1, 4 failure.
    // `its` restarts whenever the active block shrinks below the last deflation point (the do-while of the classic code).
    for (int guard = 0; guard < 2000; ++guard) {
        if (h->nn < 0 || !h->ok) break;
        xq.one([=]() {
            int nn = h->nn, l;
            double s, x, y, w, p, q, z, r, u, v;
            for (l = nn; l >= 1; --l) {
                s = dabs(HA(l - 1, l - 1)) + dabs(HA(l, l));
                if (s == 0.0) s = h->anorm;
                if (dabs(HA(l, l - 1)) + s == s) {
                    HA(l, l - 1) = 0.0;
                    break;
                }
            }
            h->l = l;
            x = HA(nn, nn);
            if (l == nn) {
                h->wr[nn] = x + h->t;
                h->wi[nn] = 0.0;
                h->nn = nn - 1;
                h->mode = 1;
            } else {
                y = HA(nn - 1, nn - 1);
                w = HA(nn, nn - 1) * HA(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = sqrt(dabs(q));
                    x += h->t;
                    if (q >= 0.0) {
                        z = p + ((p < 0 || (p == 0 && 1.0 / p < 0)) ? -z : z);  // copysign(z, p)
                        h->wr[nn - 1] = h->wr[nn] = x + z;
                        if (z != 0.0) h->wr[nn] = x - w / z;
                        h->wi[nn - 1] = h->wi[nn] = 0.0;
                    } else {
                        h->wr[nn - 1] = h->wr[nn] = x + p;
                        h->wi[nn] = z;
                        h->wi[nn - 1] = -z;
                    }
                    h->nn = nn - 2;
                    h->mode = 2;
                } else if (h->its == 90) {
                    h->ok = 0;
                    h->mode = 4;
                } else {
                    if (h->its == 10 || h->its == 20 || h->its == 40) {   // exceptional shift
                        h->t += x;
                        for (int i = 0; i <= nn; ++i) HA(i, i) -= x;
                        s = dabs(HA(nn, nn - 1)) + dabs(HA(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++h->its;
                    int m;
                    p = q = r = 0.0;
                    for (m = nn - 2; m >= l; --m) {
                        z = HA(m, m);
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / HA(m + 1, m) + HA(m, m + 1);
                        q = HA(m + 1, m + 1) - z - r - s;
                        r = HA(m + 2, m + 1);
                        s = dabs(p) + dabs(q) + dabs(r);
                        p /= s;
                        q /= s;
                        r /= s;
                        if (m == l) break;
                        u = dabs(HA(m, m - 1)) * (dabs(q) + dabs(r));
                        v = dabs(p) * (dabs(HA(m - 1, m - 1)) + dabs(z) + dabs(HA(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        HA(i, i - 2) = 0.0;
                        if (i != m + 2) HA(i, i - 3) = 0.0;
                    }
                    h->m = m;
                    h->p = p;
                    h->q = q;
                    h->r = r;
                    h->x = x;
                    h->mode = 3;
                }
            }
            // the do-while of the classic code ends when l >= nn - 1 (with the new nn): the iteration count restarts
            if (h->mode != 3 && !(h->l < h->nn - 1)) h->its = 0;
        });
        if (h->mode != 3) continue;
        const int nn = h->nn, l = h->l, m0 = h->m;
        for (int k = m0; k <= nn - 1; ++k) {
            xq.one([=]() {
                double p = h->p, q = h->q, r = h->r, x = h->x, s;
                if (k != m0) {
                    p = HA(k, k - 1);
                    q = HA(k + 1, k - 1);
                    r = 0.0;
                    if (k != nn - 1) r = HA(k + 2, k - 1);
                    if ((x = dabs(p) + dabs(q) + dabs(r)) != 0.0) {
                        const double ix = 1.0 / x;   // (one division where the classic code has three: a long dependent chain on one lane)
                        p *= ix;
                        q *= ix;
                        r *= ix;
                    }
                }
                const double sq = sqrt(p * p + q * q + r * r);
                s = (p < 0 || (p == 0 && 1.0 / p < 0)) ? -sq : sq;  // copysign(sq, p)
                h->apply = 0;
                if (s != 0.0) {
                    if (k == m0) {
                        if (l != m0) HA(k, k - 1) = -HA(k, k - 1);
                    } else {
                        HA(k, k - 1) = -s * x;
                    }
                    p += s;
                    const double is = 1.0 / s, ip = 1.0 / p;
                    h->x = p * is;
                    h->y = q * is;
                    h->z = r * is;
                    h->q = q * ip;
                    h->r = r * ip;
                    h->apply = 1;
                } else {
                    h->x = x;
                }
            });
            if (!h->apply) continue;
            xq.each(nn - k + 1, [=](int e) {   // rows k, k+1, k+2 at column j
                const int j = k + e;
                double p = HA(k, j) + h->q * HA(k + 1, j);
                if (k != nn - 1) {
                    p += h->r * HA(k + 2, j);
                    HA(k + 2, j) -= p * h->z;
                }
                HA(k + 1, j) -= p * h->y;
                HA(k, j) -= p * h->x;
            });
            xq.each((nn < k + 3 ? nn : k + 3) - l + 1, [=](int e) {   // columns k, k+1, k+2 at row i
                const int i = l + e;
                double p = h->x * HA(i, k) + h->y * HA(i, k + 1);
                if (k != nn - 1) {
                    p += h->z * HA(i, k + 2);
                    HA(i, k + 2) -= p * h->r;
                }
                HA(i, k + 1) -= p * h->q;
                HA(i, k) -= p;
            });
        }
    }
    xq.one([=]() {
        if (h->nn >= 0) h->ok = 0;   // (the guard ran out)
    });
#undef HA
}

// right eigenvector of `a` (10 x 10 row-major) for the real eigenvalue lambda: inverse iteration with partial-pivot LU
HYPO_HD void eigenvector10(const double *a, double lambda, double *x) {
    const int n = 10;
    double M[100];
    int piv[10];
    double scale = 0.0;
    for (int i = 0; i < 100; ++i) {
        M[i] = a[i];
        scale = dmax(scale, dabs(a[i]));
    }
    const double shift = lambda + 1e-10 * dmax(dabs(lambda), scale > 0 ? scale : 1.0);
    for (int i = 0; i < n; ++i) M[i * n + i] -= shift;
    const double tiny = 1e-300 + 1e-16 * scale;
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r)
            if (dabs(M[r * n + c]) > dabs(M[p * n + c])) p = r;
        piv[c] = p;
        if (p != c)
            for (int j = 0; j < n; ++j) {
                const double t = M[c * n + j];
                M[c * n + j] = M[p * n + j];
                M[p * n + j] = t;
            }
        if (dabs(M[c * n + c]) < tiny) M[c * n + c] = tiny;
        for (int r = c + 1; r < n; ++r) {
            const double f = M[r * n + c] / M[c * n + c];
            M[r * n + c] = f;
            for (int j = c + 1; j < n; ++j) M[r * n + j] -= f * M[c * n + j];
        }
    }
    for (int i = 0; i < n; ++i) x[i] = 1.0;
    for (int it = 0; it < 3; ++it) {
        for (int c = 0; c < n; ++c) {
            if (piv[c] != c) {
                const double t = x[c];
                x[c] = x[piv[c]];
                x[piv[c]] = t;
            }
            for (int r = c + 1; r < n; ++r) x[r] -= M[r * n + c] * x[c];
        }
        for (int c = n - 1; c >= 0; --c) {
            for (int j = c + 1; j < n; ++j) x[c] -= M[c * n + j] * x[j];
            x[c] /= M[c * n + c];
        }
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += x[i] * x[i];
        nrm = sqrt(nrm);
        if (!(nrm > 0.0) || !dfinite(nrm)) break;
        for (int i = 0; i < n; ++i) x[i] /= nrm;
    }
}

struct Ess5Work {  // scratch of one hypothesis
    double A[5][9], AtA[81], Vv[81], Bw[81], lam[9], cs[12], red[24];
    double basis[9][4];
    double Ep[9][20], EEt[9][20], half_trace[20];
    double polys[10][20];
    double action[100], mult[10];
    Hqr10 hq;
    int ord[9], perm[10], flag, ok, n_out;
};

// essential matrices (row-major 3 x 3 each, at most 10) from five correspondences of normalised image points
// p1 (5 x 2) and p2 (5 x 2); n_models receives the count
template <class X>
HYPO_HD void essential5(const X &x, Ess5Work *w, const double *p1, const double *p2, double *models /* 10 x 9 */, int *n_models) {
    // null space of the 5 x 9 epipolar constraint matrix (essential.cpp:119-131): rows h = p1 p2^T flattened row-wise
    HYPO_STAMP(0);
    x.each(45, [=](int e) {
        const int i = e / 9, j = (e % 9) / 3, k = e % 3;
        const double a = j == 0 ? p1[2 * i] : (j == 1 ? p1[2 * i + 1] : 1.0), b = k == 0 ? p2[2 * i] : (k == 1 ? p2[2 * i + 1] : 1.0);
        w->A[i][3 * j + k] = a * b;
    });
    x.each(81, [=](int e) {
        const int i = e / 9, j = e % 9;
        double s = 0;
        for (int k = 0; k < 5; ++k) s += w->A[k][i] * w->A[k][j];
        w->AtA[e] = s;
    });
    HYPO_STAMP(1);
    jacobi_rr<9>(x, w->AtA, w->Bw, w->Vv, w->lam, w->cs, w->red, &w->flag);
    HYPO_STAMP(2);
    x.one([=]() { ascending(9, w->lam, w->ord); });
    // basis columns: singular vectors 5..8 in JacobiSVD's descending order = ascending eigenvalues 3, 2, 1, 0
    x.each(36, [=](int e) {
        const int r = e / 4, c = e % 4;
        w->basis[r][c] = w->Vv[9 * r + w->ord[3 - c]];
    });
    // E(x, y, z) = x Ex + y Ey + z Ez + Ew with E_c = to_matrix(basis.col(c)) (COLUMNS of E are the 3-segments)
    x.each(9, [=](int e) {
        const int i = e / 3, j = e % 3;
        double *p = w->Ep[e];
        poly_zero(p);
        p[MX] = w->basis[3 * j + i][0];
        p[MY] = w->basis[3 * j + i][1];
        p[MZ] = w->basis[3 * j + i][2];
        p[MI] = w->basis[3 * j + i][3];
    });
    x.each(9, [=](int e) {   // E E^T
        const int i = e / 3, j = e % 3;
        double s[20], t[20], m[20];
        poly_zero(s);
        for (int k = 0; k < 3; ++k) {
            poly_mul(w->Ep[3 * i + k], w->Ep[3 * j + k], m);
            poly_add(s, m, t);
            for (int c = 0; c < 20; ++c) s[c] = t[c];
        }
        for (int c = 0; c < 20; ++c) w->EEt[e][c] = s[c];
    });
    x.one([=]() {
        for (int c = 0; c < 20; ++c) w->half_trace[c] = 0.5 * ((w->EEt[0][c] + w->EEt[4][c]) + w->EEt[8][c]);
    });
    x.each(10, [=](int e) {   // the nine trace constraints and the determinant
        double s[20], t[20], m[20];
        if (e < 9) {
            const int i = e / 3, j = e % 3;
            poly_zero(s);
            for (int k = 0; k < 3; ++k) {
                poly_mul(w->EEt[3 * i + k], w->Ep[3 * k + j], m);
                poly_add(s, m, t);
                for (int c = 0; c < 20; ++c) s[c] = t[c];
            }
            poly_mul(w->half_trace, w->Ep[e], m);
            poly_sub(s, m, t);
            for (int c = 0; c < 20; ++c) w->polys[e][c] = t[c];
        } else {
            double a[20], b[20], d[20], u[20];
            // Ep00 (Ep11 Ep22 - Ep12 Ep21) - Ep01 (Ep10 Ep22 - Ep12 Ep20) + Ep02 (Ep10 Ep21 - Ep11 Ep20)
            poly_mul(w->Ep[4], w->Ep[8], a); poly_mul(w->Ep[5], w->Ep[7], b); poly_sub(a, b, d); poly_mul(w->Ep[0], d, s);
            poly_mul(w->Ep[3], w->Ep[8], a); poly_mul(w->Ep[5], w->Ep[6], b); poly_sub(a, b, d); poly_mul(w->Ep[1], d, t);
            poly_sub(s, t, u);
            poly_mul(w->Ep[3], w->Ep[7], a); poly_mul(w->Ep[4], w->Ep[6], b); poly_sub(a, b, d); poly_mul(w->Ep[2], d, t);
            poly_add(u, t, s);
            for (int c = 0; c < 20; ++c) w->polys[9][c] = s[c];
        }
    });
    HYPO_STAMP(3);
    // Gauss-Jordan with the reference's row-permutation bookkeeping (essential.cpp:167-190).  A pivot step is three steps here: the
    // pivot row is chosen and the column of multipliers filed (one lane); every COLUMN of the 10 x 20 system is then updated by its
    // own item (the ten row updates of an item are independent loads and stores: they pipeline); the back substitution likewise.
    x.one([=]() {
        for (int i = 0; i < 10; ++i) w->perm[i] = i;
    });
    for (int i = 0; i < 10; ++i) {
        x.one([=]() {
            int *perm = w->perm;
            for (int j = i + 1; j < 10; ++j)
                if (dabs(w->polys[perm[i]][i]) < dabs(w->polys[perm[j]][i])) {
                    const int t = perm[i];
                    perm[i] = perm[j];
                    perm[j] = t;
                }
            // mult[0] = the pivot (0: the step is skipped), mult[j] = the entry of row perm[j] in the pivot column
            w->mult[0] = w->polys[perm[i]][i];
            for (int j = i + 1; j < 10; ++j) w->mult[j] = w->polys[perm[j]][i];
        });
        x.each(20, [=](int c) {
            const double d = w->mult[0];
            if (d == 0.0) return;
            const int *perm = w->perm;
            const double pv = w->polys[perm[i]][c] / d;
            w->polys[perm[i]][c] = pv;
            HYPO_UNROLL
            for (int j = 1; j < 10; ++j)
                if (j > i) w->polys[perm[j]][c] -= pv * w->mult[j];
        });
    }
    for (int i = 9; i > 0; --i) {
        x.one([=]() {
            for (int j = 0; j < i; ++j) w->mult[j] = w->polys[w->perm[j]][i];
        });
        x.each(20, [=](int c) {
            const int *perm = w->perm;
            const double pv = w->polys[perm[i]][c];
            HYPO_UNROLL
            for (int j = 0; j < 9; ++j)
                if (j < i) w->polys[perm[j]][c] -= pv * w->mult[j];
        });
    }
    // action matrix; its eigenvalues
    x.one([=]() {
        int *perm = w->perm;
        for (int i = 0; i < 100; ++i) w->action[i] = 0.0;
        const int rows[6] = {XXX, XXY, XYY, XXZ, XYZ, XZZ};
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 10; ++c) w->action[r * 10 + c] = -w->polys[perm[rows[r]]][XX + c];
        w->action[6 * 10 + (XX - XX)] = 1.0;
        w->action[7 * 10 + (XY - XX)] = 1.0;
        w->action[8 * 10 + (XZ - XX)] = 1.0;
        w->action[9 * 10 + (MX - XX)] = 1.0;
        for (int i = 0; i < 100; ++i) w->hq.a[i] = w->action[i];
    });
    real_eigenvalues10(x, &w->hq);
    x.one([=]() {
        w->ok = w->hq.ok;
        // output slots in eigenvalue order: slot of eigenvalue i = number of real eigenvalues before it
        int n = 0;
        for (int i = 0; i < 10; ++i) {
            w->perm[i] = -1;
            if (w->ok && dabs(w->hq.wi[i]) < 1.0e-10) w->perm[i] = n++;
        }
        w->n_out = n;
        *n_models = n;
    });
    HYPO_STAMP(4);
    x.each(10, [=](int i) {   // one real eigenvalue each: eigenvector -> (x, y, z) -> E
        const int slot = w->perm[i];
        if (slot < 0) return;
        double h[10];
        eigenvector10(w->action, w->hq.wr[i], h);
        const double ww = h[MI - XX];
        const double sx = h[MX - XX] / ww, sy = h[MY - XX] / ww, sz = h[MZ - XX] / ww;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                const int k = 3 * c + r;  // to_matrix: column c = segment c
                models[9 * slot + 3 * r + c] = w->basis[k][0] * sx + w->basis[k][1] * sy + w->basis[k][2] * sz + w->basis[k][3];
            }
    });
    HYPO_STAMP(5);
}

}  // namespace hypo
