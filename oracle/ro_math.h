/*
 * ORACLE (test infrastructure, NOT product code) -- small FP64 linear algebra
 * used by the CPU restatement of rd_vio's hot path.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may use anything under oracle/.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be
 * built in this container (needs Eigen, Ceres, OpenCV, yaml-cpp -- all absent).
 * The restatement is pinned only by self-consistency (finite differences,
 * algebraic identities) -- see DESIGN.md.
 *
 * Conventions follow the reference:
 *  - quaternions are stored (x, y, z, w) like Eigen's coeffs()
 *    (src/rdvio_estimation/src/solver.cpp:90-91)
 *  - matrices are row-major double arrays unless stated otherwise
 */
#ifndef RO_MATH_H
#define RO_MATH_H

#include <math.h>
#include <string.h>

#define RO_GRAVITY 9.80665 /* src/rdvio/include/rdvio/types.h:26 */

/* ---------------- vec3 ---------------- */
static inline void v3_set(double *o, double x, double y, double z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void v3_copy(double *o, const double *a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline void v3_add(double *o, const double *a, const double *b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3_sub(double *o, const double *a, const double *b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3_scale(double *o, const double *a, double s) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; }
static inline double v3_dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline double v3_norm(const double *a) { return sqrt(v3_dot(a, a)); }
static inline void v3_cross(double *o, const double *a, const double *b) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void v3_normalize(double *o, const double *a) {
    double n = v3_norm(a);
    if (n > 0) { o[0] = a[0] / n; o[1] = a[1] / n; o[2] = a[2] / n; } else { v3_copy(o, a); }
}

/* ---------------- generic small dense (row-major) ---------------- */
/* C(m x n) = A(m x k) * B(k x n) */
static inline void mat_mul(double *C, const double *A, const double *B, int m, int k, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
/* C(m x n) = A(k x m)^T * B(k x n) */
static inline void mat_mul_tn(double *C, const double *A, const double *B, int k, int m, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[l * m + i] * B[l * n + j];
            C[i * n + j] = s;
        }
}
/* C(m x n) = A(m x k) * B(n x k)^T */
static inline void mat_mul_nt(double *C, const double *A, const double *B, int m, int k, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[j * k + l];
            C[i * n + j] = s;
        }
}
static inline void mat_transpose(double *T, const double *A, int m, int n) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) T[j * m + i] = A[i * n + j];
}
static inline void m3_identity(double *M) { memset(M, 0, 9 * sizeof(double)); M[0] = M[4] = M[8] = 1.0; }
static inline void m3_mul(double *C, const double *A, const double *B) { double t[9]; mat_mul(t, A, B, 3, 3, 3); memcpy(C, t, sizeof t); }
static inline void m3_transpose(double *T, const double *A) { double t[9]; mat_transpose(t, A, 3, 3); memcpy(T, t, sizeof t); }
static inline void m3_mulv(double *o, const double *A, const double *v) {
    double x = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double y = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double z = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}

/* hat: src/rdvio_geometry/include/rdvio/geometry/lie_algebra.h:6-9 */
static inline void hat(double *M, const double *w) {
    M[0] = 0;     M[1] = -w[2]; M[2] = w[1];
    M[3] = w[2];  M[4] = 0;     M[5] = -w[0];
    M[6] = -w[1]; M[7] = w[0];  M[8] = 0;
}

/* ---------------- quaternion (x,y,z,w) ---------------- */
static inline void q_identity(double *q) { q[0] = q[1] = q[2] = 0; q[3] = 1; }
static inline void q_copy(double *o, const double *q) { memcpy(o, q, 4 * sizeof(double)); }
static inline void q_conj(double *o, const double *q) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }
/* Hamilton product a*b (Eigen operator*) */
static inline void q_mul(double *o, const double *a, const double *b) {
    double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static inline void q_normalize(double *o, const double *q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    o[0] = q[0] / n; o[1] = q[1] / n; o[2] = q[2] / n; o[3] = q[3] / n;
}
/* v' = q * v  (Eigen _transformVector: v + 2w(u x v) + 2 u x (u x v)) */
static inline void q_rot(double *o, const double *q, const double *v) {
    double uv[3], uuv[3];
    v3_cross(uv, q, v);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    v3_cross(uuv, q, uv);
    o[0] = v[0] + q[3] * uv[0] + uuv[0];
    o[1] = v[1] + q[3] * uv[1] + uuv[1];
    o[2] = v[2] + q[3] * uv[2] + uuv[2];
}
static inline void q_rot_inv(double *o, const double *q, const double *v) {
    double c[4]; q_conj(c, q); q_rot(o, c, v);
}
/* rotation matrix of a (unit) quaternion, Eigen toRotationMatrix */
static inline void q_to_mat(double *R, const double *q) {
    double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* expmap: lie_algebra.h:11-16 (AngleAxis(|w|, w.stableNormalized()) -> quaternion) */
static inline void expmap(double *q, const double *w) {
    double n = v3_norm(w);
    if (n > 0) {
        double s = sin(0.5 * n) / n;
        q[0] = w[0] * s; q[1] = w[1] * s; q[2] = w[2] * s; q[3] = cos(0.5 * n);
    } else {
        q_identity(q);
    }
}
/* logmap: lie_algebra.h:18-21 (Eigen AngleAxis(q): angle = 2 atan2(|v|, |w|), axis flipped if w<0) */
static inline void logmap(double *w, const double *q) {
    double n = v3_norm(q);
    if (n != 0) {
        double angle = 2.0 * atan2(n, fabs(q[3]));
        if (q[3] < 0) n = -n;
        double s = angle / n;
        w[0] = q[0] * s; w[1] = q[1] * s; w[2] = q[2] * s;
    } else {
        w[0] = w[1] = w[2] = 0;
    }
}

void ro_right_jacobian(double *J, const double *w);          /* lie_algebra.cpp:5-45 */
void ro_s2_tangential_basis(double *b1, double *b2, const double *x); /* lie_algebra.cpp:47-56 */
int ro_inverse(double *Ainv, const double *A, int n);         /* partial-pivot LU inverse */
int ro_cholesky_lower(double *L, const double *A, int n);     /* LLT, A = L L^T */
void ro_sym_eig(double *evals, double *V, const double *A, int n); /* cyclic Jacobi; V columns = eigenvectors, ascending */

#endif
