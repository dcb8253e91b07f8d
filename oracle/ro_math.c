/*
 * ORACLE (test infrastructure, NOT product code) -- see ro_math.h header.
 * PARITY UNPINNED (no reference fixtures exist; reference not buildable here).
 */
#include "ro_math.h"
#include <float.h>
#include <stdlib.h>

/* src/rdvio_geometry/src/lie_algebra.cpp:5-45 */
void ro_right_jacobian(double *J, const double *w) {
    const double root2_eps = sqrt(DBL_EPSILON);
    const double root4_eps = sqrt(root2_eps);
    const double qdrt720 = sqrt(sqrt(720.0));
    const double qdrt5040 = sqrt(sqrt(5040.0));
    const double sqrt24 = sqrt(24.0);
    const double sqrt120 = sqrt(120.0);

    double angle = v3_norm(w);
    double cangle = cos(angle), sangle = sin(angle);
    double angle2 = angle * angle;

    double cos_term;
    if (angle > root4_eps * qdrt720) {
        cos_term = (1 - cangle) / angle2;
    } else {
        cos_term = 0.5;
        if (angle > root2_eps * sqrt24) cos_term -= angle2 / 24.0;
    }
    double sin_term;
    if (angle > root4_eps * qdrt5040) {
        sin_term = (angle - sangle) / (angle * angle2);
    } else {
        sin_term = 1.0 / 6.0;
        if (angle > root2_eps * sqrt120) sin_term -= angle2 / 120.0;
    }
    double H[9], H2[9];
    hat(H, w);
    m3_mul(H2, H, H);
    for (int i = 0; i < 9; ++i) J[i] = -cos_term * H[i] + sin_term * H2[i];
    J[0] += 1; J[4] += 1; J[8] += 1;
}

/* src/rdvio_geometry/src/lie_algebra.cpp:47-56 */
void ro_s2_tangential_basis(double *b1, double *b2, const double *x) {
    int d = 0;
    for (int i = 1; i < 3; ++i)
        if (fabs(x[i]) > fabs(x[d])) d = i;
    double e[3] = {0, 0, 0};
    e[(d + 1) % 3] = 1.0;
    double c[3];
    v3_cross(c, x, e);
    v3_normalize(b1, c);
    v3_cross(c, x, b1);
    v3_normalize(b2, c);
}

/* general inverse by LU with partial pivoting (what Eigen's .inverse() does for n > 4) */
int ro_inverse(double *Ainv, const double *A, int n) {
    double *M = (double *)malloc(sizeof(double) * n * 2 * n);
    int ok = 1;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            M[i * 2 * n + j] = A[i * n + j];
            M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(M[c * 2 * n + c]);
        for (int r = c + 1; r < n; ++r) {
            double v = fabs(M[r * 2 * n + c]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) { ok = 0; }
        if (piv != c)
            for (int j = 0; j < 2 * n; ++j) {
                double t = M[c * 2 * n + j]; M[c * 2 * n + j] = M[piv * 2 * n + j]; M[piv * 2 * n + j] = t;
            }
        double d = M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) M[c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = M[r * 2 * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * n; ++j) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Ainv[i * n + j] = M[i * 2 * n + n + j];
    free(M);
    return ok;
}

/* A = L L^T (lower); returns 0 if not positive definite */
int ro_cholesky_lower(double *L, const double *A, int n) {
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0)) return 0;
        double d = sqrt(s);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / d;
        }
    }
    return 1;
}

/* symmetric eigendecomposition by cyclic Jacobi rotations.
 * evals ascending (like Eigen::SelfAdjointEigenSolver), V row-major with eigenvectors in columns. */
void ro_sym_eig(double *evals, double *V, const double *Ain, int n) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0.0) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) evals[i] = A[i * n + i];
    /* sort ascending (selection sort, swap columns) */
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j)
            if (evals[j] < evals[m]) m = j;
        if (m != i) {
            double t = evals[i]; evals[i] = evals[m]; evals[m] = t;
            for (int k = 0; k < n; ++k) {
                double v = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = v;
            }
        }
    }
    free(A);
}
