// Host-side geometry for the per-frame orchestration (rows A4, A5, A17, A18 of SURVEY.md section 8): small fixed-size
// FP64 algebra plus the two-view solvers Frame::track_keypoints runs between the LK kernel and the BA kernels.
// Dependency-free C++17 (the reference uses Eigen; it is not available in this build image).
//
// Reference behaviour restated here:
//   apply_k / remove_k / triangulate_point      /root/reference/src/rdvio_geometry/include/rdvio/geometry/stereo.h:7-20,83-93
//   find_essential_matrix / find_rotation_matrix /root/reference/src/rdvio_geometry/src/stereo.cpp:38-91
//   Ransac<>                                     /root/reference/src/rdvio_util/include/rdvio/util/ransac.h:8-103
//   LotBox                                       /root/reference/src/rdvio_util/include/rdvio/util/random.h:79-126
//   solve_essential_5pt (Groebner action matrix) /root/reference/src/rdvio_geometry/src/essential.cpp:8-299
//   solve_rotation_2pt (Wahba / Kabsch)          /root/reference/src/rdvio_geometry/include/rdvio/geometry/wahba.h:8-26
// Where the reference calls Eigen::JacobiSVD / Eigen::EigenSolver the same quantities are obtained with a cyclic
// Jacobi eigensolver on A^T A and a Hessenberg-QR eigenvalue iteration + inverse iteration; the sign / basis freedom
// of those decompositions does not reach the results used downstream (inlier masks, R = V E U^T, dehomogenised points).
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <random>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "../../csrc/hypo_solvers.hpp"

namespace rdvio_pipe {

struct V2 { double x = 0, y = 0; };
struct V3 { double x = 0, y = 0, z = 0; };
struct Q4 { double x = 0, y = 0, z = 0, w = 1; };
struct M3 { double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; };  // row-major

inline V2 operator-(const V2 &a, const V2 &b) { return {a.x - b.x, a.y - b.y}; }
inline double norm(const V2 &a) { return std::sqrt(a.x * a.x + a.y * a.y); }
inline double sqnorm(const V2 &a) { return a.x * a.x + a.y * a.y; }

inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(const V3 &a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(double s, const V3 &a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(const V3 &a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(const V3 &a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(const V3 &a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(const V3 &a) { return a / norm(a); }
inline V2 hnormalized(const V3 &a) { return {a.x / a.z, a.y / a.z}; }

inline Q4 conj(const Q4 &q) { return {-q.x, -q.y, -q.z, q.w}; }
inline Q4 operator*(const Q4 &a, const Q4 &b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Q4 normalized(const Q4 &q) {
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return {q.x / n, q.y / n, q.z / n, q.w / n};
}
// q * v (Eigen's Quaternion::_transformVector)
inline V3 rot(const Q4 &q, const V3 &v) {
    const V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}
inline M3 to_mat(const Q4 &q) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 R;
    R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
    R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
    R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
    return R;
}
// lie_algebra.h:9-15 (Eigen::AngleAxis path): exp(w) as a unit quaternion
inline Q4 expmap(const V3 &w) {
    const double th = norm(w);
    if (th == 0.0) return Q4{0, 0, 0, 1};  // AngleAxis(0, NaN-axis) guard: identity
    const double s = std::sin(0.5 * th) / th;
    return {s * w.x, s * w.y, s * w.z, std::cos(0.5 * th)};
}
inline V3 operator*(const M3 &A, const V3 &v) {
    return {A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
            A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
inline M3 operator*(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A.m[3 * i + k] * B.m[3 * k + j];
            C.m[3 * i + j] = s;
        }
    return C;
}
inline M3 transpose(const M3 &A) {
    M3 T;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T.m[3 * i + j] = A.m[3 * j + i];
    return T;
}
inline double det(const M3 &A) {
    return A.m[0] * (A.m[4] * A.m[8] - A.m[5] * A.m[7]) - A.m[1] * (A.m[3] * A.m[8] - A.m[5] * A.m[6]) +
           A.m[2] * (A.m[3] * A.m[7] - A.m[4] * A.m[6]);
}

// stereo.h:7-14.  K row-major 3x3.
inline V2 apply_k(const V3 &p, const double *K) { return {p.x / p.z * K[0] + K[2], p.y / p.z * K[4] + K[5]}; }
inline V3 remove_k(const V2 &p, const double *K) { return normalized(V3{(p.x - K[2]) / K[0], (p.y - K[5]) / K[4], 1.0}); }

// ------------------------------------------------------------------------------------------------------------------
// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major, destroyed); V columns = eigenvectors
// ------------------------------------------------------------------------------------------------------------------
inline void sym_eigen(int n, double *A, double *V, double *lam) {
    for (int i = 0; i < n * n; ++i) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? dg : off) += A[i * n + j] * A[i * n + j];
        if (off <= 1e-300 || off <= 1e-32 * dg) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) lam[i] = A[i * n + i];
}

// indices of the eigenvalues in ascending order
inline std::vector<int> ascending_order(int n, const double *lam) {
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lam[a] < lam[b]; });
    return idx;
}

// SVD of a 3x3 matrix A = U diag(s) V^T, s descending, U and V orthogonal (completed by cross products when A is
// rank deficient).  Used for Wahba's problem where only V E U^T matters.
inline void svd3(const M3 &A, M3 &U, double s[3], M3 &V) {
    double AtA[9], Vv[9], lam[3];
    const M3 At = transpose(A);
    const M3 P = At * A;
    for (int i = 0; i < 9; ++i) AtA[i] = P.m[i];
    sym_eigen(3, AtA, Vv, lam);
    std::vector<int> ord = ascending_order(3, lam);
    std::reverse(ord.begin(), ord.end());
    V3 v[3], u[3];
    for (int c = 0; c < 3; ++c) {
        v[c] = V3{Vv[0 * 3 + ord[c]], Vv[1 * 3 + ord[c]], Vv[2 * 3 + ord[c]]};
        s[c] = std::sqrt(std::max(lam[ord[c]], 0.0));
    }
    v[2] = cross(v[0], v[1]);  // right-handed completion (also fixes a degenerate third vector)
    const double tol = 1e-12 * std::max(s[0], 1e-300);
    u[0] = s[0] > tol ? (A * v[0]) / s[0] : V3{1, 0, 0};
    if (s[1] > tol) {
        u[1] = (A * v[1]) / s[1];
        u[1] = normalized(u[1] - dot(u[1], u[0]) * u[0]);
    } else {
        const V3 a = std::fabs(u[0].x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0};
        u[1] = normalized(cross(u[0], a));
    }
    if (s[2] > tol) {
        u[2] = (A * v[2]) / s[2];
        u[2] = u[2] - dot(u[2], u[0]) * u[0];
        u[2] = normalized(u[2] - dot(u[2], u[1]) * u[1]);
    } else {
        u[2] = cross(u[0], u[1]);
    }
    for (int c = 0; c < 3; ++c) {
        U.m[0 * 3 + c] = u[c].x; U.m[1 * 3 + c] = u[c].y; U.m[2 * 3 + c] = u[c].z;
        V.m[0 * 3 + c] = v[c].x; V.m[1 * 3 + c] = v[c].y; V.m[2 * 3 + c] = v[c].z;
    }
}

// wahba.h:8-26:  h(p2) = R h(p1)
inline M3 solve_rotation_2pt(const std::array<V3, 2> &p1, const std::array<V3, 2> &p2) {
    // (csrc/hypo_solvers.hpp: the source the device kernel runs)
    const double a[6] = {p1[0].x, p1[0].y, p1[0].z, p1[1].x, p1[1].y, p1[1].z}, b[6] = {p2[0].x, p2[0].y, p2[0].z, p2[1].x, p2[1].y, p2[1].z};
    M3 R;
    hypo::rotation2(a, b, R.m);
    return R;
}

// essential.h:14-19
inline double essential_geometric_error(const M3 &E, const V2 &p1, const V2 &p2) {
    const V3 Ep1 = E * V3{p1.x, p1.y, 1.0};
    const double r = p2.x * Ep1.x + p2.y * Ep1.y + Ep1.z;
    return r * r / (Ep1.x * Ep1.x + Ep1.y * Ep1.y);
}

// ------------------------------------------------------------------------------------------------------------------
// eigenvalues of a general real n x n matrix: elimination to Hessenberg form + QR with implicit double shifts (the
// classic EISPACK elmhes / hqr recipe); right eigenvectors of the real eigenvalues by inverse iteration.
// ------------------------------------------------------------------------------------------------------------------
inline bool real_eigenvalues(int n, std::vector<double> a, std::vector<double> &wr, std::vector<double> &wi) {
    auto A = [&](int i, int j) -> double & { return a[(size_t)i * n + j]; };
    for (int m = 1; m < n - 1; ++m) {
        double x = 0.0;
        int i = m;
        for (int j = m; j < n; ++j)
            if (std::fabs(A(j, m - 1)) > std::fabs(x)) { x = A(j, m - 1); i = j; }
        if (i != m) {
            for (int j = m - 1; j < n; ++j) std::swap(A(i, j), A(m, j));
            for (int j = 0; j < n; ++j) std::swap(A(j, i), A(j, m));
        }
        if (x != 0.0)
            for (i = m + 1; i < n; ++i) {
                double y = A(i, m - 1);
                if (y != 0.0) {
                    y /= x;
                    A(i, m - 1) = y;
                    for (int j = m; j < n; ++j) A(i, j) -= y * A(m, j);
                    for (int j = 0; j < n; ++j) A(j, m) += y * A(j, i);
                }
            }
    }
    for (int i = 2; i < n; ++i)
        for (int j = 0; j < i - 1; ++j) A(i, j) = 0.0;
    wr.assign(n, 0.0);
    wi.assign(n, 0.0);
    double anorm = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = std::max(i - 1, 0); j < n; ++j) anorm += std::fabs(A(i, j));
    int nn = n - 1;
    double t = 0.0, p = 0, q = 0, r = 0, s = 0, x = 0, y = 0, z = 0, w = 0, u = 0, v = 0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {
                s = std::fabs(A(l - 1, l - 1)) + std::fabs(A(l, l));
                if (s == 0.0) s = anorm;
                if (std::fabs(A(l, l - 1)) + s == s) { A(l, l - 1) = 0.0; break; }
            }
            x = A(nn, nn);
            if (l == nn) {
                wr[nn] = x + t;
                wi[nn--] = 0.0;
            } else {
                y = A(nn - 1, nn - 1);
                w = A(nn, nn - 1) * A(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = std::sqrt(std::fabs(q));
                    x += t;
                    if (q >= 0.0) {
                        z = p + std::copysign(z, p);
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.0) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn - 1] = -(wi[nn] = z);
                    }
                    nn -= 2;
                } else {
                    if (its == 90) return false;
                    if (its == 10 || its == 20 || its == 40) {
                        t += x;
                        for (int i = 0; i <= nn; ++i) A(i, i) -= x;
                        s = std::fabs(A(nn, nn - 1)) + std::fabs(A(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = A(m, m);
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / A(m + 1, m) + A(m, m + 1);
                        q = A(m + 1, m + 1) - z - r - s;
                        r = A(m + 2, m + 1);
                        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        u = std::fabs(A(m, m - 1)) * (std::fabs(q) + std::fabs(r));
                        v = std::fabs(p) * (std::fabs(A(m - 1, m - 1)) + std::fabs(z) + std::fabs(A(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        A(i, i - 2) = 0.0;
                        if (i != m + 2) A(i, i - 3) = 0.0;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = A(k, k - 1);
                            q = A(k + 1, k - 1);
                            r = 0.0;
                            if (k != nn - 1) r = A(k + 2, k - 1);
                            if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        if ((s = std::copysign(std::sqrt(p * p + q * q + r * r), p)) != 0.0) {
                            if (k == m) {
                                if (l != m) A(k, k - 1) = -A(k, k - 1);
                            } else {
                                A(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s; y = q / s; z = r / s;
                            q /= p; r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = A(k, j) + q * A(k + 1, j);
                                if (k != nn - 1) { p += r * A(k + 2, j); A(k + 2, j) -= p * z; }
                                A(k + 1, j) -= p * y;
                                A(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * A(i, k) + y * A(i, k + 1);
                                if (k != nn - 1) { p += z * A(i, k + 2); A(i, k + 2) -= p * r; }
                                A(i, k + 1) -= p * q;
                                A(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return true;
}

// right eigenvector of `a` (n x n row-major) for the real eigenvalue lambda: inverse iteration with partial-pivot LU
inline std::vector<double> eigenvector_for(int n, const std::vector<double> &a, double lambda) {
    std::vector<double> M(a);
    double scale = 0.0;
    for (double v : a) scale = std::max(scale, std::fabs(v));
    const double shift = lambda + 1e-10 * std::max(std::fabs(lambda), scale > 0 ? scale : 1.0);
    for (int i = 0; i < n; ++i) M[(size_t)i * n + i] -= shift;
    std::vector<int> piv(n);
    const double tiny = 1e-300 + 1e-16 * scale;
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)p * n + c])) p = r;
        piv[c] = p;
        if (p != c)
            for (int j = 0; j < n; ++j) std::swap(M[(size_t)c * n + j], M[(size_t)p * n + j]);
        if (std::fabs(M[(size_t)c * n + c]) < tiny) M[(size_t)c * n + c] = tiny;
        for (int r = c + 1; r < n; ++r) {
            const double f = M[(size_t)r * n + c] / M[(size_t)c * n + c];
            M[(size_t)r * n + c] = f;
            for (int j = c + 1; j < n; ++j) M[(size_t)r * n + j] -= f * M[(size_t)c * n + j];
        }
    }
    std::vector<double> x(n, 1.0);
    for (int it = 0; it < 3; ++it) {
        for (int c = 0; c < n; ++c) {
            if (piv[c] != c) std::swap(x[c], x[piv[c]]);
            for (int r = c + 1; r < n; ++r) x[r] -= M[(size_t)r * n + c] * x[c];
        }
        for (int c = n - 1; c >= 0; --c) {
            for (int j = c + 1; j < n; ++j) x[c] -= M[(size_t)c * n + j] * x[j];
            x[c] /= M[(size_t)c * n + c];
        }
        double nrm = 0.0;
        for (double v : x) nrm += v * v;
        nrm = std::sqrt(nrm);
        if (!(nrm > 0.0) || !std::isfinite(nrm)) break;
        for (double &v : x) v /= nrm;
    }
    return x;
}

// ------------------------------------------------------------------------------------------------------------------
// five-point essential matrix (essential.cpp:8-299): null-space basis, ten cubic constraints in GRevLex order,
// Gauss-Jordan to the action matrix of multiplication by x, real eigenvectors -> (x, y, z).
// ------------------------------------------------------------------------------------------------------------------
// The solver is csrc/hypo_solvers.hpp (hypo::essential5), the same source the device kernels run -- here with the serial executor.
inline std::vector<M3> solve_essential_5pt(const std::array<V2, 5> &pts1, const std::array<V2, 5> &pts2) {
    double p1[10], p2[10], models[90];
    for (int i = 0; i < 5; ++i) {
        p1[2 * i] = pts1[i].x; p1[2 * i + 1] = pts1[i].y;
        p2[2 * i] = pts2[i].x; p2[2 * i + 1] = pts2[i].y;
    }
    int n = 0;
    hypo::Ess5Work work;
    hypo::essential5(hypo::SerialExec{}, &work, p1, p2, models, &n);
    std::vector<M3> results((size_t)n);
    for (int k = 0; k < n; ++k)
        for (int q = 0; q < 9; ++q) results[(size_t)k].m[q] = models[9 * k + q];
    return results;
}

// ------------------------------------------------------------------------------------------------------------------
// LotBox + Ransac (random.h:79-126, ransac.h:8-103).  std::default_random_engine and
// std::uniform_int_distribution<size_t> are the same library types the reference draws from, seeded identically.
// ------------------------------------------------------------------------------------------------------------------
class LotBox {
  public:
    explicit LotBox(size_t size) : cap(0), lots(size) { std::iota(lots.begin(), lots.end(), 0); }
    size_t draw_without_replacement() {
        if (remaining() > 1) {
            std::swap(lots[cap], lots[next(cap, lots.size() - 1)]);
            return lots[cap++];
        } else if (remaining() == 1) {
            cap++;
            return lots.back();
        }
        return size_t(-1);
    }
    void refill_all() { cap = 0; }
    size_t remaining() const { return lots.size() - cap; }
    void seed(unsigned int value) { engine.seed(value); }

  private:
    size_t next(size_t left, size_t right) {
        return distribution(engine, std::uniform_int_distribution<size_t>::param_type(left, right));
    }
    size_t cap;
    std::vector<size_t> lots;
    std::default_random_engine engine;
    std::uniform_int_distribution<size_t> distribution{0, std::numeric_limits<size_t>::max()};
};

// Hypothesis generation and scoring of a RANSAC gate behind the backend (rdvio_backend::ransac_generate_score / ransac_fetch; the
// HIP product implements them: rdvio_hip_ransac_generate_score).  kind / pa / pb / threshold as that entry takes them.
struct RansacDevice {
    int (*generate)(void *user, int kind, int n_points, int points_changed, const double *pa, const double *pb, double threshold, int n_iterations,
                    const int32_t *samples, int32_t *models_per_iteration, double *models, int32_t *inlier_counts) = nullptr;
    int (*fetch)(void *user, int model, uint8_t *mask) = nullptr;
    void *user = nullptr;
    int kind = 0;
    const double *pa = nullptr, *pb = nullptr;
    double threshold = 0.0;
};

// Generic RANSAC loop (ransac.h:31-76).  solve(sample indices) -> models; inlier(model, i) -> bool.
// Evaluated a batch of iterations at a time: the samples do not depend on the scores (the lot box is seeded per call and draws in
// a fixed order), only the NUMBER of iterations does, through the adaptive iter_max.  The hypotheses of the next RANSAC_BATCH
// iterations are generated and scored together -- on the device when `dev` offers the hooks, in place otherwise -- and the
// reference's accept / early-exit decisions are replayed on the inlier counts in iteration order.  Both roads run this control
// flow over bit-identical models (csrc/hypo_solvers.hpp) and inlier decisions.
constexpr size_t RANSAC_BATCH = 8;

template <size_t DoF, class Model, class SolveFn, class InlierFn>
Model ransac(size_t size, double confidence, size_t max_iteration, int seed, SolveFn solve, InlierFn inlier, std::vector<char> &inlier_mask,
             Model model = Model(), const RansacDevice *dev = nullptr) {
    LotBox lotbox(size);
    lotbox.seed((unsigned int)seed);
    const double K = std::log(std::max(1 - confidence, 1.0e-5));
    size_t inlier_count = 0;
    inlier_mask.assign(size, 0);
    if (size < DoF) return model;
    inlier_mask.clear();  // ransac.h leaves the member mask empty until a model beats zero inliers
    const bool on_device = dev && dev->generate && dev->fetch;
    size_t iter_max = max_iteration;
    bool first_batch = true;
    for (size_t iter0 = 0; iter0 < iter_max; iter0 += RANSAC_BATCH) {
        const size_t B = std::min(RANSAC_BATCH, iter_max - iter0);
        std::vector<std::array<size_t, DoF>> samples(B);
        for (size_t b = 0; b < B; ++b) {
            lotbox.refill_all();
            for (size_t si = 0; si < DoF; ++si) samples[b][si] = lotbox.draw_without_replacement();
        }
        std::vector<Model> models;
        std::vector<size_t> first_of(B + 1, 0), counts;
        std::vector<std::vector<char>> masks;   // host road only
        if (on_device) {
            std::vector<int32_t> flat_s(B * DoF), per_iter(B), cnt(B * 10);
            for (size_t b = 0; b < B; ++b)
                for (size_t si = 0; si < DoF; ++si) flat_s[b * DoF + si] = (int32_t)samples[b][si];
            std::vector<double> flat_m(B * 10 * 9);
            if (dev->generate(dev->user, dev->kind, (int)size, first_batch ? 1 : 0, dev->pa, dev->pb, dev->threshold, (int)B, flat_s.data(), per_iter.data(),
                              flat_m.data(), cnt.data()) != 0)   // (0 = RDVIO_OK)
                throw std::runtime_error("backend ransac_generate_score failed");
            first_batch = false;
            for (size_t b = 0; b < B; ++b) first_of[b + 1] = first_of[b] + (size_t)per_iter[b];
            models.resize(first_of[B]);
            counts.resize(first_of[B]);
            for (size_t k = 0; k < models.size(); ++k) {
                for (int q = 0; q < 9; ++q) models[k].m[q] = flat_m[9 * k + q];
                counts[k] = (size_t)cnt[k];
            }
        } else {
            for (size_t b = 0; b < B; ++b) {
                const std::vector<Model> ms = solve(samples[b]);
                models.insert(models.end(), ms.begin(), ms.end());
                first_of[b + 1] = models.size();
            }
            counts.assign(models.size(), 0);
            masks.resize(models.size());
            for (size_t k = 0; k < models.size(); ++k) {
                masks[k].assign(size, 0);
                for (size_t i = 0; i < size; ++i)
                    if (inlier(models[k], i)) {
                        counts[k]++;
                        masks[k][i] = 1;
                    }
            }
        }
        long best = -1;
        for (size_t b = 0; b < B && iter0 + b < iter_max; ++b)
            for (size_t k = first_of[b]; k < first_of[b + 1]; ++k)
                if (counts[k] > inlier_count) {
                    model = models[k];
                    inlier_count = counts[k];
                    best = (long)k;
                    const double ratio = inlier_count / (double)size;
                    const double N = K / std::log(1 - std::pow(ratio, 5));
                    if (N < (double)iter_max) iter_max = (size_t)std::ceil(N);
                }
        if (best >= 0) {
            if (on_device) {
                std::vector<uint8_t> m8(size);
                if (dev->fetch(dev->user, (int)best, m8.data()) != 0) throw std::runtime_error("backend ransac_fetch failed");
                inlier_mask.assign(m8.begin(), m8.end());
            } else {
                inlier_mask.swap(masks[(size_t)best]);
            }
        }
    }
    return model;
}

// stereo.cpp:38-66
inline M3 find_essential_matrix(const std::vector<V2> &p1, const std::vector<V2> &p2, std::vector<char> &mask, double threshold = 1.0,
                                double confidence = 0.999, size_t max_iteration = 1000, int seed = 0, RansacDevice *dev = nullptr) {
    const double t1 = 3.84;
    const double thr = 2.0 * t1 * threshold * threshold;
    auto solve = [&](const std::array<size_t, 5> &s) {
        std::array<V2, 5> a, b;
        for (int i = 0; i < 5; ++i) { a[i] = p1[s[i]]; b[i] = p2[s[i]]; }
        return solve_essential_5pt(a, b);
    };
    auto inl = [&](const M3 &E, size_t i) {
        return essential_geometric_error(E, p1[i], p2[i]) + essential_geometric_error(transpose(E), p2[i], p1[i]) <= thr;
    };
    std::vector<double> fa, fb;
    if (dev) {
        fa.resize(2 * p1.size());
        fb.resize(2 * p2.size());
        for (size_t i = 0; i < p1.size(); ++i) { fa[2 * i] = p1[i].x; fa[2 * i + 1] = p1[i].y; fb[2 * i] = p2[i].x; fb[2 * i + 1] = p2[i].y; }
        dev->kind = 0; dev->pa = fa.data(); dev->pb = fb.data(); dev->threshold = thr;
    }
    return ransac<5, M3>(p1.size(), confidence, max_iteration, seed, solve, inl, mask, M3(), dev);
}

// stereo.cpp:68-91.  The inlier test acos((R p1) . p2) <= threshold is decided as cos(threshold) <= (R p1) . p2 <= 1
// (hypo::rotation_inlier: the same predicate on the host and on the device)
inline M3 find_rotation_matrix(const std::vector<V3> &p1, const std::vector<V3> &p2, std::vector<char> &mask, double threshold = 1.0,
                               double confidence = 0.999, size_t max_iteration = 1000, int seed = 0, RansacDevice *dev = nullptr) {
    const double t2 = 5.99;
    const double cos_thr = std::cos(t2 * threshold * threshold);
    auto solve = [&](const std::array<size_t, 2> &s) {
        return std::vector<M3>{solve_rotation_2pt({p1[s[0]], p1[s[1]]}, {p2[s[0]], p2[s[1]]})};
    };
    auto inl = [&](const M3 &R, size_t i) {
        const double a[3] = {p1[i].x, p1[i].y, p1[i].z}, b[3] = {p2[i].x, p2[i].y, p2[i].z};
        return hypo::rotation_inlier(R.m, a, b, cos_thr);
    };
    std::vector<double> fa, fb;
    if (dev) {
        fa.resize(3 * p1.size());
        fb.resize(3 * p2.size());
        for (size_t i = 0; i < p1.size(); ++i) {
            fa[3 * i] = p1[i].x; fa[3 * i + 1] = p1[i].y; fa[3 * i + 2] = p1[i].z;
            fb[3 * i] = p2[i].x; fb[3 * i + 1] = p2[i].y; fb[3 * i + 2] = p2[i].z;
        }
        dev->kind = 2; dev->pa = fa.data(); dev->pb = fb.data(); dev->threshold = cos_thr;
    }
    return ransac<2, M3>(p1.size(), confidence, max_iteration, seed, solve, inl, mask, M3(), dev);
}

// stereo.h:83-93: N-view DLT.  Ps: 3x4 row-major projection matrices.  Returns the homogeneous point (smallest right
// singular vector of the 2m x 4 system = eigenvector of A^T A with the smallest eigenvalue).
inline std::array<double, 4> triangulate_point(const std::vector<std::array<double, 12>> &Ps, const std::vector<V3> &points) {
    double AtA[16] = {0};
    for (size_t i = 0; i < points.size(); ++i) {
        const double *P = Ps[i].data();
        const double pt[3] = {points[i].x, points[i].y, points[i].z};
        for (int rrow = 0; rrow < 2; ++rrow) {
            double row[4];
            for (int c = 0; c < 4; ++c) row[c] = pt[rrow] * P[8 + c] - pt[2] * P[4 * rrow + c];
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) AtA[4 * a + b] += row[a] * row[b];
        }
    }
    double V[16], lam[4];
    sym_eigen(4, AtA, V, lam);
    const int k = ascending_order(4, lam)[0];
    return {V[0 * 4 + k], V[1 * 4 + k], V[2 * 4 + k], V[3 * 4 + k]};
}

// PoissonDiskFilter<2> (/root/reference/src/rdvio_util/include/rdvio/util/poisson_disk_filter.h:8-112), kept literally:
// a sparse grid with ONE point index per cell (later points overwrite, :20-24) and the reference's neighbourhood walk,
// which skips the first cell of the window and visits one cell past its end (:77-92).
class PoissonDisk2 {
  public:
    explicit PoissonDisk2(double radius) : r2_(radius * radius), cell_(radius / std::sqrt(2.0)), span_((int)std::ceil(std::sqrt(2.0))) {}
    void preset_point(const V2 &p) {
        grid_[key(ix(p.x), ix(p.y))] = pts_.size();
        pts_.push_back(p);
    }
    bool permit_point(const V2 &p) const {
        const int cx = ix(p.x), cy = ix(p.y);
        const int bx = cx - span_, by = cy - span_, ex = cx + span_, ey = cy + span_;
        int x0 = bx, y0 = by;
        while (y0 <= ey) {
            ++x0;
            if (x0 > ex) {
                x0 = bx;
                ++y0;
            }
            auto it = grid_.find(key(x0, y0));
            if (it != grid_.end() && sqnorm(p - pts_[it->second]) < r2_) return false;
        }
        return true;
    }

  private:
    int ix(double v) const { return (int)std::floor(v / cell_); }
    static int64_t key(int x, int y) { return ((int64_t)x << 32) ^ (uint32_t)y; }
    double r2_, cell_;
    int span_;
    std::vector<V2> pts_;
    std::unordered_map<int64_t, size_t> grid_;
};

// ------------------------------------------------------------------------------------------------------------------
// pieces used by the SfM / IMU initializer (initializer.cpp)
// ------------------------------------------------------------------------------------------------------------------
// lie_algebra.h:18-21: Eigen::AngleAxis(q) -> angle * axis
inline V3 logmap(const Q4 &q) {
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (n == 0.0) return {0, 0, 0};
    const double angle = 2.0 * std::atan2(n, std::fabs(q.w));
    if (q.w < 0) n = -n;
    return {angle * q.x / n, angle * q.y / n, angle * q.z / n};
}
inline M3 inverse3(const M3 &A) {
    const double d = det(A);
    M3 I;
    I.m[0] = (A.m[4] * A.m[8] - A.m[5] * A.m[7]) / d; I.m[1] = (A.m[2] * A.m[7] - A.m[1] * A.m[8]) / d; I.m[2] = (A.m[1] * A.m[5] - A.m[2] * A.m[4]) / d;
    I.m[3] = (A.m[5] * A.m[6] - A.m[3] * A.m[8]) / d; I.m[4] = (A.m[0] * A.m[8] - A.m[2] * A.m[6]) / d; I.m[5] = (A.m[2] * A.m[3] - A.m[0] * A.m[5]) / d;
    I.m[6] = (A.m[3] * A.m[7] - A.m[4] * A.m[6]) / d; I.m[7] = (A.m[1] * A.m[6] - A.m[0] * A.m[7]) / d; I.m[8] = (A.m[0] * A.m[4] - A.m[1] * A.m[3]) / d;
    return I;
}
// quaternion of a rotation matrix (Eigen's Quaternion(Matrix3) recipe)
inline Q4 from_mat(const M3 &R) {
    const double t = R.m[0] + R.m[4] + R.m[8];
    Q4 q;
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        q.w = 0.5 * s;
        s = 0.5 / s;
        q.x = (R.m[7] - R.m[5]) * s; q.y = (R.m[2] - R.m[6]) * s; q.z = (R.m[3] - R.m[1]) * s;
    } else {
        int i = 0;
        if (R.m[4] > R.m[0]) i = 1;
        if (R.m[8] > R.m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R.m[4 * i] - R.m[4 * j] - R.m[4 * k] + 1.0);
        double v[3];
        v[i] = 0.5 * s;
        s = 0.5 / s;
        q.w = (R.m[3 * k + j] - R.m[3 * j + k]) * s;
        v[j] = (R.m[3 * j + i] + R.m[3 * i + j]) * s;
        v[k] = (R.m[3 * k + i] + R.m[3 * i + k]) * s;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}
// Eigen::Quaternion::FromTwoVectors
inline Q4 from_two_vectors(const V3 &a, const V3 &b) {
    const V3 v0 = normalized(a), v1 = normalized(b);
    const double c = dot(v1, v0);
    if (c < -1.0 + 1e-12) {  // nearly opposite: any axis orthogonal to v0
        V3 axis = cross(v0, std::fabs(v0.x) < 0.9 ? V3{1, 0, 0} : V3{0, 1, 0});
        axis = normalized(axis);
        return {axis.x, axis.y, axis.z, 0.0};
    }
    const V3 axis = cross(v0, v1);
    const double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
    return {axis.x * invs, axis.y * invs, axis.z * invs, s * 0.5};
}

// minimum-norm least squares x = argmin |A x - b| (rows x cols, row-major) through the eigen-decomposition of A^T A
// (stands in for fullPivHouseholderQr().solve / JacobiSVD::solve)
inline std::vector<double> least_squares(int rows, int cols, const std::vector<double> &A, const std::vector<double> &b) {
    std::vector<double> AtA((size_t)cols * cols, 0.0), Atb(cols, 0.0), V((size_t)cols * cols), lam(cols);
    for (int r = 0; r < rows; ++r)
        for (int i = 0; i < cols; ++i) {
            const double ai = A[(size_t)r * cols + i];
            if (ai == 0.0) continue;
            Atb[i] += ai * b[r];
            for (int j = 0; j < cols; ++j) AtA[(size_t)i * cols + j] += ai * A[(size_t)r * cols + j];
        }
    sym_eigen(cols, AtA.data(), V.data(), lam.data());
    double lmax = 0.0;
    for (double l : lam) lmax = std::max(lmax, l);
    std::vector<double> x(cols, 0.0);
    for (int k = 0; k < cols; ++k) {
        if (!(lam[k] > 1e-13 * lmax)) continue;
        double c = 0.0;
        for (int i = 0; i < cols; ++i) c += V[(size_t)i * cols + k] * Atb[i];
        c /= lam[k];
        for (int i = 0; i < cols; ++i) x[i] += c * V[(size_t)i * cols + k];
    }
    return x;
}

// homography.h:16-20
inline double homography_geometric_error(const M3 &H, const V2 &p1, const V2 &p2) {
    const V3 q = H * V3{p1.x, p1.y, 1.0};
    return sqnorm(p2 - V2{q.x / q.z, q.y / q.z});
}

// homography.cpp:89-158: normalised 4-point DLT
inline M3 solve_homography_4pt(const std::array<V2, 4> &p1, const std::array<V2, 4> &p2) {
    const double sqrt2 = std::sqrt(2.0);
    V2 ma{0, 0}, mb{0, 0};
    for (int i = 0; i < 4; ++i) { ma.x += p1[i].x; ma.y += p1[i].y; mb.x += p2[i].x; mb.y += p2[i].y; }
    ma.x /= 4; ma.y /= 4; mb.x /= 4; mb.y /= 4;
    double sa = 0, sb = 0;
    for (int i = 0; i < 4; ++i) { sa += norm(p1[i] - ma); sb += norm(p2[i] - mb); }
    sa = 1.0 / (sqrt2 * sa);
    sb = 1.0 / (sqrt2 * sb);
    double A[8][9] = {{0}};
    for (int i = 0; i < 4; ++i) {
        const V2 a{(p1[i].x - ma.x) * sa, (p1[i].y - ma.y) * sa}, b{(p2[i].x - mb.x) * sb, (p2[i].y - mb.y) * sb};
        A[2 * i][1] = -a.x; A[2 * i][2] = a.x * b.y; A[2 * i][4] = -a.y; A[2 * i][5] = a.y * b.y; A[2 * i][7] = -1; A[2 * i][8] = b.y;
        A[2 * i + 1][0] = a.x; A[2 * i + 1][2] = -a.x * b.x; A[2 * i + 1][3] = a.y; A[2 * i + 1][5] = -a.y * b.x; A[2 * i + 1][6] = 1; A[2 * i + 1][8] = -b.x;
    }
    double AtA[81], V[81], lam[9];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            double s = 0;
            for (int k = 0; k < 8; ++k) s += A[k][i] * A[k][j];
            AtA[9 * i + j] = s;
        }
    sym_eigen(9, AtA, V, lam);
    const int kmin = ascending_order(9, lam)[0];
    M3 NH;  // to_matrix: column c = segment c
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) NH.m[3 * r + c] = V[9 * (3 * c + r) + kmin];
    const M3 Nb{{1 / sb, 0, mb.x, 0, 1 / sb, mb.y, 0, 0, 1}}, Na{{sa, 0, -sa * ma.x, 0, sa, -sa * ma.y, 0, 0, 1}};
    return Nb * NH * Na;
}

// stereo.cpp:93-120
inline M3 find_homography_matrix(const std::vector<V2> &p1, const std::vector<V2> &p2, std::vector<char> &mask, double threshold, double confidence,
                                 size_t max_iteration, int seed) {
    const double t2 = 5.99;
    auto solve = [&](const std::array<size_t, 4> &s) {
        return std::vector<M3>{solve_homography_4pt({p1[s[0]], p1[s[1]], p1[s[2]], p1[s[3]]}, {p2[s[0]], p2[s[1]], p2[s[2]], p2[s[3]]})};
    };
    const double thr = 2.0 * t2 * threshold * threshold;
    auto inl = [&](const M3 &H, size_t i) { return homography_geometric_error(H, p1[i], p2[i]) + homography_geometric_error(inverse3(H), p2[i], p1[i]) <= thr; };
    return ransac<4, M3>(p1.size(), confidence, max_iteration, seed, solve, inl, mask);
}

// homography.cpp:5-87.  Returns false for a pure rotation.
inline bool decompose_homography(const M3 &H, M3 &R1, M3 &R2, V3 &T1, V3 &T2, V3 &n1, V3 &n2) {
    M3 U, V;
    double sv[3];
    svd3(H, U, sv, V);
    M3 Hn;
    for (int i = 0; i < 9; ++i) Hn.m[i] = H.m[i] / sv[1];
    M3 S = transpose(Hn) * Hn;
    S.m[0] -= 1; S.m[4] -= 1; S.m[8] -= 1;
    bool pure_rotation = true;
    for (int i = 0; i < 9 && pure_rotation; ++i)
        if (std::fabs(S.m[i]) > 1e-3) pure_rotation = false;
    auto Sx = [&](int i, int j) { return S.m[3 * i + j]; };
    if (pure_rotation) {
        R1 = U * transpose(V);
        if (det(R1) < 0)
            for (double &v : R1.m) v = -v;
        R2 = R1;
        T1 = T2 = n1 = n2 = V3{0, 0, 0};
        return false;
    }
    const double Ms00 = Sx(1, 2) * Sx(1, 2) - Sx(1, 1) * Sx(2, 2), Ms11 = Sx(0, 2) * Sx(0, 2) - Sx(0, 0) * Sx(2, 2),
                 Ms22 = Sx(0, 1) * Sx(0, 1) - Sx(0, 0) * Sx(1, 1);
    const double r00 = std::sqrt(Ms00), r11 = std::sqrt(Ms11), r22 = std::sqrt(Ms22);
    const double tr = Sx(0, 0) + Sx(1, 1) + Sx(2, 2);
    const double nu = 2.0 * std::sqrt(1 + tr - Ms00 - Ms11 - Ms22);
    const double tenormsq = 2 + tr - nu;
    V3 ts1, ts2;
    if (Sx(0, 0) > Sx(1, 1) && Sx(0, 0) > Sx(2, 2)) {
        const double e = ((Sx(0, 1) * Sx(0, 2) - Sx(0, 0) * Sx(1, 2)) < 0) ? -1 : 1;
        n1 = {Sx(0, 0), Sx(0, 1) + r22, Sx(0, 2) + e * r11};
        n2 = {Sx(0, 0), Sx(0, 1) - r22, Sx(0, 2) - e * r11};
        ts1 = norm(n1) * n2 / Sx(0, 0);
        ts2 = norm(n2) * n1 / Sx(0, 0);
    } else if (Sx(1, 1) > Sx(0, 0) && Sx(1, 1) > Sx(2, 2)) {
        const double e = ((Sx(1, 1) * Sx(0, 2) - Sx(0, 1) * Sx(1, 2)) < 0) ? -1 : 1;
        n1 = {Sx(0, 1) + r22, Sx(1, 1), Sx(1, 2) - e * r00};
        n2 = {Sx(0, 1) - r22, Sx(1, 1), Sx(1, 2) + e * r00};
        ts2 = norm(n2) * n1 / Sx(1, 1);
        ts1 = norm(n1) * n2 / Sx(1, 1);
    } else {
        const double e = ((Sx(1, 2) * Sx(0, 2) - Sx(0, 1) * Sx(2, 2)) < 0) ? -1 : 1;
        n1 = {Sx(0, 2) + e * r11, Sx(1, 2) + r00, Sx(2, 2)};
        n2 = {Sx(0, 2) - e * r11, Sx(1, 2) - r00, Sx(2, 2)};
        ts1 = norm(n1) * n2 / Sx(2, 2);
        ts2 = norm(n2) * n1 / Sx(2, 2);
    }
    n1 = normalized(n1);
    n2 = normalized(n2);
    ts1 = ts1 - tenormsq * n1;
    ts2 = ts2 - tenormsq * n2;
    auto outer_sub = [&](const V3 &t, const V3 &n) {  // I - (t / nu) n^T
        M3 M;
        const double tv[3] = {t.x / nu, t.y / nu, t.z / nu}, nv[3] = {n.x, n.y, n.z};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) M.m[3 * i + j] = (i == j ? 1.0 : 0.0) - tv[i] * nv[j];
        return M;
    };
    R1 = Hn * outer_sub(ts1, n1);
    R2 = Hn * outer_sub(ts2, n2);
    T1 = R1 * (0.5 * ts1);
    T2 = R2 * (0.5 * ts2);
    return true;
}

// essential.cpp:266-284 (JacobiSVD branch)
inline void decompose_essential(const M3 &E, M3 &R1, M3 &R2, V3 &T) {
    M3 U, V;
    double sv[3];
    svd3(E, U, sv, V);
    M3 VT = transpose(V);
    if (det(U) < 0)
        for (double &v : U.m) v = -v;
    if (det(VT) < 0)
        for (double &v : VT.m) v = -v;
    const M3 W{{0, 1, 0, -1, 0, 0, 0, 0, 1}};
    R1 = U * W * VT;
    R2 = U * transpose(W) * VT;
    T = {U.m[2], U.m[5], U.m[8]};
}

}  // namespace rdvio_pipe
