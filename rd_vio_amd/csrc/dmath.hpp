// FP64 device math for the estimation kernels: 3-vectors, quaternions (x,y,z,w), 3x3 row-major
// matrices and the Lie-algebra helpers of the reference
// (/root/reference/src/rdvio_geometry/include/rdvio/geometry/lie_algebra.h:6-21,
//  /root/reference/src/rdvio_geometry/src/lie_algebra.cpp:5-56).
#pragma once
#include <hip/hip_runtime.h>

#define DM __device__ __forceinline__

struct V3 {
    double x, y, z;
};
struct Q4 {
    double x, y, z, w;
};
struct M3 {
    double m[9];
};

DM V3 v3(double x, double y, double z) { return V3{x, y, z}; }
DM V3 v3_load(const double *p) { return V3{p[0], p[1], p[2]}; }
DM void v3_store(double *p, const V3 &a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
DM V3 operator+(const V3 &a, const V3 &b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DM V3 operator-(const V3 &a, const V3 &b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DM V3 operator-(const V3 &a) { return V3{-a.x, -a.y, -a.z}; }
DM V3 operator*(double s, const V3 &a) { return V3{s * a.x, s * a.y, s * a.z}; }
DM V3 operator*(const V3 &a, double s) { return V3{s * a.x, s * a.y, s * a.z}; }
DM V3 operator/(const V3 &a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
DM double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DM V3 cross(const V3 &a, const V3 &b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DM double norm(const V3 &a) { return sqrt(dot(a, a)); }

DM Q4 q_load(const double *p) { return Q4{p[0], p[1], p[2], p[3]}; }
DM void q_store(double *p, const Q4 &q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
DM Q4 q_identity() { return Q4{0, 0, 0, 1}; }
DM Q4 conj(const Q4 &q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
DM Q4 operator*(const Q4 &a, const Q4 &b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
DM Q4 normalized(const Q4 &q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return Q4{q.x / n, q.y / n, q.z / n, q.w / n};
}
// q * v (Eigen's Quaternion::_transformVector)
DM V3 rot(const Q4 &q, const V3 &v) {
    V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}
DM V3 rot_inv(const Q4 &q, const V3 &v) { return rot(conj(q), v); }

DM M3 to_mat(const Q4 &q) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3 R;
    R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
    R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
    R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
    return R;
}
DM M3 m3_identity() {
    M3 I;
#pragma unroll
    for (int i = 0; i < 9; ++i) I.m[i] = 0;
    I.m[0] = I.m[4] = I.m[8] = 1;
    return I;
}
DM M3 transpose(const M3 &A) {
    M3 T;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T.m[i * 3 + j] = A.m[j * 3 + i];
    return T;
}
DM M3 operator*(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C.m[i * 3 + j] = A.m[i * 3] * B.m[j] + A.m[i * 3 + 1] * B.m[3 + j] + A.m[i * 3 + 2] * B.m[6 + j];
    return C;
}
DM V3 operator*(const M3 &A, const V3 &v) {
    return V3{A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
              A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
DM M3 operator*(double s, const M3 &A) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 9; ++i) C.m[i] = s * A.m[i];
    return C;
}
DM M3 operator+(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] + B.m[i];
    return C;
}
DM M3 operator-(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] - B.m[i];
    return C;
}
DM M3 m3_load(const double *p) {
    M3 A;
#pragma unroll
    for (int i = 0; i < 9; ++i) A.m[i] = p[i];
    return A;
}
// lie_algebra.h:6-9
DM M3 hat(const V3 &w) {
    M3 H;
    H.m[0] = 0;    H.m[1] = -w.z; H.m[2] = w.y;
    H.m[3] = w.z;  H.m[4] = 0;    H.m[5] = -w.x;
    H.m[6] = -w.y; H.m[7] = w.x;  H.m[8] = 0;
    return H;
}
// sin and cos for the rotation algebra.  The arguments here are rotation increments and residual angles -- almost always
// far inside [-pi/4, pi/4], where the two kernel polynomials of fdlibm (k_sin.c / k_cos.c, < 1 ulp) need no argument
// reduction.  The library routines carry a Payne-Hanek
// reduction that is never taken here but is executed around (two calls, sin and cos, each with its own reduction and
// range tests); they remain the path for larger arguments.  (Inlined: a call would turn the caller into a non-leaf
// function that saves its callee-saved registers to scratch.)
DM double sin_large(double x) { return sin(x); }
DM double cos_large(double x) { return cos(x); }
DM void sincos_rot(double x, double &sn, double &cs) {
    if (fabs(x) <= 0.7853981633974483) {
        const double z = x * x;
        {
            const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                         S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
            const double v = z * x;
            const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
            sn = x + v * (S1 + z * r);
        }
        {
            const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                         C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
            const double w = z * z;
            const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
            const double hz = 0.5 * z, a = 1.0 - hz;
            cs = a + (((1.0 - a) - hz) + z * r);
        }
    } else {
        sn = sin_large(x);
        cs = cos_large(x);
    }
}

// lie_algebra.h:11-16
DM Q4 expmap(const V3 &w) {
    double n = norm(w);
    if (n > 0) {
        double sh, ch;
        sincos_rot(0.5 * n, sh, ch);
        const double s = sh / n;
        return Q4{w.x * s, w.y * s, w.z * s, ch};
    }
    return q_identity();
}
// lie_algebra.h:18-21 (Eigen AngleAxis(q))
DM V3 logmap(const Q4 &q) {
    double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (n != 0) {
        double angle = 2.0 * atan2(n, fabs(q.w));
        if (q.w < 0) n = -n;
        double s = angle / n;
        return V3{q.x * s, q.y * s, q.z * s};
    }
    return V3{0, 0, 0};
}
// lie_algebra.cpp:5-45
DM M3 right_jacobian(const V3 &w) {
    const double root2_eps = 1.4901161193847656e-08;   // sqrt(DBL_EPSILON)
    const double root4_eps = 1.220703125e-04;          // sqrt(sqrt(DBL_EPSILON))
    const double qdrt720 = 5.180044444112967;          // 720^(1/4)
    const double qdrt5040 = 8.425717895368383;         // 5040^(1/4)
    const double sqrt24 = 4.898979485566356;
    const double sqrt120 = 10.954451150103322;
    double angle = norm(w);
    double cangle, sangle;
    sincos_rot(angle, sangle, cangle);
    double angle2 = angle * angle;
    double cos_term, sin_term;
    if (angle > root4_eps * qdrt720) {
        cos_term = (1 - cangle) / angle2;
    } else {
        cos_term = 0.5;
        if (angle > root2_eps * sqrt24) cos_term -= angle2 / 24.0;
    }
    if (angle > root4_eps * qdrt5040) {
        sin_term = (angle - sangle) / (angle * angle2);
    } else {
        sin_term = 1.0 / 6.0;
        if (angle > root2_eps * sqrt120) sin_term -= angle2 / 120.0;
    }
    M3 H = hat(w);
    M3 H2 = H * H;
    M3 J;
#pragma unroll
    for (int i = 0; i < 9; ++i) J.m[i] = -cos_term * H.m[i] + sin_term * H2.m[i];
    J.m[0] += 1; J.m[4] += 1; J.m[8] += 1;
    return J;
}
// closed-form 3x3 inverse (adjugate), what Eigen uses for fixed 3x3
DM M3 inverse3(const M3 &A) {
    const double *a = A.m;
    double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    double id = 1.0 / det;
    M3 R;
    R.m[0] = c00 * id; R.m[1] = (a[2] * a[7] - a[1] * a[8]) * id; R.m[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    R.m[3] = c01 * id; R.m[4] = (a[0] * a[8] - a[2] * a[6]) * id; R.m[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    R.m[6] = c02 * id; R.m[7] = (a[1] * a[6] - a[0] * a[7]) * id; R.m[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return R;
}
