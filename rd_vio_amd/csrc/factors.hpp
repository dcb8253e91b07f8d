// Device-side factor evaluation (FP64), shared by the stand-alone evaluation kernels and the solver.
// Each function cites the reference code it follows (paths relative to /root/reference).
#pragma once
#include "dmath.hpp"

enum { ST_Q = 0, ST_P = 4, ST_V = 7, ST_BG = 10, ST_BA = 13 };
enum { ES_Q = 0, ES_P = 3, ES_V = 6, ES_BG = 9, ES_BA = 12 };
enum { PRE_T = 0, PRE_Q = 1, PRE_P = 5, PRE_V = 8, PRE_COV = 11, PRE_SIC = 236, PRE_JAC = 461 };
enum { EX_CQ = 0, EX_CP = 4, EX_IQ = 7, EX_IP = 11 };

// QuaternionParameterization::Plus on a 16-double frame state
// (src/rdvio_estimation/include/rdvio/estimation/ceres/quaternion_parameterization.h:11-17)
DM void state_plus(const double *s, const double *d15, double *o) {
    Q4 q = normalized(q_load(s) * expmap(v3_load(d15)));
    q_store(o, q);
#pragma unroll
    for (int i = 0; i < 12; ++i) o[4 + i] = s[4 + i] + d15[3 + i];
}

// CeresReprojectionErrorFactor::Evaluate
// (src/rdvio_estimation/include/rdvio/estimation/ceres/reprojection_factor.h:24-89).
// Jt/Jr: 2x6 row-major (theta, p) of the target / anchor frame, Jd: 2 (inverse depth).
template <bool WITH_JAC>
DM void reprojection_factor(const double *__restrict__ st, const double *__restrict__ sr, const double *__restrict__ T9,
                            const double *__restrict__ zref3, double rho, const double *__restrict__ extr,
                            const double *__restrict__ W, double *r, double *Jt, double *Jr, double *Jd) {
    const Q4 qcs = q_load(extr + EX_CQ);
    const V3 pcs = v3_load(extr + EX_CP);
    const Q4 q_t = q_load(st + ST_Q), q_r = q_load(sr + ST_Q);
    const V3 p_t = v3_load(st + ST_P), p_r = v3_load(sr + ST_P);
    const M3 T = m3_load(T9);
    const V3 z_ref = v3_load(zref3);
    const V3 y_ref = z_ref / rho;
    const V3 y_rc = rot(qcs, y_ref) + pcs;
    const V3 x = rot(q_r, y_rc) + p_r;
    const V3 y_tc = rot_inv(q_t, x - p_t);
    const V3 y_t = rot_inv(qcs, y_tc - pcs);
    const M3 Tt = transpose(T);
    const V3 u = Tt * y_t;
    const double h0 = u.x / u.z, h1 = u.y / u.z;
    const double w00 = W[0], w01 = W[1], w10 = W[2], w11 = W[3];
    r[0] = w00 * h0 + w01 * h1;
    r[1] = w10 * h0 + w11 * h1;
    if (!WITH_JAC) return;
    const double iz = 1.0 / u.z, z2 = u.z * u.z;
    const double d02 = -u.x / z2, d12 = -u.y / z2;
    const double wd[6] = {w00 * iz, w01 * iz, w00 * d02 + w01 * d12, w10 * iz, w11 * iz, w10 * d02 + w11 * d12};
    double A[6], Bm[6], C[6], D[6], Mt[6], Mr[6];
    // (Jacobian chain with explicit FMAs: the linearisation is issue-bound, and nothing downstream relies on an exact
    // cancellation in J; the residual above keeps the reference's unfused arithmetic)
#define RDVIO_MUL23(OUT, IN, MAT)                                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 3; ++j) OUT[i * 3 + j] = \
        __builtin_fma(IN[i * 3 + 2], (MAT).m[6 + j], __builtin_fma(IN[i * 3 + 1], (MAT).m[3 + j], IN[i * 3] * (MAT).m[j]));
    RDVIO_MUL23(A, wd, Tt)
    const M3 Rcs = to_mat(qcs);
    const M3 RcsT = transpose(Rcs);
    const M3 RtT = transpose(to_mat(q_t));
    const M3 Rr = to_mat(q_r);
    RDVIO_MUL23(Bm, A, RcsT)  // dr/dy_tgt_center
    RDVIO_MUL23(C, Bm, RtT)   // dr/dx
    RDVIO_MUL23(D, C, Rr)     // dr/dy_ref_center
    const M3 Ht = hat(y_tc), Hr = hat(y_rc);
    RDVIO_MUL23(Mt, Bm, Ht)
    RDVIO_MUL23(Mr, D, Hr)
#undef RDVIO_MUL23
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jt[i * 6 + j] = Mt[i * 3 + j];
            Jt[i * 6 + 3 + j] = -C[i * 3 + j];
            Jr[i * 6 + j] = -Mr[i * 3 + j];
            Jr[i * 6 + 3 + j] = C[i * 3 + j];
        }
    const V3 t3 = Rcs * y_ref;
    Jd[0] = -(D[0] * t3.x + D[1] * t3.y + D[2] * t3.z) / rho;
    Jd[1] = -(D[3] * t3.x + D[4] * t3.y + D[5] * t3.z) / rho;
}

// Residual of CeresReprojectionErrorFactor::Evaluate (reprojection_factor.h:24-49) from per-frame camera poses
// (R_wc row-major 3x3 | c_w, see camera_pose_of) instead of per-factor quaternion algebra: the cost-only evaluation of a
// trust-region step is issue-bound on one CU, and this form needs ~150 instead of ~350 FP64 instructions per factor.
// Same mathematics as reprojection_factor<false>, different rounding (one reciprocal instead of three divisions by rho).
DM void camera_pose_of(const double *__restrict__ s16, const double *__restrict__ extr, double *__restrict__ cam12) {
    const Q4 q = q_load(s16 + ST_Q);
    const M3 Rwc = to_mat(q * q_load(extr + EX_CQ));
    const V3 c = v3_load(s16 + ST_P) + rot(q, v3_load(extr + EX_CP));
#pragma unroll
    for (int i = 0; i < 9; ++i) cam12[i] = Rwc.m[i];
    cam12[9] = c.x; cam12[10] = c.y; cam12[11] = c.z;
}
DM void reprojection_residual(const double *__restrict__ cam_t, const double *__restrict__ cam_r, const double *__restrict__ T9,
                              const double *__restrict__ zref3, double rho, const double *__restrict__ W, double *r) {
    const double inv = 1.0 / rho;
    const double y0 = zref3[0] * inv, y1 = zref3[1] * inv, y2 = zref3[2] * inv;
    // world point, relative to the target camera centre
    const double dx = cam_r[0] * y0 + cam_r[1] * y1 + cam_r[2] * y2 + cam_r[9] - cam_t[9];
    const double dy = cam_r[3] * y0 + cam_r[4] * y1 + cam_r[5] * y2 + cam_r[10] - cam_t[10];
    const double dz = cam_r[6] * y0 + cam_r[7] * y1 + cam_r[8] * y2 + cam_r[11] - cam_t[11];
    // into the target camera (R_wc^T), then into the tangent frame of the observation (T^T)
    const double t0 = cam_t[0] * dx + cam_t[3] * dy + cam_t[6] * dz;
    const double t1 = cam_t[1] * dx + cam_t[4] * dy + cam_t[7] * dz;
    const double t2 = cam_t[2] * dx + cam_t[5] * dy + cam_t[8] * dz;
    const double u0 = T9[0] * t0 + T9[3] * t1 + T9[6] * t2;
    const double u1 = T9[1] * t0 + T9[4] * t1 + T9[7] * t2;
    const double u2 = T9[2] * t0 + T9[5] * t1 + T9[8] * t2;
    const double iz = 1.0 / u2;
    const double h0 = u0 * iz, h1 = u1 * iz;
    r[0] = W[0] * h0 + W[1] * h1;
    r[1] = W[2] * h0 + W[3] * h1;
}

// CeresRotationPriorFactor::Evaluate (src/rdvio_estimation/include/rdvio/estimation/ceres/rotation_factor.h:22-58);
// J: 2x3 row-major wrt theta of the target frame.
template <bool WITH_JAC>
DM void rotation_prior_factor(const double *q_tgt4, const double *q_ref4, const double *zref3, const double *T9,
                              const double *extr, const double *W, double *r, double *J) {
    const Q4 qcs = q_load(extr + EX_CQ);
    const V3 pcs = v3_load(extr + EX_CP);
    const Q4 q_t = q_load(q_tgt4), q_r = q_load(q_ref4);
    const V3 z_rc = rot(qcs, v3_load(zref3)) + pcs;  // translation added to a bearing: reference quirk (:34)
    const V3 z_tc = rot_inv(q_t, rot(q_r, z_rc));
    const V3 z_t = rot_inv(qcs, z_tc - pcs);
    const M3 Tt = transpose(m3_load(T9));
    const V3 u = Tt * z_t;
    const double h0 = u.x / u.z, h1 = u.y / u.z;
    r[0] = W[0] * h0 + W[1] * h1;
    r[1] = W[2] * h0 + W[3] * h1;
    if (!WITH_JAC) return;
    const double iz = 1.0 / u.z, z2 = u.z * u.z;
    const double d02 = -u.x / z2, d12 = -u.y / z2;
    const double wd[6] = {W[0] * iz, W[1] * iz, W[0] * d02 + W[1] * d12, W[2] * iz, W[3] * iz, W[2] * d02 + W[3] * d12};
    const M3 RcsT = transpose(to_mat(qcs));
    const M3 H = hat(z_tc);
    double A[6], Bm[6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i * 3 + j] = wd[i * 3] * Tt.m[j] + wd[i * 3 + 1] * Tt.m[3 + j] + wd[i * 3 + 2] * Tt.m[6 + j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Bm[i * 3 + j] = A[i * 3] * RcsT.m[j] + A[i * 3 + 1] * RcsT.m[3 + j] + A[i * 3 + 2] * RcsT.m[6 + j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) J[i * 3 + j] = Bm[i * 3] * H.m[j] + Bm[i * 3 + 1] * H.m[3 + j] + Bm[i * 3 + 2] * H.m[6 + j];
}

DM void blk3_set(double *M, int ld, int r0, int c0, const M3 &B, double s) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[(r0 + i) * ld + c0 + j] = s * B.m[i * 3 + j];
}

// rotation rows of the preintegration Jacobian (preintegration_factor.h:84-86, 111-113, 131-133)
__device__ __attribute__((noinline)) void preintegration_rotation_blocks(const double *si, const double *sj, const double *pre,
                                                                          const double *extr, V3 th, V3 r_th, double *Gi, double *Gj) {
    const Q4 q_ci = q_load(si + ST_Q), iq = q_load(extr + EX_IQ);
    const M3 Jrinv = inverse3(right_jacobian(r_th));
    {
        const Q4 q_j = q_load(sj + ST_Q) * iq;
        blk3_set(Gi, 15, ES_Q, ES_Q, Jrinv * (to_mat(conj(q_j)) * to_mat(q_ci)), -1.0);
    }
    blk3_set(Gj, 15, ES_Q, ES_Q, Jrinv * to_mat(conj(iq)), 1.0);
    const M3 ERt = to_mat(conj(expmap(r_th)));
    blk3_set(Gi, 15, ES_Q, ES_BG, ((Jrinv * ERt) * right_jacobian(th)) * m3_load(pre + PRE_JAC), -1.0);
}

// position / velocity rows against the two poses and velocities (preintegration_factor.h:88-109, 135-147)
__device__ __attribute__((noinline)) void preintegration_translation_blocks(const double *si, const double *sj, const double *extr,
                                                                             double dt, double *Gi, double *Gj) {
    const V3 g = v3(0, 0, -9.80665);
    const Q4 q_ci = q_load(si + ST_Q), q_cj = q_load(sj + ST_Q), iq = q_load(extr + EX_IQ);
    const V3 ip = v3_load(extr + EX_IP);
    const V3 p_ci = v3_load(si + ST_P), v_i = v3_load(si + ST_V), v_j = v3_load(sj + ST_V);
    const V3 p_j = v3_load(sj + ST_P) + rot(q_cj, ip);
    {
        const M3 IqT = to_mat(conj(iq)), RciT = transpose(to_mat(q_ci));
        blk3_set(Gi, 15, ES_P, ES_Q, IqT * hat(RciT * (p_j - p_ci - dt * v_i - (0.5 * dt * dt) * g)), 1.0);
        blk3_set(Gi, 15, ES_V, ES_Q, IqT * hat(RciT * (v_j - v_i - dt * g)), 1.0);
    }
    const M3 RiT = to_mat(conj(q_ci * iq));
    blk3_set(Gi, 15, ES_P, ES_P, RiT, -1.0);
    blk3_set(Gi, 15, ES_P, ES_V, RiT, -dt);
    blk3_set(Gi, 15, ES_V, ES_V, RiT, -1.0);
    blk3_set(Gj, 15, ES_P, ES_Q, (RiT * to_mat(q_cj)) * hat(ip), -1.0);
    blk3_set(Gj, 15, ES_P, ES_P, RiT, 1.0);
    blk3_set(Gj, 15, ES_V, ES_V, RiT, 1.0);
}

// translation block group + the bias columns (the preintegration's own Jacobians and identities): everything of the
// Jacobian that does not depend on the residual
__device__ __attribute__((noinline)) void preintegration_translation_bias_blocks(const double *si, const double *sj, const double *pre,
                                                                                const double *extr, double *Gi, double *Gj) {
    preintegration_translation_blocks(si, sj, extr, pre[PRE_T], Gi, Gj);
    const M3 I3 = m3_identity();
    blk3_set(Gi, 15, ES_P, ES_BG, m3_load(pre + PRE_JAC + 9), -1.0);
    blk3_set(Gi, 15, ES_V, ES_BG, m3_load(pre + PRE_JAC + 27), -1.0);
    blk3_set(Gi, 15, ES_BG, ES_BG, I3, -1.0);
    blk3_set(Gi, 15, ES_P, ES_BA, m3_load(pre + PRE_JAC + 18), -1.0);
    blk3_set(Gi, 15, ES_V, ES_BA, m3_load(pre + PRE_JAC + 36), -1.0);
    blk3_set(Gi, 15, ES_BA, ES_BA, I3, -1.0);
    blk3_set(Gj, 15, ES_BG, ES_BG, I3, 1.0);
    blk3_set(Gj, 15, ES_BA, ES_BA, I3, 1.0);
}

// CeresPreIntegrationErrorFactor::Evaluate, UNWHITENED part
// (src/rdvio_estimation/include/rdvio/estimation/ceres/preintegration_factor.h:19-153):
// e (15) and, if requested, the 15x15 blocks Gi, Gj (tangent columns; must be zero-initialised by the caller)
// before the left multiplication by delta.sqrt_inv_cov (:155 and the per-block products).
// ROT_ONLY: the translation / bias block groups are left to preintegration_translation_bias_blocks (the solver runs them
// on otherwise idle lanes of another wavefront, in parallel with this function's residual and rotation rows).
template <bool WITH_JAC, bool ROT_ONLY = false>
__device__ __attribute__((noinline)) void preintegration_unwhitened(const double *si, const double *sj, const double *pre, const double *bias_lin,
                                  const double *extr, double *e, double *Gi, double *Gj) {
    const V3 g = v3(0, 0, -9.80665);
    const Q4 q_ci = q_load(si + ST_Q), q_cj = q_load(sj + ST_Q);
    const V3 p_ci = v3_load(si + ST_P), v_i = v3_load(si + ST_V), bg_i = v3_load(si + ST_BG), ba_i = v3_load(si + ST_BA);
    const V3 p_cj = v3_load(sj + ST_P), v_j = v3_load(sj + ST_V), bg_j = v3_load(sj + ST_BG), ba_j = v3_load(sj + ST_BA);
    const Q4 iq = q_load(extr + EX_IQ);
    const V3 ip = v3_load(extr + EX_IP);
    const Q4 q_i = q_ci * iq, q_j = q_cj * iq;
    const V3 p_i = p_ci + rot(q_ci, ip), p_j = p_cj + rot(q_cj, ip);
    const double dt = pre[PRE_T];
    const V3 dbg = bg_i - v3_load(bias_lin), dba = ba_i - v3_load(bias_lin + 3);
    // (the five 3x3 bias Jacobians of the preintegration are read where they are used -- first for the residual, again
    // for the blocks they fill -- instead of being held across the whole evaluation: one lane evaluates a factor, and
    // 45 extra live doubles are what tipped the register allocation into scratch)
    V3 th, r_th;
    {
        const Q4 dq = q_load(pre + PRE_Q);
        th = m3_load(pre + PRE_JAC) * dbg;
        const Q4 corr = dq * expmap(th);
        r_th = logmap(conj(corr) * conj(q_i) * q_j);
    }
    const V3 wp = p_j - p_i - dt * v_i - (0.5 * dt * dt) * g, wv = v_j - v_i - dt * g;
    {
        const V3 dp = v3_load(pre + PRE_P), dv = v3_load(pre + PRE_V);
        const V3 r_p = rot_inv(q_i, wp) - (dp + m3_load(pre + PRE_JAC + 9) * dbg + m3_load(pre + PRE_JAC + 18) * dba);
        const V3 r_v = rot_inv(q_i, wv) - (dv + m3_load(pre + PRE_JAC + 27) * dbg + m3_load(pre + PRE_JAC + 36) * dba);
        v3_store(e + ES_Q, r_th);
        v3_store(e + ES_P, r_p);
        v3_store(e + ES_V, r_v);
        v3_store(e + ES_BG, bg_j - bg_i);
        v3_store(e + ES_BA, ba_j - ba_i);
    }
    if (!WITH_JAC) return;

    // Each group of blocks is its own (noinline) function that re-reads the few state entries it needs (the states are
    // LDS-resident): a factor is evaluated by ONE lane, so the live set of the whole Jacobian -- ~100 doubles of
    // rotations, translations and trigonometric temporaries -- does not fit the register file in one piece.
    preintegration_rotation_blocks(si, sj, pre, extr, th, r_th, Gi, Gj);
    if (!ROT_ONLY) preintegration_translation_bias_blocks(si, sj, pre, extr, Gi, Gj);
}

// CeresMarginalizationFactor::Evaluate, per-frame part
// (src/rdvio_estimation/include/rdvio/estimation/ceres/marginalization_factor.h:29-52):
// e15 = [log(q0^-1 q); p-p0; v-v0; bg-bg0; ba-ba0] and Jr(e_theta)^-1.
DM void marginalization_frame_error(const double *s, const double *lin, double *e15, M3 *Jrinv) {
    const V3 eth = logmap(conj(q_load(lin + ST_Q)) * q_load(s + ST_Q));
    v3_store(e15, eth);
#pragma unroll
    for (int i = 0; i < 12; ++i) e15[3 + i] = s[4 + i] - lin[4 + i];
    if (Jrinv) *Jrinv = inverse3(right_jacobian(eth));
}
