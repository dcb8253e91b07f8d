/*
 * ORACLE (test infrastructure, NOT product code) -- CPU restatement of the
 * in-tree estimation arithmetic of rd_vio (rows A7-A13 of SURVEY.md section 8a).
 * PARITY UNPINNED: see rdvio_oracle.h.
 */
#include "rdvio_oracle.h"
#include "ro_math.h"
#include <stdlib.h>

enum { ES_Q = 0, ES_P = 3, ES_V = 6, ES_BG = 9, ES_BA = 12, ES = 15 };
enum { ST_Q = 0, ST_P = 4, ST_V = 7, ST_BG = 10, ST_BA = 13 };
enum { EX_CQ = 0, EX_CP = 4, EX_IQ = 7, EX_IP = 11 };

void ro_expmap(const double *w, double *q) { expmap(q, w); }
void ro_logmap(const double *q, double *w) { logmap(w, q); }
void ro_right_jacobian_c(const double *w, double *J) { ro_right_jacobian(J, w); }

/* local_tangent = [b1 b2 z] (ceres/reprojection_factor.h:19-21), row-major 3x3 */
void ro_tangent_frame(const double *z, double *T) {
    double b1[3], b2[3];
    ro_s2_tangential_basis(b1, b2, z);
    for (int i = 0; i < 3; ++i) {
        T[i * 3 + 0] = b1[i];
        T[i * 3 + 1] = b2[i];
        T[i * 3 + 2] = z[i];
    }
}

/* QuaternionParameterization::Plus, ceres/quaternion_parameterization.h:11-17 */
void ro_quat_plus(const double *q, const double *delta, double *out) {
    double e[4], t[4];
    expmap(e, delta);
    q_mul(t, q, e);
    q_normalize(out, t);
}

/* ------------------------------------------------------------------ A7 */
/* block helpers on a row-major matrix with leading dimension ld */
static void blk_set(double *M, int ld, int r0, int c0, const double *B, double s) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[(r0 + i) * ld + c0 + j] = s * B[i * 3 + j];
}

/* PreIntegrator::increment, preintegrator.cpp:22-76 */
static void preint_increment(double *pre, double dt, const double *w_raw, const double *a_raw,
                             const double *bg, const double *ba, const double *noise, int cj, int cc) {
    double *dq = pre + RO_PREINT_Q, *dp = pre + RO_PREINT_P, *dv = pre + RO_PREINT_V;
    double *cov = pre + RO_PREINT_COV;
    double *dq_dbg = pre + RO_PREINT_JAC, *dp_dbg = dq_dbg + 9, *dp_dba = dq_dbg + 18,
           *dv_dbg = dq_dbg + 27, *dv_dba = dq_dbg + 36;
    const double *cov_w = noise, *cov_a = noise + 9, *cov_bg = noise + 18, *cov_ba = noise + 27;

    double w[3], a[3], wdt[3];
    v3_sub(w, w_raw, bg);
    v3_sub(a, a_raw, ba);
    v3_scale(wdt, w, dt);

    double R[9], Ha[9], RHa[9], eq[4], eqc[4], ERt[9], Jr[9];
    q_to_mat(R, dq);
    hat(Ha, a);
    m3_mul(RHa, R, Ha);
    expmap(eq, wdt);
    q_conj(eqc, eq);
    q_to_mat(ERt, eqc); /* expmap(w dt).conjugate().matrix() */
    ro_right_jacobian(Jr, wdt);

    if (cc) {
        double A[81], B[54], Q[36];
        memset(A, 0, sizeof A);
        for (int i = 0; i < 9; ++i) A[i * 9 + i] = 1.0;
        blk_set(A, 9, ES_Q, ES_Q, ERt, 1.0);
        blk_set(A, 9, ES_V, ES_Q, RHa, -dt);
        blk_set(A, 9, ES_P, ES_Q, RHa, -0.5 * dt * dt);
        double I3[9];
        m3_identity(I3);
        blk_set(A, 9, ES_P, ES_V, I3, dt);
        memset(B, 0, sizeof B);
        blk_set(B, 6, ES_Q, 0, Jr, dt);
        blk_set(B, 6, ES_V, 3, R, dt);
        blk_set(B, 6, ES_P, 3, R, 0.5 * dt * dt);
        double inv_dt = 1.0 / fmax(dt, 1.0e-7);
        memset(Q, 0, sizeof Q);
        blk_set(Q, 6, 0, 0, cov_w, inv_dt);
        blk_set(Q, 6, 3, 3, cov_a, inv_dt);

        double C9[81], AC[81], ACAt[81], BQ[54], BQBt[81];
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) C9[i * 9 + j] = cov[i * 15 + j];
        mat_mul(AC, A, C9, 9, 9, 9);
        mat_mul_nt(ACAt, AC, A, 9, 9, 9);
        mat_mul(BQ, B, Q, 9, 6, 6);
        mat_mul_nt(BQBt, BQ, B, 9, 6, 9);
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) cov[i * 15 + j] = ACAt[i * 9 + j] + BQBt[i * 9 + j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                cov[(ES_BG + i) * 15 + ES_BG + j] += cov_bg[i * 3 + j] * dt;
                cov[(ES_BA + i) * 15 + ES_BA + j] += cov_ba[i * 3 + j] * dt;
            }
    }

    if (cj) {
        double T[9];
        /* dp_dbg += dt*dv_dbg - 0.5 dt^2 R hat(a) dq_dbg */
        m3_mul(T, RHa, dq_dbg);
        for (int i = 0; i < 9; ++i) dp_dbg[i] += dt * dv_dbg[i] - 0.5 * dt * dt * T[i];
        /* dp_dba += dt*dv_dba - 0.5 dt^2 R */
        for (int i = 0; i < 9; ++i) dp_dba[i] += dt * dv_dba[i] - 0.5 * dt * dt * R[i];
        /* dv_dbg -= dt R hat(a) dq_dbg */
        for (int i = 0; i < 9; ++i) dv_dbg[i] -= dt * T[i];
        /* dv_dba -= dt R */
        for (int i = 0; i < 9; ++i) dv_dba[i] -= dt * R[i];
        /* dq_dbg = exp(w dt)^T dq_dbg - dt Jr(w dt) */
        m3_mul(T, ERt, dq_dbg);
        for (int i = 0; i < 9; ++i) dq_dbg[i] = T[i] - dt * Jr[i];
    }

    double qa[3], nq[4];
    q_rot(qa, dq, a);
    pre[RO_PREINT_T] += dt;
    for (int i = 0; i < 3; ++i) dp[i] = dp[i] + dt * dv[i] + 0.5 * dt * dt * qa[i];
    for (int i = 0; i < 3; ++i) dv[i] = dv[i] + dt * qa[i];
    q_mul(nq, dq, eq);
    q_normalize(dq, nq);
}

/* PreIntegrator::integrate + compute_sqrt_inv_cov, preintegrator.cpp:78-100 */
int ro_preintegrate(int n, const double *imu, double t_end, const double *bg, const double *ba,
                    const double *noise, int cj, int cc, double *pre) {
    if (n == 0) return 0;
    memset(pre, 0, sizeof(double) * RO_PREINT_SIZE);
    pre[RO_PREINT_Q + 3] = 1.0;
    for (int i = 0; i + 1 < n; ++i) {
        const double *d = imu + 7 * i;
        preint_increment(pre, imu[7 * (i + 1)] - d[0], d + 1, d + 4, bg, ba, noise, cj, cc);
    }
    const double *d = imu + 7 * (n - 1);
    preint_increment(pre, t_end - d[0], d + 1, d + 4, bg, ba, noise, cj, cc);
    if (cc) {
        double inv[225], L[225];
        ro_inverse(inv, pre + RO_PREINT_COV, 15);
        ro_cholesky_lower(L, inv, 15);
        mat_transpose(pre + RO_PREINT_SIC, L, 15, 15);
    }
    return 1;
}

/* PreIntegrator::predict, preintegrator.cpp:102-112 */
void ro_preint_predict(const double *pre, const double *si, double *sj) {
    const double g[3] = {0, 0, -RO_GRAVITY};
    double dt = pre[RO_PREINT_T];
    double qdv[3], qdp[3];
    q_rot(qdv, si + ST_Q, pre + RO_PREINT_V);
    q_rot(qdp, si + ST_Q, pre + RO_PREINT_P);
    for (int i = 0; i < 3; ++i) {
        sj[ST_BG + i] = si[ST_BG + i];
        sj[ST_BA + i] = si[ST_BA + i];
        sj[ST_V + i] = si[ST_V + i] + g[i] * dt + qdv[i];
        sj[ST_P + i] = si[ST_P + i] + 0.5 * g[i] * dt * dt + si[ST_V + i] * dt + qdp[i];
    }
    q_mul(sj + ST_Q, si + ST_Q, pre + RO_PREINT_Q);
}

/* ------------------------------------------------------------------ A8/A9 */
/* CeresReprojectionErrorFactor::Evaluate, ceres/reprojection_factor.h:24-89 */
void ro_reprojection_eval(int nf, const int32_t *tgt, const int32_t *ref, const int32_t *lm,
                          const double *tangent, const double *z_ref_all, const double *inv_depth_all,
                          const double *states, const double *extr, const double *W,
                          double *r_out, double *Jt_out, double *Jr_out, double *Jd_out) {
    const double *qcs = extr + EX_CQ, *pcs = extr + EX_CP;
    double Rcs[9], RcsT[9];
    q_to_mat(Rcs, qcs);
    m3_transpose(RcsT, Rcs);
    for (int k = 0; k < nf; ++k) {
        const double *st = states + 16 * tgt[k], *sr = states + 16 * ref[k];
        const double *T = tangent + 9 * k;
        const double *z_ref = z_ref_all + 3 * lm[k];
        double rho = inv_depth_all[lm[k]];

        double y_ref[3], y_rc[3], x[3], d[3], y_tc[3], y_t[3], u[3];
        for (int i = 0; i < 3; ++i) y_ref[i] = z_ref[i] / rho; /* z_ref / inv_depth */
        q_rot(y_rc, qcs, y_ref);
        v3_add(y_rc, y_rc, pcs);
        q_rot(x, sr + ST_Q, y_rc);
        v3_add(x, x, sr + ST_P);
        v3_sub(d, x, st + ST_P);
        q_rot_inv(y_tc, st + ST_Q, d);
        v3_sub(d, y_tc, pcs);
        q_rot_inv(y_t, qcs, d);
        /* u = T^T y_t */
        for (int c = 0; c < 3; ++c) u[c] = T[0 * 3 + c] * y_t[0] + T[1 * 3 + c] * y_t[1] + T[2 * 3 + c] * y_t[2];
        double h0 = u[0] / u[2], h1 = u[1] / u[2];
        r_out[2 * k + 0] = W[0] * h0 + W[1] * h1;
        r_out[2 * k + 1] = W[2] * h0 + W[3] * h1;

        if (Jt_out || Jr_out || Jd_out) {
            double dproj[6] = {1.0 / u[2], 0.0, -u[0] / (u[2] * u[2]), 0.0, 1.0 / u[2], -u[1] / (u[2] * u[2])};
            double WD[6], Tt[9], A[6], B[6], C[6], D[6], Rt[9], RtT[9], Rr[9], H[9], M[6];
            mat_mul(WD, W, dproj, 2, 2, 3);
            m3_transpose(Tt, T);
            mat_mul(A, WD, Tt, 2, 3, 3);      /* dr_dy_tgt */
            mat_mul(B, A, RcsT, 2, 3, 3);     /* dr_dy_tgt_center */
            q_to_mat(Rt, st + ST_Q);
            m3_transpose(RtT, Rt);
            mat_mul(C, B, RtT, 2, 3, 3);      /* dr_dx */
            q_to_mat(Rr, sr + ST_Q);
            mat_mul(D, C, Rr, 2, 3, 3);       /* dr_dy_ref_center */
            if (Jt_out) {
                double *J = Jt_out + 12 * k;
                hat(H, y_tc);
                mat_mul(M, B, H, 2, 3, 3);
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 3; ++j) {
                        J[i * 6 + j] = M[i * 3 + j];
                        J[i * 6 + 3 + j] = -C[i * 3 + j];
                    }
            }
            if (Jr_out) {
                double *J = Jr_out + 12 * k;
                hat(H, y_rc);
                mat_mul(M, D, H, 2, 3, 3);
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 3; ++j) {
                        J[i * 6 + j] = -M[i * 3 + j];
                        J[i * 6 + 3 + j] = C[i * 3 + j];
                    }
            }
            if (Jd_out) {
                double t3[3];
                m3_mulv(t3, Rcs, y_ref);
                for (int i = 0; i < 2; ++i)
                    Jd_out[2 * k + i] = -(D[i * 3] * t3[0] + D[i * 3 + 1] * t3[1] + D[i * 3 + 2] * t3[2]) / rho;
            }
        }
    }
}

/* ------------------------------------------------------------------ A10 */
/* CeresRotationPriorFactor::Evaluate, ceres/rotation_factor.h:22-58 */
void ro_rotation_prior_eval(const double *q_tgt, const double *q_ref, const double *z_ref,
                            const double *T, const double *extr, const double *W, double *r, double *J) {
    const double *qcs = extr + EX_CQ, *pcs = extr + EX_CP;
    double z_rc[3], t[3], z_tc[3], d[3], z_t[3], u[3];
    q_rot(z_rc, qcs, z_ref);
    v3_add(z_rc, z_rc, pcs); /* translation added to a bearing: reference quirk, rotation_factor.h:34 */
    q_rot(t, q_ref, z_rc);
    q_rot_inv(z_tc, q_tgt, t);
    v3_sub(d, z_tc, pcs);
    q_rot_inv(z_t, qcs, d);
    for (int c = 0; c < 3; ++c) u[c] = T[0 * 3 + c] * z_t[0] + T[1 * 3 + c] * z_t[1] + T[2 * 3 + c] * z_t[2];
    double h0 = u[0] / u[2], h1 = u[1] / u[2];
    r[0] = W[0] * h0 + W[1] * h1;
    r[1] = W[2] * h0 + W[3] * h1;
    if (J) {
        double dproj[6] = {1.0 / u[2], 0.0, -u[0] / (u[2] * u[2]), 0.0, 1.0 / u[2], -u[1] / (u[2] * u[2])};
        double WD[6], Tt[9], A[6], B[6], Rcs[9], RcsT[9], H[9];
        mat_mul(WD, W, dproj, 2, 2, 3);
        m3_transpose(Tt, T);
        mat_mul(A, WD, Tt, 2, 3, 3);
        q_to_mat(Rcs, qcs);
        m3_transpose(RcsT, Rcs);
        mat_mul(B, A, RcsT, 2, 3, 3);
        hat(H, z_tc);
        mat_mul(J, B, H, 2, 3, 3);
    }
}

/* ------------------------------------------------------------------ A11 */
/* CeresPreIntegrationErrorFactor::Evaluate, ceres/preintegration_factor.h:19-160 */
void ro_preintegration_eval(const double *si, const double *sj, const double *pre, const double *bias_lin,
                            const double *extr, double *r, double *Ji, double *Jj) {
    const double g[3] = {0, 0, -RO_GRAVITY};
    const double *q_ci = si + ST_Q, *p_ci = si + ST_P, *v_i = si + ST_V, *bg_i = si + ST_BG, *ba_i = si + ST_BA;
    const double *q_cj = sj + ST_Q, *p_cj = sj + ST_P, *v_j = sj + ST_V, *bg_j = sj + ST_BG, *ba_j = sj + ST_BA;
    const double *iq = extr + EX_IQ, *ip = extr + EX_IP;
    const double *bg0 = bias_lin, *ba0 = bias_lin + 3;

    double q_i[4], q_j[4], p_i[3], p_j[3], t[3];
    q_mul(q_i, q_ci, iq);
    q_rot(t, q_ci, ip); v3_add(p_i, p_ci, t);
    q_mul(q_j, q_cj, iq);
    q_rot(t, q_cj, ip); v3_add(p_j, p_cj, t);

    double dt = pre[RO_PREINT_T];
    const double *dq = pre + RO_PREINT_Q, *dp = pre + RO_PREINT_P, *dv = pre + RO_PREINT_V;
    const double *dq_dbg = pre + RO_PREINT_JAC, *dp_dbg = dq_dbg + 9, *dp_dba = dq_dbg + 18,
                 *dv_dbg = dq_dbg + 27, *dv_dba = dq_dbg + 36;
    const double *S = pre + RO_PREINT_SIC;
    double dbg[3], dba[3];
    v3_sub(dbg, bg_i, bg0);
    v3_sub(dba, ba_i, ba0);

    double e[15];
    /* r_theta = log((dq * exp(dq_dbg dbg))^-1 * q_i^-1 * q_j) */
    double th[3], eq[4], a4[4], a4c[4], qic[4], b4[4], c4[4];
    m3_mulv(th, dq_dbg, dbg);
    expmap(eq, th);
    q_mul(a4, dq, eq);
    q_conj(a4c, a4);
    q_conj(qic, q_i);
    q_mul(b4, a4c, qic);
    q_mul(c4, b4, q_j);
    logmap(e + ES_Q, c4);
    /* r_p */
    double d3[3], rp[3], c1[3], c2[3];
    for (int i = 0; i < 3; ++i) d3[i] = p_j[i] - p_i[i] - dt * v_i[i] - 0.5 * dt * dt * g[i];
    q_rot_inv(rp, q_i, d3);
    m3_mulv(c1, dp_dbg, dbg);
    m3_mulv(c2, dp_dba, dba);
    for (int i = 0; i < 3; ++i) e[ES_P + i] = rp[i] - (dp[i] + c1[i] + c2[i]);
    /* r_v */
    for (int i = 0; i < 3; ++i) d3[i] = v_j[i] - v_i[i] - dt * g[i];
    q_rot_inv(rp, q_i, d3);
    m3_mulv(c1, dv_dbg, dbg);
    m3_mulv(c2, dv_dba, dba);
    for (int i = 0; i < 3; ++i) e[ES_V + i] = rp[i] - (dv[i] + c1[i] + c2[i]);
    for (int i = 0; i < 3; ++i) {
        e[ES_BG + i] = bg_j[i] - bg_i[i];
        e[ES_BA + i] = ba_j[i] - ba_i[i];
    }

    if (Ji && Jj) {
        double Gi[225], Gj[225]; /* unwhitened 15x15 */
        memset(Gi, 0, sizeof Gi);
        memset(Gj, 0, sizeof Gj);
        double Jr[9], Jrinv[9], RjT[9], Rci[9], RciT[9], Rcj[9], RiT[9], IqT[9], M[9], N[9], H[9];
        ro_right_jacobian(Jr, e + ES_Q);
        ro_inverse(Jrinv, Jr, 3);
        double qjc[4];
        q_conj(qjc, q_j);
        q_to_mat(RjT, qjc);
        q_to_mat(Rci, q_ci);
        m3_transpose(RciT, Rci);
        q_to_mat(Rcj, q_cj);
        q_to_mat(RiT, qic);
        double iqc[4];
        q_conj(iqc, iq);
        q_to_mat(IqT, iqc);
        /* d/dtheta_i */
        m3_mul(M, RjT, Rci);
        m3_mul(N, Jrinv, M);
        blk_set(Gi, 15, ES_Q, ES_Q, N, -1.0);
        for (int i = 0; i < 3; ++i) d3[i] = p_j[i] - p_ci[i] - dt * v_i[i] - 0.5 * dt * dt * g[i];
        m3_mulv(t, RciT, d3);
        hat(H, t);
        m3_mul(M, IqT, H);
        blk_set(Gi, 15, ES_P, ES_Q, M, 1.0);
        for (int i = 0; i < 3; ++i) d3[i] = v_j[i] - v_i[i] - dt * g[i];
        m3_mulv(t, RciT, d3);
        hat(H, t);
        m3_mul(M, IqT, H);
        blk_set(Gi, 15, ES_V, ES_Q, M, 1.0);
        /* d/dp_i */
        blk_set(Gi, 15, ES_P, ES_P, RiT, -1.0);
        /* d/dv_i */
        blk_set(Gi, 15, ES_P, ES_V, RiT, -dt);
        blk_set(Gi, 15, ES_V, ES_V, RiT, -1.0);
        /* d/dbg_i */
        double er[4], erc[4], ERt[9], Jrb[9];
        expmap(er, e + ES_Q);
        q_conj(erc, er);
        q_to_mat(ERt, erc);
        ro_right_jacobian(Jrb, th);
        m3_mul(M, Jrinv, ERt);
        m3_mul(N, M, Jrb);
        m3_mul(M, N, dq_dbg);
        blk_set(Gi, 15, ES_Q, ES_BG, M, -1.0);
        blk_set(Gi, 15, ES_P, ES_BG, dp_dbg, -1.0);
        blk_set(Gi, 15, ES_V, ES_BG, dv_dbg, -1.0);
        double I3[9];
        m3_identity(I3);
        blk_set(Gi, 15, ES_BG, ES_BG, I3, -1.0);
        /* d/dba_i */
        blk_set(Gi, 15, ES_P, ES_BA, dp_dba, -1.0);
        blk_set(Gi, 15, ES_V, ES_BA, dv_dba, -1.0);
        blk_set(Gi, 15, ES_BA, ES_BA, I3, -1.0);
        /* d/dtheta_j */
        m3_mul(M, Jrinv, IqT);
        blk_set(Gj, 15, ES_Q, ES_Q, M, 1.0);
        hat(H, ip);
        m3_mul(M, RiT, Rcj);
        m3_mul(N, M, H);
        blk_set(Gj, 15, ES_P, ES_Q, N, -1.0);
        /* d/dp_j, d/dv_j, d/dbg_j, d/dba_j */
        blk_set(Gj, 15, ES_P, ES_P, RiT, 1.0);
        blk_set(Gj, 15, ES_V, ES_V, RiT, 1.0);
        blk_set(Gj, 15, ES_BG, ES_BG, I3, 1.0);
        blk_set(Gj, 15, ES_BA, ES_BA, I3, 1.0);
        mat_mul(Ji, S, Gi, 15, 15, 15);
        mat_mul(Jj, S, Gj, 15, 15, 15);
    }
    mat_mul(r, S, e, 15, 15, 1);
}

/* ------------------------------------------------------------------ A12 */
/* CeresMarginalizationFactor::Evaluate, ceres/marginalization_factor.h:27-72 */
void ro_marginalization_eval(int np, const double *states, const double *lin, const double *S,
                             const double *f, double *r, double *J) {
    int D = 15 * np;
    double *e = (double *)malloc(sizeof(double) * D);
    double *E = J ? (double *)calloc((size_t)D * D, sizeof(double)) : NULL;
    for (int i = 0; i < np; ++i) {
        const double *s = states + 16 * i, *l = lin + 16 * i;
        double lc[4], dq4[4];
        q_conj(lc, l + ST_Q);
        q_mul(dq4, lc, s + ST_Q);
        logmap(e + 15 * i + ES_Q, dq4);
        for (int k = 0; k < 3; ++k) {
            e[15 * i + ES_P + k] = s[ST_P + k] - l[ST_P + k];
            e[15 * i + ES_V + k] = s[ST_V + k] - l[ST_V + k];
            e[15 * i + ES_BG + k] = s[ST_BG + k] - l[ST_BG + k];
            e[15 * i + ES_BA + k] = s[ST_BA + k] - l[ST_BA + k];
        }
        if (E) {
            double Jr[9], Jri[9];
            ro_right_jacobian(Jr, e + 15 * i + ES_Q);
            ro_inverse(Jri, Jr, 3);
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) E[(15 * i + a) * D + 15 * i + b] = Jri[a * 3 + b];
            for (int a = 3; a < 15; ++a) E[(15 * i + a) * D + 15 * i + a] = 1.0;
        }
    }
    if (J) mat_mul(J, S, E, D, D, D);
    for (int i = 0; i < D; ++i) {
        double s = 0;
        for (int j = 0; j < D; ++j) s += S[i * D + j] * e[j];
        r[i] = s + f[i];
    }
    free(e);
    free(E);
}

/* ------------------------------------------------------------------ A13 */
/* CeresMarginalizationFactor::marginalize(0), ceres/marginalization_factor.h:74-475 */
void ro_marginalize(const ro_marg_problem *pb, double *S_out, double *f_out, double *lin_out,
                    double *Lambda_out, double *eta_out) {
    int nfm = pb->nframes;
    int N = 15 * nfm;
    double *Lam = (double *)calloc((size_t)N * N, sizeof(double));
    double *eta = (double *)calloc((size_t)N, sizeof(double));
    /* frame_indices: victim (map index 0) permuted last (:95-105) */
    int *fidx = (int *)malloc(sizeof(int) * nfm);
    for (int i = 0; i < nfm; ++i) fidx[i] = (i == 0) ? nfm - 1 : i - 1;

    /* (i) current prior: J^T J, J^T r (:107-161) */
    {
        int np = pb->np, D = 15 * np;
        double *ps = (double *)calloc((size_t)16 * (np > 0 ? np : 1), sizeof(double));
        for (int i = 0; i < np; ++i) memcpy(ps + 16 * i, pb->states + 16 * pb->prior_frames[i], 16 * sizeof(double));
        double *r = (double *)malloc(sizeof(double) * D);
        double *J = (double *)malloc(sizeof(double) * D * D);
        ro_marginalization_eval(np, ps, pb->lin, pb->S, pb->f, r, J);
        for (int i = 0; i < np; ++i) {
            int fi = fidx[pb->prior_frames[i]];
            for (int j = 0; j < np; ++j) {
                int fj = fidx[pb->prior_frames[j]];
                for (int a = 0; a < 15; ++a)
                    for (int b = 0; b < 15; ++b) {
                        double s = 0;
                        for (int k = 0; k < D; ++k) s += J[k * D + 15 * i + a] * J[k * D + 15 * j + b];
                        Lam[(15 * fi + a) * N + 15 * fj + b] += s;
                    }
            }
            for (int a = 0; a < 15; ++a) {
                double s = 0;
                for (int k = 0; k < D; ++k) s += J[k * D + 15 * i + a] * r[k];
                eta[15 * fi + a] += s;
            }
        }
        free(ps); free(r); free(J);
    }

    /* (ii) preintegration factor between frames 0 and 1 (:163-231; loop j = index..index+1 with j == 0 skipped) */
    if (nfm >= 2 && pb->preint01) {
        const double *si = pb->states, *sj = pb->states + 16;
        double r[15], Ji[225], Jj[225];
        /* bias linearisation = live frame members -> dbg = dba = 0 (preintegration_factor.h:37-38) */
        ro_preintegration_eval(si, sj, pb->preint01, si + ST_BG, pb->extr, r, Ji, Jj);
        int fi = fidx[0], fj = fidx[1];
        const double *Js[2] = {Ji, Jj};
        int fs[2] = {fi, fj};
        for (int x = 0; x < 2; ++x) {
            for (int y = 0; y < 2; ++y)
                for (int a = 0; a < 15; ++a)
                    for (int b = 0; b < 15; ++b) {
                        double s = 0;
                        for (int k = 0; k < 15; ++k) s += Js[x][k * 15 + a] * Js[y][k * 15 + b];
                        Lam[(15 * fs[x] + a) * N + 15 * fs[y] + b] += s;
                    }
            for (int a = 0; a < 15; ++a) {
                double s = 0;
                for (int k = 0; k < 15; ++k) s += Js[x][k * 15 + a] * r[k];
                eta[15 * fs[x] + a] += s;
            }
        }
    }

    /* (iii) reprojection factors + landmark info (:233-380) */
    int nl = pb->nlm;
    double *lmat = (double *)calloc(nl, sizeof(double));
    double *lvec = (double *)calloc(nl, sizeof(double));
    double *lh = (double *)calloc((size_t)nl * nfm * 6, sizeof(double)); /* h[lm][frame_index][6] */
    char *lhset = (char *)calloc((size_t)nl * nfm, 1);
    char *lused = (char *)calloc(nl, 1);
    if (pb->nfac > 0) {
        double *r = (double *)malloc(sizeof(double) * 2 * pb->nfac);
        double *Jt = (double *)malloc(sizeof(double) * 12 * pb->nfac);
        double *Jr = (double *)malloc(sizeof(double) * 12 * pb->nfac);
        double *Jd = (double *)malloc(sizeof(double) * 2 * pb->nfac);
        ro_reprojection_eval(pb->nfac, pb->tgt, pb->ref, pb->lm, pb->tangent, pb->z_ref, pb->inv_depth,
                             pb->states, pb->extr, pb->sqrt_inv_cov, r, Jt, Jr, Jd);
        for (int k = 0; k < pb->nfac; ++k) {
            int ft = fidx[pb->tgt[k]], fr = fidx[pb->ref[k]], l = pb->lm[k];
            const double *A = Jt + 12 * k, *B = Jr + 12 * k, *rk = r + 2 * k, *d = Jd + 2 * k;
            const double *Js[2] = {A, B};
            int fs[2] = {ft, fr};
            /* all 16 3x3 blocks == the four 6x6 products over (theta,p) slots */
            for (int x = 0; x < 2; ++x) {
                for (int y = 0; y < 2; ++y)
                    for (int a = 0; a < 6; ++a)
                        for (int b = 0; b < 6; ++b)
                            Lam[(15 * fs[x] + a) * N + 15 * fs[y] + b] +=
                                Js[x][a] * Js[y][b] + Js[x][6 + a] * Js[y][6 + b];
                for (int a = 0; a < 6; ++a) eta[15 * fs[x] + a] += Js[x][a] * rk[0] + Js[x][6 + a] * rk[1];
            }
            lused[l] = 1;
            lmat[l] += d[0] * d[0] + d[1] * d[1];
            lvec[l] += d[0] * rk[0] + d[1] * rk[1];
            lhset[l * nfm + ft] = 1;
            lhset[l * nfm + fr] = 1;
            for (int a = 0; a < 6; ++a) {
                lh[((size_t)l * nfm + ft) * 6 + a] += d[0] * A[a] + d[1] * A[6 + a];
                lh[((size_t)l * nfm + fr) * 6 + a] += d[0] * B[a] + d[1] * B[6 + a];
            }
        }
        free(r); free(Jt); free(Jr); free(Jd);
    }

    /* (iv) landmark Schur (:382-398) */
    for (int l = 0; l < nl; ++l) {
        if (!lused[l]) continue;
        double inv = 1.0 / lmat[l];
        if (!isfinite(inv)) continue;
        for (int i = 0; i < nfm; ++i) {
            if (!lhset[l * nfm + i]) continue;
            const double *hi = lh + ((size_t)l * nfm + i) * 6;
            for (int j = 0; j < nfm; ++j) {
                if (!lhset[l * nfm + j]) continue;
                const double *hj = lh + ((size_t)l * nfm + j) * 6;
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) Lam[(15 * i + a) * N + 15 * j + b] -= hi[a] * inv * hj[b];
            }
            for (int a = 0; a < 6; ++a) eta[15 * i + a] -= hi[a] * inv * lvec[l];
        }
    }

    /* (v) frame Schur (:400-438) */
    int last = nfm - 1, R = 15 * last;
    double Mmm[225], Minv[225];
    for (int a = 0; a < 15; ++a)
        for (int b = 0; b < 15; ++b) Mmm[a * 15 + b] = Lam[(R + a) * N + R + b];
    ro_inverse(Minv, Mmm, 15);
    double *Lr = (double *)malloc(sizeof(double) * R * R);
    double *er = (double *)malloc(sizeof(double) * R);
    double *T = (double *)malloc(sizeof(double) * R * 15); /* Lam_rm * Minv */
    for (int i = 0; i < R; ++i)
        for (int b = 0; b < 15; ++b) {
            double s = 0;
            for (int a = 0; a < 15; ++a) s += Lam[i * N + R + a] * Minv[a * 15 + b];
            T[i * 15 + b] = s;
        }
    for (int i = 0; i < R; ++i) {
        for (int j = 0; j < R; ++j) {
            double s = 0;
            for (int a = 0; a < 15; ++a) s += T[i * 15 + a] * Lam[(R + a) * N + j];
            Lr[i * R + j] = Lam[i * N + j] - s;
        }
        double s = 0;
        for (int a = 0; a < 15; ++a) s += T[i * 15 + a] * eta[R + a];
        er[i] = eta[i] - s;
    }
    if (Lambda_out) memcpy(Lambda_out, Lr, sizeof(double) * R * R);
    if (eta_out) memcpy(eta_out, er, sizeof(double) * R);

    /* (vi) new prior via symmetric eigendecomposition (:440-474) */
    double *ev = (double *)malloc(sizeof(double) * R);
    double *V = (double *)malloc(sizeof(double) * R * R);
    ro_sym_eig(ev, V, Lr, R);
    for (int i = 0; i < R; ++i) {
        double lam = ev[i] > 1.0e-8 ? ev[i] : 0.0;
        double lam_inv = ev[i] > 1.0e-8 ? 1.0 / ev[i] : 0.0;
        double sl = sqrt(lam), sli = sqrt(lam_inv);
        double s = 0;
        for (int j = 0; j < R; ++j) {
            S_out[i * R + j] = sl * V[j * R + i];
            s += V[j * R + i] * er[j];
        }
        f_out[i] = sli * s;
    }
    for (int i = 1; i < nfm; ++i) memcpy(lin_out + 16 * (i - 1), pb->states + 16 * i, 16 * sizeof(double));

    free(ev); free(V); free(Lr); free(er); free(T);
    free(lmat); free(lvec); free(lh); free(lhset); free(lused);
    free(Lam); free(eta); free(fidx);
}
