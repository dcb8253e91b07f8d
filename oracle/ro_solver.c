/*
 * ORACLE (test infrastructure, NOT product code) -- CPU restatement of the non-linear solve behind
 * rdvio::Solver::solve (src/rdvio_estimation/src/solver.cpp:180-194):
 *     ceres::Solve, TRUST_REGION + DOGLEG (TRADITIONAL_DOGLEG), SPARSE_SCHUR, CauchyLoss(1.0) on the
 *     visual factors (solver.cpp:37-38,116-141), no loss on inertial / marginalisation factors
 *     (:143-178), QuaternionParameterization manifold (quaternion_parameterization.h:10-39),
 *     max_num_iterations = solver.iteration_limit, update_state_every_iteration = true.
 *
 * Ceres is an unpinned third-party dependency (>= 2.1, absent from /root/reference and from this
 * image).  This file restates the published algorithm of its TrustRegionMinimizer / DoglegStrategy /
 * Schur-eliminated normal equations with Ceres' default options:
 *   initial_trust_region_radius 1e4, max 1e16, min 1e-32, min_relative_decrease 1e-3,
 *   function_tolerance 1e-6, gradient_tolerance 1e-10, parameter_tolerance 1e-8,
 *   jacobi_scaling on (computed at iteration 0), min/max_lm_diagonal 1e-6/1e32,
 *   dogleg mu in [1e-8, 1] with x10 increase, monotonic steps, max 5 consecutive invalid steps,
 *   robust loss via the Corrector with rho'' <= 0 (Cauchy) => sqrt(rho') scaling of r and J.
 * PARITY UNPINNED (no Ceres build or fixture is available to check against).
 *
 * Bias linearisation of the preintegration factor: the reference reads bg_i0/ba_i0 from the LIVE
 * frame members (preintegration_factor.h:37-38), i.e. from the user state that Ceres refreshes in its
 * state-updating callback at the start of the iteration AFTER a successful step.  `user` below tracks
 * exactly that copy.
 */
#include "rdvio_oracle.h"
#include "ro_math.h"
#include <float.h>
#include <stdio.h>
#include <stdlib.h>

enum { ST_Q = 0, ST_P = 4, ST_V = 7, ST_BG = 10, ST_BA = 13 };

typedef struct {
    const ro_ba_problem *pb;
    int nfr, nl, nf, npre, nrot, np, D;
    int nfree;   /* free frames */
    int N;       /* 15 * nfree */
    int *fcol;   /* frame -> free slot or -1 */
    int *lfree;  /* landmark -> 1 if variable */
    /* stored linearisation (robustified) */
    double *r_f, *Jt, *Jr, *Jd; /* reprojection */
    double *r_p, *Ji, *Jj;       /* preintegration */
    double *r_m, *Jm;            /* marginalisation prior (D, D x D) */
    double *r_r, *Jro;           /* rotation prior (2, 2x3) */
    /* normal equations */
    double *H, *g;              /* N x N, N */
    double *lm_m, *lm_g, *lm_h; /* nl, nl, nl x nfree x 6 */
    /* scratch residuals for cost-only evaluations (must not clobber the stored linearisation) */
    double *c_f, *c_p, *c_m, *c_r;
} lin_t;

static void state_plus(const double *s, const double *d15, double *o) {
    double e[4], t[4];
    expmap(e, d15);
    q_mul(t, s, e);
    q_normalize(o, t);
    for (int i = 0; i < 12; ++i) o[4 + i] = s[4 + i] + d15[3 + i];
}

/* Evaluate cost (and, if L != NULL-linearise, the robustified residuals/Jacobians + normal equations). */
static double evaluate(lin_t *L, const double *states, const double *invd, const double *user, int want_jac) {
    const ro_ba_problem *pb = L->pb;
    double cost = 0.0;
    double *r_f = want_jac ? L->r_f : L->c_f, *r_p = want_jac ? L->r_p : L->c_p;
    double *r_m = want_jac ? L->r_m : L->c_m, *r_r = want_jac ? L->r_r : L->c_r;
    /* reprojection factors with CauchyLoss(1) */
    if (L->nf > 0) {
        ro_reprojection_eval(L->nf, pb->tgt, pb->ref, pb->lm, pb->tangent, pb->z_ref, invd, states, pb->extr,
                             pb->sqrt_inv_cov, r_f, want_jac ? L->Jt : NULL, want_jac ? L->Jr : NULL,
                             want_jac ? L->Jd : NULL);
        for (int k = 0; k < L->nf; ++k) {
            double *r = r_f + 2 * k;
            double s = r[0] * r[0] + r[1] * r[1];
            double sum = 1.0 + s, inv = 1.0 / sum;
            cost += 0.5 * log(sum);
            if (want_jac) {
                double rho1 = inv > DBL_MIN ? inv : DBL_MIN;
                double sc = sqrt(rho1);
                for (int i = 0; i < 12; ++i) { L->Jt[12 * k + i] *= sc; L->Jr[12 * k + i] *= sc; }
                L->Jd[2 * k] *= sc; L->Jd[2 * k + 1] *= sc;
                r[0] *= sc; r[1] *= sc;
            }
        }
    }
    for (int k = 0; k < L->nrot; ++k) {
        double *r = r_r + 2 * k;
        ro_rotation_prior_eval(states + 16 * pb->rot_tgt[k] + ST_Q, states + 16 * pb->rot_ref[k] + ST_Q,
                               pb->rot_zref + 3 * k, pb->rot_tangent + 9 * k, pb->extr, pb->sqrt_inv_cov, r,
                               want_jac ? L->Jro + 6 * k : NULL);
        double s = r[0] * r[0] + r[1] * r[1];
        double sum = 1.0 + s, inv = 1.0 / sum;
        cost += 0.5 * log(sum);
        if (want_jac) {
            double sc = sqrt(inv > DBL_MIN ? inv : DBL_MIN);
            for (int i = 0; i < 6; ++i) L->Jro[6 * k + i] *= sc;
            r[0] *= sc; r[1] *= sc;
        }
    }
    for (int k = 0; k < L->npre; ++k) {
        int i = pb->pre_i[k], j = pb->pre_j[k];
        ro_preintegration_eval(states + 16 * i, states + 16 * j, pb->preint + (size_t)RO_PREINT_SIZE * k,
                               user + 16 * i + ST_BG, pb->extr, r_p + 15 * k, want_jac ? L->Ji + 225 * k : NULL,
                               want_jac ? L->Jj + 225 * k : NULL);
        double s = 0;
        for (int a = 0; a < 15; ++a) s += r_p[15 * k + a] * r_p[15 * k + a];
        cost += 0.5 * s;
    }
    if (L->np > 0) {
        double *ps = (double *)malloc(sizeof(double) * 16 * L->np);
        for (int i = 0; i < L->np; ++i) memcpy(ps + 16 * i, states + 16 * pb->prior_frames[i], 16 * sizeof(double));
        ro_marginalization_eval(L->np, ps, pb->lin, pb->S, pb->f, r_m, want_jac ? L->Jm : NULL);
        free(ps);
        double s = 0;
        for (int a = 0; a < L->D; ++a) s += r_m[a] * r_m[a];
        cost += 0.5 * s;
    }
    if (!want_jac) return cost;
    /* frame_fixed == 2: pose constant, motion free (FT_FIX_POSE without FT_FIX_MOTION, solver.cpp:92-97).  The frame
     * keeps its 15 columns; the 6 pose columns of every Jacobian are zeroed, which leaves them decoupled with zero
     * gradient (the dogleg diagonal's lower clamp keeps the reduced system positive definite; their step is 0). */
    for (int k = 0; k < L->nf; ++k) {
        if (pb->frame_fixed[pb->tgt[k]] == 2) for (int i = 0; i < 12; ++i) L->Jt[12 * k + i] = 0.0;
        if (pb->frame_fixed[pb->ref[k]] == 2) for (int i = 0; i < 12; ++i) L->Jr[12 * k + i] = 0.0;
    }
    for (int k = 0; k < L->nrot; ++k)
        if (pb->frame_fixed[pb->rot_tgt[k]] == 2) for (int i = 0; i < 6; ++i) L->Jro[6 * k + i] = 0.0;
    for (int k = 0; k < L->npre; ++k) {
        if (pb->frame_fixed[pb->pre_i[k]] == 2) for (int q = 0; q < 15; ++q) for (int a = 0; a < 6; ++a) L->Ji[225 * k + 15 * q + a] = 0.0;
        if (pb->frame_fixed[pb->pre_j[k]] == 2) for (int q = 0; q < 15; ++q) for (int a = 0; a < 6; ++a) L->Jj[225 * k + 15 * q + a] = 0.0;
    }
    for (int i = 0; i < L->np; ++i)
        if (pb->frame_fixed[pb->prior_frames[i]] == 2)
            for (int q = 0; q < L->D; ++q) for (int a = 0; a < 6; ++a) L->Jm[(size_t)q * L->D + 15 * i + a] = 0.0;

    /* ---- normal equations: H (pose block), g, landmark scalars and couplings ---- */
    int N = L->N, nfree = L->nfree;
    memset(L->H, 0, sizeof(double) * N * N);
    memset(L->g, 0, sizeof(double) * N);
    memset(L->lm_m, 0, sizeof(double) * L->nl);
    memset(L->lm_g, 0, sizeof(double) * L->nl);
    memset(L->lm_h, 0, sizeof(double) * (size_t)L->nl * nfree * 6);
    for (int k = 0; k < L->nf; ++k) {
        int ct = L->fcol[pb->tgt[k]], cr = L->fcol[pb->ref[k]], l = pb->lm[k];
        const double *Js[2] = {L->Jt + 12 * k, L->Jr + 12 * k};
        int cs[2] = {ct, cr};
        const double *r = L->r_f + 2 * k, *d = L->Jd + 2 * k;
        for (int x = 0; x < 2; ++x) {
            if (cs[x] < 0) continue;
            for (int y = 0; y < 2; ++y) {
                if (cs[y] < 0) continue;
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b)
                        L->H[(15 * cs[x] + a) * N + 15 * cs[y] + b] += Js[x][a] * Js[y][b] + Js[x][6 + a] * Js[y][6 + b];
            }
            for (int a = 0; a < 6; ++a) L->g[15 * cs[x] + a] += Js[x][a] * r[0] + Js[x][6 + a] * r[1];
        }
        if (L->lfree[l]) {
            L->lm_m[l] += d[0] * d[0] + d[1] * d[1];
            L->lm_g[l] += d[0] * r[0] + d[1] * r[1];
            for (int x = 0; x < 2; ++x) {
                if (cs[x] < 0) continue;
                for (int a = 0; a < 6; ++a)
                    L->lm_h[((size_t)l * nfree + cs[x]) * 6 + a] += d[0] * Js[x][a] + d[1] * Js[x][6 + a];
            }
        }
    }
    for (int k = 0; k < L->nrot; ++k) {
        int c = L->fcol[pb->rot_tgt[k]];
        if (c < 0) continue;
        const double *J = L->Jro + 6 * k, *r = L->r_r + 2 * k;
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) L->H[(15 * c + a) * N + 15 * c + b] += J[a] * J[b] + J[3 + a] * J[3 + b];
            L->g[15 * c + a] += J[a] * r[0] + J[3 + a] * r[1];
        }
    }
    for (int k = 0; k < L->npre; ++k) {
        const double *Js[2] = {L->Ji + 225 * k, L->Jj + 225 * k};
        int cs[2] = {L->fcol[pb->pre_i[k]], L->fcol[pb->pre_j[k]]};
        const double *r = L->r_p + 15 * k;
        for (int x = 0; x < 2; ++x) {
            if (cs[x] < 0) continue;
            for (int y = 0; y < 2; ++y) {
                if (cs[y] < 0) continue;
                for (int a = 0; a < 15; ++a)
                    for (int b = 0; b < 15; ++b) {
                        double s = 0;
                        for (int q = 0; q < 15; ++q) s += Js[x][q * 15 + a] * Js[y][q * 15 + b];
                        L->H[(15 * cs[x] + a) * N + 15 * cs[y] + b] += s;
                    }
            }
            for (int a = 0; a < 15; ++a) {
                double s = 0;
                for (int q = 0; q < 15; ++q) s += Js[x][q * 15 + a] * r[q];
                L->g[15 * cs[x] + a] += s;
            }
        }
    }
    if (L->np > 0) {
        int D = L->D;
        for (int i = 0; i < L->np; ++i) {
            int ci = L->fcol[pb->prior_frames[i]];
            if (ci < 0) continue;
            for (int j = 0; j < L->np; ++j) {
                int cj = L->fcol[pb->prior_frames[j]];
                if (cj < 0) continue;
                for (int a = 0; a < 15; ++a)
                    for (int b = 0; b < 15; ++b) {
                        double s = 0;
                        for (int q = 0; q < D; ++q) s += L->Jm[q * D + 15 * i + a] * L->Jm[q * D + 15 * j + b];
                        L->H[(15 * ci + a) * N + 15 * cj + b] += s;
                    }
            }
            for (int a = 0; a < 15; ++a) {
                double s = 0;
                for (int q = 0; q < D; ++q) s += L->Jm[q * D + 15 * i + a] * L->r_m[q];
                L->g[15 * ci + a] += s;
            }
        }
    }
    return cost;
}

/* || J x ||^2 and (J x).r over all residual blocks; xp: N pose entries, xl: nl landmark entries (unscaled J) */
static void jx_products(const lin_t *L, const double *xp, const double *xl, double *jx_sq, double *jx_dot_r) {
    const ro_ba_problem *pb = L->pb;
    double sq = 0, dr = 0;
    for (int k = 0; k < L->nf; ++k) {
        int ct = L->fcol[pb->tgt[k]], cr = L->fcol[pb->ref[k]], l = pb->lm[k];
        double v[2] = {0, 0};
        for (int row = 0; row < 2; ++row) {
            if (ct >= 0) for (int a = 0; a < 6; ++a) v[row] += L->Jt[12 * k + 6 * row + a] * xp[15 * ct + a];
            if (cr >= 0) for (int a = 0; a < 6; ++a) v[row] += L->Jr[12 * k + 6 * row + a] * xp[15 * cr + a];
            if (L->lfree[l]) v[row] += L->Jd[2 * k + row] * xl[l];
        }
        sq += v[0] * v[0] + v[1] * v[1];
        dr += v[0] * L->r_f[2 * k] + v[1] * L->r_f[2 * k + 1];
    }
    for (int k = 0; k < L->nrot; ++k) {
        int c = L->fcol[pb->rot_tgt[k]];
        if (c < 0) continue;
        for (int row = 0; row < 2; ++row) {
            double v = 0;
            for (int a = 0; a < 3; ++a) v += L->Jro[6 * k + 3 * row + a] * xp[15 * c + a];
            sq += v * v;
            dr += v * L->r_r[2 * k + row];
        }
    }
    for (int k = 0; k < L->npre; ++k) {
        int ci = L->fcol[pb->pre_i[k]], cj = L->fcol[pb->pre_j[k]];
        for (int row = 0; row < 15; ++row) {
            double v = 0;
            if (ci >= 0) for (int a = 0; a < 15; ++a) v += L->Ji[225 * k + 15 * row + a] * xp[15 * ci + a];
            if (cj >= 0) for (int a = 0; a < 15; ++a) v += L->Jj[225 * k + 15 * row + a] * xp[15 * cj + a];
            sq += v * v;
            dr += v * L->r_p[15 * k + row];
        }
    }
    if (L->np > 0) {
        int D = L->D;
        for (int row = 0; row < D; ++row) {
            double v = 0;
            for (int i = 0; i < L->np; ++i) {
                int c = L->fcol[pb->prior_frames[i]];
                if (c < 0) continue;
                for (int a = 0; a < 15; ++a) v += L->Jm[row * D + 15 * i + a] * xp[15 * c + a];
            }
            sq += v * v;
            dr += v * L->r_m[row];
        }
    }
    *jx_sq = sq;
    *jx_dot_r = dr;
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

int ro_ba_solve(const ro_ba_problem *pb, int max_iterations, double *states_io, double *inv_depth_io,
                ro_ba_summary *sum) {
    lin_t Ls, *L = &Ls;
    memset(L, 0, sizeof *L);
    L->pb = pb;
    L->nfr = pb->n_frames; L->nl = pb->n_landmarks; L->nf = pb->n_factors; L->npre = pb->n_preint;
    L->nrot = pb->n_rot; L->np = pb->np; L->D = 15 * pb->np;
    L->fcol = (int *)malloc(sizeof(int) * L->nfr);
    L->nfree = 0;
    for (int i = 0; i < L->nfr; ++i) L->fcol[i] = (pb->frame_fixed[i] == 1) ? -1 : L->nfree++;
    L->N = 15 * L->nfree;
    int N = L->N, nl = L->nl, nfree = L->nfree;
    L->lfree = (int *)calloc(nl > 0 ? nl : 1, sizeof(int));
    for (int k = 0; k < L->nf; ++k)
        if (!pb->lm_fixed[pb->lm[k]]) L->lfree[pb->lm[k]] = 1; /* landmarks without factors are not in the program */
#define ALLOC(n) ((double *)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(double)))
    L->r_f = ALLOC(2 * L->nf); L->Jt = ALLOC(12 * L->nf); L->Jr = ALLOC(12 * L->nf); L->Jd = ALLOC(2 * L->nf);
    L->r_p = ALLOC(15 * L->npre); L->Ji = ALLOC(225 * L->npre); L->Jj = ALLOC(225 * L->npre);
    L->r_m = ALLOC(L->D); L->Jm = ALLOC((size_t)L->D * L->D);
    L->r_r = ALLOC(2 * L->nrot); L->Jro = ALLOC(6 * L->nrot);
    L->c_f = ALLOC(2 * L->nf); L->c_p = ALLOC(15 * L->npre); L->c_m = ALLOC(L->D); L->c_r = ALLOC(2 * L->nrot);
    L->H = ALLOC((size_t)N * N); L->g = ALLOC(N);
    L->lm_m = ALLOC(nl); L->lm_g = ALLOC(nl); L->lm_h = ALLOC((size_t)nl * nfree * 6);

    double *x = (double *)malloc(sizeof(double) * 16 * L->nfr), *xc = (double *)malloc(sizeof(double) * 16 * L->nfr);
    double *user = (double *)malloc(sizeof(double) * 16 * L->nfr);
    double *xd = ALLOC(nl), *xdc = ALLOC(nl);
    memcpy(x, states_io, sizeof(double) * 16 * L->nfr);
    memcpy(user, x, sizeof(double) * 16 * L->nfr);
    memcpy(xd, inv_depth_io, sizeof(double) * nl);

    double *sig_p = ALLOC(N), *sig_l = ALLOC(nl);          /* jacobi scaling */
    double *diag_p = ALLOC(N), *diag_l = ALLOC(nl);        /* dogleg diagonal (scaled space) */
    double *grad_p = ALLOC(N), *grad_l = ALLOC(nl);        /* dogleg gradient_ = g_s / diagonal */
    double *gn_p = ALLOC(N), *gn_l = ALLOC(nl);            /* gauss_newton_step_ (D-scaled) */
    double *step_p = ALLOC(N), *step_l = ALLOC(nl);        /* trust-region step (scaled variables) */
    double *S = ALLOC((size_t)N * N), *Lc = ALLOC((size_t)N * N), *rhs = ALLOC(N), *yp = ALLOC(N), *yl = ALLOC(nl);
    double *tp = ALLOC(N), *tl = ALLOC(nl), *wl = ALLOC(nl);

    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_step_norm = 0;
    int reuse = 0, iteration = 0, invalid_steps = 0, last_successful = 0;
    int term = RO_TERM_NO_CONVERGENCE;

    /* x_norm over the free parameter blocks in ambient space */
    double x_norm;
#define X_NORM(out, st, dep)                                                              \
    do {                                                                                  \
        double s__ = 0;                                                                   \
        for (int i = 0; i < L->nfr; ++i)                                                  \
            if (L->fcol[i] >= 0) for (int a = (pb->frame_fixed[i] == 2 ? 7 : 0); a < 16; ++a) s__ += (st)[16 * i + a] * (st)[16 * i + a]; \
        for (int l = 0; l < nl; ++l) if (L->lfree[l]) s__ += (dep)[l] * (dep)[l];        \
        (out) = sqrt(s__);                                                                \
    } while (0)
    X_NORM(x_norm, x, xd);

    double x_cost = evaluate(L, x, xd, user, 1);
    double initial_cost = x_cost;
    /* jacobi scaling from the iteration-0 Jacobian: 1 / (1 + sqrt(column squared norm)) */
    for (int i = 0; i < N; ++i) sig_p[i] = 1.0 / (1.0 + sqrt(L->H[i * N + i]));
    for (int l = 0; l < nl; ++l) sig_l[l] = 1.0 / (1.0 + sqrt(L->lm_m[l]));
    double grad_max;
    /* gradient_max_norm = || x - Plus(x, -g) ||_inf (g unscaled, tangent space) */
#define GRAD_MAX(out)                                                                      \
    do {                                                                                   \
        double m__ = 0;                                                                    \
        for (int i = 0; i < L->nfr; ++i) {                                                 \
            int c = L->fcol[i];                                                            \
            if (c < 0) continue;                                                           \
            double d15[15], o[16];                                                         \
            for (int a = 0; a < 15; ++a) d15[a] = -L->g[15 * c + a];                       \
            state_plus(x + 16 * i, d15, o);                                                \
            for (int a = (pb->frame_fixed[i] == 2 ? 7 : 0); a < 16; ++a) { double e__ = fabs(x[16 * i + a] - o[a]); if (e__ > m__) m__ = e__; } \
        }                                                                                  \
        for (int l = 0; l < nl; ++l) if (L->lfree[l]) { double e__ = fabs(L->lm_g[l]); if (e__ > m__) m__ = e__; } \
        (out) = m__;                                                                       \
    } while (0)
    GRAD_MAX(grad_max);

    int n_success = 0;
    if (N == 0 && nl == 0) { term = RO_TERM_CONVERGENCE; goto done; }
    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (last_successful) memcpy(user, x, sizeof(double) * 16 * L->nfr); /* state-updating callback */
        if (iteration >= max_iterations) { term = RO_TERM_NO_CONVERGENCE; break; }
        if (grad_max <= 1e-10) { term = RO_TERM_CONVERGENCE; break; }
        if (radius <= 1e-32) { term = RO_TERM_CONVERGENCE; break; }
        iteration++;
        last_successful = 0;

        /* ---- DoglegStrategy::ComputeStep ---- */
        int solve_ok = 1;
        if (!reuse) {
            reuse = 1;
            double gsq = 0;
            for (int i = 0; i < N; ++i) {
                diag_p[i] = sqrt(clampd(sig_p[i] * sig_p[i] * L->H[i * N + i], 1e-6, 1e32));
                grad_p[i] = sig_p[i] * L->g[i] / diag_p[i];
                gsq += grad_p[i] * grad_p[i];
            }
            for (int l = 0; l < nl; ++l) {
                if (!L->lfree[l]) { diag_l[l] = 1; grad_l[l] = 0; continue; }
                diag_l[l] = sqrt(clampd(sig_l[l] * sig_l[l] * L->lm_m[l], 1e-6, 1e32));
                grad_l[l] = sig_l[l] * L->lm_g[l] / diag_l[l];
                gsq += grad_l[l] * grad_l[l];
            }
            /* Cauchy point: alpha = |gradient_|^2 / |J_s D^-1 gradient_|^2 */
            for (int i = 0; i < N; ++i) tp[i] = sig_p[i] * grad_p[i] / diag_p[i];
            for (int l = 0; l < nl; ++l) tl[l] = L->lfree[l] ? sig_l[l] * grad_l[l] / diag_l[l] : 0.0;
            double jsq, jdr;
            jx_products(L, tp, tl, &jsq, &jdr);
            alpha = gsq / jsq;
            /* Gauss-Newton step with the mu-regularised normal equations, landmarks eliminated (Schur) */
            solve_ok = 0;
            while (mu < 1.0) {
                for (int l = 0; l < nl; ++l) {
                    if (!L->lfree[l]) { wl[l] = 0; continue; }
                    double C = sig_l[l] * sig_l[l] * L->lm_m[l] + mu * diag_l[l] * diag_l[l];
                    wl[l] = sig_l[l] * sig_l[l] / C;
                }
                for (int i = 0; i < N; ++i) {
                    for (int j = 0; j < N; ++j) S[i * N + j] = L->H[i * N + j];
                    rhs[i] = L->g[i];
                }
                for (int l = 0; l < nl; ++l) {
                    if (!L->lfree[l]) continue;
                    const double *h = L->lm_h + (size_t)l * nfree * 6;
                    for (int fi = 0; fi < nfree; ++fi) {
                        const double *hi = h + 6 * fi;
                        int nz = 0;
                        for (int a = 0; a < 6; ++a) if (hi[a] != 0.0) nz = 1;
                        if (!nz) continue;
                        for (int fj = 0; fj < nfree; ++fj) {
                            const double *hj = h + 6 * fj;
                            for (int a = 0; a < 6; ++a)
                                for (int b = 0; b < 6; ++b) S[(15 * fi + a) * N + 15 * fj + b] -= hi[a] * wl[l] * hj[b];
                        }
                        for (int a = 0; a < 6; ++a) rhs[15 * fi + a] -= hi[a] * wl[l] * L->lm_g[l];
                    }
                }
                for (int i = 0; i < N; ++i) {
                    for (int j = 0; j < N; ++j) S[i * N + j] *= sig_p[i] * sig_p[j];
                    S[i * N + i] += mu * diag_p[i] * diag_p[i];
                    rhs[i] *= sig_p[i];
                }
                int ok = (N == 0) ? 1 : ro_cholesky_lower(Lc, S, N);
                if (ok) {
                    for (int i = 0; i < N; ++i) { /* forward */
                        double s = rhs[i];
                        for (int k = 0; k < i; ++k) s -= Lc[i * N + k] * yp[k];
                        yp[i] = s / Lc[i * N + i];
                    }
                    for (int i = N - 1; i >= 0; --i) { /* backward */
                        double s = yp[i];
                        for (int k = i + 1; k < N; ++k) s -= Lc[k * N + i] * yp[k];
                        yp[i] = s / Lc[i * N + i];
                    }
                    for (int l = 0; l < nl; ++l) {
                        if (!L->lfree[l]) { yl[l] = 0; continue; }
                        const double *h = L->lm_h + (size_t)l * nfree * 6;
                        double s = L->lm_g[l];
                        for (int fi = 0; fi < nfree; ++fi)
                            for (int a = 0; a < 6; ++a) s -= h[6 * fi + a] * sig_p[15 * fi + a] * yp[15 * fi + a];
                        double C = sig_l[l] * sig_l[l] * L->lm_m[l] + mu * diag_l[l] * diag_l[l];
                        yl[l] = sig_l[l] * s / C;
                    }
                    for (int i = 0; i < N; ++i) if (!isfinite(yp[i])) ok = 0;
                    for (int l = 0; l < nl; ++l) if (!isfinite(yl[l])) ok = 0;
                }
                if (!ok) { mu *= 10.0; continue; }
                solve_ok = 1;
                break;
            }
            if (solve_ok) {
                for (int i = 0; i < N; ++i) gn_p[i] = -yp[i] * diag_p[i];
                for (int l = 0; l < nl; ++l) gn_l[l] = -yl[l] * diag_l[l];
            }
        }
        int step_valid = 0;
        double model_cost_change = 0;
        if (solve_ok) {
            /* ComputeTraditionalDoglegStep */
            double gnorm = 0, gn_norm = 0, gdotgn = 0;
            for (int i = 0; i < N; ++i) { gnorm += grad_p[i] * grad_p[i]; gn_norm += gn_p[i] * gn_p[i]; gdotgn += grad_p[i] * gn_p[i]; }
            for (int l = 0; l < nl; ++l) { gnorm += grad_l[l] * grad_l[l]; gn_norm += gn_l[l] * gn_l[l]; gdotgn += grad_l[l] * gn_l[l]; }
            gnorm = sqrt(gnorm); gn_norm = sqrt(gn_norm);
            double ca, cb; /* step = ca * gradient_ + cb * gauss_newton */
            if (gn_norm <= radius) { ca = 0; cb = 1; dogleg_step_norm = gn_norm; }
            else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0; dogleg_step_norm = radius; }
            else {
                double b_dot_a = -alpha * gdotgn;
                double a_sq = pow(alpha * gnorm, 2.0);
                double bma_sq = a_sq - 2 * b_dot_a + pow(gn_norm, 2);
                double c = b_dot_a - a_sq;
                double d = sqrt(c * c + bma_sq * (pow(radius, 2.0) - a_sq));
                double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
                ca = -alpha * (1.0 - beta); cb = beta;
                dogleg_step_norm = -1; /* computed below */
            }
            double sn = 0;
            for (int i = 0; i < N; ++i) { double v = ca * grad_p[i] + cb * gn_p[i]; sn += v * v; step_p[i] = v / diag_p[i]; }
            for (int l = 0; l < nl; ++l) { double v = ca * grad_l[l] + cb * gn_l[l]; sn += v * v; step_l[l] = v / diag_l[l]; }
            if (dogleg_step_norm < 0) dogleg_step_norm = sqrt(sn);
            /* model_cost_change = -(J_s step)'(f + J_s step / 2) */
            for (int i = 0; i < N; ++i) tp[i] = sig_p[i] * step_p[i];
            for (int l = 0; l < nl; ++l) tl[l] = L->lfree[l] ? sig_l[l] * step_l[l] : 0.0;
            double jsq, jdr;
            jx_products(L, tp, tl, &jsq, &jdr);
            model_cost_change = -(jdr + 0.5 * jsq);
            step_valid = model_cost_change > 0.0;
        }
        if (getenv("RO_SOLVER_DEBUG")) fprintf(stderr, "it %d cost %.9g solve_ok %d mu %.3g radius %.3g mcc %.6g gradmax %.3g reuse %d\n", iteration, x_cost, solve_ok, mu, radius, model_cost_change, grad_max, reuse);
        if (!step_valid) {
            /* HandleInvalidStep + DoglegStrategy::StepIsInvalid */
            if (++invalid_steps >= 5) { term = RO_TERM_FAILURE; break; }
            mu *= 10.0;
            reuse = 0;
            continue;
        }
        invalid_steps = 0;
        /* candidate = Plus(x, delta), delta = step * jacobi scaling  (tp/tl hold it) */
        for (int i = 0; i < L->nfr; ++i) {
            int c = L->fcol[i];
            if (c < 0) memcpy(xc + 16 * i, x + 16 * i, 16 * sizeof(double));
            else {
                state_plus(x + 16 * i, tp + 15 * c, xc + 16 * i);
                if (pb->frame_fixed[i] == 2) memcpy(xc + 16 * i, x + 16 * i, 7 * sizeof(double)); /* constant pose block */
            }
        }
        for (int l = 0; l < nl; ++l) xdc[l] = xd[l] + (L->lfree[l] ? tl[l] : 0.0);
        double cand_cost = evaluate(L, xc, xdc, user, 0);
        if (!isfinite(cand_cost)) cand_cost = DBL_MAX;
        /* ParameterToleranceReached: ambient-space step norm */
        double step_norm = 0;
        for (int i = 0; i < L->nfr; ++i)
            if (L->fcol[i] >= 0) for (int a = (pb->frame_fixed[i] == 2 ? 7 : 0); a < 16; ++a) { double e = x[16 * i + a] - xc[16 * i + a]; step_norm += e * e; }
        for (int l = 0; l < nl; ++l) if (L->lfree[l]) { double e = xd[l] - xdc[l]; step_norm += e * e; }
        step_norm = sqrt(step_norm);
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { term = RO_TERM_CONVERGENCE; break; }
        /* FunctionToleranceReached */
        double cost_change = x_cost - cand_cost;
        if (fabs(cost_change) <= 1e-6 * x_cost) { term = RO_TERM_CONVERGENCE; break; }
        /* IsStepSuccessful */
        if (getenv("RO_SOLVER_DEBUG")) fprintf(stderr, "   cand %.9g step_norm %.3g\n", cand_cost, step_norm);
        double rel = (cand_cost >= DBL_MAX) ? -DBL_MAX : (x_cost - cand_cost) / model_cost_change;
        if (rel > 1e-3) {
            memcpy(x, xc, sizeof(double) * 16 * L->nfr);
            memcpy(xd, xdc, sizeof(double) * nl);
            X_NORM(x_norm, x, xd);
            x_cost = evaluate(L, x, xd, user, 1); /* user state still holds the previous point here */
            GRAD_MAX(grad_max);
            last_successful = 1;
            n_success++;
            /* DoglegStrategy::StepAccepted */
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
            mu = fmax(1e-8, 2.0 * mu / 10.0);
            reuse = 0;
        } else {
            radius *= 0.5; /* StepRejected */
            reuse = 1;
        }
    }
done:
    memcpy(states_io, x, sizeof(double) * 16 * L->nfr);
    memcpy(inv_depth_io, xd, sizeof(double) * nl);
    if (sum) {
        sum->iterations = iteration;
        sum->successful_steps = n_success;
        sum->initial_cost = initial_cost;
        sum->final_cost = x_cost;
        sum->termination = term;
    }
    free(L->fcol); free(L->lfree); free(L->r_f); free(L->Jt); free(L->Jr); free(L->Jd); free(L->r_p); free(L->Ji);
    free(L->Jj); free(L->r_m); free(L->Jm); free(L->r_r); free(L->Jro); free(L->H); free(L->g); free(L->lm_m);
    free(L->lm_g); free(L->lm_h); free(L->c_f); free(L->c_p); free(L->c_m); free(L->c_r); free(x); free(xc); free(user); free(xd); free(xdc); free(sig_p); free(sig_l);
    free(diag_p); free(diag_l); free(grad_p); free(grad_l); free(gn_p); free(gn_l); free(step_p); free(step_l);
    free(S); free(Lc); free(rhs); free(yp); free(yl); free(tp); free(tl); free(wl);
    return term != RO_TERM_FAILURE;
}
