/*
 * ORACLE -- CPU restatement of rd_vio's hot path (test infrastructure only).
 *
 * This library is the CHECKER for the HIP path in rd_vio_amd/: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (librdvio_hip.so) never links, loads or falls back to it.
 *
 * PARITY UNPINNED: /root/reference has no tests, fixtures or golden vectors and
 * cannot be compiled here (Eigen / Ceres / OpenCV / yaml-cpp absent), so this
 * restatement is anchored on the reference's source text (cited per function)
 * and on self-consistency checks in tests/ (finite differences, Schur-vs-dense
 * identities, closed forms).  The OpenCV and Ceres arithmetic it restates
 * (LK, CLAHE, pyramid, GFTT; dogleg/Schur) follows their published algorithms
 * (unpinned versions: OpenCV 4.x, Ceres >= 2.1).
 *
 * Array layouts (shared with include/rdvio_hip.h):
 *   frame state   double[16] : q(x,y,z,w) p(3) v(3) bg(3) ba(3)
 *   extrinsics    double[14] : cam q_cs(4) p_cs(3), imu q_cs(4) p_cs(3)
 *   preint        double[RO_PREINT_SIZE] : t, q(4), p(3), v(3), cov(225), sqrt_inv_cov(225),
 *                                          dq_dbg(9) dp_dbg(9) dp_dba(9) dv_dbg(9) dv_dba(9)
 *   error state   theta(0..2) p(3..5) v(6..8) bg(9..11) ba(12..14)   (estimation/state.h:11-18)
 *   matrices are row-major.
 */
#ifndef RDVIO_ORACLE_H
#define RDVIO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RO_STATE_SIZE 16
#define RO_ES_SIZE 15
#define RO_PREINT_SIZE 506
#define RO_PREINT_T 0
#define RO_PREINT_Q 1
#define RO_PREINT_P 5
#define RO_PREINT_V 8
#define RO_PREINT_COV 11
#define RO_PREINT_SIC 236
#define RO_PREINT_JAC 461

/* ---- lie algebra helpers exposed for tests (lie_algebra.{h,cpp}) ---- */
void ro_expmap(const double *w, double *q);
void ro_logmap(const double *q, double *w);
void ro_right_jacobian_c(const double *w, double *J);
void ro_tangent_frame(const double *z, double *T /*3x3 row-major [b1 b2 z] columns*/);
void ro_quat_plus(const double *q, const double *delta, double *out); /* quaternion_parameterization.h:11-17 */

/* ---- A7: PreIntegrator (preintegrator.cpp:7-112) ---- */
/* imu: n x 7 (t, w[3], a[3]); noise: cov_w, cov_a, cov_bg, cov_ba (4 x 9 row-major) */
int ro_preintegrate(int n, const double *imu, double t_end, const double *bg, const double *ba,
                    const double *noise, int compute_jacobian, int compute_covariance, double *preint);
/* PreIntegrator::predict, preintegrator.cpp:102-112 */
void ro_preint_predict(const double *preint, const double *state_i, double *state_j);

/* ---- A8/A9: reprojection factor (ceres/reprojection_factor.h:12-121) ---- */
/* For each factor k: r[2k..], Jt[12k..] (2x6: theta_tgt,p_tgt), Jr[12k..] (2x6: theta_ref,p_ref), Jd[2k..].
 * J pointers may be NULL (residual only). */
void ro_reprojection_eval(int nf, const int32_t *tgt, const int32_t *ref, const int32_t *lm,
                          const double *tangent /*nf x 9*/, const double *z_ref /*nl x 3*/,
                          const double *inv_depth /*nl*/, const double *states /*nframes x 16*/,
                          const double *extr /*14*/, const double *sqrt_inv_cov /*4*/,
                          double *r, double *Jt, double *Jr, double *Jd);

/* ---- A10: rotation prior (ceres/rotation_factor.h:11-67) ---- */
void ro_rotation_prior_eval(const double *q_tgt, const double *q_ref, const double *z_ref,
                            const double *tangent, const double *extr, const double *sqrt_inv_cov,
                            double *r /*2*/, double *J /*2x3*/);

/* ---- A11: preintegration error factor (ceres/preintegration_factor.h:11-163) ---- */
/* bias_lin = (bg_i0, ba_i0) linearisation biases; r[15], Ji[15x15], Jj[15x15] (tangent columns) */
void ro_preintegration_eval(const double *state_i, const double *state_j, const double *preint,
                            const double *bias_lin /*6*/, const double *extr,
                            double *r, double *Ji, double *Jj);

/* ---- A12: marginalisation prior Evaluate (ceres/marginalization_factor.h:27-72) ---- */
/* np frames; lin = np x 16 linearisation states; S = D x D (D = 15 np); f = D.
 * r[D], J[D x D] (tangent columns); J may be NULL */
void ro_marginalization_eval(int np, const double *states /*np x 16, the prior's frames in order*/,
                             const double *lin, const double *S, const double *f, double *r, double *J);

/* ---- A13: CeresMarginalizationFactor::marginalize(0) (ceres/marginalization_factor.h:74-475) ---- */
typedef struct {
    int nframes;              /* frames in the map (victim = frame 0) */
    const double *states;     /* nframes x 16 */
    const double *extr;       /* 14 */
    const double *sqrt_inv_cov; /* 4 */
    /* current prior */
    int np;                   /* frames covered by the prior */
    const int32_t *prior_frames; /* np map-frame indices */
    const double *lin;        /* np x 16 */
    const double *S;          /* (15 np)^2 */
    const double *f;          /* 15 np */
    /* preintegration between map frames 0 and 1 (keyframe_preintegration of frame 1); NULL if nframes < 2 */
    const double *preint01;
    /* reprojection factors of victim-observed tracks (host-selected: marginalization_factor.h:233-380) */
    int nfac;
    const int32_t *tgt, *ref, *lm;
    const double *tangent;
    int nlm;
    const double *z_ref;
    const double *inv_depth;
} ro_marg_problem;
/* outputs: S_out ((15 (nframes-1))^2), f_out, lin_out ((nframes-1) x 16);
 * optional Lambda_out / eta_out = reduced information matrix / vector before the eigen step */
void ro_marginalize(const ro_marg_problem *pb, double *S_out, double *f_out, double *lin_out,
                    double *Lambda_out, double *eta_out);

/* ================= image side (rows A1-A3; OpenCV arithmetic restated, see ro_image.c) ================= */
#define RO_MAX_LEVELS 4      /* level_num() == 3 -> levels 0..3, opencv_image.h:19 */
#define RO_LK_WIN 21         /* Size(21,21), opencv_image.cpp:96 */
#define RO_PYR_BORDER 32     /* >= winSize+1; OpenCV pads by winSize=21, the extra columns are never read */

/* One padded arena per frame.  Level l of the u8 image lives at img_off[l] (bytes) with row stride
 * stride[l] and a `border`-pixel frame (BORDER_REFLECT_101); derivatives are interleaved int16 (dx,dy)
 * at deriv_off[l] (int16 elements), same stride in pixels, zero border (BORDER_CONSTANT). */
typedef struct {
    int32_t levels;
    int32_t border;
    int32_t w[RO_MAX_LEVELS], h[RO_MAX_LEVELS], stride[RO_MAX_LEVELS];
    int64_t img_off[RO_MAX_LEVELS], deriv_off[RO_MAX_LEVELS];
    int64_t img_bytes, deriv_elems;
} ro_pyr_layout;

void ro_pyr_layout_init(int w, int h, int max_level, ro_pyr_layout *L);
void ro_clahe(const uint8_t *src, int w, int h, int src_stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t *dst, int dst_stride);
void ro_build_pyramid(const uint8_t *img, int w, int h, int img_stride, const ro_pyr_layout *L, uint8_t *pyr_img,
                      int16_t *pyr_deriv);
void ro_preprocess(const uint8_t *gray, int w, int h, int stride, double clip, int tiles_x, int tiles_y,
                   const ro_pyr_layout *L, uint8_t *pyr_img, int16_t *pyr_deriv);
void ro_lk_flow(const ro_pyr_layout *L, const uint8_t *prev_img, const int16_t *prev_deriv, const uint8_t *next_img,
                int n, const float *prev_xy, float *next_xy, uint8_t *status, int max_iter, double eps);
void ro_track_keypoints(const ro_pyr_layout *L, const uint8_t *cur_img, const int16_t *cur_deriv,
                        const uint8_t *nxt_img, const int16_t *nxt_deriv, int n, const double *curr,
                        double *next, int has_guess, uint8_t *status);
void ro_harris_response(const uint8_t *img, int w, int h, int stride, double k, float *resp);
int ro_good_features(const uint8_t *img, int w, int h, int stride, int max_corners, double quality, double min_dist,
                     double k, float *out_xy, float *out_resp);
int ro_detect_keypoints(const uint8_t *img, int w, int h, int stride, int max_corners, double min_dist_poisson,
                        double *keypoints, int n_existing);

/* ================= A14: the non-linear solve (Ceres restated, see ro_solver.c) ================= */
#define RO_TERM_CONVERGENCE 0
#define RO_TERM_NO_CONVERGENCE 1
#define RO_TERM_FAILURE 2

/* One Solver problem in SoA form (what Solver::add_* assemble through pointers, solver.cpp:88-178).
 * "prior" factor flavours are expressed through the fixed flags: a ReprojectionPriorFactor is a
 * reprojection factor whose anchor frame and landmark are fixed; a PreIntegrationPriorFactor has frame i fixed. */
typedef struct {
    int n_frames;
    const uint8_t *frame_fixed;   /* 1 = constant (FT_FIX_POSE|FT_FIX_MOTION, or not a parameter of this solve); 2 = pose constant, motion free (FT_FIX_POSE only) */
    const double *extr;           /* 14 */
    const double *sqrt_inv_cov;   /* 4 */
    int n_landmarks;
    const uint8_t *lm_fixed;      /* 1 = constant inverse depth */
    const double *z_ref;          /* n_landmarks x 3 */
    int n_factors;                /* reprojection factors (CauchyLoss) */
    const int32_t *tgt, *ref, *lm;
    const double *tangent;        /* n_factors x 9 */
    int n_rot;                    /* rotation priors (CauchyLoss), ceres/rotation_factor.h */
    const int32_t *rot_tgt, *rot_ref;
    const double *rot_zref;       /* n_rot x 3 */
    const double *rot_tangent;    /* n_rot x 9 */
    int n_preint;                 /* preintegration factors (no loss) */
    const int32_t *pre_i, *pre_j;
    const double *preint;         /* n_preint x RO_PREINT_SIZE */
    int np;                       /* marginalisation prior over np frames (0 = none) */
    const int32_t *prior_frames;
    const double *lin, *S, *f;
} ro_ba_problem;

typedef struct {
    int iterations;        /* trust-region iterations performed (Ceres iteration counter) */
    int successful_steps;
    double initial_cost, final_cost;
    int termination;
} ro_ba_summary;

/* states_io: n_frames x 16, inv_depth_io: n_landmarks; updated in place like Solver::solve (solver.cpp:191). */
int ro_ba_solve(const ro_ba_problem *pb, int max_iterations, double *states_io, double *inv_depth_io,
                ro_ba_summary *summary);

#ifdef __cplusplus
}
#endif
#endif
