/*
 * rdvio_hip.h -- C ABI of the MI355X (gfx950) implementation of rd_vio's data-parallel hot path.
 *
 * This is the drop-in boundary: plain C, POD pointers and sizes, int status
 * returns (0 = ok), no exceptions, no torch/Eigen/OpenCV types.  It sits exactly at the two plugin
 * seams the reference already has:
 *
 *   seam 1  abstract rdvio::Image            src/rdvio/include/rdvio/types.h:153-177
 *           (sole implementation OpenCvImage, src/rdvio_extra/src/opencv_image.cpp:38-161)
 *   seam 2  rdvio::Solver facade + factors   src/rdvio_estimation/include/rdvio/estimation/solver.h:15-70
 *           MarginalizationFactor::marginalize   .../marginalization_factor.h:9-12
 *           PreIntegrator                        .../preintegrator.h:10-47
 *
 * INTEGRATION.md shows the reference-side subclasses (HipImage : rdvio::Image, the Solver facade and
 * the MarginalizationFactor subclass) a maintainer adds to bind these entry points.
 *
 * Conventions (same as the reference):
 *   quaternion (x,y,z,w); frame state double[16] = q(4) p(3) v(3) bg(3) ba(3);
 *   error state theta(0..2) p(3..5) v(6..8) bg(9..11) ba(12..14); matrices row-major;
 *   extrinsics double[14] = camera q_cs(4) p_cs(3), imu q_cs(4) p_cs(3).
 * Host entry points take host pointers and copy; *_dev entry points take device pointers (HBM-resident
 * data, e.g. torch tensors' data_ptr()) and only enqueue work on the context's stream.
 */
#ifndef RDVIO_HIP_H
#define RDVIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDVIO_OK 0
#define RDVIO_ERR_INVALID 1   /* bad argument / shape mismatch (checked on the host before any launch) */
#define RDVIO_ERR_HIP 2       /* a HIP runtime call failed; see rdvio_hip_last_error() */
#define RDVIO_ERR_CAPACITY 3  /* problem larger than the context was created for */
#define RDVIO_ERR_TIMEOUT 4   /* a device-side bounded wait expired (multi-workgroup solve); outputs hold the last accepted point */

#define RDVIO_MAX_LEVELS 4    /* OpenCvImage::level_num() == 3 -> levels 0..3 (opencv_image.h:19) */
#define RDVIO_LK_WIN 21       /* Size(21,21), opencv_image.cpp:96 */
#define RDVIO_PYR_BORDER 32
#define RDVIO_STATE_SIZE 16
#define RDVIO_ES_SIZE 15
#define RDVIO_PREINT_SIZE 506 /* t, q(4), p(3), v(3), cov(225), sqrt_inv_cov(225), dq_dbg dp_dbg dp_dba dv_dbg dv_dba (5x9) */

typedef struct rdvio_hip_ctx rdvio_hip_ctx;

/* Padded pyramid arena of one frame (replaces the std::vector<cv::Mat> image_pyramid of
 * OpenCvImage, opencv_image.h:40): level l of the u8 image at img_off[l] (bytes), row stride
 * stride[l] pixels, `border` pixels of BORDER_REFLECT_101 around it; Scharr derivatives as
 * interleaved int16 (dx,dy) at deriv_off[l] (int16 elements) with a zero border. */
typedef struct {
    int32_t levels;
    int32_t border;
    int32_t w[RDVIO_MAX_LEVELS], h[RDVIO_MAX_LEVELS], stride[RDVIO_MAX_LEVELS];
    int64_t img_off[RDVIO_MAX_LEVELS], deriv_off[RDVIO_MAX_LEVELS];
    int64_t img_bytes, deriv_elems;
} rdvio_pyr_layout;

/* ------------------------------------------------------------------ context */
/* One context per process/stream (the reference's process-global state makes it one Odometry per
 * process anyway).  `stream` is a hipStream_t to enqueue on (NULL = context-owned stream). */
int rdvio_hip_ctx_create(rdvio_hip_ctx **out, int device, int max_width, int max_height, int max_features,
                         int max_window, int max_factors, void *stream);
void rdvio_hip_ctx_destroy(rdvio_hip_ctx *ctx);
const char *rdvio_hip_last_error(const rdvio_hip_ctx *ctx);
int rdvio_hip_sync(rdvio_hip_ctx *ctx);
/* Lanes.  The reference runs the feature tracker and the frontend (sliding-window tracker) on two worker threads
 * (src/rdvio/src/handler.cpp:35-50); here each gets a HIP stream, and marginalisation -- whose result is first read by the
 * NEXT refine_window (sliding_window_tracker.cpp:339-347, 226-300) -- a third.  Every entry point enqueues on the lane of
 * its seam and touches only that lane's staging buffers:
 *   RDVIO_LANE_FRONTEND  Image::preprocess / track_keypoints / detect_keypoints, PreIntegrator::integrate, unit entries
 *   RDVIO_LANE_SOLVER    rdvio_hip_ba_*            (Solver::solve)
 *   RDVIO_LANE_MARG      rdvio_hip_marginalize*    (MarginalizationFactor::marginalize)
 * All lanes are the context's stream (everything serial, in call order) until a lane gets a stream of its own with
 * rdvio_hip_ctx_set_lane_stream (stream == NULL: the context creates and owns one).  Host entry points synchronise only
 * their own lane; rdvio_hip_lane_wait(lane, on_lane) makes work enqueued on `lane` from now on wait for everything enqueued
 * on `on_lane` so far (a device-side dependency, no host wait); rdvio_hip_sync waits for all lanes. */
#define RDVIO_LANE_FRONTEND 0
#define RDVIO_LANE_SOLVER 1
#define RDVIO_LANE_MARG 2
int rdvio_hip_ctx_set_lane_stream(rdvio_hip_ctx *ctx, int lane, void *stream);
/* gives the solver and the marginalisation lane a context-owned stream each unless they already have one (a threaded
 * pipeline calls this: two host threads on one stream would wait for each other's kernels) */
int rdvio_hip_ctx_ensure_lane_streams(rdvio_hip_ctx *ctx);
/* How the library's host-side waits wait: 0 (default) = spinning (hipStreamSynchronize: lowest latency, one core per waiting
 * thread); 1 = blocking on an event created with hipEventBlockingSync (the waiting thread sleeps) -- for processes that drive
 * more sequences than they have cores (rdvio_hip_run_sequences switches its contexts to 1 for the run when n_seq exceeds
 * the hardware concurrency). */
int rdvio_hip_ctx_set_wait_mode(rdvio_hip_ctx *ctx, int blocking);
int rdvio_hip_lane_wait(rdvio_hip_ctx *ctx, int lane, int on_lane);
/* Binds the calling thread to the context's device (hipSetDevice): call once on every host thread other than the creating one
 * before it uses the context (a new thread starts on device 0). */
int rdvio_hip_ctx_attach_thread(rdvio_hip_ctx *ctx);
int rdvio_hip_lane_sync(rdvio_hip_ctx *ctx, int lane);
int rdvio_hip_pyr_layout_init(int width, int height, int max_level, rdvio_pyr_layout *out);
const char *rdvio_hip_version(void);

/* ------------------------------------------------------------------ seam 1: rdvio::Image */
/* Image::preprocess (types.h:160; opencv_image.cpp:156-161): CLAHE(clip, tiles) in place, then the
 * 4-level pyramid with Scharr derivatives, into image slot `slot` (0/1: the tracker keeps exactly two
 * consecutive frames alive, feature_tracker.cpp:94). */
int rdvio_hip_image_preprocess(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray, int width, int height,
                               int stride, double clahe_clip, int tiles_x, int tiles_y);
int rdvio_hip_image_preprocess_dev(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray_dev, int width,
                                   int height, int stride, double clahe_clip, int tiles_x, int tiles_y);
/* The same in two halves, for a caller that receives the image before the tracker asks for it (Odometry::addFrame only enqueues,
 * handler.cpp:113-138): upload copies the pixels into the slot's pinned staging buffer (the caller's buffer is free when the call
 * returns) and enqueues the copy to the device; preprocess_uploaded enqueues the kernels on what was uploaded.  Neither waits. */
int rdvio_hip_image_upload(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray, int width, int height, int stride);
int rdvio_hip_image_preprocess_uploaded(rdvio_hip_ctx *ctx, int slot, double clahe_clip, int tiles_x, int tiles_y);
/* copy a slot's arenas back (parity tests) */
int rdvio_hip_image_download(rdvio_hip_ctx *ctx, int slot, uint8_t *pyr_img, int16_t *pyr_deriv);

/* Image::track_keypoints (types.h:165-168; opencv_image.cpp:75-154): forward LK from the initial guess,
 * 20-px border and rows/4 max-flow rejection, backward LK, 0.5-px forward-backward check.
 * curr_xy / next_xy: n x 2 doubles (pixels); has_guess == 0 <=> the reference's empty next_keypoints.
 * status[i] != 0 <=> survivor; next_xy is written only for survivors (opencv_image.cpp:148-153). */
int rdvio_hip_track_keypoints(rdvio_hip_ctx *ctx, int slot_curr, int slot_next, int n, const double *curr_xy,
                              double *next_xy, int has_guess, uint8_t *status);
int rdvio_hip_track_keypoints_dev(rdvio_hip_ctx *ctx, int slot_curr, int slot_next, int n,
                                  const double *curr_xy_dev, double *next_xy_dev, int has_guess,
                                  uint8_t *status_dev);
/* one cv::calcOpticalFlowPyrLK call (float points, OPTFLOW_USE_INITIAL_FLOW), exposed for unit parity */
int rdvio_hip_lk_flow(rdvio_hip_ctx *ctx, int slot_prev, int slot_next, int n, const float *prev_xy,
                      float *next_xy, uint8_t *status, int max_iter, double eps);

/* Image::detect_keypoints (types.h:162-164; opencv_image.cpp:38-73): GFTT-Harris (quality 1e-3,
 * minDistance 20, block 3, k 0.04) -> response order -> Poisson-disk thinning against the existing
 * points at `min_distance` -> 20-px border.  keypoints: in/out n x 2 doubles with capacity rows. */
int rdvio_hip_detect_keypoints(rdvio_hip_ctx *ctx, int slot, double *keypoints, int n_existing, int capacity,
                               int max_points, double min_distance, int *n_out);
/* Harris response map of level 0 of a slot (float, width*height) -- unit parity */
int rdvio_hip_harris_response(rdvio_hip_ctx *ctx, int slot, float *resp);
/* Image::release_image_buffer (types.h:170) */
int rdvio_hip_image_release(rdvio_hip_ctx *ctx, int slot);

/* ------------------------------------------------------------------ seam 2: estimation */
/* PreIntegrator::integrate (preintegrator.cpp:78-95): imu = n x 7 (t, gyro, acc), noise = cov_w cov_a
 * cov_bg cov_ba (4 x 9).  nseg independent segments in one launch: seg_off[nseg+1] sample offsets,
 * t_end/bg/ba per segment.  out: nseg x RDVIO_PREINT_SIZE. */
int rdvio_hip_preintegrate(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off, const double *imu,
                           const double *t_end, const double *bg, const double *ba, const double *noise,
                           int compute_jacobian, int compute_covariance, double *preint_out);
/* The same call for a second host thread: the estimator of a threaded pipeline (rdvio_pipeline_config::threading == 2)
 * integrates while the tracker does -- this entry enqueues on RDVIO_LANE_SOLVER and has staging buffers of its own, so the
 * two never meet. */
int rdvio_hip_preintegrate_estimator(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off, const double *imu,
                                     const double *t_end, const double *bg, const double *ba, const double *noise,
                                     int compute_jacobian, int compute_covariance, double *preint_out);
/* The same in two halves (the estimator's per-frame integration sits at the head of its step: _begin enqueues upload, kernel and
 * download on the solver lane and returns; _end waits and copies the nseg records out).  One integration in flight per context. */
int rdvio_hip_preintegrate_estimator_begin(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off, const double *imu, const double *t_end,
                                           const double *bg, const double *ba, const double *noise, int compute_jacobian, int compute_covariance);
int rdvio_hip_preintegrate_estimator_end(rdvio_hip_ctx *ctx, double *preint_out);
/* Same with every array resident in HBM: imu_dev (n x 7), par_dev (nseg x 7: t_end, bg, ba), noise_dev (36),
 * seg_off_dev (nseg+1), out_dev (nseg x RDVIO_PREINT_SIZE).  Only enqueues the kernel. */
int rdvio_hip_preintegrate_dev(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off_dev, const double *imu_dev,
                               const double *par_dev, const double *noise_dev, int compute_jacobian,
                               int compute_covariance, double *preint_out_dev);

/* One Solver problem in SoA form: what Solver::add_frame_states / add_track_states / add_factor assemble
 * through pointers (solver.cpp:88-178).  Which states and factors enter each solve is the caller's
 * (host) decision, exactly as in sliding_window_tracker.cpp:101-125 (localize_newframe), :226-300
 * (refine_window) and :349-444 (refine_subwindow).  The reference's "prior" factor flavours are the same
 * factors with constant blocks: ReprojectionPriorFactor = reprojection factor whose anchor frame and
 * landmark are fixed (reprojection_factor.h:99-121); PreIntegrationPriorFactor = frame i fixed
 * (preintegration_factor.h:165-198).  Factors must be ordered by landmark (lm non-decreasing). */
typedef struct {
    int32_t n_frames;
    const double *states;          /* n_frames x 16 (initial values) */
    const uint8_t *frame_fixed;    /* n_frames: 1 = constant block (FT_FIX_POSE|FT_FIX_MOTION, solver.cpp:92-113,
                                      or a frame that is not a parameter of this solve); 2 = pose constant, motion
                                      free (FT_FIX_POSE alone: the initializer's first keyframe, initializer.cpp:82) */
    const double *extr;            /* 14 */
    const double *sqrt_inv_cov;    /* 2 x 2 */
    int32_t n_landmarks;
    const double *z_ref;           /* n_landmarks x 3: bearing in the anchor frame */
    const double *inv_depth;       /* n_landmarks (initial values) */
    const uint8_t *lm_fixed;       /* n_landmarks: 1 = constant */
    int32_t n_factors;             /* reprojection factors, CauchyLoss(1.0) (solver.cpp:116-132) */
    const int32_t *tgt, *ref, *lm; /* target frame, anchor frame, landmark */
    const double *tangent;         /* n_factors x 9: [b1 b2 z_obs] (reprojection_factor.h:16-22) */
    int32_t n_rot;                 /* rotation priors, CauchyLoss(1.0) (solver.cpp:134-141; rotation_factor.h) */
    const int32_t *rot_tgt, *rot_ref;
    const double *rot_zref;        /* n_rot x 3 */
    const double *rot_tangent;     /* n_rot x 9 */
    int32_t n_preint;              /* preintegration factors, no loss (solver.cpp:143-170) */
    const int32_t *pre_i, *pre_j;
    const double *preint;          /* n_preint x RDVIO_PREINT_SIZE */
    int32_t n_prior;               /* frames covered by the marginalisation prior (0 = none; solver.cpp:172-178) */
    const int32_t *prior_frames;   /* n_prior frame indices */
    const double *prior_lin;       /* n_prior x 16 linearisation states */
    const double *prior_S;         /* (15 n_prior)^2 sqrt information */
    const double *prior_f;         /* 15 n_prior */
    /* Optional, fused PreIntegrator::integrate (preintegrator.cpp:78-95) of the records this solve uses: refine_window re-integrates
     * every keyframe interval and refine_subwindow every subframe interval right before their solve
     * (sliding_window_tracker.cpp:283-301, 379-384).  With n_pre_jobs == n_preint > 0 the records are not read from `preint` (may be
     * NULL) but integrated on the device from the raw samples (covariance and bias Jacobians on) into the solve's own record slots --
     * no host hop between the two kernels -- and handed back through job_preint_out by rdvio_hip_ba_fetch / rdvio_hip_ba_solve. */
    int32_t n_pre_jobs;            /* 0 or n_preint */
    const int32_t *job_seg_off;    /* n_pre_jobs + 1 sample offsets into job_imu */
    const double *job_imu;         /* samples x 7 (t, gyro, acc) */
    const double *job_par;         /* n_pre_jobs x 7: t_end, bg(3), ba(3) */
    const double *job_noise;       /* 36: cov_w cov_a cov_bg cov_ba */
    double *job_preint_out;        /* n_pre_jobs x RDVIO_PREINT_SIZE (host) */
} rdvio_ba_problem;

typedef struct {
    int32_t iterations;            /* trust-region iterations (Ceres' iteration counter) */
    int32_t successful_steps;
    double initial_cost, final_cost;
    int32_t termination;           /* 0 convergence, 1 iteration limit, 2 failure */
} rdvio_ba_summary;

/* CeresReprojectionErrorFactor::Evaluate for every factor (reprojection_factor.h:24-89).
 * r: F x 2; Jt: F x 2 x 6 (theta_tgt, p_tgt); Jr: F x 2 x 6 (theta_ref, p_ref); Jd: F x 2.
 * The J pointers may be NULL. */
int rdvio_hip_reprojection_eval(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, double *r, double *Jt,
                                double *Jr, double *Jd);

/* CeresRotationPriorFactor::Evaluate for every rotation prior of the problem (ceres/rotation_factor.h:22-58;
 * created by refine_subwindow for valid untriangulated tracks, sliding_window_tracker.cpp:389-404).
 * r: n_rot x 2; J: n_rot x 2 x 3 (theta of the target frame; the reference's 2 x 4 block has a zero last column) or NULL. */
int rdvio_hip_rotation_prior_eval(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, double *r, double *J);

/* Solver::solve (solver.cpp:180-194): ceres::Solve with TRUST_REGION/DOGLEG, SPARSE_SCHUR (landmarks
 * eliminated), max_num_iterations = solver.iteration_limit.  The whole trust-region loop runs on the device.
 * states_out (n_frames x 16) and inv_depth_out (n_landmarks) receive the optimised values (the reference
 * updates Frame/Track members in place, solver.cpp:191). */
int rdvio_hip_ba_solve(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, int max_iterations, double *states_out,
                       double *inv_depth_out, rdvio_ba_summary *summary);
/* Split form for HBM-resident operation: upload once, then (re)solve from the uploaded initial values
 * without any host->device traffic; fetch copies the result back. */
#define RDVIO_BA_SLOTS 2 /* e.g. slot 0 = window BA (refine_window), slot 1 = localize_newframe */
int rdvio_hip_ba_upload(rdvio_hip_ctx *ctx, int slot, const rdvio_ba_problem *pb);
/* the result copies of rdvio_hip_ba_fetch without its wait: with several solves in flight, enqueue them all, then fetch -- the first
 * fetch waits once for everything */
int rdvio_hip_ba_fetch_enqueue(rdvio_hip_ctx *ctx, int slot);
/* rdvio_hip_ba_upload for a solve that CONTINUES another one: the initial state of frame `to_frame` of this problem is the result
 * of frame `from_frame` of the solve in `from_slot`, copied on the device in stream order -- the host packs and uploads this
 * problem while the other solve is still running and fetches both afterwards (localize_newframe -> refine_subwindow,
 * sliding_window_tracker.cpp:80-99: the second graph does not depend on the first result, only its starting point does).
 * pb->states' row to_frame is ignored.  Waits for this slot's previous upload only, not for the lane. */
int rdvio_hip_ba_upload_chained(rdvio_hip_ctx *ctx, int slot, const rdvio_ba_problem *pb, int from_slot, int from_frame, int to_frame);
int rdvio_hip_ba_solve_resident(rdvio_hip_ctx *ctx, int slot, int max_iterations);
int rdvio_hip_ba_fetch(rdvio_hip_ctx *ctx, int slot, double *states_out, double *inv_depth_out,
                       rdvio_ba_summary *summary);

/* Unit parity of the estimation kernels' inner pieces (SURVEY.md 8b's `build_normal_schur`): ONE linearisation of the problem at
 * its initial values with the device routines the solver runs -- every factor's Evaluate, J^T J / J^T r assembly, landmark
 * elimination -- and the pieces copied out.  lin_states (n_frames x 16, may be NULL = the initial states) are the states whose
 * biases the preintegration factors are linearised about (the reference reads them live from the frames,
 * ceres/preintegration_factor.h:37-38).  robust_loss != 0 keeps CauchyLoss(1.0) on the reprojection / rotation factors
 * (residual and Jacobian scaled by sqrt(rho')), 0 evaluates them plain.  Any output pointer may be NULL.
 *   r_preint / J_preint  CeresPreIntegrationErrorFactor::Evaluate (ceres/preintegration_factor.h:19-160): whitened residual (15) and
 *                        Jacobians wrt the 15-dim tangents (theta p v bg ba) of frame i and frame j (two row-major 15 x 15 per factor)
 *   r_prior / J_prior    CeresMarginalizationFactor::Evaluate (ceres/marginalization_factor.h:27-72): D = 15 n_prior
 *   H, g                 J^T J and J^T r over the free frames' 15-dim blocks (N = 15 x free frames, frame order), all factors
 *   lm_info, lm_grad     the landmarks' scalar blocks of the same system
 *   S_reduced, c_reduced the landmark-eliminated (Schur) system H - A^T W A, g - A^T W g_l */
typedef struct {
    double *r_preint, *J_preint;
    double *r_prior, *J_prior;
    double *H, *g;
    double *lm_info, *lm_grad;
    double *S_reduced, *c_reduced;
    int32_t N; /* out */
} rdvio_ba_linearization;
int rdvio_hip_ba_linearize(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, const double *lin_states, int robust_loss,
                           rdvio_ba_linearization *out);

/* Large solves run on a team of workgroups whose members must all be resident.  When a member does not answer within the
 * bounded wait (a device shared with other work), the solve is repeated once on one workgroup inside rdvio_hip_ba_fetch instead of
 * failing; this counts those repeats.  A process with more than one live context on a device keeps every solve on one workgroup. */
/* diagnostic: how the last rdvio_hip_detect_keypoints selected on the device -- 0 one pass over every Harris maximum, 1 the maxima of
 * the top response bins sufficed (partial sort), 2 they did not and a second pass took everything; -1 host road / no call yet */
int rdvio_hip_debug_last_select_path(const rdvio_hip_ctx *ctx);
/* diagnostic: the selection kernel's own clock (10 ns units since its start) after the key load, the sort, the cell lists and the
 * greedy pass of the last call, and its number of candidates */
int rdvio_hip_debug_last_select_stamps(const rdvio_hip_ctx *ctx, int32_t *out5);
long rdvio_hip_ctx_team_retries(const rdvio_hip_ctx *ctx);

/* Measurement: live timing of the dominant kernel.  on = k > 0: the launches of one in k PAIRS of consecutive ba_solve_kernel
 * launches (a frame's two solves, picked by a hash of the pair's index; 1: every launch) carry HIP start / stop events on the solver lane, read at the fetch that follows;
 * 0 switches it off.  A timed launch costs the lane a few microseconds (its stop event keeps the next command from following it
 * directly); bench.py times every launch.  get returns the sums over the TIMED launches since
 * timing was switched on:
 * out4 = { launches, kernel milliseconds, algorithmic FP64 flops (SURVEY.md 8d per-unit figures x the units of each launch:
 * (successful steps + 1) linearisations + iterations cost evaluations), solver iterations }. */
int rdvio_hip_ctx_set_kernel_timing(rdvio_hip_ctx *ctx, int on);
int rdvio_hip_ctx_get_kernel_timing(rdvio_hip_ctx *ctx, double *out4);

/* MarginalizationFactor::marginalize(0) (marginalization_factor.h:9-12; ceres/marginalization_factor.h:74-475;
 * called by Map::marginalize_frame, map.cpp:50-62).  Frame 0 of `states` is the victim.  The caller passes the
 * current prior, keyframe_preintegration of frame 1 (NULL if none), and the reprojection factors of every
 * TT_VALID, keyframe-anchored track the victim observes (all of the track's non-anchor observations that are
 * still in the window, :233-380), ordered by landmark. */
typedef struct {
    int32_t n_frames;              /* frames in the map (W+1) */
    const double *states;          /* n_frames x 16 */
    const double *extr;            /* 14 */
    const double *sqrt_inv_cov;    /* 2 x 2 */
    int32_t n_prior;               /* frames covered by the current prior */
    const int32_t *prior_frames;
    const double *prior_lin, *prior_S, *prior_f;
    const double *preint01;        /* RDVIO_PREINT_SIZE or NULL */
    int32_t n_landmarks;
    const double *z_ref, *inv_depth;
    int32_t n_factors;
    const int32_t *tgt, *ref, *lm;
    const double *tangent;
} rdvio_marg_problem;

/* Outputs: the new prior over frames 1..n_frames-1: S_out (R x R, R = 15 (n_frames-1)), f_out (R), lin_out
 * ((n_frames-1) x 16 = the current states of the retained frames).  Lambda_out / eta_out (optional) are the
 * reduced information matrix / vector before the sqrt step.  S_out is A sqrt factor: S^T S and S^T f equal the
 * reference's (eigenvalues <= 1e-8 removed); S itself is only defined up to an orthogonal left factor, exactly
 * like the reference's eigenvector-based factor.  force_eigen != 0 forces the literal eigendecomposition path;
 * *used_fast_path reports whether the Cholesky path produced the factor. */
int rdvio_hip_marginalize(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb, int force_eigen, double *S_out,
                          double *f_out, double *lin_out, double *Lambda_out, double *eta_out, int *used_fast_path);
int rdvio_hip_marginalize_upload(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb);
int rdvio_hip_marginalize_resident(rdvio_hip_ctx *ctx, int force_eigen);
int rdvio_hip_marginalize_fetch(rdvio_hip_ctx *ctx, double *S_out, double *f_out, double *lin_out,
                                double *Lambda_out, double *eta_out, int *used_fast_path);

/* ------------------------------------------------------------------ RD path: PARSAC hypothesis scoring (row A19 / N2) */
/* The inner loop of Parsac<>::solve / IMU_Parsac<>::solve (src/rdvio_util/include/rdvio/util/parsac.h:128-160, 215-262;
 * imu_parsac.h:93-150, 233-280) for a batch of hypotheses: per model the error test of every correspondence, the inlier
 * mask, the number of inliers (and of inliers the IMU prior model shares), the inlier count of every occupied grid bin and
 * the coverage-weighted score -- float arithmetic in the reference's order, so scores compare bit for bit.  Sampling
 * (rand()-ordered) and model generation (5-point essential / EPnP) stay with the caller.
 *   kind 0  model = essential matrix E (9, row-major); pa = p1, pb = p2 (n x 2, normalised image points);
 *           error = d(E, p1, p2) + d(E^T, p2, p1)          (src/rdvio_geometry/src/stereo.cpp:137-141, essential.h:14-19)
 *   kind 1  model = [R | t] (R row-major 9, then t 3); pa = X (n x 3), pb = x (n x 2);
 *           error = |x - proj(R X + t)|^2                  (src/rdvio_geometry/include/rdvio/geometry/pnp.h:89-93, 187-189)
 * The grid of the solve is passed flattened: data_to_valid (point -> occupied bin), valid_sizes, bin_xy (centre of each
 * occupied bin), lens_weight (1 - dynamic_probability^(0.1 mean track length) per occupied bin, NULL for plain PARSAC),
 * prior_mask (inliers of the IMU prior model, NULL for plain PARSAC).  Runs on RDVIO_LANE_SOLVER. */
#define RDVIO_PARSAC_MAX_BINS 400
#define RDVIO_PARSAC_MAX_MODELS 320 /* e.g. 32 five-point samples x 10 essential matrices */
typedef struct {
    int32_t kind, n_points;
    int32_t points_changed;        /* != 0: pa / pb / grid / prior_mask differ from the previous call (re-upload) */
    const double *pa, *pb;
    double threshold;
    int32_t n_valid;
    const int32_t *data_to_valid, *valid_sizes;
    const double *bin_xy;          /* n_valid x 2 */
    const float *lens_weight;      /* n_valid or NULL */
    const uint8_t *prior_mask;     /* n_points or NULL */
    int32_t n_models;
    const double *models;          /* n_models x 9 (kind 0) or x 12 (kind 1) */
} rdvio_parsac_batch;
typedef struct {
    int32_t count;                 /* inliers */
    int32_t effective;             /* inliers shared with the prior model (= count without a prior) */
    float score;
    int32_t pad_;
} rdvio_parsac_result;
int rdvio_hip_parsac_score(rdvio_hip_ctx *ctx, const rdvio_parsac_batch *batch, rdvio_parsac_result *results);
/* The same with the hypotheses GENERATED on the device (row N2): sampling stays with the caller (the reference's samplers draw
 * in a fixed rand() / default_random_engine order), `samples` holds the point indices of n_iterations minimal samples -- six per
 * iteration for kind 1 (solve_pnp_6pt, src/rdvio_geometry/include/rdvio/geometry/pnp.h:11-48: EPnP on float32 copies, float32
 * Rodrigues round trip; one pose per iteration), five for kind 0 (solve_essential_5pt, src/rdvio_geometry/src/essential.cpp:286-298;
 * up to ten essential matrices per iteration).  One launch solves all samples (one wavefront each, csrc/hypo_solvers.hpp: the
 * source the host road runs, so models are bit-identical) and the scoring kernel reads the models where they were written.
 * batch->models / n_models are ignored.  Out: models_per_iteration (n_iterations), then packed in iteration order models
 * (x 12 or x 9 doubles; capacity n_iterations x (1 | 10)) and results; rdvio_hip_parsac_fetch takes packed indices. */
int rdvio_hip_parsac_generate_score(rdvio_hip_ctx *ctx, const rdvio_parsac_batch *batch, int n_iterations, const int32_t *samples,
                                    int32_t *models_per_iteration, double *models, rdvio_parsac_result *results);
/* inlier mask (n_points) and per-occupied-bin inlier counts (n_valid) of model `model` of the last scored batch */
int rdvio_hip_parsac_fetch(rdvio_hip_ctx *ctx, int model, uint8_t *mask, int32_t *bin_inliers);

/* ------------------------------------------------------------------ the tracker's two-view gates */
/* Frame::track_keypoints (src/rdvio_map/src/frame.cpp:108-161) between the LK kernel and the estimator: the 5-point essential RANSAC
 * (stereo.cpp:38-66; kind 0: pa = p1, pb = p2 normalised image points n x 2, threshold on the symmetric epipolar error), the
 * 2-point rotation RANSAC (stereo.cpp:68-91, wahba.h:8-26; kind 2: pa = p1, pb = p2 unit bearings n x 3, threshold = cos of the
 * angular threshold: a point is an inlier when cos(threshold) <= (R p1) . p2 <= 1, i.e. acos(...) <= threshold) and the
 * track-length Poisson-disk thinning.  Like the PARSAC entries: the caller draws the samples (std::default_random_engine order,
 * random.h:79-126) and replays the accept / early-exit decisions of ransac.h:31-76 on the returned inlier counts; one launch solves
 * the minimal problems of n_iterations samples (5 resp. 2 point indices each) and scores every model.  Out, packed in iteration
 * order: models (9 doubles each; capacity n_iterations x (10 | 1)), inlier_counts; rdvio_hip_ransac_fetch copies a model's mask.
 * Frontend lane, staging of its own (a threaded pipeline's estimator may be in rdvio_hip_parsac_* at the same time). */
int rdvio_hip_ransac_generate_score(rdvio_hip_ctx *ctx, int kind, int n_points, int points_changed, const double *pa, const double *pb,
                                    double threshold, int n_iterations, const int32_t *samples, int32_t *models_per_iteration, double *models,
                                    int32_t *inlier_counts);
int rdvio_hip_ransac_fetch(rdvio_hip_ctx *ctx, int model, uint8_t *mask);
/* frame.cpp:134-161: xy (n_points x 2 pixels in the next image), the n_order surviving keypoints in processing order (longest track
 * first -- the caller sorts like the reference's std::sort), trash[i] != 0 for a TT_TRASH track; keep[k] = 1 when order[k] passes
 * PoissonDiskFilter<2>(radius) against the keypoints kept before it and is not trash. */
int rdvio_hip_thin_tracks(rdvio_hip_ctx *ctx, int width, int height, double radius, int n_points, const double *xy, int n_order,
                          const int32_t *order, const uint8_t *trash, uint8_t *keep);

/* ------------------------------------------------------------------ multi-sequence driver */
/* One camera frame of the hot path over inputs that are resident in HBM, as one call: preprocess -> LK -> detect ->
 * preintegration (frame segment without covariance, then nseg - 1 keyframe segments with) on the frontend lane;
 * localize_newframe (BA slot 1) -> refine_window (slot 0) on the solver lane -> marginalisation on its lane -- the per-frame call
 * order of FeatureTracker::run (feature_tracker.cpp:26-111) and SlidingWindowTracker::track (sliding_window_tracker.cpp:80-99).
 * k = frame index (image slot parity, image k % n_images).  overlap != 0: the estimation of frame k is enqueued first and the
 * frontend of the next image runs beside it (handler.cpp:35-50), the host waits on the frontend and solver lanes; 0: back to
 * back, one wait.  The BA / marginalisation problems are the ones last uploaded (rdvio_hip_ba_upload, _marginalize_upload). */
typedef struct rdvio_frame_step {
    rdvio_hip_ctx *ctx;
    int32_t width, height, stride, n_images;
    const uint8_t *const *images_dev;       /* n_images device images (u8, width x height) */
    int32_t n_features, keypoints_capacity;
    const double *curr_xy_dev;              /* n_features x 2 */
    double *next_xy_dev;
    uint8_t *status_dev;
    double *keypoints_host;                 /* detect output, keypoints_capacity x 2 */
    double min_distance;
    int32_t nseg, ba_iterations;
    const int32_t *seg_off_dev;             /* nseg + 1 */
    const double *imu_dev, *par_dev, *noise_dev;
    double *preint_out_dev;                 /* nseg x RDVIO_PREINT_SIZE */
    int32_t overlap, reserved;
} rdvio_frame_step;
int rdvio_hip_frame_step(const rdvio_frame_step *step, int k);
/* n_seq independent sequences (one context each; the reference is one process per sequence), one host thread per
 * sequence: `warmup` untimed frames each, a common start, then `steps` frames each.  elapsed_s = common start -> last frame of the
 * slowest sequence; per_sequence_s (n_seq, may be NULL) = each sequence's own time.  wait_mode: how the threads wait for the
 * device during the run (rdvio_hip_ctx_set_wait_mode): 0 spin, 1 block, -1 = block when 2 n_seq exceeds the host's cores; the
 * contexts are back in spin mode afterwards.  Returns the first error met (0 = none). */
int rdvio_hip_run_sequences(const rdvio_frame_step *seqs, int n_seq, int warmup, int steps, int wait_mode, double *elapsed_s,
                            double *per_sequence_s);

#ifdef __cplusplus
}
#endif
#endif
