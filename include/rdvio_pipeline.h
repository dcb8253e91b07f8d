/*
 * rdvio_pipeline.h -- C ABI of the per-frame host orchestration around the HIP hot path: the push-image / push-IMU /
 * get-pose surface of rdvio::Odometry on top of librdvio_hip.so.
 *
 * Reference interfaces replaced (file:line under /root/reference):
 *   rdvio::Odometry                     src/rdvio/include/rdvio/rdvio.hpp:25-115   (addFrame, addMotion, addGyro, addAcc,
 *                                                                                   transform_world_cam, state)
 *   Handler (sensor interleave)         src/rdvio/src/handler.cpp:15-227
 *   FeatureTracker::run                 src/rdvio/src/feature_tracker.cpp:26-111   (row A6)
 *   Frame::track_keypoints / detect     src/rdvio_map/src/frame.cpp:55-172         (rows A4, A5)
 *   Frontend::run                       src/rdvio/src/frontend.cpp:27-70
 *   SlidingWindowTracker                src/rdvio/src/sliding_window_tracker.cpp:16-456 (row A16)
 *   Track::triangulate                  src/rdvio_map/src/track.cpp:46-76          (row A18)
 *   Map::marginalize_frame              src/rdvio_map/src/map.cpp:50-62
 *
 * The orchestration is host C++ (like the reference's); everything data-parallel goes through the `rdvio_backend`
 * function table, whose product implementation is the HIP library (rdvio_pipeline_create_hip).  The table exists so
 * that the test-suite can run the SAME orchestration over the CPU oracle and compare trajectories and feature index
 * sets (SURVEY.md 8d metrics 2 and 3); the product never constructs a non-HIP backend.
 *
 * Initialisation: Initializer::initialize (src/rdvio/src/initializer.cpp:72-560: two-view SfM + IMU alignment + BA) runs
 * unless bootstrap states were supplied with rdvio_pipeline_set_init_states, which then replace its SfM / alignment
 * stages.  parsac_flag enables the RD dynamic-outlier path (IMU-PARSAC over EPnP hypotheses, PARSAC essential checks).
 */
#ifndef RDVIO_PIPELINE_H
#define RDVIO_PIPELINE_H

#include <stdint.h>

#include "rdvio_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The two plugin seams of the reference (rdvio::Image, rdvio::Solver / PreIntegrator / MarginalizationFactor) as a
 * function table.  All functions return RDVIO_OK (0) or an RDVIO_ERR_* code. */
typedef struct rdvio_backend {
    void *user;
    /* rdvio::Image (types.h:153-177).  An image handle owns a private copy of the pixels (OpenCvImage::image). */
    int (*image_create)(void *user, const uint8_t *gray, int width, int height, int stride, void **image_out);
    int (*image_preprocess)(void *user, void *image, double clahe_clip, int tiles_x, int tiles_y);
    int (*image_detect)(void *user, void *image, double *keypoints, int n_existing, int capacity, int max_points,
                        double min_distance, int *n_out);
    int (*image_track)(void *user, void *image_curr, void *image_next, int n, const double *curr_xy, double *next_xy,
                       int has_guess, uint8_t *status);
    void (*image_release)(void *user, void *image); /* release_image_buffer */
    void (*image_destroy)(void *user, void *image);
    /* PreIntegrator::integrate for nseg independent segments (refine_window integrates every keyframe interval before
     * one solve): seg_off[nseg+1] sample offsets into imu (n x 7: t, gyro, acc), t_end[nseg], bg / ba [3 nseg],
     * noise 4 x 9 (cov_w cov_a cov_bg cov_ba), preint_out nseg x RDVIO_PREINT_SIZE */
    int (*preintegrate)(void *user, int nseg, const int32_t *seg_off, const double *imu, const double *t_end, const double *bg,
                        const double *ba, const double *noise, int compute_jacobian, int compute_covariance, double *preint_out);
    /* Solver::solve on the SoA problem; states_out / inv_depth_out receive the optimised values */
    int (*ba_solve)(void *user, const rdvio_ba_problem *pb, int max_iterations, double *states_out, double *inv_depth_out,
                    rdvio_ba_summary *summary);
    /* MarginalizationFactor::marginalize(0) */
    int (*marginalize)(void *user, const rdvio_marg_problem *pb, double *S_out, double *f_out, double *lin_out);
    const char *(*last_error)(void *user);
    void (*destroy)(void *user); /* optional: called by rdvio_pipeline_destroy */
    /* optional (NULL = the orchestration scores on the host, parsac.hpp): PARSAC / IMU-PARSAC hypothesis scoring of a
     * batch of models and the mask / bin counts of one of them (rdvio_hip_parsac_score / _fetch) */
    int (*parsac_score)(void *user, const rdvio_parsac_batch *batch, rdvio_parsac_result *results);
    int (*parsac_fetch)(void *user, int model, uint8_t *mask, int32_t *bin_inliers);
    /* optional, for threaded pipelines (rdvio_pipeline_config::threading == 2).  The feature tracker calls the image
     * functions and `preintegrate` on the caller's thread; the estimator (Frontend: sliding-window tracker) calls
     * ba_solve / marginalize / parsac_* and -- when set -- `preintegrate_estimator` on the worker thread, concurrently.
     * preintegrate_estimator: the same contract as preintegrate on staging buffers of its own (NULL = preintegrate is
     * safe to call from both threads at once); thread_attach: called once on the worker thread before its first backend
     * call (a HIP backend binds the thread to its device). */
    int (*preintegrate_estimator)(void *user, int nseg, const int32_t *seg_off, const double *imu, const double *t_end, const double *bg,
                                  const double *ba, const double *noise, int compute_jacobian, int compute_covariance, double *preint_out);
    int (*thread_attach)(void *user);
    /* optional pair (NULL = marginalize): the same call in two halves -- begin enqueues, end waits and copies out.  The new prior
     * is first read by the NEXT refine_window (sliding_window_tracker.cpp:339-347, 226-300), several frames later: the
     * orchestration calls end only then, so the marginalisation runs beside the frames in between. */
    int (*marginalize_begin)(void *user, const rdvio_marg_problem *pb);
    int (*marginalize_end)(void *user, double *S_out, double *f_out, double *lin_out);
    /* optional (NULL = the orchestration's host code): the tracker's two-view gates and track-length thinning behind the backend
     * (rdvio_hip_ransac_generate_score / _ransac_fetch / rdvio_hip_thin_tracks; Frame::track_keypoints, frame.cpp:108-161) */
    int (*ransac_generate_score)(void *user, int kind, int n_points, int points_changed, const double *pa, const double *pb, double threshold,
                                 int n_iterations, const int32_t *samples, int32_t *models_per_iteration, double *models, int32_t *inlier_counts);
    int (*ransac_fetch)(void *user, int model, uint8_t *mask);
    int (*thin_tracks)(void *user, int width, int height, double radius, int n_points, const double *xy, int n_order, const int32_t *order,
                       const uint8_t *trash, uint8_t *keep);
    /* optional (NULL = the orchestration's host solvers generate the hypotheses): rdvio_hip_parsac_generate_score */
    int (*parsac_generate_score)(void *user, const rdvio_parsac_batch *batch, int n_iterations, const int32_t *samples,
                                 int32_t *models_per_iteration, double *models, rdvio_parsac_result *results);
    /* optional pair (NULL = preintegrate_estimator / preintegrate): the estimator's integration of the new frame in two halves --
     * begin enqueues, end waits and copies the records out; the sliding-window map's share of mirror_frame runs in between */
    int (*preintegrate_estimator_begin)(void *user, int nseg, const int32_t *seg_off, const double *imu, const double *t_end, const double *bg,
                                        const double *ba, const double *noise, int compute_jacobian, int compute_covariance);
    int (*preintegrate_estimator_end)(void *user, double *preint_out);
    /* optional pair (NULL = ba_solve): Solver::solve in two halves on one of two slots (0 / 1).  begin packs, uploads and enqueues;
     * with chain_from_slot >= 0 the initial state of frame chain_to_frame is the RESULT of frame chain_from_frame of the solve begun
     * in that slot (copied behind the backend: the orchestration builds the second graph while the first solve runs --
     * localize_newframe -> refine_subwindow); end waits and copies out.  Ends may come in any order after the begins. */
    int (*ba_solve_begin)(void *user, int slot, const rdvio_ba_problem *pb, int max_iterations, int chain_from_slot, int chain_from_frame,
                          int chain_to_frame);
    int (*ba_solve_end)(void *user, int slot, double *states_out, double *inv_depth_out, rdvio_ba_summary *summary);
} rdvio_backend;

/* rdvio::Config (types.h:85-151) with the defaults of src/rdvio/src/config.cpp; rdvio_pipeline_config_default fills
 * them.  Matrices row-major, quaternions (x, y, z, w). */
typedef struct rdvio_pipeline_config {
    int32_t width, height;
    double K[9];
    double q_bc[4], p_bc[3]; /* camera_to_body_rotation / translation */
    double q_bi[4], p_bi[3]; /* imu_to_body_rotation / translation */
    double q_bo[4], p_bo[3]; /* output_to_body */
    double keypoint_noise_cov[4];
    double gyroscope_noise_cov[9], accelerometer_noise_cov[9], gyroscope_bias_noise_cov[9], accelerometer_bias_noise_cov[9];
    int32_t sliding_window_size, sliding_window_subframe_size, sliding_window_force_keyframe_landmarks,
        sliding_window_tracker_frequent;
    double feature_tracker_min_keypoint_distance;
    int32_t feature_tracker_max_keypoint_detection, feature_tracker_max_init_frames, feature_tracker_max_frames;
    double feature_tracker_clahe_clip_limit;
    int32_t feature_tracker_clahe_width, feature_tracker_clahe_height, feature_tracker_predict_keypoints;
    int32_t initializer_keyframe_num, initializer_keyframe_gap;
    int32_t initializer_min_matches, initializer_min_triangulation, initializer_min_landmarks;
    double initializer_min_parallax;
    int32_t solver_iteration_limit;
    double rotation_misalignment_threshold, rotation_ransac_threshold;
    int32_t random;
    int32_t parsac_flag; /* RD dynamic-outlier handling (judge_track_status / update_track_status) */
    int32_t parsac_keyframe_check_size;
    /* The reference's tracker / frontend split (handler.cpp:35-50, CMake option THREADING):
     *   0  inline, the reference's THREADING=OFF: the frontend's step for frame k runs inside the tracker's step for
     *      frame k, which therefore sees the states optimised with frame k - 1;
     *   1  the pipelined schedule, executed on one thread: the frontend's step for frame k is issued when the tracker has
     *      finished frame k and its results (latest optimised state, track tags) become visible to the tracker when it has
     *      finished frame k + 1 -- one admissible interleaving of the reference's THREADING=ON, made deterministic;
     *   2  the same schedule with the frontend's step on a worker thread, concurrent with the tracker's next frame.
     * 1 and 2 produce identical results by construction (the two steps share no mutable state between hand-overs); the
     * CPU path of the comparison runs 1, the product 2. */
    int32_t threading;
    /* initializer.refine_imu (initializer.cpp:373): 0 skips refine_scale_velocity_via_gravity */
    int32_t initializer_refine_imu;
    /* Where the feature tracker's two-view RANSAC gates and its track-length thinning run (frame.cpp:108-161):
     *   0  on the host (geom.hpp) -- the default: at the reference's sizes (<= 150 points, <= 8 five-point hypotheses per
     *      batch) one host core is faster than a launch + round trip per gate;
     *   1  behind the backend's ransac_generate_score / ransac_fetch / thin_tracks hooks when it offers them (the HIP
     *      product: hypothesis generation and scoring on the frontend lane).
     * Both roads run the same solvers (csrc/hypo_solvers.hpp) and give bit-identical gates. */
    int32_t tracker_gates_on_backend;
} rdvio_pipeline_config;

void rdvio_pipeline_config_default(rdvio_pipeline_config *cfg);

typedef struct rdvio_pipeline rdvio_pipeline;

/* Product entry point: the pipeline over the HIP context (the context must outlive the pipeline). */
int rdvio_pipeline_create_hip(rdvio_pipeline **out, const rdvio_pipeline_config *cfg, rdvio_hip_ctx *ctx);
/* Generic entry point (function table copied). */
int rdvio_pipeline_create(rdvio_pipeline **out, const rdvio_pipeline_config *cfg, const rdvio_backend *backend);
void rdvio_pipeline_destroy(rdvio_pipeline *p);
const char *rdvio_pipeline_last_error(const rdvio_pipeline *p);

/* OPTIONAL bootstrap states of the first keyframes: n rows of (t, q(4), p(3), v(3), bg(3), ba(3)) = 17 doubles, body
 * frame in the world frame with gravity along -z.  A keyframe whose timestamp matches a row within 1e-6 s takes that
 * state and the SfM / IMU-alignment stages of the initializer are skipped; n = 0 restores the full initializer. */
int rdvio_pipeline_set_init_states(rdvio_pipeline *p, int n, const double *rows17);

/* Odometry::addFrame (rdvio.hpp:41-56): gray u8 image; pose_out (may be NULL) = predicted output pose q(4) p(3), all
 * zeros before the first optimised state (handler.cpp:178-181). */
int rdvio_pipeline_add_frame(rdvio_pipeline *p, double t, const uint8_t *gray, int width, int height, int stride,
                             double *pose_out);
/* Odometry::addMotion / addGyro / addAcc (rdvio.hpp:58-69): gyro is pushed first */
int rdvio_pipeline_add_motion(rdvio_pipeline *p, double t, const double *acc, const double *gyro);
int rdvio_pipeline_add_gyro(rdvio_pipeline *p, double t, const double *gyro);
int rdvio_pipeline_add_acc(rdvio_pipeline *p, double t, const double *acc);
/* Odometry::state: 0 initialising, 1 tracking, 2 crash, 3 unknown */
int rdvio_pipeline_state(const rdvio_pipeline *p);
/* Handler::get_latest_state (handler.cpp:190-206): time + body pose of the newest tracked frame; returns 0 if none */
int rdvio_pipeline_latest_state(const rdvio_pipeline *p, double *t, double *pose7);
/* full state (q p v bg ba) of the newest frame of the sliding window (SlidingWindowTracker::get_latest_state) */
int rdvio_pipeline_window_state(const rdvio_pipeline *p, double *t, double *state16);
/* Odometry::transform_world_cam (rdvio.hpp:71-77): 4x4 row-major */
int rdvio_pipeline_transform_world_cam(const rdvio_pipeline *p, double *T16);
/* Odometry::local_map: valid triangulated landmarks (world frame, without the reference's axis swap); returns count */
int rdvio_pipeline_local_map(const rdvio_pipeline *p, double *xyz, int capacity);

/* Diagnostics for the index-parity metric: keypoints of the newest frame of the feature-tracking map.
 * track_ids[i] = id of the track through keypoint i or -1; xy in pixels.  Returns the number of keypoints. */
int rdvio_pipeline_last_frame_keypoints(const rdvio_pipeline *p, int64_t *track_ids, double *xy, int capacity);
/* counters: [0] frames tracked, [1] window solves, [2] keyframes inserted, [3] marginalisations, [4] localisations,
 * [5] subwindow solves, [6] frame id of the newest tracked frame, [7] tracks in the window map, [8] largest number of
 * frames in one solve, [9] largest number of reprojection factors in one solve, [10] solver iterations summed over all
 * solves, then (microseconds, calls) pairs of the time spent inside backend calls: [11,12] preprocess, [13,14] detect,
 * [15,16] track, [17,18] preintegrate, [19,20] ba_solve, [21,22] marginalize, [23,24] image_create; [25] frames tagged
 * FT_NO_TRANSLATION, [26] rotation-prior factors summed over all solves, [27] IMU-PARSAC judgements run, [28] tracks
 * switched to non-static by update_track_status */
int rdvio_pipeline_counters(const rdvio_pipeline *p, int64_t *out29);

/* The reference's test_euroc loop (examples/test_euroc.cpp:46-95: IMU samples and camera frames pushed in timestamp order)
 * over a stream that is resident in host memory, as one call -- what a benchmark or an offline run drives instead of paying an
 * FFI crossing per sample.  Every IMU sample with t <= frame_t[k] is pushed (add_motion) before frame k (add_frame), the
 * remaining samples after the last frame.  Outputs (each may be NULL) get one row per camera frame the feature tracker has
 * consumed, in processing order: the keypoint table of the newest tracked frame (rdvio_pipeline_last_frame_keypoints), the
 * output pose (rdvio_pipeline_latest_state), the window state (rdvio_pipeline_window_state; NaN rows before the first),
 * rdvio_pipeline_state, and the time since the start of the call at which the frame was done. */
typedef struct rdvio_replay {
    int32_t n_frames, width, height, stride;
    const uint8_t *const *frames; /* n_frames gray images */
    const double *frame_t;        /* n_frames */
    int32_t n_imu;
    const double *imu;            /* n_imu x 7: t, gyro (3), acc (3) */
    int32_t kp_capacity;          /* rows of kp_ids / kp_xy per frame */
    int64_t *kp_ids;              /* n_frames x kp_capacity */
    double *kp_xy;                /* n_frames x kp_capacity x 2 */
    int32_t *kp_n;                /* n_frames */
    double *latest;               /* n_frames x 8: t, q(4), p(3) */
    double *window;               /* n_frames x 17: t, state16 */
    int32_t *sys_state;           /* n_frames */
    double *done_s;               /* n_frames */
    int32_t frames_processed;     /* out */
    double elapsed_s;             /* out: whole call */
    /* A stream can be replayed in segments (warm-up, then a timed region): after the last frame of the segment IMU samples are
     * pushed until the feature tracker has consumed every frame pushed so far (flush == 1: only until then; flush == 2: all of
     * them, the end of the stream), and imu_consumed says where the next segment's IMU starts.  flush == 0: none. */
    int32_t flush;
    int32_t imu_consumed;         /* out */
} rdvio_replay;
int rdvio_pipeline_replay(rdvio_pipeline *p, rdvio_replay *r);
/* waits until the frontend's step in flight (threading == 2) has finished; every frame pushed so far is then fully processed */
int rdvio_pipeline_drain(rdvio_pipeline *p);

#ifdef __cplusplus
}
#endif
#endif
