// Internal context of librdvio_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/rdvio_hip.h"
#include "solver_ws.hpp"

#define RDVIO_NUM_SLOTS 2
#define RDVIO_PARSAC_MASKS_INLINE (256 * 1024)  // masks + bin counts of a whole batch travel with its results up to this size
#define RDVIO_MAX_TILES 256  // CLAHE tile grid (8x8 in configs/setting.yaml:17-19)

struct ImageSlot {
    uint8_t *pyr_img = nullptr;    // padded u8 arena (all levels)
    int16_t *pyr_deriv = nullptr;  // padded int16x2 arena (all levels)
    rdvio_pyr_layout L{};
    int w = 0, h = 0;
    bool valid = false;
};

struct HarrisCand {
    float v;
    int32_t idx;
};

struct rdvio_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // lane 0 (frontend): Image::*, PreIntegrator::integrate, unit-parity entry points
    bool own_stream = false;
    // lanes: [0] frontend (= stream), [1] solver (Solver::solve), [2] marginalisation.  All three are `stream` until
    // rdvio_hip_ctx_set_lane_stream gives a lane a stream of its own; each lane has its own staging buffers, so calls on
    // different lanes never share mutable state.
    hipStream_t lane[3] = {nullptr, nullptr, nullptr};
    bool own_lane[3] = {false, false, false};
    hipEvent_t lane_ev[3] = {nullptr, nullptr, nullptr};
    // host waits: spinning (hipStreamSynchronize: lowest latency, one core per waiting thread) or blocking on an event created
    // with hipEventBlockingSync (the waiting thread sleeps: many sequences per process, more waiting threads than cores)
    bool blocking_wait = false;
    hipEvent_t wait_ev[3] = {nullptr, nullptr, nullptr};
    int max_w = 0, max_h = 0, max_feat = 0, max_window = 0, max_factors = 0;
    bool force_host_select = false;  // env RDVIO_HOST_SELECT=1 (read at context creation)
    int helper_min_factors = 4096;  // problems with at least this many factors launch the team (env RDVIO_HELPER_MIN_FACTORS)
    int solver_wgs = 8;  // workgroups per solver launch for problems with >= RDVIO_HELPER_MIN_FACTORS factors (env RDVIO_SOLVER_WGS)
    rdvio_pyr_layout maxL{};

    ImageSlot slots[RDVIO_NUM_SLOTS];
    uint8_t *gray = nullptr;       // device staging of a slot-less upload (max_w*max_h)
    // host images travel through pinned memory, one staging pair per image slot: the copy out of the caller's buffer is
    // synchronous (the caller may free it when the call returns), the copy to the device asynchronous (gray_ev[s] says when the
    // pinned buffer may be overwritten)
    uint8_t *gray_pinned[RDVIO_NUM_SLOTS] = {nullptr, nullptr};
    uint8_t *gray_slot[RDVIO_NUM_SLOTS] = {nullptr, nullptr};
    hipEvent_t gray_ev[RDVIO_NUM_SLOTS] = {nullptr, nullptr};
    int gray_w[RDVIO_NUM_SLOTS] = {0, 0}, gray_h[RDVIO_NUM_SLOTS] = {0, 0};
    uint8_t *clahe_lut = nullptr;  // RDVIO_MAX_TILES x 256
    float *harris = nullptr;       // max_w*max_h
    uint32_t *harris_scalars = nullptr;  // [0] ordered-uint max, [1] candidate count
    HarrisCand *harris_cand = nullptr;
    int harris_cand_cap = 0;
    // device-side selection (select_kernels.hip): header + accepted new keypoints in one buffer (one D2H), corner list,
    // uploaded existing keypoints
    int32_t *sel_hdr = nullptr;      // [0] candidates, [1] capacity flags (host road when != 0), [2] corners, [3] new keypoints
    double *sel_new = nullptr;       // = (double *)(sel_hdr + 16)
    int last_select_path = -1;       // rdvio_hip_debug_last_select_path
    int32_t last_select_stamps[5] = {0, 0, 0, 0, 0};   // gftt_select_kernel: 10-ns stamps after load / sort / cell lists / greedy pass, candidates
    float *sel_corners = nullptr;
    double *sel_existing = nullptr;

    // LK device buffers
    double *lk_curr = nullptr, *lk_next = nullptr;
    float *lk_prevf = nullptr, *lk_nextf = nullptr;
    uint8_t *lk_status = nullptr;

    // estimation device buffers
    double *ba_states = nullptr, *ba_extr = nullptr, *ba_zref = nullptr, *ba_invd = nullptr, *ba_tangent = nullptr;
    int32_t *ba_idx = nullptr;  // tgt | ref | lm, each max_factors
    double *ba_r = nullptr, *ba_Jt = nullptr, *ba_Jr = nullptr, *ba_Jd = nullptr;
    double *pre_out = nullptr, *pre_blob = nullptr;  // results; one staging blob (par | noise | samples | offsets)
    int pre_max_samples = 0, pre_max_seg = 0;
    // the estimator's preintegrations (rdvio_hip_preintegrate_estimator, solver lane): staging of their own
    double *pre2_out = nullptr, *pre2_blob = nullptr;
    void *pre2_pinned = nullptr;
    size_t pre2_pinned_bytes = 0;
    size_t pre2_pending[2] = {0, 0};   // rdvio_hip_preintegrate_estimator_begin / _end: where the records in flight will land

    // BA solver: pinned input blob, device arena (inputs + scratch), workspace descriptor
    struct BaSlot {
        void *host = nullptr, *arena = nullptr;
        SolverWs ws{};
        size_t in_states_off = 0, in_invd_off = 0, in_bytes = 0, host_bytes = 0;
        bool ready = false;
        // fused preintegration jobs of the uploaded problem (rdvio_ba_problem::n_pre_jobs): device views of the raw samples and
        // where the records go back to
        bool retried = false;   // a team solve whose helpers stayed silent was repeated on the leader alone
        bool down_enqueued = false;   // rdvio_hip_ba_fetch_enqueue has put this launch's result copies on the lane
        hipEvent_t up_ev = nullptr;   // behind this slot's last upload (rdvio_hip_ba_upload_chained waits for it, not for the lane)
        size_t user0_off = 0;
        int n_jobs = 0;
        const int32_t *job_off = nullptr;
        const double *job_imu = nullptr, *job_par = nullptr, *job_noise = nullptr;
        double *job_out_host = nullptr;
        // SURVEY 8(d) algorithmic flops of the problem: per linearisation / per trial-step cost evaluation (rdvio_ba_prepare)
        double flops_lin = 0.0, flops_eval = 0.0;
        // live kernel timing (rdvio_hip_ctx_set_kernel_timing): events around the launch in flight
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        bool timed_launch = false;
    } ba[RDVIO_BA_SLOTS];
    size_t ba_host_bytes = 0, ba_arena_bytes = 0;
    // rdvio_hip_ba_upload_chained: the continuing solve's inputs (and its preintegration jobs) travel on a stream of their own,
    // beside the solve they continue; the solver lane waits for chain_ev before the continuing kernel
    hipStream_t chain_stream = nullptr;
    hipEvent_t chain_ev = nullptr;
    // marginalisation: a solver slot of its own (the linearisation is shared with the solver) + tail scratch
    BaSlot marg;
    size_t marg_bytes = 0;

    // Hypothesis generation / scoring state (parsac_kernels.hip): pinned blob (inputs | results), device mirror, outputs.  Two
    // instances, one per caller: [0] the estimator's PARSAC solves on the solver lane, [1] the tracker's two-view RANSAC gates
    // on the frontend lane -- they run concurrently in a threaded pipeline and share nothing.
    struct PsState {
        int lane = RDVIO_LANE_SOLVER;
        void *host = nullptr, *dev = nullptr;
        void *down_dev = nullptr;   // device image of the blob's result part: everything a batch brings back is contiguous -- ONE copy
        size_t in_bytes = 0, down_bytes = 0;
        int max_points = 0;
        uint8_t *masks = nullptr;           // RDVIO_PARSAC_MAX_MODELS x max_points
        int32_t *bins = nullptr;            // RDVIO_PARSAC_MAX_MODELS x n_valid
        rdvio_parsac_result *results = nullptr;
        int n = 0, kind = -1, nv = 0, nm = 0;
        bool has_prior = false, has_lens = false;
        int slot_of[RDVIO_PARSAC_MAX_MODELS] = {0};   // model index of the last batch (as the caller counts) -> device slot
        // masks and bin counts of the last batch on the host already (they rode along with the results): offsets into host, 0 = no
        size_t masks_host = 0, bins_host = 0;
    } ps[2];
    // track-length thinning (rdvio_hip_thin_tracks; frontend lane)
    void *thin_host = nullptr, *thin_dev = nullptr;
    size_t thin_bytes = 0;

    // pinned host staging
    void *pinned = nullptr;
    size_t pinned_bytes = 0;

    // live kernel timing of the dominant kernel (ba_solve_kernel): HIP events on the solver lane around every launch,
    // read at the fetch that follows; sums since the last reset
    bool counted = false;    // this context is in the device's live-context count
    long team_retries = 0;   // solves repeated on one workgroup after a helper time-out (rdvio_hip_ctx_team_retries)
    int kernel_timing = 0;       // 0 off; k > 0: every k-th launch is bracketed by events (rdvio_hip_ctx_set_kernel_timing)
    long kt_seen = 0;            // launches since timing was switched on
    double kt_launches = 0.0, kt_ms = 0.0, kt_flops = 0.0, kt_iterations = 0.0;

    char err[512] = {0};
};

inline int rdvio_fail(rdvio_hip_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

// every host-side wait of the library goes through here (rdvio_hip_ctx_set_wait_mode).  Blocking waits use one event per
// lane (two host threads may wait on two lanes at once: tracker and estimator of a threaded pipeline).
static inline hipError_t rdvio_wait(rdvio_hip_ctx *ctx, hipStream_t st) {
    if (!ctx->blocking_wait || !ctx->wait_ev[0]) return hipStreamSynchronize(st);
    int l = 0;
    for (int k = 1; k < 3; ++k)
        if (st == ctx->lane[k] && st != ctx->lane[0]) l = k;
    const hipError_t e = hipEventRecord(ctx->wait_ev[l], st);
    return e != hipSuccess ? e : hipEventSynchronize(ctx->wait_ev[l]);
}

#define RDVIO_HIP_CHECK(ctx, expr)                                                                     \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return rdvio_fail(ctx, RDVIO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                              __FILE__, __LINE__);                                                     \
    } while (0)

// contexts alive on a device (capi.hip): more than one means the device is shared between sequences
int rdvio_live_contexts(int device);

// solver_host.hip: validate + index + pack a BA problem into `slot` (capacity `cap` bytes of pinned blob and arena);
// with_marg_tail also carves the marginalisation tail's scratch and outputs
int rdvio_ba_prepare(rdvio_hip_ctx *ctx, rdvio_hip_ctx::BaSlot &slot, const rdvio_ba_problem *pb, size_t cap, bool with_marg_tail);

// kernel launchers (defined in the .hip files)
int rdvio_launch_preprocess(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray_dev, int w, int h, int stride,
                            double clip, int tiles_x, int tiles_y);
int rdvio_launch_track(rdvio_hip_ctx *ctx, int slot_curr, int slot_next, int n, const double *curr_dev,
                       double *next_dev, int has_guess, uint8_t *status_dev);
int rdvio_launch_lk_flow(rdvio_hip_ctx *ctx, int slot_prev, int slot_next, int n, const float *prev_dev,
                         float *next_dev, uint8_t *status_dev, int max_iter, double eps);
int rdvio_launch_harris(rdvio_hip_ctx *ctx, int slot);
int rdvio_launch_harris_candidates(rdvio_hip_ctx *ctx, int slot, double quality);
int rdvio_launch_select(rdvio_hip_ctx *ctx, int slot, int max_corners, double gftt_min_dist, double poisson_radius, int n_existing);
int rdvio_launch_reprojection(rdvio_hip_ctx *ctx, int nf, int with_jac);
int rdvio_launch_rotation_prior(rdvio_hip_ctx *ctx, int n, int with_jac);
int rdvio_launch_preintegrate(rdvio_hip_ctx *ctx, hipStream_t st, int nseg, const int32_t *off, const double *imu, const double *par,
                              const double *noise, int cj, int cc, double *out);
