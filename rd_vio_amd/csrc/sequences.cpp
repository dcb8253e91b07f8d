// Multi-sequence driver: one camera frame of the resident hot path as ONE call (rdvio_hip_frame_step), and n independent
// sequences -- one context and one host thread each -- driven through K frames (rdvio_hip_run_sequences).
//
// Why it exists: a VIO sequence is sequential in itself and occupies one compute unit for most of a frame (the persistent
// single-workgroup solver), so one sequence cannot fill 256 CUs; independent sequences can share the device.  The reference
// cannot hold two sequences in a process (process-global state, SURVEY F9: one process per sequence,
// src/rdvio_util/include/rdvio/util/identifiable.h:22-29); here all state is per context, so sequences are threads.
// The frame step is the per-frame call order of the reference's two workers (feature tracker: feature_tracker.cpp:26-111;
// frontend: sliding_window_tracker.cpp:80-99) over inputs that are already resident in HBM.
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/rdvio_hip.h"

extern "C" int rdvio_hip_frame_step(const rdvio_frame_step *d, int k) {
    if (!d || !d->ctx || d->n_images <= 0 || k < 0) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx *c = d->ctx;
    int rc;
    // frame k's estimation: localize_newframe (slot 1), refine_window (slot 0; reads the prior the previous marginalisation
    // wrote), slide_window -> marginalize (reads the window solve's states)
    auto estimator = [&]() -> int {
        if ((rc = rdvio_hip_ba_solve_resident(c, 1, d->ba_iterations))) return rc;
        if ((rc = rdvio_hip_lane_wait(c, RDVIO_LANE_SOLVER, RDVIO_LANE_MARG))) return rc;
        if ((rc = rdvio_hip_ba_solve_resident(c, 0, d->ba_iterations))) return rc;
        if ((rc = rdvio_hip_lane_wait(c, RDVIO_LANE_MARG, RDVIO_LANE_SOLVER))) return rc;
        return rdvio_hip_marginalize_resident(c, 0);
    };
    // the image side and the preintegration (frame segment without covariance, keyframe segments with: the reference's two
    // call sites, feature_tracker.cpp:82-84 and sliding_window_tracker.cpp:294)
    auto frontend = [&]() -> int {
        const int cur = k % 2, prv = (k + 1) % 2;
        int n_out = 0;
        if ((rc = rdvio_hip_image_preprocess_dev(c, cur, d->images_dev[k % d->n_images], d->width, d->height, d->stride, 6.0, 8, 8))) return rc;
        if ((rc = rdvio_hip_track_keypoints_dev(c, prv, cur, d->n_features, d->curr_xy_dev, d->next_xy_dev, 0, d->status_dev))) return rc;
        if ((rc = rdvio_hip_detect_keypoints(c, cur, d->keypoints_host, 0, d->keypoints_capacity, d->n_features, d->min_distance, &n_out))) return rc;
        if ((rc = rdvio_hip_preintegrate_dev(c, 1, d->seg_off_dev, d->imu_dev, d->par_dev, d->noise_dev, 0, 0, d->preint_out_dev))) return rc;
        return rdvio_hip_preintegrate_dev(c, d->nseg - 1, d->seg_off_dev + 1, d->imu_dev, d->par_dev + 7, d->noise_dev, 1, 1,
                                          d->preint_out_dev + RDVIO_PREINT_SIZE);
    };
    if (d->overlap) {
        // the solver / marginalisation lanes work on frame k's estimation while the frontend lane runs the next image
        // (handler.cpp:35-50); the host waits for what host logic consumes: keypoints and states
        if ((rc = estimator())) return rc;
        if ((rc = frontend())) return rc;
        if ((rc = rdvio_hip_lane_sync(c, RDVIO_LANE_FRONTEND))) return rc;
        return rdvio_hip_lane_sync(c, RDVIO_LANE_SOLVER);
    }
    if ((rc = frontend())) return rc;
    if ((rc = estimator())) return rc;
    return rdvio_hip_sync(c);
}

extern "C" int rdvio_hip_run_sequences(const rdvio_frame_step *seqs, int n_seq, int warmup, int steps, int wait_mode,
                                       double *elapsed_s, double *per_sequence_s) {
    if (!seqs || n_seq <= 0 || n_seq > 1024 || warmup < 0 || steps <= 0) return RDVIO_ERR_INVALID;
    // wait mode for the run: 0 spin, 1 block, -1 (default) block when there are more sequences than half the cores (a spinning
    // waiter holds a core, and the runtime's own threads need some too)
    const unsigned cores = std::thread::hardware_concurrency();
    const bool blocking = wait_mode < 0 ? (cores == 0 || 2u * (unsigned)n_seq > cores) : wait_mode != 0;
    for (int i = 0; i < n_seq; ++i)
        if (int rc = rdvio_hip_ctx_set_wait_mode(seqs[i].ctx, blocking ? 1 : 0)) return rc;
    using clock = std::chrono::steady_clock;
    std::atomic<int> arrived{0}, failed{0};
    std::atomic<bool> go{false};
    std::vector<clock::time_point> done(n_seq);
    clock::time_point t0;
    auto worker = [&](int i) {
        // a new thread starts on device 0: bind it to the context's device before anything creates an event or launches
        int rc = rdvio_hip_ctx_attach_thread(seqs[i].ctx);
        for (int k = 0; k < warmup && !rc; ++k) rc = rdvio_hip_frame_step(&seqs[i], k);
        if (!rc) rc = rdvio_hip_sync(seqs[i].ctx);
        if (rc) failed.store(rc);
        arrived.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();   // common start
        for (int k = warmup; k < warmup + steps && !rc && !failed.load(); ++k) rc = rdvio_hip_frame_step(&seqs[i], k);
        if (!rc) rc = rdvio_hip_sync(seqs[i].ctx);
        if (rc) failed.store(rc);
        done[i] = clock::now();
    };
    std::vector<std::thread> th;
    th.reserve(n_seq);
    for (int i = 0; i < n_seq; ++i) th.emplace_back(worker, i);
    while (arrived.load() < n_seq) std::this_thread::yield();
    t0 = clock::now();
    go.store(true, std::memory_order_release);
    for (auto &t : th) t.join();
    double worst = 0.0;
    for (int i = 0; i < n_seq; ++i) {
        const double s = std::chrono::duration<double>(done[i] - t0).count();
        if (per_sequence_s) per_sequence_s[i] = s;
        if (s > worst) worst = s;
    }
    if (elapsed_s) *elapsed_s = worst;   // common start -> the last sequence's last frame
    for (int i = 0; i < n_seq; ++i) (void)rdvio_hip_ctx_set_wait_mode(seqs[i].ctx, 0);
    return failed.load();
}
