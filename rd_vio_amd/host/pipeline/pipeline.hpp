// The per-frame orchestration: Handler -> FeatureTracker -> Frontend -> (bootstrap | SlidingWindowTracker).
//
// Threading (rdvio_pipeline_config::threading).  The reference runs FeatureTracker and Frontend either synchronously
// (THREADING=OFF: FeatureTracker::track_frame and Frontend::issue_frame call run() directly, feature_tracker.cpp:113-118,
// frontend.cpp:72-77) or on two polling worker threads that meet under the map mutex (handler.cpp:35-50) -- in which case
// what the tracker sees of the frontend's results depends on timing.  Here:
//   0  inline (THREADING=OFF);
//   1  the pipelined schedule on one thread;
//   2  the pipelined schedule with the frontend's step on a worker thread.
// The pipelined schedule fixes the interleaving: when the tracker has finished frame k it (a) waits for the frontend's step
// for frame k-1 and PUBLISHES its results to the tracker's side (latest optimised state, deferred track tags), (b) does the
// part of mirror_frame that touches both maps, (c) starts the frontend's step for frame k.  Between two hand-overs the two
// steps share no mutable state: the frontend's step works on the sliding-window map and a snapshot, the tracker on the
// feature-tracking map and what was published.  1 and 2 therefore compute the same thing; 2 overlaps them.
#pragma once

#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <optional>
#include <thread>
#include <tuple>

#include "map.hpp"

namespace rdvio_pipe {

// Written by the tracker's side: frames_tracked, no_translation_frames, backend classes 0-2 and 6; by the frontend's
// step: everything else -- no field has two writers.
struct Counters {
    int64_t frames_tracked = 0, window_solves = 0, keyframes = 0, marginalizations = 0, localizations = 0, subwindow_solves = 0;
    int64_t max_problem_frames = 0, max_problem_factors = 0, solver_iterations = 0;
    int64_t no_translation_frames = 0, rotation_prior_factors = 0, parsac_judgements = 0, tracks_marked_dynamic = 0;
    // seconds spent inside backend calls: preprocess, detect, track, preintegrate, ba_solve, marginalize, image_create
    double backend_seconds[7] = {0, 0, 0, 0, 0, 0, 0};
    int64_t backend_calls[7] = {0, 0, 0, 0, 0, 0, 0};
};

// scope timer of one backend call class
struct BackendTimer {
    Counters &c;
    int k;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    BackendTimer(Counters &c, int k) : c(c), k(k) {}
    ~BackendTimer() {
        c.backend_seconds[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        c.backend_calls[k]++;
    }
};

// host-side stage profile (diagnostic: RDVIO_PIPELINE_PROF=1 prints it to stderr when the pipeline is destroyed): inclusive
// wall time of the orchestration's stages, backend calls included -- subtract the backend counters for the host share
struct HostProf {
    static constexpr int N = 19;
    static constexpr const char *names[N] = {"tracker.run", "tracker.track_keypoints", "tracker.detect_keypoints", "frontend.run", "swt.mirror_frame",
                                             "swt.localize_newframe", "swt.refine_window", "swt.refine_subwindow", "swt.marginalize_frame0",
                                             "swt.track_landmark+manage", "ba.solve (assembly + backend)", "rd path",
                                             "frontend step (either thread)", "hand-over: wait for the step", "hand-over: publish + tags",
                                             "tracker.gates (host)", "mirror packet (tracker's side, ahead of the hand-over)",
                                             "swt.refine_window: graph", "swt.refine_window: culling"};
    double seconds[N] = {};
    long calls[N] = {};
    bool on = false;
};
struct HostTimer {
    HostProf &p;
    int k;
    std::chrono::steady_clock::time_point t0;
    HostTimer(HostProf &p, int k) : p(p), k(k) {
        if (p.on) t0 = std::chrono::steady_clock::now();
    }
    ~HostTimer() {
        if (p.on) {
            p.seconds[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            p.calls[k]++;
        }
    }
};

// diagnostic (RDVIO_PIPELINE_PROF): solver work by call site -- 0 localize_newframe, 1 refine_window, 2 refine_subwindow, 3 other
struct SolveProf {
    long calls[4] = {0, 0, 0, 0}, iterations[4] = {0, 0, 0, 0}, factors[4] = {0, 0, 0, 0}, frames[4] = {0, 0, 0, 0};
    double seconds[4] = {0, 0, 0, 0};
};

struct Shared {  // what every stage needs
    HostProf prof;
    SolveProf solve_prof;
    rdvio_pipeline_config cfg;
    Backend backend;
    IdGenerator ids;
    std::atomic<uint64_t> ba_stamps{0};   // one stamp per BaBuilder (Track::ba_stamp)
    Counters counters;
    std::vector<std::array<double, 17>> init_states;
    // bin confidences of the two PARSAC call sites (function-local statics in the reference: pnp.h:195, stereo.cpp:147)
    std::vector<float> pnp_bin_confidences = std::vector<float>(400, 0.5f), essential_bin_confidences = std::vector<float>(400, 0.5f);
    size_t pnp_iterations_hint = 0;   // iterations the last IMU-PARSAC solve replayed (sizes the next solve's first device batch)
};

// SoA export of one Solver problem (row A16): frames / landmarks / factors by index, ordered by landmark
class BaBuilder {
  public:
    explicit BaBuilder(Shared &sh);
    ~BaBuilder();
    // Solver::add_frame_states: constancy from the frame's FT_FIX_POSE / FT_FIX_MOTION tags (solver.cpp:88-114)
    int add_frame_states(Frame *frame, bool with_motion = true);
    // a frame whose values are read by a factor but which is not a parameter of this solve
    int add_constant_frame(Frame *frame);
    int add_track_states(Track *track, bool constant);
    void add_reprojection_error(Frame *frame, size_t keypoint_index);   // Solver::add_factor(ReprojectionErrorFactor *)
    void add_reprojection_prior(Frame *frame, Track *track);            // create_reprojection_prior_factor
    void add_reprojection_prior(Frame *frame, Track *track, size_t keypoint_index);   // (the caller knows the track's keypoint in `frame`)
    void add_rotation_prior(Frame *frame, Track *track);                // create_rotation_prior_factor
    void add_preintegration(Frame *frame_i, Frame *frame_j, const PreIntegrator &pre, bool prior);
    // PreIntegrator::integrate(t, bg, ba, true, true) + add_preintegration in one: the integration runs inside the solve call
    // (rdvio_ba_problem::n_pre_jobs) when every preintegration factor of the solve comes this way, through the backend's
    // separate entry otherwise.  false: no samples, no factor (preintegrator.cpp:80-81)
    bool add_integrated_preintegration(Frame *frame_i, Frame *frame_j, PreIntegrator &pre, double t, const V3 &bg, const V3 &ba);
    void add_marginalization(const MarginalizationPrior *prior) { this->prior = prior; }
    bool solve(rdvio_ba_summary *summary = nullptr);                    // Solver::solve + in-place state update
    // the same in two halves when the backend offers them (can_begin): begin packs and enqueues, end waits and updates the
    // states.  `chain`: a builder whose solve has been begun; this solve starts frame `chain_frame` from THAT solve's result
    // (handed over behind the backend), so it can be built and begun while the first one is still running.
    bool can_begin() const;
    void solve_begin(int slot, const BaBuilder *chain = nullptr, Frame *chain_frame = nullptr);
    bool solve_end(rdvio_ba_summary *summary = nullptr);
    int kind = 3;                                                       // call site, for the diagnostic profile

  private:
    struct Fac { int tgt, ref, lm; const double *tangent; };
    struct Rot { int tgt, ref; V3 zref; const double *tangent; };
    struct Pre { int i, j; const double *delta; PreIntegrator *job; double t; V3 bg, ba; };
    struct Packed;
    void pack();
    bool apply(rdvio_ba_summary *summary);
    int frame_index(Frame *frame) const;
    std::unique_ptr<Packed> packed;
    Shared &sh;
    std::vector<Frame *> frames;
    std::vector<uint8_t> frame_fixed;
    std::unordered_map<const Frame *, int> fidx;
    std::vector<Track *> lms;
    std::vector<uint8_t> lm_fixed;
    uint64_t stamp;   // a track knows its index in THIS builder while Track::ba_stamp == stamp (no hash map on the assembly path)
    std::vector<Fac> facs;
    std::vector<Rot> rots;
    std::vector<Pre> pres;
    const MarginalizationPrior *prior = nullptr;
};

// One step of the frontend (Frontend::run's tracking branch, frontend.cpp:47-66) as a unit of work: prepared at the
// hand-over (both maps quiescent), executed inline or on the worker, its results published at the next hand-over.
// What mirror_frame (sliding_window_tracker.cpp:29-78) takes from the FEATURE-TRACKING map for one new frame, gathered on the
// tracker's side: the frame's clone with the IMU samples since the sliding-window map's newest frame, and which keypoints of
// that newest frame continue into the new one.  Depends on the feature-tracking map and the newest frame's id only, so the
// tracker builds it while the previous step is still running.
struct MirrorPacket {
    size_t frame_i_id = nil, frame_j_id = nil;
    bool found = false;                                  // both frames are in the feature-tracking map
    std::unique_ptr<Frame> curr_frame;
    std::vector<std::pair<uint32_t, uint32_t>> matches;  // (keypoint of frame i, keypoint of frame j)
    std::vector<Track *> ft_tracks;                      // the feature-tracking map's track of each match
};

struct FrontendJob {
    size_t frame_id = nil;
    bool mirrored = false;                   // mirror_frame found both frames (sliding_window_tracker.cpp:33-36)
    MirrorPacket packet;
    size_t track_ids[2] = {0, 0};            // ids drawn at the hand-over for the tracks the step creates while mirroring
    Frame *new_frame_i = nullptr, *new_frame_j = nullptr;
    // update_track_status reads, per keypoint of the new frame, whether the FEATURE-TRACKING map's track exists (bit 0) and
    // is TT_STATIC (bit 1) (sliding_window_tracker.cpp:731-757); taken at the hand-over
    std::vector<uint8_t> old_track_flags;
    // ---- results
    bool ok = true;
    std::tuple<double, PoseState, MotionState> latest_state;
    std::vector<size_t> old_tracks_nonstatic;  // keypoints of the new frame whose feature-tracking-map track loses TT_STATIC
    // the sliding-window map's newest frame when the step ended, for the next hand-over: its id and, per keypoint, whether it has
    // a track (bit 0) and whether that track is TT_TRASH and not TT_STATIC (bit 1)
    size_t newest_id = nil;
    std::vector<uint8_t> newest_flags;
    std::exception_ptr error;
};

class SlidingWindowTracker {
  public:
    SlidingWindowTracker(std::unique_ptr<Map> keyframe_map, Shared &sh);
    // mirror_frame (sliding_window_tracker.cpp:29-78) in two halves: the part that reads and tags the feature-tracking map
    // (hand-over, caller's thread) and the preintegration + prediction of the new frame (the frontend's step)
    //   gather (tracker's side, any time after the tracker finished the frame)  ->  mirror_frame_handover (hand-over: the tags
    //   mirror_frame writes INTO the feature-tracking map, ids for the tracks to come)  ->  mirror_frame_apply + _finish (the
    //   frontend's step: the sliding-window map's side, preintegration, prediction)
    static void gather_mirror_packet(const Map *feature_tracking_map, size_t frame_i_id, size_t frame_j_id, MirrorPacket &out);
    static void mirror_frame_handover(IdGenerator &ids, const std::vector<uint8_t> &newest_flags, const Map *feature_tracking_map, bool parsac, FrontendJob &job);
    void mirror_frame_begin(FrontendJob &job);     // (issues the new frame's preintegration; _apply runs beside it)
    void mirror_frame_apply(FrontendJob &job);
    void mirror_frame_finish(FrontendJob &job);
    void newest_frame_summary(size_t &id, std::vector<uint8_t> &flags) const;
    bool track(FrontendJob &job);
    std::tuple<double, PoseState, MotionState> get_latest_state() const;
    std::unique_ptr<Map> map;

  private:
    Frame *localize_newframe(BaBuilder &solver);
    bool manage_keyframe();
    void track_landmark();
    void refine_window();
    void slide_window();
    void refine_subwindow(BaBuilder *after = nullptr, Frame *after_frame = nullptr);
    void marginalize_frame0();
    // RD dynamic-outlier path (parsac_flag; sliding_window_tracker.cpp:487-769)
    bool judge_track_status();
    void update_track_status(FrontendJob &job);
    bool filter_parsac_2d2d(Frame *frame_i, Frame *frame_j, std::vector<char> &mask, std::vector<size_t> &pts_to_index);
    Shared &sh;
    double m_th = 0.0;
};

// Bootstrap of the window from externally supplied keyframe states (the reference's Initializer without its SfM /
// IMU-alignment stages).  mirror_keyframe_map and the closing BA follow initializer.cpp:20-140.
class Initializer {
  public:
    explicit Initializer(Shared &sh) : sh(sh) {}
    void mirror_keyframe_map(Map *feature_tracking_map, size_t init_frame_id);
    std::unique_ptr<SlidingWindowTracker> initialize();

  private:
    bool bootstrap_from_supplied_states();   // rdvio_pipeline_set_init_states path
    bool init_sfm();                          // initializer.cpp:142-366
    bool init_imu();                          // :368-381
    void solve_gyro_bias();                   // :383-409
    void solve_gravity_scale_velocity();      // :411-447
    void refine_scale_velocity_via_gravity(); // :449-503
    void reset_states();
    void preintegrate();
    bool apply_init(bool apply_ba = false, bool apply_velocity = true);  // :522-560
    Shared &sh;
    std::unique_ptr<Map> map;
    V3 bg, ba, gravity;
    double scale = 1.0;
    std::vector<V3> velocities;
};

class FeatureTracker;

class Frontend {
  public:
    Frontend(FeatureTracker *ft, Shared &sh);
    ~Frontend();
    void issue_frame(Frame *frame);
    // what the feature tracker sees: the state PUBLISHED at the last hand-over (in inline mode: the newest)
    std::tuple<double, size_t, PoseState, MotionState> get_latest_state() const { return latest_state; }
    int get_system_state() const { return initializer ? 0 : (sliding_window_tracker ? 1 : 3); }
    // waits for the step in flight (threading == 2); publishes nothing -- for readers of the sliding-window map
    void drain();
    std::unique_ptr<SlidingWindowTracker> sliding_window_tracker;

  private:
    void run();
    void execute(FrontendJob &job);   // the frontend's step proper (either thread)
    void publish();                   // the finished step's results -> the tracker's side
    void worker_main();
    FeatureTracker *feature_tracker;
    Shared &sh;
    std::unique_ptr<Initializer> initializer;
    std::deque<size_t> pending_frame_ids;
    std::tuple<double, size_t, PoseState, MotionState> latest_state;
    // the step in flight / finished but not yet published
    std::unique_ptr<FrontendJob> job;
    // the sliding-window map's newest frame as of the last published step (FrontendJob::newest_id / newest_flags)
    size_t newest_id = nil;
    std::vector<uint8_t> newest_flags;
    // worker (threading == 2): one slot, phase 0 idle, 1 posted, 2 done, 3 quit; waits spin briefly, then sleep
    std::thread worker;
    std::atomic<int> phase{0};
    std::mutex mtx;
    std::condition_variable cv;
};

class FeatureTracker {
  public:
    explicit FeatureTracker(Shared &sh);
    void set_frontend(Frontend *f) { frontend = f; }
    void track_frame(std::unique_ptr<Frame> frame);
    std::optional<std::tuple<double, PoseState, MotionState>> get_latest_state() const { return latest_state; }
    std::unique_ptr<Map> map;

  private:
    void run();
    void detect_keypoints(Frame *frame);                       // Frame::detect_keypoints, frame.cpp:55-72
    void track_keypoints(Frame *frame, Frame *next_frame);     // Frame::track_keypoints, frame.cpp:74-172
    Shared &sh;
    Frontend *frontend = nullptr;
    std::deque<std::unique_ptr<Frame>> frames;
    std::optional<std::tuple<double, PoseState, MotionState>> latest_state;
};

class Handler {
  public:
    explicit Handler(Shared &sh);
    PoseState track_gyroscope(double t, double x, double y, double z);
    PoseState track_accelerometer(double t, double x, double y, double z);
    PoseState track_camera(std::shared_ptr<ImageRef> image);
    std::tuple<double, PoseState> get_latest_state() const;
    size_t queued_frames() const { return frames.size(); }   // pushed, waiting for the IMU sample that releases them
    FeatureTracker feature_tracker;
    Frontend frontend;

  private:
    struct Gyro { double t; V3 w; };
    struct Acc { double t; V3 a; };
    void track_imu(const ImuData &imu);
    PoseState predict_pose(double t);
    Shared &sh;
    std::deque<Gyro> gyroscopes;
    std::deque<Acc> accelerometers;
    std::deque<ImuData> imus, frontal_imus;
    std::deque<std::unique_ptr<Frame>> frames;
};

}  // namespace rdvio_pipe

struct rdvio_pipeline {
    rdvio_pipe::Shared shared;
    std::unique_ptr<rdvio_pipe::Handler> handler;
    std::string error;
};
