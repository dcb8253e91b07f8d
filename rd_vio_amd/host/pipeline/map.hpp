// Frame / Track / Map of the per-frame orchestration and the host mirrors of PreIntegrator and MarginalizationFactor.
//
// Behaviour follows /root/reference/src/rdvio_map/src/{frame,track,map}.cpp and
// /root/reference/src/rdvio_estimation/include/rdvio/estimation/marginalization_factor.h:9-40; the representation is
// this build's own: ids come from a per-pipeline generator (the reference's are process-global statics), a track keeps
// its observations in an id-ordered map of (frame, keypoint index), and every observation carries the 3x3 tangent frame
// that the reference stores inside its eagerly created CeresReprojectionErrorFactor (reprojection_factor.h:16-22), so
// the BA graph can be exported as SoA index lists without touching factor objects.
#pragma once

#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/rdvio_pipeline.h"
#include "geom.hpp"

namespace rdvio_pipe {

constexpr double GRAVITY_NOMINAL = 9.80665;
constexpr size_t nil = size_t(-1);
// layout of a preintegration record (include/rdvio_hip.h: RDVIO_PREINT_SIZE)
constexpr int PRE_T = 0, PRE_Q = 1, PRE_P = 5, PRE_V = 8, PRE_JAC = 461;

struct ImuData {
    double t;
    V3 w, a;
};
struct ExtrinsicParams {
    Q4 q_cs;
    V3 p_cs;
};
struct PoseState {
    Q4 q;
    V3 p;
};
struct MotionState {
    V3 v, bg, ba;
};

enum FrameTag { FT_KEYFRAME = 0, FT_NO_TRANSLATION, FT_FIX_POSE, FT_FIX_MOTION };
enum TrackTag { TT_VALID = 0, TT_TRIANGULATED, TT_FIX_INVD, TT_TRASH, TT_STATIC, TT_OUTLIER, TT_TEMP };

struct IdGenerator {
    size_t next_frame = 0, next_track = 0;
};

// Who calls: the feature tracker (the caller's thread) or the estimator (Frontend: SlidingWindowTracker / Initializer --
// the worker thread when the pipeline runs threaded).  The two never share a staging buffer or a counter.
enum CallerLane { LANE_TRACKER = 0, LANE_ESTIMATOR = 1 };

struct Backend {
    rdvio_backend fn;
    std::string error;
    double preintegrate_seconds[2] = {0.0, 0.0};  // PreIntegrator::integrate is called from map-level code; per CallerLane
    int64_t preintegrate_calls[2] = {0, 0};
    void check(int rc, const char *what);  // throws std::runtime_error on failure
};

// rdvio::Image: handle + timestamp; the backend image is destroyed with the last frame clone that shares it
struct ImageRef {
    Backend *backend = nullptr;
    void *handle = nullptr;
    double t = 0.0;
    int width = 0, height = 0;
    ~ImageRef() {
        if (backend && handle) backend->fn.image_destroy(backend->fn.user, handle);
    }
};

class Frame;

// PreIntegrator (preintegrator.h:10-47): the raw samples live here, the integration runs behind the backend
struct PreIntegrator {
    std::vector<ImuData> data;
    double noise[36] = {0};                   // cov_w cov_a cov_bg cov_ba
    std::vector<double> delta = std::vector<double>(RDVIO_PREINT_SIZE, 0.0);
    PreIntegrator() { delta[PRE_Q + 3] = 1.0; }
    bool integrate(Backend &be, CallerLane lane, double t, const V3 &bg, const V3 &ba, bool compute_jacobian, bool compute_covariance);
    // the estimator's integration (Jacobians and covariance) in two halves when the backend offers them: begin enqueues, end
    // collects (otherwise end is the whole call).  Same result as integrate(be, LANE_ESTIMATOR, t, bg, ba, true, true).
    void integrate_begin(Backend &be, double t, const V3 &bg, const V3 &ba);
    bool integrate_end(Backend &be);
    void predict(const Frame *old_frame, Frame *new_frame) const;  // preintegrator.cpp:102-112
    // the same integrate() for several independent integrators in one backend call (all with the same noise model and
    // flags); ok[i] = false where data is empty
    struct Job { PreIntegrator *pre; double t; V3 bg, ba; };
    static std::vector<char> integrate_batch(Backend &be, CallerLane lane, const std::vector<Job> &jobs, bool compute_jacobian, bool compute_covariance);
    // key of the integration `delta` holds (samples, end time, biases, flags): integrate() with the same key is a no-op --
    // the reference integrates the new frame in mirror_frame and again, from identical inputs, in judge_track_status
    // (sliding_window_tracker.cpp:73-76, 586-590)
    struct Pending { bool active = false, in_flight = false; double t = 0.0; V3 bg{}, ba{}; } pending;
    struct Key {
        size_t n = 0;
        double t = 0, t_first = 0, t_last = 0;
        V3 bg, ba;
        bool cj = false, cc = false, valid = false;
    } key;
    double dt() const { return delta[PRE_T]; }
    Q4 dq() const { return {delta[PRE_Q], delta[PRE_Q + 1], delta[PRE_Q + 2], delta[PRE_Q + 3]}; }
    V3 dp() const { return {delta[PRE_P], delta[PRE_P + 1], delta[PRE_P + 2]}; }
    V3 dv() const { return {delta[PRE_V], delta[PRE_V + 1], delta[PRE_V + 2]}; }
};

class Track;
class Map;

class Frame {
  public:
    explicit Frame(IdGenerator &ids) : id_(++ids.next_frame) {}
    size_t id() const { return id_; }
    bool tag(FrameTag f) const { return (tags_ >> f) & 1u; }
    void set_tag(FrameTag f, bool v) { tags_ = v ? (tags_ | (1u << f)) : (tags_ & ~(1u << f)); }

    std::unique_ptr<Frame> clone() const;  // frame.cpp:20-38
    size_t keypoint_num() const { return bearings.size(); }
    void append_keypoint(const V3 &bearing);
    const V3 &get_keypoint(size_t i) const { return bearings[i]; }
    Track *get_track(size_t i) const { return tracks[i]; }
    Track *get_track(size_t i, Map *allocation_map);  // creates the track if absent (frame.cpp:46-55)
    PoseState get_pose(const ExtrinsicParams &sensor) const { return {pose.q * sensor.q_cs, pose.p + rot(pose.q, sensor.p_cs)}; }
    void set_pose(const ExtrinsicParams &sensor, const PoseState &ps) {
        pose.q = ps.q * conj(sensor.q_cs);
        pose.p = ps.p - rot(pose.q, sensor.p_cs);
    }
    void get_state(double *s16) const;
    void set_state(const double *s16);

    Map *map = nullptr;
    double K[9] = {0};
    double sqrt_inv_cov[4] = {0};
    std::shared_ptr<ImageRef> image;
    PoseState pose;
    MotionState motion;
    ExtrinsicParams camera, imu;
    PreIntegrator preintegration, keyframe_preintegration;
    std::vector<std::unique_ptr<Frame>> subframes;

    std::vector<V3> bearings;
    std::vector<Track *> tracks;
    std::vector<std::array<double, 9>> tangents;  // per observation: [b1 b2 z] columns, row-major (valid when tracked)

  private:
    Frame(size_t id, unsigned tags) : id_(id), tags_(tags) {}
    size_t id_;
    unsigned tags_ = 0;
};

class Track {
  public:
    Track(IdGenerator &ids, Map *m) : map(m), id_(++ids.next_track) { set_tag(TT_STATIC, true); }  // track.cpp:7
    Track(size_t reserved_id, Map *m) : map(m), id_(reserved_id) { set_tag(TT_STATIC, true); }   // an id taken from the generator earlier (Map::reserved_track_ids)
    size_t id() const { return id_; }
    bool tag(TrackTag f) const { return (tags_ >> f) & 1u; }
    void set_tag(TrackTag f, bool v) { tags_ = v ? (tags_ | (1u << f)) : (tags_ & ~(1u << f)); }
    bool all_tagged(std::initializer_list<TrackTag> fs) const {
        for (TrackTag f : fs)
            if (!tag(f)) return false;
        return true;
    }

    size_t keypoint_num() const { return refs.size(); }
    std::pair<Frame *, size_t> first_keypoint() const { return refs.begin()->second; }
    Frame *first_frame() const { return refs.begin()->second.first; }
    Frame *last_frame() const { return refs.rbegin()->second.first; }
    const std::map<size_t, std::pair<Frame *, size_t>> &keypoint_map() const { return refs; }
    bool has_keypoint(const Frame *f) const { return refs.count(f->id()) > 0; }
    size_t get_keypoint_index(const Frame *f) const {
        auto it = refs.find(f->id());
        return it == refs.end() ? nil : it->second.second;
    }
    void add_keypoint(Frame *frame, size_t keypoint_index);                 // track.cpp:14-23
    void remove_keypoint(Frame *frame, bool suicide_if_empty = true);       // track.cpp:25-44
    std::optional<V3> triangulate();                                        // track.cpp:46-76
    V3 get_landmark_point() const;                                          // track.cpp:90-95
    void set_landmark_point(const V3 &p);                                   // track.cpp:97-101

    Map *map;
    size_t map_index = 0;
    double inv_depth = 0.0;
    size_t m_life = 0;
    // scratch of the solver's graph assembly (BaBuilder): this track's state index in the builder whose stamp is ba_stamp
    mutable int ba_index = -1;
    mutable uint64_t ba_stamp = 0;

  private:
    size_t id_;
    unsigned tags_ = 0;
    std::map<size_t, std::pair<Frame *, size_t>> refs;  // ordered by frame id (compare<Frame *>, types.h:36-40)
};

// MarginalizationFactor (marginalization_factor.h:9-40): the sqrt prior and its linearisation points
struct MarginalizationPrior {
    std::vector<Frame *> frames;
    std::vector<double> lin;  // frames x 16
    std::vector<double> S, f; // (15 frames)^2, 15 frames
    // a marginalisation is under way behind the backend (marginalize_begin): S / f / lin hold their new sizes and get their
    // values at the first read (ready())
    bool pending = false;
    void ready(Backend &be);
};

class Map {
  public:
    explicit Map(IdGenerator &ids) : ids(ids) {}
    ~Map();
    size_t frame_num() const { return frames.size(); }
    Frame *get_frame(size_t i) const { return frames[i].get(); }
    void attach_frame(std::unique_ptr<Frame> frame, size_t position = nil);
    std::unique_ptr<Frame> detach_frame(size_t index);
    void untrack_frame(Frame *frame);
    void erase_frame(size_t index);
    size_t frame_index_by_id(size_t id) const;
    size_t track_num() const { return tracks.size(); }
    Track *get_track(size_t i) const { return tracks[i].get(); }
    Track *create_track();
    void erase_track(Track *track);
    void prune_tracks(const std::function<bool(const Track *)> &condition);
    void recycle_track(Track *track);
    void drop_front_frame();  // frames.erase(begin) after a marginalisation (map.cpp:61)

    IdGenerator &ids;
    // ids set aside for the tracks this map creates next (first, one past the last): the sliding-window map creates its tracks
    // inside the frontend's step, concurrently with the feature tracker's own -- the ids it may use were drawn at the hand-over,
    // in the order a single thread would have drawn them
    size_t reserved_track_ids[2] = {0, 0};
    std::unique_ptr<MarginalizationPrior> marginalization_factor;

  private:
    std::deque<std::unique_ptr<Frame>> frames;
    std::vector<std::unique_ptr<Track>> tracks;
};

// lie_algebra.cpp:47-56 + reprojection_factor.h:16-22: [b1 b2 z] as the columns of a row-major 3x3
std::array<double, 9> tangent_frame(const V3 &z);

}  // namespace rdvio_pipe
