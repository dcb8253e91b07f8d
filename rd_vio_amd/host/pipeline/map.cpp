// Frame / Track / Map behaviour (see map.hpp for the reference file:line of each method).
#include "map.hpp"

#include <chrono>
#include <stdexcept>

namespace rdvio_pipe {

void Backend::check(int rc, const char *what) {
    if (rc == RDVIO_OK) return;
    const char *msg = fn.last_error ? fn.last_error(fn.user) : "";
    error = std::string(what) + " failed (code " + std::to_string(rc) + "): " + (msg ? msg : "");
    throw std::runtime_error(error);
}

// lie_algebra.cpp:47-56
std::array<double, 9> tangent_frame(const V3 &x) {
    const double xv[3] = {x.x, x.y, x.z};
    int d = 0;
    for (int i = 1; i < 3; ++i)
        if (std::fabs(xv[i]) > std::fabs(xv[d])) d = i;
    V3 unit{0, 0, 0};
    const int u = (d + 1) % 3;
    (u == 0 ? unit.x : (u == 1 ? unit.y : unit.z)) = 1.0;
    const V3 b1 = normalized(cross(x, unit));
    const V3 b2 = normalized(cross(x, b1));
    return {b1.x, b2.x, x.x, b1.y, b2.y, x.y, b1.z, b2.z, x.z};
}

void MarginalizationPrior::ready(Backend &be) {
    if (!pending) return;
    pending = false;
    be.check(be.fn.marginalize_end(be.fn.user, S.data(), f.data(), lin.data()), "marginalize (end)");
}

// ---------------------------------------------------------------------------------------------- PreIntegrator
bool PreIntegrator::integrate(Backend &be, CallerLane lane, double t, const V3 &bg, const V3 &ba, bool compute_jacobian, bool compute_covariance) {
    return integrate_batch(be, lane, {Job{this, t, bg, ba}}, compute_jacobian, compute_covariance)[0] != 0;
}

std::vector<char> PreIntegrator::integrate_batch(Backend &be, CallerLane lane, const std::vector<Job> &jobs, bool compute_jacobian, bool compute_covariance) {
    std::vector<char> ok(jobs.size(), 0);
    std::vector<int32_t> off(1, 0);
    std::vector<double> imu, t_end, bg, ba;
    std::vector<size_t> which;
    auto same = [](const V3 &a, const V3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
    for (size_t j = 0; j < jobs.size(); ++j) {
        const std::vector<ImuData> &data = jobs[j].pre->data;
        if (data.empty()) continue;  // preintegrator.cpp:80-81
        ok[j] = 1;
        const Key &k = jobs[j].pre->key;
        if (k.valid && k.n == data.size() && k.t == jobs[j].t && k.t_first == data.front().t && k.t_last == data.back().t && same(k.bg, jobs[j].bg) &&
            same(k.ba, jobs[j].ba) && k.cj == compute_jacobian && k.cc == compute_covariance)
            continue;  // `delta` already holds exactly this integration
        which.push_back(j);
        for (const ImuData &d : data) imu.insert(imu.end(), {d.t, d.w.x, d.w.y, d.w.z, d.a.x, d.a.y, d.a.z});
        off.push_back((int32_t)(imu.size() / 7));
        t_end.push_back(jobs[j].t);
        bg.insert(bg.end(), {jobs[j].bg.x, jobs[j].bg.y, jobs[j].bg.z});
        ba.insert(ba.end(), {jobs[j].ba.x, jobs[j].ba.y, jobs[j].ba.z});
    }
    if (which.empty()) return ok;
    std::vector<double> out(which.size() * RDVIO_PREINT_SIZE);
    {
        const auto t0 = std::chrono::steady_clock::now();
        struct Acc {
            Backend &be;
            int lane;
            std::chrono::steady_clock::time_point t0;
            ~Acc() {
                be.preintegrate_seconds[lane] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                be.preintegrate_calls[lane]++;
            }
        } acc{be, (int)lane, t0};
        // the estimator's integrations have an entry of their own when the backend separates the two callers' staging
        auto fn = (lane == LANE_ESTIMATOR && be.fn.preintegrate_estimator) ? be.fn.preintegrate_estimator : be.fn.preintegrate;
        be.check(fn(be.fn.user, (int)which.size(), off.data(), imu.data(), t_end.data(), bg.data(), ba.data(), jobs[which[0]].pre->noise,
                    compute_jacobian ? 1 : 0, compute_covariance ? 1 : 0, out.data()),
                 "preintegrate");
    }
    for (size_t k = 0; k < which.size(); ++k) {
        const Job &job = jobs[which[k]];
        std::copy(out.begin() + (std::ptrdiff_t)(k * RDVIO_PREINT_SIZE), out.begin() + (std::ptrdiff_t)((k + 1) * RDVIO_PREINT_SIZE), job.pre->delta.begin());
        job.pre->key = Key{job.pre->data.size(), job.t, job.pre->data.front().t, job.pre->data.back().t, job.bg, job.ba, compute_jacobian, compute_covariance, true};
    }
    return ok;
}

void PreIntegrator::integrate_begin(Backend &be, double t, const V3 &bg, const V3 &ba) {
    pending = Pending{true, false, t, bg, ba};
    if (!(be.fn.preintegrate_estimator_begin && be.fn.preintegrate_estimator_end) || data.empty()) return;
    auto same = [](const V3 &a, const V3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
    if (key.valid && key.n == data.size() && key.t == t && key.t_first == data.front().t && key.t_last == data.back().t && same(key.bg, bg) && same(key.ba, ba) &&
        key.cj && key.cc)
        return;   // `delta` already holds exactly this integration
    std::vector<double> imu;
    imu.reserve(7 * data.size());
    for (const ImuData &d : data) imu.insert(imu.end(), {d.t, d.w.x, d.w.y, d.w.z, d.a.x, d.a.y, d.a.z});
    const int32_t off[2] = {0, (int32_t)data.size()};
    const double bgv[3] = {bg.x, bg.y, bg.z}, bav[3] = {ba.x, ba.y, ba.z};
    const auto t0 = std::chrono::steady_clock::now();
    be.check(be.fn.preintegrate_estimator_begin(be.fn.user, 1, off, imu.data(), &t, bgv, bav, noise, 1, 1), "preintegrate (begin)");
    be.preintegrate_seconds[LANE_ESTIMATOR] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pending.in_flight = true;
}

bool PreIntegrator::integrate_end(Backend &be) {
    const Pending p = pending;
    pending = Pending{};
    if (!p.active) return false;
    if (!p.in_flight) return integrate(be, LANE_ESTIMATOR, p.t, p.bg, p.ba, true, true);
    const auto t0 = std::chrono::steady_clock::now();
    be.check(be.fn.preintegrate_estimator_end(be.fn.user, delta.data()), "preintegrate (end)");
    be.preintegrate_seconds[LANE_ESTIMATOR] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    be.preintegrate_calls[LANE_ESTIMATOR]++;
    key = Key{data.size(), p.t, data.front().t, data.back().t, p.bg, p.ba, true, true, true};
    return true;
}

void PreIntegrator::predict(const Frame *old_frame, Frame *new_frame) const {
    const V3 gravity{0, 0, -GRAVITY_NOMINAL};
    const double t = dt();
    new_frame->motion.bg = old_frame->motion.bg;
    new_frame->motion.ba = old_frame->motion.ba;
    new_frame->motion.v = old_frame->motion.v + gravity * t + rot(old_frame->pose.q, dv());
    new_frame->pose.p = old_frame->pose.p + 0.5 * gravity * t * t + old_frame->motion.v * t + rot(old_frame->pose.q, dp());
    new_frame->pose.q = old_frame->pose.q * dq();
}

// ---------------------------------------------------------------------------------------------- Frame
std::unique_ptr<Frame> Frame::clone() const {
    std::unique_ptr<Frame> f(new Frame(id_, tags_));
    std::copy(K, K + 9, f->K);
    std::copy(sqrt_inv_cov, sqrt_inv_cov + 4, f->sqrt_inv_cov);
    f->image = image;
    f->pose = pose;
    f->motion = motion;
    f->camera = camera;
    f->imu = imu;
    f->preintegration = preintegration;
    // keyframe_preintegration keeps the noise model (the reference copies `preintegration` over it before use)
    std::copy(preintegration.noise, preintegration.noise + 36, f->keyframe_preintegration.noise);
    f->bearings = bearings;
    f->tracks.assign(bearings.size(), nullptr);
    f->tangents.assign(bearings.size(), std::array<double, 9>{});
    f->map = nullptr;
    return f;
}

void Frame::append_keypoint(const V3 &bearing) {
    bearings.push_back(bearing);
    tracks.push_back(nullptr);
    tangents.push_back(std::array<double, 9>{});
}

Track *Frame::get_track(size_t i, Map *allocation_map) {
    if (!allocation_map) allocation_map = map;
    if (tracks[i] == nullptr) {
        Track *track = allocation_map->create_track();
        track->add_keypoint(this, i);
    }
    return tracks[i];
}

void Frame::get_state(double *s) const {
    s[0] = pose.q.x; s[1] = pose.q.y; s[2] = pose.q.z; s[3] = pose.q.w;
    s[4] = pose.p.x; s[5] = pose.p.y; s[6] = pose.p.z;
    s[7] = motion.v.x; s[8] = motion.v.y; s[9] = motion.v.z;
    s[10] = motion.bg.x; s[11] = motion.bg.y; s[12] = motion.bg.z;
    s[13] = motion.ba.x; s[14] = motion.ba.y; s[15] = motion.ba.z;
}
void Frame::set_state(const double *s) {
    pose.q = {s[0], s[1], s[2], s[3]};
    pose.p = {s[4], s[5], s[6]};
    motion.v = {s[7], s[8], s[9]};
    motion.bg = {s[10], s[11], s[12]};
    motion.ba = {s[13], s[14], s[15]};
}

// ---------------------------------------------------------------------------------------------- Track
void Track::add_keypoint(Frame *frame, size_t keypoint_index) {
    refs[frame->id()] = {frame, keypoint_index};
    frame->tracks[keypoint_index] = this;
    frame->tangents[keypoint_index] = tangent_frame(frame->bearings[keypoint_index]);  // create_reprojection_error_factor
    if (tag(TT_TRIANGULATED)) m_life++;
    else m_life = 1;
}

void Track::remove_keypoint(Frame *frame, bool suicide_if_empty) {
    const size_t keypoint_index = refs.at(frame->id()).second;
    std::optional<V3> landmark;
    if (frame == first_frame()) landmark = get_landmark_point();
    frame->tracks[keypoint_index] = nullptr;
    refs.erase(frame->id());
    if (!refs.empty()) {
        if (landmark.has_value()) set_landmark_point(landmark.value());
    } else {
        set_tag(TT_VALID, false);
        if (suicide_if_empty) map->recycle_track(this);
    }
}

std::optional<V3> Track::triangulate() {
    std::vector<std::array<double, 12>> Ps;
    std::vector<V3> ps;
    for (const auto &kv : refs) {
        Frame *frame = kv.second.first;
        const PoseState pose = frame->get_pose(frame->camera);
        const M3 R = to_mat(conj(pose.q));
        const V3 T = -(R * pose.p);
        Ps.push_back({R.m[0], R.m[1], R.m[2], T.x, R.m[3], R.m[4], R.m[5], T.y, R.m[6], R.m[7], R.m[8], T.z});
        ps.push_back(frame->get_keypoint(kv.second.second));
    }
    const std::array<double, 4> h = triangulate_point(Ps, ps);
    for (size_t i = 0; i < ps.size(); ++i) {
        const double *P = Ps[i].data();
        const double qz = P[8] * h[0] + P[9] * h[1] + P[10] * h[2] + P[11] * h[3];
        if (!(qz * h[3] > 0)) return {};
    }
    m_life = 1;
    return V3{h[0] / h[3], h[1] / h[3], h[2] / h[3]};
}

V3 Track::get_landmark_point() const {
    const auto [frame, keypoint_index] = first_keypoint();
    const PoseState camera = frame->get_pose(frame->camera);
    return rot(camera.q, frame->get_keypoint(keypoint_index)) / inv_depth + camera.p;
}

void Track::set_landmark_point(const V3 &p) {
    const auto [frame, keypoint_index] = first_keypoint();
    (void)keypoint_index;
    const PoseState camera = frame->get_pose(frame->camera);
    inv_depth = 1.0 / norm(rot(conj(camera.q), p - camera.p));
}

// ---------------------------------------------------------------------------------------------- Map
Map::~Map() {
    // tracks reference frames and vice versa; drop the links before either container goes away
    tracks.clear();
    frames.clear();
}

void Map::attach_frame(std::unique_ptr<Frame> frame, size_t position) {
    frame->map = this;
    if (position == nil) frames.emplace_back(std::move(frame));
    else frames.emplace(frames.begin() + (std::ptrdiff_t)position, std::move(frame));
}

std::unique_ptr<Frame> Map::detach_frame(size_t index) {
    std::unique_ptr<Frame> frame = std::move(frames[index]);
    frames.erase(frames.begin() + (std::ptrdiff_t)index);
    frame->map = nullptr;
    return frame;
}

void Map::untrack_frame(Frame *frame) {
    for (size_t i = 0; i < frame->keypoint_num(); ++i)
        if (Track *track = frame->get_track(i)) track->remove_keypoint(frame);
}

void Map::erase_frame(size_t index) {
    untrack_frame(frames[index].get());
    detach_frame(index);
}

void Map::drop_front_frame() { frames.erase(frames.begin()); }

size_t Map::frame_index_by_id(size_t id) const {
    auto it = std::lower_bound(frames.begin(), frames.end(), id,
                               [](const std::unique_ptr<Frame> &f, size_t v) { return f->id() < v; });
    if (it == frames.end()) return nil;
    if (id < (*it)->id()) return nil;
    return (size_t)std::distance(frames.begin(), it);
}

Track *Map::create_track() {
    std::unique_ptr<Track> track = reserved_track_ids[0] < reserved_track_ids[1] ? std::make_unique<Track>(reserved_track_ids[0]++, this)
                                                                                 : std::make_unique<Track>(ids, this);
    track->map_index = tracks.size();
    tracks.emplace_back(std::move(track));
    return tracks.back().get();
}

void Map::erase_track(Track *track) {
    while (track->keypoint_num() > 0) track->remove_keypoint(track->keypoint_map().begin()->second.first, false);
    recycle_track(track);
}

void Map::prune_tracks(const std::function<bool(const Track *)> &condition) {
    std::vector<Track *> doomed;
    for (size_t i = 0; i < track_num(); ++i)
        if (Track *track = get_track(i); condition(track)) doomed.push_back(track);
    for (Track *track : doomed) erase_track(track);
}

void Map::recycle_track(Track *track) {
    if (track->map_index != tracks.back()->map_index) {
        tracks[track->map_index].swap(tracks.back());
        tracks[track->map_index]->map_index = track->map_index;
    }
    tracks.pop_back();
}

}  // namespace rdvio_pipe
