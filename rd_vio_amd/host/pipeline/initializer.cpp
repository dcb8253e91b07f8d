// Initializer (/root/reference/src/rdvio/src/initializer.cpp): keyframe mirroring (:20-70), vision-only SfM (:142-366),
// visual-inertial alignment (:368-560) and the closing BA that hands the window to the SlidingWindowTracker (:72-140).
// When bootstrap states were supplied (rdvio_pipeline_set_init_states) they replace the SfM / IMU-alignment stages.
#include <unordered_set>

#include "pipeline.hpp"

namespace rdvio_pipe {

void Initializer::mirror_keyframe_map(Map *ftmap, size_t init_frame_id) {
    const size_t last = ftmap->frame_index_by_id(init_frame_id);
    const size_t gap = (size_t)sh.cfg.initializer_keyframe_gap;
    const size_t distance = gap * ((size_t)sh.cfg.initializer_keyframe_num - 1);
    if (last == nil || last < distance) {
        map.reset();
        return;
    }
    const size_t first = last - distance;
    std::vector<size_t> indices;
    for (size_t i = 0; i < (size_t)sh.cfg.initializer_keyframe_num; ++i) indices.push_back(first + i * gap);
    map = std::make_unique<Map>(sh.ids);
    for (size_t index : indices) map->attach_frame(ftmap->get_frame(index)->clone());
    for (size_t j = 1; j < map->frame_num(); ++j) {
        Frame *old_i = ftmap->get_frame(indices[j - 1]), *old_j = ftmap->get_frame(indices[j]);
        Frame *new_i = map->get_frame(j - 1), *new_j = map->get_frame(j);
        for (size_t ki = 0; ki < old_i->keypoint_num(); ++ki)
            if (Track *track = old_i->get_track(ki))
                if (size_t kj = track->get_keypoint_index(old_j); kj != nil) new_i->get_track(ki, nullptr)->add_keypoint(new_j, kj);
        new_j->preintegration.data.clear();
        for (size_t f = indices[j - 1]; f < indices[j]; ++f) {
            const std::vector<ImuData> &old_data = ftmap->get_frame(f + 1)->preintegration.data;
            new_j->preintegration.data.insert(new_j->preintegration.data.end(), old_data.begin(), old_data.end());
        }
    }
}

bool Initializer::bootstrap_from_supplied_states() {
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        const std::array<double, 17> *row = nullptr;
        for (const auto &r : sh.init_states)
            if (std::fabs(r[0] - frame->image->t) < 1.0e-6) { row = &r; break; }
        if (!row) return false;
        frame->set_state(row->data() + 1);
    }
    // triangulate every track of the keyframe map (initializer.cpp:305-315), drop the failures (:361-364)
    for (size_t i = 0; i < map->track_num(); ++i) {
        Track *track = map->get_track(i);
        if (track->tag(TT_VALID)) continue;
        if (auto p = track->triangulate()) {
            track->set_landmark_point(p.value());
            track->set_tag(TT_VALID, true);
            track->set_tag(TT_TRIANGULATED, true);
        }
    }
    map->prune_tracks([](const Track *track) { return !track->tag(TT_VALID); });
    return true;
}

std::unique_ptr<SlidingWindowTracker> Initializer::initialize() {
    if (!map) return nullptr;
    if (!sh.init_states.empty()) {
        if (!bootstrap_from_supplied_states()) return nullptr;
    } else {
        if (!init_sfm()) return nullptr;
        if (!init_imu()) return nullptr;
    }
    // closing visual-inertial BA (initializer.cpp:82-127)
    map->get_frame(0)->set_tag(FT_FIX_POSE, true);
    BaBuilder solver(sh);
    for (size_t i = 0; i < map->frame_num(); ++i) solver.add_frame_states(map->get_frame(i));
    std::unordered_set<Track *> visited;
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        for (size_t j = 0; j < frame->keypoint_num(); ++j) {
            Track *track = frame->get_track(j);
            if (!track || !track->tag(TT_VALID) || visited.count(track)) continue;
            visited.insert(track);
            solver.add_track_states(track, false);
        }
    }
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        for (size_t j = 0; j < frame->keypoint_num(); ++j) {
            Track *track = frame->get_track(j);
            if (!track || !track->all_tagged({TT_VALID, TT_TRIANGULATED}) || frame == track->first_frame()) continue;
            solver.add_reprojection_error(frame, j);
        }
    }
    for (size_t j = 1; j < map->frame_num(); ++j) {
        Frame *frame_i = map->get_frame(j - 1), *frame_j = map->get_frame(j);
        if (frame_j->preintegration.integrate(sh.backend, LANE_ESTIMATOR, frame_j->image->t, frame_i->motion.bg, frame_i->motion.ba, true, true))
            solver.add_preintegration(frame_i, frame_j, frame_j->preintegration, false);
    }
    solver.solve();
    for (size_t i = 0; i < map->frame_num(); ++i) map->get_frame(i)->set_tag(FT_KEYFRAME, true);
    return std::make_unique<SlidingWindowTracker>(std::move(map), sh);
}

// ---------------------------------------------------------------------------------------------------------------------
// vision-only structure from motion between the first and the last keyframe (initializer.cpp:142-366)
// ---------------------------------------------------------------------------------------------------------------------
bool Initializer::init_sfm() {
    Frame *init_frame_i = map->get_frame(0), *init_frame_j = map->get_frame(map->frame_num() - 1);
    double total_parallax = 0;
    int common = 0;
    std::vector<std::pair<size_t, size_t>> matches;
    std::vector<V2> pi, pj;
    for (size_t ki = 0; ki < init_frame_i->keypoint_num(); ++ki) {
        Track *track = init_frame_i->get_track(ki);
        if (!track) continue;
        const size_t kj = track->get_keypoint_index(init_frame_j);
        if (kj == nil) continue;
        pi.push_back(hnormalized(init_frame_i->get_keypoint(ki)));
        pj.push_back(hnormalized(init_frame_j->get_keypoint(kj)));
        matches.emplace_back(ki, kj);
        total_parallax += norm(apply_k(init_frame_i->get_keypoint(ki), init_frame_i->K) - apply_k(init_frame_j->get_keypoint(kj), init_frame_j->K));
        common++;
    }
    if (common < sh.cfg.initializer_min_matches) return false;
    total_parallax /= std::max(common, 1);
    if (total_parallax < sh.cfg.initializer_min_parallax) return false;

    std::vector<M3> Rs;
    std::vector<V3> Ts;
    std::vector<char> mask;
    M3 RH1, RH2;
    V3 TH1, TH2, nH1, nH2;
    const double thr = 0.7 / init_frame_i->K[0];
    const M3 H = find_homography_matrix(pi, pj, mask, thr, 0.999, 1000, sh.cfg.random);
    if (!decompose_homography(H, RH1, RH2, TH1, TH2, nH1, nH2)) return false;  // pure rotation
    TH1 = normalized(TH1);
    TH2 = normalized(TH2);
    Rs.insert(Rs.end(), {RH1, RH1, RH2, RH2});
    Ts.insert(Ts.end(), {TH1, -TH1, TH2, -TH2});
    M3 RE1, RE2;
    V3 TE;
    const M3 E = find_essential_matrix(pi, pj, mask, thr, 0.999, 1000, sh.cfg.random);
    decompose_essential(E, RE1, RE2, TE);
    TE = normalized(TE);
    Rs.insert(Rs.end(), {RE1, RE1, RE2, RE2});
    Ts.insert(Ts.end(), {TE, -TE, TE, -TE});

    // [1.1] triangulate with every candidate, keep the best (:207-262)
    std::vector<std::vector<V3>> points(Rs.size());
    std::vector<std::vector<char>> status(Rs.size());
    std::vector<size_t> counts(Rs.size(), 0);
    std::vector<double> scores(Rs.size(), 0.0);
    size_t best = 0;
    for (size_t i = 0; i < Rs.size(); ++i) {
        points[i].resize(pi.size());
        status[i].assign(pi.size(), 0);
        const std::array<double, 12> P1{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        const M3 &R = Rs[i];
        const std::array<double, 12> P2{R.m[0], R.m[1], R.m[2], Ts[i].x, R.m[3], R.m[4], R.m[5], Ts[i].y, R.m[6], R.m[7], R.m[8], Ts[i].z};
        for (size_t j = 0; j < pi.size(); ++j) {
            const std::array<double, 4> q = triangulate_point({P1, P2}, {V3{pi[j].x, pi[j].y, 1.0}, V3{pj[j].x, pj[j].y, 1.0}});
            const V3 q1{q[0], q[1], q[2]};
            const V3 q2 = R * q1 + q[3] * Ts[i];
            if (q1.z * q[3] > 0 && q2.z * q[3] > 0 && q1.z / q[3] < 100 && q2.z / q[3] < 100) {
                points[i][j] = q1 / q[3];
                status[i][j] = 1;
                counts[i]++;
                scores[i] += 0.5 * (sqnorm(hnormalized(q1) - pi[j]) + sqnorm(hnormalized(q2) - pj[j]));
            }
        }
        if (counts[i] > (size_t)sh.cfg.initializer_min_triangulation && scores[i] < scores[best]) best = i;
        else if (counts[i] > counts[best]) best = i;
    }
    const M3 init_R = Rs[best];
    const V3 init_T = Ts[best];
    if (counts[best] < (size_t)sh.cfg.initializer_min_triangulation) return false;

    // [2] sfm map: first frame at the origin, last frame from (R, T) (:271-289)
    PoseState pose;
    init_frame_i->set_pose(init_frame_i->camera, pose);
    pose.q = from_mat(transpose(init_R));
    pose.p = -(transpose(init_R) * init_T);
    init_frame_j->set_pose(init_frame_j->camera, pose);
    for (size_t k = 0; k < points[best].size(); ++k) {
        if (!status[best][k]) continue;
        Track *track = init_frame_i->get_track(matches[k].first);
        track->set_landmark_point(points[best][k]);
        track->set_tag(TT_VALID, true);
        track->set_tag(TT_TRIANGULATED, true);
    }
    // [2.2] the frames in between by PnP-style solves against the triangulated points (:291-309)
    for (size_t j = 1; j + 1 < map->frame_num(); ++j) {
        Frame *frame_i = map->get_frame(j - 1), *frame_j = map->get_frame(j);
        frame_j->set_pose(frame_j->camera, frame_i->get_pose(frame_i->camera));
        BaBuilder solver(sh);
        solver.add_frame_states(frame_j);
        for (size_t k = 0; k < frame_j->keypoint_num(); ++k) {
            Track *track = frame_j->get_track(k);
            if (!track || !track->has_keypoint(map->get_frame(0))) continue;
            if (track->tag(TT_VALID) && track->tag(TT_TRIANGULATED)) solver.add_reprojection_prior(frame_j, track);
        }
        solver.solve();
    }
    // [2.3] triangulate more points (:311-321)
    for (size_t i = 0; i < map->track_num(); ++i) {
        Track *track = map->get_track(i);
        if (track->tag(TT_VALID)) continue;
        if (auto p = track->triangulate()) {
            track->set_landmark_point(p.value());
            track->set_tag(TT_VALID, true);
            track->set_tag(TT_TRIANGULATED, true);
        }
    }
    // [3.1] vision-only bundle adjustment, first pose fixed (:325-358)
    map->get_frame(0)->set_tag(FT_FIX_POSE, true);
    BaBuilder solver(sh);
    for (size_t i = 0; i < map->frame_num(); ++i) solver.add_frame_states(map->get_frame(i), false);
    std::unordered_set<Track *> visited;
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        for (size_t j = 0; j < frame->keypoint_num(); ++j) {
            Track *track = frame->get_track(j);
            if (!track || !track->tag(TT_VALID) || visited.count(track)) continue;
            visited.insert(track);
            solver.add_track_states(track, false);
        }
    }
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        for (size_t j = 0; j < frame->keypoint_num(); ++j) {
            Track *track = frame->get_track(j);
            if (!track || !track->all_tagged({TT_VALID, TT_TRIANGULATED}) || frame == track->first_frame()) continue;
            solver.add_reprojection_error(frame, j);
        }
    }
    if (!solver.solve()) return false;
    // [3.2] (landmark.reprojection_error is never written by the reference, so only the validity test acts, :361-364)
    map->prune_tracks([](const Track *track) { return !track->tag(TT_VALID); });
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// visual-inertial alignment (initializer.cpp:368-560)
// ---------------------------------------------------------------------------------------------------------------------
bool Initializer::init_imu() {
    reset_states();
    solve_gyro_bias();
    solve_gravity_scale_velocity();
    if (scale < 0.001 || scale > 1.0) return false;
    if (sh.cfg.initializer_refine_imu) {   // initializer.cpp:373 (true by default, config.cpp:52)
        refine_scale_velocity_via_gravity();
        if (scale < 0.001 || scale > 1.0) return false;
    }
    return apply_init();
}

void Initializer::reset_states() {
    bg = ba = gravity = V3{0, 0, 0};
    scale = 1;
    velocities.assign(map->frame_num(), V3{0, 0, 0});
}

void Initializer::preintegrate() {
    std::vector<PreIntegrator::Job> jobs;
    for (size_t j = 1; j < map->frame_num(); ++j) jobs.push_back({&map->get_frame(j)->preintegration, map->get_frame(j)->image->t, bg, ba});
    (void)PreIntegrator::integrate_batch(sh.backend, LANE_ESTIMATOR, jobs, true, false);
}

void Initializer::solve_gyro_bias() {
    preintegrate();
    std::vector<double> A(9, 0.0), b(3, 0.0);
    for (size_t j = 1; j < map->frame_num(); ++j) {
        const Frame *frame_i = map->get_frame(j - 1), *frame_j = map->get_frame(j);
        const PoseState pose_i = frame_i->get_pose(frame_i->imu), pose_j = frame_j->get_pose(frame_j->imu);
        const Q4 dq = frame_j->preintegration.dq();
        const double *J = frame_j->preintegration.delta.data() + PRE_JAC;  // dq_dbg, row-major 3x3
        const V3 r = logmap(conj(pose_i.q * dq) * pose_j.q);
        const double rv[3] = {r.x, r.y, r.z};
        for (int a = 0; a < 3; ++a) {
            for (int c = 0; c < 3; ++c) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += J[3 * k + a] * J[3 * k + c];
                A[3 * a + c] += s;
            }
            for (int k = 0; k < 3; ++k) b[a] += J[3 * k + a] * rv[k];
        }
    }
    const std::vector<double> x = least_squares(3, 3, A, b);
    bg = {x[0], x[1], x[2]};
}

void Initializer::solve_gravity_scale_velocity() {
    preintegrate();
    const int N = (int)map->frame_num(), rows = (N - 1) * 6, cols = 3 + 1 + 3 * N;
    std::vector<double> A((size_t)rows * cols, 0.0), b(rows, 0.0);
    auto at = [&](int r, int c) -> double & { return A[(size_t)r * cols + c]; };
    for (int j = 1; j < N; ++j) {
        const int i = j - 1;
        const Frame *frame_i = map->get_frame(i), *frame_j = map->get_frame(j);
        const PreIntegrator &d = frame_j->preintegration;
        const PoseState ci = frame_i->get_pose(frame_i->camera), cj = frame_j->get_pose(frame_j->camera);
        const double dt = d.dt();
        const V3 dpos = cj.p - ci.p;
        const V3 b0 = rot(frame_i->pose.q, d.dp()) + (rot(frame_j->pose.q, frame_j->camera.p_cs) - rot(frame_i->pose.q, frame_i->camera.p_cs));
        const V3 b1 = rot(frame_i->pose.q, d.dv());
        const double dp3[3] = {dpos.x, dpos.y, dpos.z}, b03[3] = {b0.x, b0.y, b0.z}, b13[3] = {b1.x, b1.y, b1.z};
        for (int k = 0; k < 3; ++k) {
            at(i * 6 + k, k) = -0.5 * dt * dt;
            at(i * 6 + k, 3) = dp3[k];
            at(i * 6 + k, 4 + i * 3 + k) = -dt;
            b[i * 6 + k] = b03[k];
            at(i * 6 + 3 + k, k) = -dt;
            at(i * 6 + 3 + k, 4 + i * 3 + k) = -1.0;
            at(i * 6 + 3 + k, 4 + j * 3 + k) = 1.0;
            b[i * 6 + 3 + k] = b13[k];
        }
    }
    const std::vector<double> x = least_squares(rows, cols, A, b);
    gravity = normalized(V3{x[0], x[1], x[2]}) * GRAVITY_NOMINAL;
    scale = x[3];
    for (int i = 0; i < N; ++i) velocities[i] = {x[4 + 3 * i], x[5 + 3 * i], x[6 + 3 * i]};
}

void Initializer::refine_scale_velocity_via_gravity() {
    const double damp = 0.1;
    preintegrate();
    const int N = (int)map->frame_num(), rows = (N - 1) * 6, cols = 2 + 1 + 3 * N;
    std::vector<double> A((size_t)rows * cols), b(rows), x;
    auto at = [&](int r, int c) -> double & { return A[(size_t)r * cols + c]; };
    for (int iter = 0; iter < 1; ++iter) {
        std::fill(A.begin(), A.end(), 0.0);
        std::fill(b.begin(), b.end(), 0.0);
        const std::array<double, 9> tf = tangent_frame(gravity);  // columns b1 b2 (z): s2_tangential_basis(gravity)
        const double Tg[3][2] = {{tf[0], tf[1]}, {tf[3], tf[4]}, {tf[6], tf[7]}};
        const double g3[3] = {gravity.x, gravity.y, gravity.z};
        for (int j = 1; j < N; ++j) {
            const int i = j - 1;
            const Frame *frame_i = map->get_frame(i), *frame_j = map->get_frame(j);
            const PreIntegrator &d = frame_j->preintegration;
            const PoseState ci = frame_i->get_pose(frame_i->camera), cj = frame_j->get_pose(frame_j->camera);
            const double dt = d.dt();
            const V3 dpos = cj.p - ci.p;
            const V3 b0 = rot(frame_i->pose.q, d.dp()) + (rot(frame_j->pose.q, frame_j->camera.p_cs) - rot(frame_i->pose.q, frame_i->camera.p_cs));
            const V3 b1 = rot(frame_i->pose.q, d.dv());
            const double dp3[3] = {dpos.x, dpos.y, dpos.z}, b03[3] = {b0.x, b0.y, b0.z}, b13[3] = {b1.x, b1.y, b1.z};
            for (int k = 0; k < 3; ++k) {
                for (int c = 0; c < 2; ++c) {
                    at(i * 6 + k, c) = -0.5 * dt * dt * Tg[k][c];
                    at(i * 6 + 3 + k, c) = -dt * Tg[k][c];
                }
                at(i * 6 + k, 2) = dp3[k];
                at(i * 6 + k, 3 + i * 3 + k) = -dt;
                b[i * 6 + k] = 0.5 * dt * dt * g3[k] + b03[k];
                at(i * 6 + 3 + k, 3 + i * 3 + k) = -1.0;
                at(i * 6 + 3 + k, 3 + j * 3 + k) = 1.0;
                b[i * 6 + 3 + k] = dt * g3[k] + b13[k];
            }
        }
        x = least_squares(rows, cols, A, b);
        const V3 step{damp * (Tg[0][0] * x[0] + Tg[0][1] * x[1]), damp * (Tg[1][0] * x[0] + Tg[1][1] * x[1]), damp * (Tg[2][0] * x[0] + Tg[2][1] * x[1])};
        gravity = normalized(gravity + step) * GRAVITY_NOMINAL;
    }
    scale = x[2];
    for (int i = 0; i < N; ++i) velocities[i] = {x[3 + 3 * i], x[4 + 3 * i], x[5 + 3 * i]};
}

bool Initializer::apply_init(bool apply_ba, bool apply_velocity) {
    const V3 gravity_nominal{0, 0, -GRAVITY_NOMINAL};
    const Q4 q = from_two_vectors(gravity, gravity_nominal);
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        PoseState imu_pose = frame->get_pose(frame->imu);
        imu_pose.q = q * imu_pose.q;
        imu_pose.p = scale * rot(q, imu_pose.p);
        frame->set_pose(frame->imu, imu_pose);
        frame->motion.v = apply_velocity ? rot(q, velocities[i]) : V3{0, 0, 0};
        frame->motion.bg = bg;
        frame->motion.ba = apply_ba ? ba : V3{0, 0, 0};
    }
    size_t final_point_num = 0;
    for (size_t i = 0; i < map->track_num(); ++i) {
        Track *track = map->get_track(i);
        if (auto p = track->triangulate()) {
            track->set_landmark_point(p.value());
            track->set_tag(TT_VALID, true);
            track->set_tag(TT_TRIANGULATED, true);
            final_point_num++;
        } else {
            track->set_tag(TT_VALID, false);
        }
    }
    return final_point_num >= (size_t)sh.cfg.initializer_min_landmarks;
}

}  // namespace rdvio_pipe
