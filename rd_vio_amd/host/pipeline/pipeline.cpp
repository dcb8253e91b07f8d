// Per-frame orchestration around the HIP hot path.  Each block cites the reference code whose behaviour it keeps;
// everything data-parallel is a call through the backend table (include/rdvio_pipeline.h).
#include "pipeline.hpp"
#include "parsac.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <unordered_set>

namespace rdvio_pipe {

namespace {
V2 v2(const double *p) { return {p[0], p[1]}; }
}  // namespace

// =====================================================================================================================
// BaBuilder: Solver::add_* -> SoA (solver.cpp:88-178), Solver::solve (solver.cpp:180-194)
// =====================================================================================================================
int BaBuilder::frame_index(Frame *frame) const {
    auto it = fidx.find(frame);
    return it == fidx.end() ? -1 : it->second;
}

int BaBuilder::add_frame_states(Frame *frame, bool with_motion) {
    if (int i = frame_index(frame); i >= 0) return i;
    // FT_FIX_POSE & FT_FIX_MOTION -> constant; FT_FIX_POSE alone -> pose constant, motion free (solver.cpp:92-113).
    // with_motion = false (the SfM bundle adjustment, initializer.cpp:321): the motion blocks are not parameters; no factor
    // of such a solve touches them, so their columns stay empty (zero gradient, zero step) -- a pose-constant frame is
    // then entirely constant.
    uint8_t fixed = 0;
    if (frame->tag(FT_FIX_POSE)) fixed = (frame->tag(FT_FIX_MOTION) || !with_motion) ? 1 : 2;
    fidx[frame] = (int)frames.size();
    frames.push_back(frame);
    frame_fixed.push_back(fixed);
    return (int)frames.size() - 1;
}

int BaBuilder::add_constant_frame(Frame *frame) {
    // a frame that a "prior" factor reads but does not differentiate.  If the frame is also a free parameter of this
    // solve, the factor sees a constant copy of its value at the start of the solve (the reference reads it live
    // through a pointer; the two differ only by that frame's own update during the solve).
    if (int i = frame_index(frame); i >= 0 && frame_fixed[i] == 1) return i;
    for (size_t i = 0; i < frames.size(); ++i)
        if (frames[i] == frame && frame_fixed[i] == 1) return (int)i;
    if (frame_index(frame) < 0) fidx[frame] = (int)frames.size();
    frames.push_back(frame);
    frame_fixed.push_back(1);
    return (int)frames.size() - 1;
}

int BaBuilder::add_track_states(Track *track, bool constant) {
    if (track->ba_stamp == stamp) return track->ba_index;
    track->ba_stamp = stamp;
    track->ba_index = (int)lms.size();
    lms.push_back(track);
    lm_fixed.push_back(constant ? 1 : 0);
    return (int)lms.size() - 1;
}

void BaBuilder::add_reprojection_error(Frame *frame, size_t keypoint_index) {
    Track *track = frame->get_track(keypoint_index);
    Frame *ref = track->first_frame();
    int t = frame_index(frame), r = frame_index(ref);
    if (t < 0) t = add_constant_frame(frame);
    if (r < 0) r = add_constant_frame(ref);
    const int l = add_track_states(track, false);
    facs.push_back({t, r, l, frame->tangents[keypoint_index].data()});
}

void BaBuilder::add_reprojection_prior(Frame *frame, Track *track) { add_reprojection_prior(frame, track, track->get_keypoint_index(frame)); }

void BaBuilder::add_reprojection_prior(Frame *frame, Track *track, size_t keypoint_index) {
    // CeresReprojectionPriorFactor (reprojection_factor.h:99-121): anchor pose and inverse depth held constant
    int t = frame_index(frame);
    if (t < 0) t = add_constant_frame(frame);
    const int r = add_constant_frame(track->first_frame());
    const int l = add_track_states(track, true);
    facs.push_back({t, r, l, frame->tangents[keypoint_index].data()});
}

void BaBuilder::add_rotation_prior(Frame *frame, Track *track) {
    int t = frame_index(frame);
    if (t < 0) t = add_constant_frame(frame);
    const auto [ref, ref_kp] = track->first_keypoint();
    const int r = add_constant_frame(ref);
    rots.push_back({t, r, ref->get_keypoint(ref_kp), frame->tangents[track->get_keypoint_index(frame)].data()});
}

void BaBuilder::add_preintegration(Frame *frame_i, Frame *frame_j, const PreIntegrator &pre, bool is_prior) {
    int i = is_prior ? add_constant_frame(frame_i) : frame_index(frame_i);
    if (i < 0) i = add_constant_frame(frame_i);
    int j = frame_index(frame_j);
    if (j < 0) j = add_constant_frame(frame_j);
    pres.push_back({i, j, pre.delta.data(), nullptr, 0.0, V3{}, V3{}});
}

bool BaBuilder::add_integrated_preintegration(Frame *frame_i, Frame *frame_j, PreIntegrator &pre, double t, const V3 &bg, const V3 &ba) {
    if (pre.data.empty()) return false;
    int i = frame_index(frame_i);
    if (i < 0) i = add_constant_frame(frame_i);
    int j = frame_index(frame_j);
    if (j < 0) j = add_constant_frame(frame_j);
    pres.push_back({i, j, pre.delta.data(), &pre, t, bg, ba});
    return true;
}

// everything Solver::solve hands to the backend, packed once (the vectors live until the result has been applied)
struct BaBuilder::Packed {
    int nfr = 0, nl = 0, nf = 0, nrot = 0, npre = 0, njobs = 0;
    std::vector<double> states, invd, zref, tangent, rot_zref, rot_tangent, preint, job_imu, job_par, states_out, invd_out;
    std::vector<int32_t> tgt, ref, lm, rot_tgt, rot_ref, pre_i, pre_j, job_off, prior_frames;
    double extr[14];
    rdvio_ba_problem pb;
    rdvio_ba_summary sm;
    int slot = -1;   // >= 0: begun behind the backend, waiting for solve_end
    std::chrono::steady_clock::time_point t0;
    double seconds = 0.0;
};

BaBuilder::BaBuilder(Shared &sh) : sh(sh), stamp(sh.ba_stamps.fetch_add(1, std::memory_order_relaxed) + 1) {}
BaBuilder::~BaBuilder() = default;

void BaBuilder::pack() {
    packed = std::make_unique<Packed>();
    Packed &P = *packed;
    const int nfr = P.nfr = (int)frames.size(), nl = P.nl = (int)lms.size();
    P.states.resize((size_t)nfr * 16);
    P.invd.resize((size_t)std::max(nl, 1));
    P.zref.resize((size_t)std::max(nl, 1) * 3);
    for (int i = 0; i < nfr; ++i) frames[i]->get_state(&P.states[16 * (size_t)i]);
    for (int l = 0; l < nl; ++l) {
        P.invd[l] = lms[l]->inv_depth;
        const auto [ref, kp] = lms[l]->first_keypoint();
        const V3 &z = ref->get_keypoint(kp);
        P.zref[3 * l] = z.x; P.zref[3 * l + 1] = z.y; P.zref[3 * l + 2] = z.z;
    }
    std::stable_sort(facs.begin(), facs.end(), [](const Fac &a, const Fac &b) { return a.lm < b.lm; });
    const int nf = P.nf = (int)facs.size(), nrot = P.nrot = (int)rots.size(), npre = P.npre = (int)pres.size();
    P.tgt.resize(std::max(nf, 1)); P.ref.resize(std::max(nf, 1)); P.lm.resize(std::max(nf, 1));
    P.tangent.resize((size_t)std::max(nf, 1) * 9);
    for (int k = 0; k < nf; ++k) {
        P.tgt[k] = facs[k].tgt; P.ref[k] = facs[k].ref; P.lm[k] = facs[k].lm;
        std::copy(facs[k].tangent, facs[k].tangent + 9, &P.tangent[9 * (size_t)k]);
    }
    P.rot_tgt.resize(std::max(nrot, 1)); P.rot_ref.resize(std::max(nrot, 1));
    P.rot_zref.resize((size_t)std::max(nrot, 1) * 3); P.rot_tangent.resize((size_t)std::max(nrot, 1) * 9);
    for (int k = 0; k < nrot; ++k) {
        P.rot_tgt[k] = rots[k].tgt; P.rot_ref[k] = rots[k].ref;
        P.rot_zref[3 * k] = rots[k].zref.x; P.rot_zref[3 * k + 1] = rots[k].zref.y; P.rot_zref[3 * k + 2] = rots[k].zref.z;
        std::copy(rots[k].tangent, rots[k].tangent + 9, &P.rot_tangent[9 * (size_t)k]);
    }
    // integrations that ride inside the solve call: only when EVERY preintegration factor asks for it (the records then fill
    // the solve's slots in order); a mixed solve integrates through the separate entry first
    int njobs = 0;
    for (const Pre &p : pres) njobs += p.job != nullptr;
    if (njobs > 0 && njobs != npre) {
        std::vector<PreIntegrator::Job> jobs;
        for (const Pre &p : pres)
            if (p.job) jobs.push_back({p.job, p.t, p.bg, p.ba});
        (void)PreIntegrator::integrate_batch(sh.backend, LANE_ESTIMATOR, jobs, true, true);
        njobs = 0;
    }
    P.njobs = njobs;
    P.pre_i.resize(std::max(npre, 1)); P.pre_j.resize(std::max(npre, 1));
    P.preint.resize((size_t)std::max(npre, 1) * RDVIO_PREINT_SIZE);
    P.job_off.assign(1, 0);
    for (int k = 0; k < npre; ++k) {
        P.pre_i[k] = pres[k].i; P.pre_j[k] = pres[k].j;
        if (njobs > 0) {
            for (const ImuData &d : pres[k].job->data) P.job_imu.insert(P.job_imu.end(), {d.t, d.w.x, d.w.y, d.w.z, d.a.x, d.a.y, d.a.z});
            P.job_off.push_back((int32_t)(P.job_imu.size() / 7));
            P.job_par.insert(P.job_par.end(), {pres[k].t, pres[k].bg.x, pres[k].bg.y, pres[k].bg.z, pres[k].ba.x, pres[k].ba.y, pres[k].ba.z});
        } else {
            std::copy(pres[k].delta, pres[k].delta + RDVIO_PREINT_SIZE, &P.preint[(size_t)k * RDVIO_PREINT_SIZE]);
        }
    }
    if (prior) const_cast<MarginalizationPrior *>(prior)->ready(sh.backend);   // a marginalisation begun frames ago lands here
    if (prior)
        for (Frame *f : prior->frames) {
            const int i = frame_index(f);
            if (i < 0) throw std::runtime_error("marginalisation prior covers a frame that is not a state of the solve");
            P.prior_frames.push_back(i);
        }
    double *extr = P.extr;
    const Frame *f0 = frames[0];
    extr[0] = f0->camera.q_cs.x; extr[1] = f0->camera.q_cs.y; extr[2] = f0->camera.q_cs.z; extr[3] = f0->camera.q_cs.w;
    extr[4] = f0->camera.p_cs.x; extr[5] = f0->camera.p_cs.y; extr[6] = f0->camera.p_cs.z;
    extr[7] = f0->imu.q_cs.x; extr[8] = f0->imu.q_cs.y; extr[9] = f0->imu.q_cs.z; extr[10] = f0->imu.q_cs.w;
    extr[11] = f0->imu.p_cs.x; extr[12] = f0->imu.p_cs.y; extr[13] = f0->imu.p_cs.z;

    rdvio_ba_problem &pb = P.pb;
    std::memset(&pb, 0, sizeof pb);
    pb.n_frames = nfr;
    pb.states = P.states.data();
    pb.frame_fixed = frame_fixed.data();
    pb.extr = extr;
    pb.sqrt_inv_cov = f0->sqrt_inv_cov;
    pb.n_landmarks = nl;
    pb.z_ref = P.zref.data();
    pb.inv_depth = P.invd.data();
    pb.lm_fixed = lm_fixed.empty() ? reinterpret_cast<const uint8_t *>("") : lm_fixed.data();
    pb.n_factors = nf;
    pb.tgt = P.tgt.data(); pb.ref = P.ref.data(); pb.lm = P.lm.data();
    pb.tangent = P.tangent.data();
    pb.n_rot = nrot;
    pb.rot_tgt = P.rot_tgt.data(); pb.rot_ref = P.rot_ref.data();
    pb.rot_zref = P.rot_zref.data(); pb.rot_tangent = P.rot_tangent.data();
    pb.n_preint = npre;
    pb.pre_i = P.pre_i.data(); pb.pre_j = P.pre_j.data(); pb.preint = P.preint.data();
    if (njobs > 0) {
        pb.n_pre_jobs = njobs;
        pb.job_seg_off = P.job_off.data();
        pb.job_imu = P.job_imu.data();
        pb.job_par = P.job_par.data();
        pb.job_noise = pres[0].job->noise;
        pb.job_preint_out = P.preint.data();
    }
    pb.n_prior = (int)P.prior_frames.size();
    if (prior) {
        pb.prior_frames = P.prior_frames.data();
        pb.prior_lin = prior->lin.data();
        pb.prior_S = prior->S.data();
        pb.prior_f = prior->f.data();
    }
    sh.counters.max_problem_frames = std::max<int64_t>(sh.counters.max_problem_frames, nfr);
    sh.counters.max_problem_factors = std::max<int64_t>(sh.counters.max_problem_factors, nf);
    sh.counters.rotation_prior_factors += nrot;
    P.states_out.resize(P.states.size());
    P.invd_out.resize(P.invd.size());
    std::memset(&P.sm, 0, sizeof P.sm);
}

// the result back into the Frame / Track members, the integrators' records, the counters
bool BaBuilder::apply(rdvio_ba_summary *summary_out) {
    Packed &P = *packed;
    const rdvio_ba_summary sm = P.sm;
    if (sh.prof.on) {
        SolveProf &sp = sh.solve_prof;
        sp.calls[kind]++; sp.iterations[kind] += sm.iterations; sp.factors[kind] += P.nf; sp.frames[kind] += P.nfr;
        sp.seconds[kind] += P.seconds;
    }
    sh.counters.solver_iterations += sm.iterations;
    if (P.njobs > 0)   // the integrators keep their records, like after PreIntegrator::integrate
        for (int k = 0; k < P.npre; ++k) {
            PreIntegrator &pre = *pres[k].job;
            std::copy(&P.preint[(size_t)k * RDVIO_PREINT_SIZE], &P.preint[(size_t)(k + 1) * RDVIO_PREINT_SIZE], pre.delta.begin());
            pre.key = PreIntegrator::Key{pre.data.size(), pres[k].t, pre.data.front().t, pre.data.back().t, pres[k].bg, pres[k].ba, true, true, true};
        }
    // the reference's parameter blocks ARE the Frame / Track members (solver.cpp:88-114): copy the result back
    for (int i = 0; i < P.nfr; ++i)
        if (frame_fixed[i] != 1) frames[i]->set_state(&P.states_out[16 * (size_t)i]);
    for (int l = 0; l < P.nl; ++l)
        if (!lm_fixed[l]) lms[l]->inv_depth = P.invd_out[l];
    if (summary_out) *summary_out = sm;
    packed.reset();
    return sm.termination != 2;
}

bool BaBuilder::solve(rdvio_ba_summary *summary_out) {
    HostTimer host_timer__(sh.prof, 10);
    pack();
    Packed &P = *packed;
    {
        BackendTimer timer(sh.counters, 4);
        const auto t0 = std::chrono::steady_clock::now();
        sh.backend.check(sh.backend.fn.ba_solve(sh.backend.fn.user, &P.pb, sh.cfg.solver_iteration_limit, P.states_out.data(), P.invd_out.data(), &P.sm),
                         "ba_solve");
        P.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return apply(summary_out);
}

bool BaBuilder::can_begin() const { return sh.backend.fn.ba_solve_begin && sh.backend.fn.ba_solve_end; }

// Solver::solve in two halves (rdvio_backend::ba_solve_begin / _end).  chain: the builder of a solve that has been begun and whose
// result for `chain_frame` is this solve's initial value for the same frame.
void BaBuilder::solve_begin(int slot, const BaBuilder *chain, Frame *chain_frame) {
    HostTimer host_timer__(sh.prof, 10);
    pack();
    Packed &P = *packed;
    int from_slot = -1, from_frame = 0, to_frame = 0;
    if (chain) {
        if (!chain->packed || chain->packed->slot < 0) throw std::logic_error("BaBuilder::solve_begin: the solve to continue has not been begun");
        from_slot = chain->packed->slot;
        from_frame = chain->frame_index(chain_frame);
        to_frame = frame_index(chain_frame);
        if (from_frame < 0 || to_frame < 0) throw std::logic_error("BaBuilder::solve_begin: the chained frame is not a state of both solves");
    }
    BackendTimer timer(sh.counters, 4);
    const auto t0 = std::chrono::steady_clock::now();
    sh.backend.check(sh.backend.fn.ba_solve_begin(sh.backend.fn.user, slot, &P.pb, sh.cfg.solver_iteration_limit, from_slot, from_frame, to_frame), "ba_solve (begin)");
    P.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    P.slot = slot;
}

bool BaBuilder::solve_end(rdvio_ba_summary *summary_out) {
    HostTimer host_timer__(sh.prof, 10);
    Packed &P = *packed;
    {
        BackendTimer timer(sh.counters, 4);
        const auto t0 = std::chrono::steady_clock::now();
        sh.backend.check(sh.backend.fn.ba_solve_end(sh.backend.fn.user, P.slot, P.states_out.data(), P.invd_out.data(), &P.sm), "ba_solve (end)");
        P.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return apply(summary_out);
}

// =====================================================================================================================
// FeatureTracker (feature_tracker.cpp:14-124) with Frame::detect_keypoints / track_keypoints (frame.cpp:55-172)
// =====================================================================================================================
FeatureTracker::FeatureTracker(Shared &sh) : sh(sh) { map = std::make_unique<Map>(sh.ids); }

void FeatureTracker::track_frame(std::unique_ptr<Frame> frame) {
    frames.emplace_back(std::move(frame));
    run();
}

void FeatureTracker::detect_keypoints(Frame *frame) {
    HostTimer host_timer__(sh.prof, 2);
    const int max_points = sh.cfg.feature_tracker_max_keypoint_detection;
    const size_t n0 = frame->bearings.size();
    std::vector<double> kps((n0 + (size_t)max_points + 8) * 2);
    for (size_t i = 0; i < n0; ++i) {
        const V2 p = apply_k(frame->bearings[i], frame->K);
        kps[2 * i] = p.x;
        kps[2 * i + 1] = p.y;
    }
    int n_out = 0;
    Backend &be = sh.backend;
    BackendTimer timer(sh.counters, 1);
    be.check(be.fn.image_detect(be.fn.user, frame->image->handle, kps.data(), (int)n0, (int)(kps.size() / 2), max_points,
                                sh.cfg.feature_tracker_min_keypoint_distance, &n_out),
             "detect_keypoints");
    for (size_t i = n0; i < (size_t)n_out; ++i) frame->append_keypoint(remove_k(v2(&kps[2 * i]), frame->K));
}

void FeatureTracker::track_keypoints(Frame *frame, Frame *next_frame) {
    HostTimer host_timer__(sh.prof, 1);
    const size_t n = frame->bearings.size();
    std::vector<V2> curr(n);
    std::vector<double> curr_xy(2 * std::max<size_t>(n, 1)), next_xy(2 * std::max<size_t>(n, 1));
    for (size_t i = 0; i < n; ++i) {
        curr[i] = apply_k(frame->bearings[i], frame->K);
        curr_xy[2 * i] = curr[i].x;
        curr_xy[2 * i + 1] = curr[i].y;
    }
    const bool predict = sh.cfg.feature_tracker_predict_keypoints != 0;
    if (predict) {
        // rotation-only prediction from the gyro preintegration (frame.cpp:82-94)
        const Q4 delta_key_q = conj(conj(frame->camera.q_cs) * frame->imu.q_cs * next_frame->preintegration.dq() *
                                    conj(next_frame->imu.q_cs) * next_frame->camera.q_cs);
        for (size_t i = 0; i < n; ++i) {
            const V2 p = apply_k(rot(delta_key_q, frame->bearings[i]), next_frame->K);
            next_xy[2 * i] = p.x;
            next_xy[2 * i + 1] = p.y;
        }
    } else {
        next_xy = curr_xy;  // the image seam starts LK at the current position when no guess is given
    }
    std::vector<uint8_t> status(std::max<size_t>(n, 1), 0);
    Backend &be = sh.backend;
    if (n > 0) {
        BackendTimer timer(sh.counters, 2);
        be.check(be.fn.image_track(be.fn.user, frame->image->handle, next_frame->image->handle, (int)n, curr_xy.data(), next_xy.data(),
                                   predict ? 1 : 0, status.data()),
                 "track_keypoints");
    }
    status.resize(n);

    std::vector<V2> curr_h(n), next_h(n);
    std::vector<V3> next_bearings(n);
    for (size_t i = 0; i < n; ++i) {
        curr_h[i] = hnormalized(frame->bearings[i]);
        next_bearings[i] = remove_k(v2(&next_xy[2 * i]), next_frame->K);
        next_h[i] = hnormalized(next_bearings[i]);
    }
    // epipolar gate over ALL points, survivors or not (frame.cpp:108-114)
    // Both gates generate and score their hypotheses behind the backend when it offers the hooks (the HIP product: one launch per
    // batch of samples on the frontend lane); sampling and the accept / early-exit replay stay here (geom.hpp, ransac<>).
    HostTimer gates_timer__(sh.prof, 15);
    RansacDevice gate{be.fn.ransac_generate_score, be.fn.ransac_fetch, be.fn.user};
    const bool gates_on_backend = sh.cfg.tracker_gates_on_backend != 0;
    RansacDevice *gate_dev = (gates_on_backend && be.fn.ransac_generate_score && be.fn.ransac_fetch) ? &gate : nullptr;
    std::vector<char> mask;
    (void)find_essential_matrix(curr_h, next_h, mask, 1.0, 0.999, 1000, 0, gate_dev);
    mask.resize(n, 0);  // (the reference indexes an empty mask when no model was found; treated as "no inliers")
    for (size_t i = 0; i < n; ++i)
        if (!mask[i]) status[i] = 0;
    const M3 R = find_rotation_matrix(frame->bearings, next_bearings, mask, (M_PI / 180.0) * sh.cfg.rotation_ransac_threshold, 0.999, 1000, 0, gate_dev);
    mask.resize(n, 0);
    std::vector<double> angles;
    for (size_t i = 0; i < n; ++i)
        if (mask[i]) angles.push_back(std::acos(dot(R * frame->bearings[i], next_bearings[i])) * 180 / M_PI);
    std::sort(angles.begin(), angles.end());
    const double misalignment = angles.size() > 0 ? angles[angles.size() * 7 / 10] : 0;
    if (misalignment < sh.cfg.rotation_misalignment_threshold) {
        next_frame->set_tag(FT_NO_TRANSLATION, true);
        sh.counters.no_translation_frames++;
    }

    // longest tracks first, Poisson-disk thinning in the next image (frame.cpp:134-161)
    std::vector<std::pair<size_t, size_t>> by_length;
    by_length.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        if (status[i] == 0) continue;
        Track *track = frame->get_track(i);
        if (track == nullptr) continue;
        by_length.emplace_back(i, track->keypoint_num());
    }
    std::sort(by_length.begin(), by_length.end(), [](const auto &a, const auto &b) { return a.second > b.second; });
    bool thinned = false;
    if (gates_on_backend && be.fn.thin_tracks && !by_length.empty()) {
        // the filter itself behind the backend: the order (std::sort's, ties included) and the TT_TRASH flags go in, a keep flag
        // per entry comes back
        std::vector<int32_t> order(by_length.size());
        std::vector<uint8_t> trash(n, 0), keep(by_length.size(), 0);
        for (size_t k = 0; k < by_length.size(); ++k) {
            order[k] = (int32_t)by_length[k].first;
            const Track *track = frame->get_track(by_length[k].first);
            trash[by_length[k].first] = (track && track->tag(TT_TRASH)) ? 1 : 0;
        }
        const int rc = be.fn.thin_tracks(be.fn.user, next_frame->image->width, next_frame->image->height, sh.cfg.feature_tracker_min_keypoint_distance, (int)n,
                                         next_xy.data(), (int)order.size(), order.data(), trash.data(), keep.data());
        if (rc == RDVIO_OK) {
            for (size_t k = 0; k < by_length.size(); ++k)
                if (!keep[k]) status[by_length[k].first] = 0;
            thinned = true;
        } else if (rc != RDVIO_ERR_CAPACITY) {
            be.check(rc, "thin_tracks");
        }   // (beyond the kernel's capacities: the host road below)
    }
    PoissonDisk2 filter(sh.cfg.feature_tracker_min_keypoint_distance);
    for (size_t k = 0; k < by_length.size() && !thinned; ++k) {
        const size_t keypoint_index = by_length[k].first;
        const V2 pt = v2(&next_xy[2 * keypoint_index]);
        Track *track = frame->get_track(keypoint_index);
        if (filter.permit_point(pt) && (!track || !track->tag(TT_TRASH))) filter.preset_point(pt);
        else status[keypoint_index] = 0;
    }
    for (size_t i = 0; i < n; ++i)
        if (status[i]) {
            const size_t next_index = next_frame->keypoint_num();
            next_frame->append_keypoint(next_bearings[i]);
            frame->get_track(i, nullptr)->add_keypoint(next_frame, next_index);
        }
}

void FeatureTracker::run() {
    if (frames.empty()) return;
    HostTimer host_timer__(sh.prof, 0);
    std::unique_ptr<Frame> frame = std::move(frames.front());
    frames.pop_front();
    Backend &be = sh.backend;
    {
        BackendTimer timer(sh.counters, 0);
        be.check(be.fn.image_preprocess(be.fn.user, frame->image->handle, sh.cfg.feature_tracker_clahe_clip_limit, sh.cfg.feature_tracker_clahe_width,
                                        sh.cfg.feature_tracker_clahe_height),
                 "preprocess");
    }

    auto [latest_optimized_time, latest_optimized_frame_id, latest_optimized_pose, latest_optimized_motion] = frontend->get_latest_state();
    (void)latest_optimized_time;
    const bool is_initialized = latest_optimized_frame_id != nil;
    const bool sliding_window_frame_tag = !is_initialized || frame->id() % (size_t)sh.cfg.sliding_window_tracker_frequent == 0;
    if (map->frame_num() > 0) {
        // The integrations of this step -- every frame after the latest optimised one (feature_tracker.cpp:45-61) and the new
        // frame (:82-84) -- read only the bias of the frame before them, and predict() copies the bias forward unchanged
        // (preintegrator.cpp:104-105): all of them integrate with the latest optimised bias and none depends on another's
        // result, so they travel to the device as ONE batch; the predictions then run in the reference's order.
        std::vector<PreIntegrator::Job> jobs;
        size_t idx = nil;
        if (is_initialized) {
            idx = map->frame_index_by_id(latest_optimized_frame_id);
            if (idx != nil) {
                Frame *latest = map->get_frame(idx);
                latest->pose = latest_optimized_pose;
                latest->motion = latest_optimized_motion;
                for (size_t j = idx + 1; j < map->frame_num(); ++j) {
                    Frame *frame_j = map->get_frame(j);
                    jobs.push_back({&frame_j->preintegration, frame_j->image->t, latest->motion.bg, latest->motion.ba});
                }
            } else {
                latest_state.reset();  // the sliding window cannot catch up (feature_tracker.cpp:62-68)
            }
        }
        Frame *last_frame = map->get_frame(map->frame_num() - 1);
        if (!last_frame->preintegration.data.empty()) {
            if (frame->preintegration.data.empty() || (frame->preintegration.data.front().t - last_frame->image->t > 1.0e-5)) {
                ImuData imu = last_frame->preintegration.data.back();
                imu.t = last_frame->image->t;
                frame->preintegration.data.insert(frame->preintegration.data.begin(), imu);
            }
        }
        // the new frame integrates with last_frame's bias as it will stand after the predictions below
        const bool last_repropagated = idx != nil && idx + 1 < map->frame_num();
        const MotionState &bias_src = last_repropagated ? map->get_frame(idx)->motion : last_frame->motion;
        jobs.push_back({&frame->preintegration, frame->image->t, bias_src.bg, bias_src.ba});
        (void)PreIntegrator::integrate_batch(be, LANE_TRACKER, jobs, false, false);
        if (idx != nil)
            for (size_t j = idx + 1; j < map->frame_num(); ++j) map->get_frame(j)->preintegration.predict(map->get_frame(j - 1), map->get_frame(j));
        track_keypoints(last_frame, frame.get());
        if (is_initialized) {
            frame->preintegration.predict(last_frame, frame.get());
            latest_state = std::make_tuple(frame->image->t, frame->pose, frame->motion);
        }
        be.fn.image_release(be.fn.user, last_frame->image->handle);
    }
    if (sliding_window_frame_tag) detect_keypoints(frame.get());
    map->attach_frame(std::move(frame));
    sh.counters.frames_tracked++;
    while (map->frame_num() > (size_t)(is_initialized ? sh.cfg.feature_tracker_max_frames : sh.cfg.feature_tracker_max_init_frames) &&
           map->get_frame(0)->id() < latest_optimized_frame_id)
        map->erase_frame(0);
    if (sliding_window_frame_tag) frontend->issue_frame(map->get_frame(map->frame_num() - 1));
}

// =====================================================================================================================
// Frontend (frontend.cpp:14-110)
// =====================================================================================================================
Frontend::Frontend(FeatureTracker *ft, Shared &sh) : feature_tracker(ft), sh(sh) {
    initializer = std::make_unique<Initializer>(sh);
    latest_state = std::make_tuple(0.0, nil, PoseState{}, MotionState{});
    if (sh.cfg.threading == 2) worker = std::thread([this] { worker_main(); });
}

Frontend::~Frontend() {
    if (worker.joinable()) {
        try { drain(); } catch (...) {}
        {
            std::lock_guard<std::mutex> lk(mtx);
            phase.store(3, std::memory_order_release);
        }
        cv.notify_all();
        worker.join();
    }
}

void Frontend::issue_frame(Frame *frame) {
    pending_frame_ids.push_back(frame->id());
    run();
}

namespace {
// a wait that spins first: a hand-over is a few hundred microseconds away when frames are replayed at full speed, a whole
// frame period away when they arrive in real time
template <class Pred>
void spin_then_sleep(std::mutex &mtx, std::condition_variable &cv, Pred ready) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; ++spins) {
        if (ready()) return;
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    std::unique_lock<std::mutex> lk(mtx);
    cv.wait(lk, ready);
}
}  // namespace

void Frontend::worker_main() {
    if (sh.backend.fn.thread_attach) (void)sh.backend.fn.thread_attach(sh.backend.fn.user);
    for (;;) {
        spin_then_sleep(mtx, cv, [this] { const int p = phase.load(std::memory_order_acquire); return p == 1 || p == 3; });
        if (phase.load(std::memory_order_acquire) == 3) return;
        execute(*job);
        {
            std::lock_guard<std::mutex> lk(mtx);
            phase.store(2, std::memory_order_release);
        }
        cv.notify_all();
    }
}

void Frontend::drain() {
    if (!worker.joinable()) return;
    spin_then_sleep(mtx, cv, [this] { return phase.load(std::memory_order_acquire) != 1; });
}

// the frontend's step proper: everything of Frontend::run's tracking branch that does not touch the feature-tracking map
void Frontend::execute(FrontendJob &j) {
    HostTimer host_timer__(sh.prof, 12);
    try {
        if (j.mirrored) sliding_window_tracker->mirror_frame_begin(j);
        sliding_window_tracker->mirror_frame_apply(j);
        if (j.mirrored) sliding_window_tracker->mirror_frame_finish(j);
        j.ok = sliding_window_tracker->track(j);
        j.latest_state = sliding_window_tracker->get_latest_state();
        sliding_window_tracker->newest_frame_summary(j.newest_id, j.newest_flags);
    } catch (...) {
        j.error = std::current_exception();
    }
}

void Frontend::publish() {
    if (!job) return;
    std::unique_ptr<FrontendJob> done = std::move(job);
    phase.store(0, std::memory_order_relaxed);
    if (done->error) std::rethrow_exception(done->error);
    if (done->ok) {
        auto [t, pose, motion] = done->latest_state;
        latest_state = std::make_tuple(t, done->frame_id, pose, motion);
        newest_id = done->newest_id;
        newest_flags.swap(done->newest_flags);
    } else {
        latest_state = std::make_tuple(0.0, nil, PoseState{}, MotionState{});
        initializer = std::make_unique<Initializer>(sh);
        sliding_window_tracker.reset();
        newest_id = nil;
        newest_flags.clear();
    }
    // update_track_status's writes to the feature-tracking map's tracks (sliding_window_tracker.cpp:751-756)
    if (!done->old_tracks_nonstatic.empty()) {
        Map *ftmap = feature_tracker->map.get();
        if (const size_t idx = ftmap->frame_index_by_id(done->frame_id); idx != nil) {
            Frame *old_frame = ftmap->get_frame(idx);
            for (size_t kp : done->old_tracks_nonstatic)
                if (kp < old_frame->keypoint_num())
                    if (Track *old_track = old_frame->get_track(kp)) old_track->set_tag(TT_STATIC, false);
        }
    }
}

void Frontend::run() {
    if (pending_frame_ids.empty()) return;
    HostTimer host_timer__(sh.prof, 3);
    const int mode = sh.cfg.threading;
    // What the next step needs from the feature-tracking map is gathered BEFORE waiting for the step in flight: it depends on
    // that step only through the id of the frame the sliding-window map will end with -- as a rule the step's own frame.  (A
    // wrong guess is noticed after the hand-over and the packet gathered again.)
    std::unique_ptr<FrontendJob> next;
    if (!initializer && sliding_window_tracker) {
        HostTimer gather_timer__(sh.prof, 16);
        next = std::make_unique<FrontendJob>();
        next->frame_id = pending_frame_ids.front();
        SlidingWindowTracker::gather_mirror_packet(feature_tracker->map.get(), job ? job->frame_id : newest_id, next->frame_id, next->packet);
    }
    // the hand-over: the previous step has finished and (pipelined schedule) its results become visible to the tracker
    {
        HostTimer wait_timer__(sh.prof, 13);
        drain();
    }
    {
        HostTimer handover_timer__(sh.prof, 14);
        if (mode != 0) publish();
    }
    if (initializer) {
        // initialisation reads the whole feature-tracking map and runs once: always inline
        const size_t pending_frame_id = pending_frame_ids.front();
        pending_frame_ids.clear();
        initializer->mirror_keyframe_map(feature_tracker->map.get(), pending_frame_id);
        if ((sliding_window_tracker = initializer->initialize())) {
            auto [t, pose, motion] = sliding_window_tracker->get_latest_state();
            latest_state = std::make_tuple(t, pending_frame_id, pose, motion);
            sliding_window_tracker->newest_frame_summary(newest_id, newest_flags);
            initializer.reset();
        }
    } else if (sliding_window_tracker) {
        job = next ? std::move(next) : std::make_unique<FrontendJob>();
        job->frame_id = pending_frame_ids.front();
        pending_frame_ids.pop_front();
        {
            HostTimer handover_timer__(sh.prof, 14);
            if (job->packet.frame_i_id != newest_id || job->packet.frame_j_id != job->frame_id)
                SlidingWindowTracker::gather_mirror_packet(feature_tracker->map.get(), newest_id, job->frame_id, job->packet);
            SlidingWindowTracker::mirror_frame_handover(sh.ids, newest_flags, feature_tracker->map.get(), sh.cfg.parsac_flag != 0, *job);
        }
        if (mode == 2) {
            {
                std::lock_guard<std::mutex> lk(mtx);
                phase.store(1, std::memory_order_release);
            }
            cv.notify_all();
        } else {
            execute(*job);
            if (mode == 0) publish();
        }
    }
}

// =====================================================================================================================
// SlidingWindowTracker (sliding_window_tracker.cpp:16-456)
// =====================================================================================================================
SlidingWindowTracker::SlidingWindowTracker(std::unique_ptr<Map> keyframe_map, Shared &sh) : map(std::move(keyframe_map)), sh(sh) {
    for (size_t j = 1; j < map->frame_num(); ++j) {
        Frame *frame_i = map->get_frame(j - 1), *frame_j = map->get_frame(j);
        frame_j->preintegration.integrate(sh.backend, LANE_ESTIMATOR, frame_j->image->t, frame_i->motion.bg, frame_i->motion.ba, true, true);
    }
}

// mirror_frame (sliding_window_tracker.cpp:29-78), the feature-tracking map's side.  Reads that map only.
void SlidingWindowTracker::gather_mirror_packet(const Map *ftmap, size_t frame_i_id, size_t frame_j_id, MirrorPacket &pk) {
    pk = MirrorPacket{};
    pk.frame_i_id = frame_i_id;
    pk.frame_j_id = frame_j_id;
    if (frame_i_id == nil || frame_j_id == nil) return;
    const size_t index_i = ftmap->frame_index_by_id(frame_i_id), index_j = ftmap->frame_index_by_id(frame_j_id);
    if (index_i == nil || index_j == nil) return;
    const Frame *old_frame_i = ftmap->get_frame(index_i), *old_frame_j = ftmap->get_frame(index_j);
    pk.curr_frame = old_frame_j->clone();
    std::vector<ImuData> &new_data = pk.curr_frame->preintegration.data;
    for (size_t index = index_j - 1; index > index_i; --index) {
        const std::vector<ImuData> &old_data = ftmap->get_frame(index)->preintegration.data;
        new_data.insert(new_data.begin(), old_data.begin(), old_data.end());
    }
    for (size_t ki = 0; ki < old_frame_i->keypoint_num(); ++ki)
        if (Track *track = old_frame_i->get_track(ki))
            if (size_t kj = track->get_keypoint_index(old_frame_j); kj != nil) {
                pk.matches.emplace_back((uint32_t)ki, (uint32_t)kj);
                pk.ft_tracks.push_back(track);
            }
    pk.found = true;
}

// The hand-over's share: what mirror_frame writes INTO the feature-tracking map (TT_TRASH of the continued tracks, from the
// sliding-window map's flags as the finished step left them), what update_track_status will read of it, and the ids of the tracks
// the sliding-window map is about to create (drawn here, between the tracker's frames, as a single thread would draw them).
void SlidingWindowTracker::mirror_frame_handover(IdGenerator &ids, const std::vector<uint8_t> &newest_flags, const Map *ftmap, bool parsac, FrontendJob &job) {
    MirrorPacket &pk = job.packet;
    job.mirrored = pk.found;
    if (!pk.found) return;
    size_t created = 0;
    for (size_t m = 0; m < pk.matches.size(); ++m) {
        const uint32_t ki = pk.matches[m].first;
        const uint8_t fl = ki < newest_flags.size() ? newest_flags[ki] : 0;
        if (!(fl & 1u)) ++created;   // (a new track: neither TT_TRASH nor anything but TT_STATIC)
        pk.ft_tracks[m]->set_tag(TT_TRASH, (fl & 2u) != 0);
    }
    job.track_ids[0] = ids.next_track + 1;
    job.track_ids[1] = ids.next_track + 1 + created;
    ids.next_track += created;
    if (parsac) {
        // what update_track_status will read of the feature-tracking map (nobody writes these bits before the step's own
        // deferred writes are published: Track::Track is the only other writer of TT_STATIC on that map)
        const Frame *old_frame_j = ftmap->get_frame(ftmap->frame_index_by_id(pk.frame_j_id));
        job.old_track_flags.assign(old_frame_j->keypoint_num(), 0);
        for (size_t j = 0; j < old_frame_j->keypoint_num(); ++j)
            if (const Track *old_track = old_frame_j->get_track(j)) job.old_track_flags[j] = (uint8_t)(1u | (old_track->tag(TT_STATIC) ? 2u : 0u));
    }
}

// the new frame's preintegration (sliding_window_tracker.cpp:63-64) is issued first: it needs the samples (in the packet) and the
// newest frame's biases (final since the last step) only, and the map work below runs while the device integrates
void SlidingWindowTracker::mirror_frame_begin(FrontendJob &job) {
    HostTimer host_timer__(sh.prof, 4);
    const Frame *keyframe = map->get_frame(map->frame_num() - 1);
    const Frame *new_frame_i = keyframe->subframes.empty() ? keyframe : keyframe->subframes.back().get();
    Frame *new_frame_j = job.packet.curr_frame.get();
    if (!new_frame_j) return;
    new_frame_j->preintegration.integrate_begin(sh.backend, new_frame_j->image->t, new_frame_i->motion.bg, new_frame_i->motion.ba);
}

// ... and the sliding-window map's side (the frontend's step)
void SlidingWindowTracker::mirror_frame_apply(FrontendJob &job) {
    HostTimer host_timer__(sh.prof, 4);
    if (!job.mirrored) return;
    MirrorPacket &pk = job.packet;
    Frame *keyframe = map->get_frame(map->frame_num() - 1);
    Frame *new_frame_i = keyframe;
    if (!keyframe->subframes.empty()) new_frame_i = keyframe->subframes.back().get();
    if (new_frame_i->id() != pk.frame_i_id) throw std::logic_error("mirror_frame: the packet was gathered against another frame");
    map->attach_frame(std::move(pk.curr_frame));
    Frame *new_frame_j = map->get_frame(map->frame_num() - 1);
    map->reserved_track_ids[0] = job.track_ids[0];
    map->reserved_track_ids[1] = job.track_ids[1];
    for (const auto &[ki, kj] : pk.matches) new_frame_i->get_track(ki, map.get())->add_keypoint(new_frame_j, kj);
    if (map->reserved_track_ids[0] != map->reserved_track_ids[1]) throw std::logic_error("mirror_frame: track ids drawn at the hand-over do not match the tracks created");
    map->prune_tracks([](const Track *track) { return track->tag(TT_TRASH) && !track->tag(TT_STATIC); });
    job.new_frame_i = new_frame_i;
    job.new_frame_j = new_frame_j;
}

void SlidingWindowTracker::newest_frame_summary(size_t &id, std::vector<uint8_t> &flags) const {
    const Frame *keyframe = map->get_frame(map->frame_num() - 1);
    const Frame *newest = keyframe->subframes.empty() ? keyframe : keyframe->subframes.back().get();
    id = newest->id();
    flags.assign(newest->keypoint_num(), 0);
    for (size_t k = 0; k < newest->keypoint_num(); ++k)
        if (const Track *track = newest->get_track(k)) flags[k] = (uint8_t)(1u | ((track->tag(TT_TRASH) && !track->tag(TT_STATIC)) ? 2u : 0u));
}

void SlidingWindowTracker::mirror_frame_finish(FrontendJob &job) {
    HostTimer host_timer__(sh.prof, 4);
    Frame *new_frame_i = job.new_frame_i, *new_frame_j = job.new_frame_j;
    if (!new_frame_j->preintegration.integrate_end(sh.backend))   // (no begin: an unmirrored path; the whole call here)
        new_frame_j->preintegration.integrate(sh.backend, LANE_ESTIMATOR, new_frame_j->image->t, new_frame_i->motion.bg, new_frame_i->motion.ba, true, true);
    new_frame_j->preintegration.predict(new_frame_i, new_frame_j);
}

bool SlidingWindowTracker::track(FrontendJob &job) {
    if (sh.cfg.parsac_flag) {
        if (judge_track_status()) update_track_status(job);
    }
    BaBuilder localize(sh);
    Frame *frame_j = localize_newframe(localize);
    if (!localize.can_begin()) {
        // one solve after the other (sliding_window_tracker.cpp:80-99 as written)
        {
            HostTimer host_timer__(sh.prof, 5);
            localize.solve();
        }
        if (manage_keyframe()) {
            track_landmark();
            refine_window();
            slide_window();
        } else {
            refine_subwindow();
        }
        return true;
    }
    // The backend takes solves in two halves.  manage_keyframe decides from tags and counts, not from the localised pose, and the
    // subwindow's graph does not depend on it either -- only the new frame's starting point does: the localisation is begun, the
    // decision taken, and on the subframe road (four frames of five) the subwindow solve is built, uploaded and begun behind it,
    // starting the new frame from the localisation's result as the DEVICE holds it.  Same numbers as one after the other.
    {
        HostTimer host_timer__(sh.prof, 5);
        localize.solve_begin(1);
    }
    if (manage_keyframe()) {
        {
            HostTimer host_timer__(sh.prof, 5);
            localize.solve_end();
        }
        track_landmark();
        refine_window();
        slide_window();
    } else {
        refine_subwindow(&localize, frame_j);
    }
    return true;
}

// the graph of localize_newframe (sliding_window_tracker.cpp:101-125); returns the new frame
Frame *SlidingWindowTracker::localize_newframe(BaBuilder &solver) {
    HostTimer host_timer__(sh.prof, 5);
    Frame *frame_i = map->get_frame(map->frame_num() - 2);
    if (!frame_i->subframes.empty()) frame_i = frame_i->subframes.back().get();
    Frame *frame_j = map->get_frame(map->frame_num() - 1);
    solver.add_frame_states(frame_j);
    solver.add_preintegration(frame_i, frame_j, frame_j->preintegration, true);
    for (size_t k = 0; k < frame_j->keypoint_num(); ++k)
        if (Track *track = frame_j->get_track(k))
            if (track->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) solver.add_reprojection_prior(frame_j, track, k);
    solver.kind = 0;
    sh.counters.localizations++;
    return frame_j;
}

bool SlidingWindowTracker::manage_keyframe() {
    HostTimer host_timer__(sh.prof, 9);
    Frame *keyframe_i = map->get_frame(map->frame_num() - 2);
    Frame *newframe_j = map->get_frame(map->frame_num() - 1);
    if (!keyframe_i->subframes.empty()) {
        if (keyframe_i->subframes.back()->tag(FT_NO_TRANSLATION)) {
            if (newframe_j->tag(FT_NO_TRANSLATION)) {
                // [T]...........<-[R]   keeps accumulating rotation-only subframes
            } else {
                keyframe_i->subframes.back()->set_tag(FT_KEYFRAME, true);
                map->attach_frame(std::move(keyframe_i->subframes.back()), map->frame_num() - 1);
                keyframe_i->subframes.pop_back();
                newframe_j->set_tag(FT_KEYFRAME, true);
                return true;
            }
        } else {
            if (newframe_j->tag(FT_NO_TRANSLATION)) {
                std::unique_ptr<Frame> frame_lifted = std::move(keyframe_i->subframes.back());
                keyframe_i->subframes.pop_back();
                frame_lifted->set_tag(FT_KEYFRAME, true);
                frame_lifted->subframes.emplace_back(map->detach_frame(map->frame_num() - 1));
                map->attach_frame(std::move(frame_lifted));
                return true;
            } else if (keyframe_i->subframes.size() >= (size_t)sh.cfg.sliding_window_subframe_size) {
                newframe_j->set_tag(FT_KEYFRAME, true);
                return true;
            }
        }
    }
    size_t mapped_landmark_count = 0;
    for (size_t k = 0; k < newframe_j->keypoint_num(); ++k)
        if (Track *track = newframe_j->get_track(k))
            if (track->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) mapped_landmark_count++;
    const bool is_keyframe = mapped_landmark_count < (size_t)sh.cfg.sliding_window_force_keyframe_landmarks;
    if (is_keyframe) {
        newframe_j->set_tag(FT_KEYFRAME, true);
        return true;
    }
    keyframe_i->subframes.emplace_back(map->detach_frame(map->frame_num() - 1));
    return false;
}

void SlidingWindowTracker::track_landmark() {
    Frame *newframe_j = map->get_frame(map->frame_num() - 1);
    for (size_t k = 0; k < newframe_j->keypoint_num(); ++k)
        if (Track *track = newframe_j->get_track(k))
            if (!track->tag(TT_TRIANGULATED)) {
                if (auto p = track->triangulate()) {
                    track->set_landmark_point(p.value());
                    track->set_tag(TT_TRIANGULATED, true);
                    track->set_tag(TT_VALID, true);
                    track->set_tag(TT_STATIC, true);
                } else {
                    track->inv_depth = -1.0;
                    track->set_tag(TT_TRIANGULATED, false);
                    track->set_tag(TT_VALID, false);
                }
            }
}

void SlidingWindowTracker::refine_window() {
    HostTimer host_timer__(sh.prof, 6);
    std::optional<HostTimer> part_timer__(std::in_place, sh.prof, 17);
    BaBuilder solver(sh);
    if (!map->marginalization_factor) {
        // MarginalizationFactor::MarginalizationFactor (marginalization_factor.h:16-32): every frame but the newest, the
        // pose of frame 0 pinned with 1e15
        auto prior = std::make_unique<MarginalizationPrior>();
        const size_t n = map->frame_num() - 1, D = 15 * n;
        prior->frames.resize(n);
        prior->lin.assign(16 * n, 0.0);
        for (size_t i = 0; i < n; ++i) {
            prior->frames[i] = map->get_frame(i);
            map->get_frame(i)->get_state(&prior->lin[16 * i]);
        }
        prior->f.assign(D, 0.0);
        prior->S.assign(D * D, 0.0);
        for (size_t a = 0; a < 6; ++a) prior->S[a * D + a] = 1.0e15;
        map->marginalization_factor = std::move(prior);
    }
    for (size_t i = 0; i < map->frame_num(); ++i) solver.add_frame_states(map->get_frame(i));
    std::unordered_set<Track *> visited;
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        for (size_t j = 0; j < frame->keypoint_num(); ++j) {
            Track *track = frame->get_track(j);
            if (!track || visited.count(track)) continue;
            visited.insert(track);
            if (!track->tag(TT_VALID) || !track->tag(TT_STATIC) || !track->first_frame()->tag(FT_KEYFRAME)) continue;
            solver.add_track_states(track, false);
        }
    }
    solver.add_marginalization(map->marginalization_factor.get());
    for (size_t i = 0; i < map->frame_num(); ++i) {
        Frame *frame = map->get_frame(i);
        for (size_t j = 0; j < frame->keypoint_num(); ++j) {
            Track *track = frame->get_track(j);
            if (!track || !track->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) continue;
            if (!track->first_frame()->tag(FT_KEYFRAME) || frame == track->first_frame()) continue;
            solver.add_reprojection_error(frame, j);
        }
    }
    // keyframe_preintegration of every interval (:283-301): the W independent integrations ride inside the solve call
    for (size_t j = 1; j < map->frame_num(); ++j) {
        Frame *frame_i = map->get_frame(j - 1), *frame_j = map->get_frame(j);
        frame_j->keyframe_preintegration = frame_j->preintegration;
        if (!frame_i->subframes.empty()) {
            std::vector<ImuData> imu_data;
            for (const auto &sub : frame_i->subframes) imu_data.insert(imu_data.end(), sub->preintegration.data.begin(), sub->preintegration.data.end());
            frame_j->keyframe_preintegration.data.insert(frame_j->keyframe_preintegration.data.begin(), imu_data.begin(), imu_data.end());
        }
        frame_j->keyframe_preintegration.key.valid = false;   // (the copied record belongs to other samples)
        (void)solver.add_integrated_preintegration(frame_i, frame_j, frame_j->keyframe_preintegration, frame_j->image->t, frame_i->motion.bg, frame_i->motion.ba);
    }
    solver.kind = 1;
    part_timer__.reset();
    solver.solve();
    sh.counters.window_solves++;
    sh.counters.keyframes++;
    part_timer__.emplace(sh.prof, 18);

    // depth / reprojection culling (:304-336)
    for (size_t k = 0; k < map->track_num(); ++k) {
        Track *track = map->get_track(k);
        if (track->tag(TT_TRIANGULATED)) {
            bool is_valid = true;
            const V3 x = track->get_landmark_point();
            double rpe = 0.0, rpe_count = 0.0;
            for (const auto &kv : track->keypoint_map()) {
                Frame *frame = kv.second.first;
                if (!frame->tag(FT_KEYFRAME)) continue;
                const PoseState pose = frame->get_pose(frame->camera);
                const V3 y = rot(conj(pose.q), x - pose.p);
                if (y.z <= 1.0e-3 || y.z > 50) {
                    is_valid = false;
                    break;
                }
                rpe += norm(apply_k(y, frame->K) - apply_k(frame->get_keypoint(kv.second.second), frame->K));
                rpe_count += 1.0;
            }
            is_valid = is_valid && (rpe / std::max(rpe_count, 1.0) < 3.0);
            track->set_tag(TT_VALID, is_valid);
        } else {
            track->inv_depth = -1.0;
        }
    }
    for (size_t k = 0; k < map->track_num(); ++k) {
        Track *track = map->get_track(k);
        if (!track->tag(TT_VALID)) track->set_tag(TT_TRASH, true);
    }
}

void SlidingWindowTracker::slide_window() {
    while (map->frame_num() > (size_t)sh.cfg.sliding_window_size) {
        Frame *frame = map->get_frame(0);
        for (auto &sub : frame->subframes) map->untrack_frame(sub.get());
        marginalize_frame0();
    }
}

// Map::marginalize_frame(0) (map.cpp:50-62) with the graph of CeresMarginalizationFactor::marginalize
// (ceres/marginalization_factor.h:95-380) exported as SoA
void SlidingWindowTracker::marginalize_frame0() {
    HostTimer host_timer__(sh.prof, 8);
    MarginalizationPrior &prior = *map->marginalization_factor;
    prior.ready(sh.backend);
    const int nfm = (int)map->frame_num();
    std::unordered_map<const Frame *, int> index;
    std::vector<double> states((size_t)nfm * 16);
    for (int i = 0; i < nfm; ++i) {
        index[map->get_frame(i)] = i;
        map->get_frame(i)->get_state(&states[16 * (size_t)i]);
    }
    std::vector<int32_t> prior_frames;
    for (Frame *f : prior.frames) prior_frames.push_back(index.at(f));
    Frame *victim = map->get_frame(0);
    std::vector<int32_t> tgt, ref, lm;
    std::vector<double> tangent, zref, invd;
    for (size_t j = 0; j < victim->keypoint_num(); ++j) {
        Track *track = victim->get_track(j);
        if (!track || !track->tag(TT_VALID)) continue;
        const auto [frame_ref, ref_kp] = track->first_keypoint();
        if (!frame_ref->tag(FT_KEYFRAME)) continue;
        auto ir = index.find(frame_ref);
        if (ir == index.end()) continue;
        const int l = (int)invd.size();
        bool any = false;
        for (const auto &kv : track->keypoint_map()) {
            Frame *frame_tgt = kv.second.first;
            if (frame_tgt == frame_ref) continue;
            auto it = index.find(frame_tgt);
            if (it == index.end()) continue;  // a subframe
            tgt.push_back(it->second);
            ref.push_back(ir->second);
            lm.push_back(l);
            const auto &tg = frame_tgt->tangents[kv.second.second];
            tangent.insert(tangent.end(), tg.begin(), tg.end());
            any = true;
        }
        if (any) {
            const V3 &z = frame_ref->get_keypoint(ref_kp);
            zref.insert(zref.end(), {z.x, z.y, z.z});
            invd.push_back(track->inv_depth);
        }
    }
    double extr[14];
    extr[0] = victim->camera.q_cs.x; extr[1] = victim->camera.q_cs.y; extr[2] = victim->camera.q_cs.z; extr[3] = victim->camera.q_cs.w;
    extr[4] = victim->camera.p_cs.x; extr[5] = victim->camera.p_cs.y; extr[6] = victim->camera.p_cs.z;
    extr[7] = victim->imu.q_cs.x; extr[8] = victim->imu.q_cs.y; extr[9] = victim->imu.q_cs.z; extr[10] = victim->imu.q_cs.w;
    extr[11] = victim->imu.p_cs.x; extr[12] = victim->imu.p_cs.y; extr[13] = victim->imu.p_cs.z;
    rdvio_marg_problem pb;
    std::memset(&pb, 0, sizeof pb);
    pb.n_frames = nfm;
    pb.states = states.data();
    pb.extr = extr;
    pb.sqrt_inv_cov = victim->sqrt_inv_cov;
    pb.n_prior = (int)prior_frames.size();
    pb.prior_frames = prior_frames.data();
    pb.prior_lin = prior.lin.data();
    pb.prior_S = prior.S.data();
    pb.prior_f = prior.f.data();
    pb.preint01 = nfm >= 2 ? map->get_frame(1)->keyframe_preintegration.delta.data() : nullptr;
    pb.n_landmarks = (int)invd.size();
    pb.z_ref = zref.data();
    pb.inv_depth = invd.data();
    pb.n_factors = (int)tgt.size();
    pb.tgt = tgt.data(); pb.ref = ref.data(); pb.lm = lm.data();
    pb.tangent = tangent.data();
    const size_t R = 15 * (size_t)(nfm - 1);
    std::vector<double> S(R * R), f(R), lin(16 * (size_t)(nfm - 1));
    bool pending = false;
    {
        BackendTimer timer(sh.counters, 5);
        if (sh.backend.fn.marginalize_begin && sh.backend.fn.marginalize_end) {
            // the new prior is read by the next refine_window: enqueue now, collect then (MarginalizationPrior::ready)
            sh.backend.check(sh.backend.fn.marginalize_begin(sh.backend.fn.user, &pb), "marginalize (begin)");
            pending = true;
        } else {
            sh.backend.check(sh.backend.fn.marginalize(sh.backend.fn.user, &pb, S.data(), f.data(), lin.data()), "marginalize");
        }
    }
    prior.frames.clear();
    for (int i = 1; i < nfm; ++i) prior.frames.push_back(map->get_frame(i));
    prior.S.swap(S);
    prior.f.swap(f);
    prior.lin.swap(lin);
    prior.pending = pending;
    for (size_t i = 0; i < victim->keypoint_num(); ++i)
        if (Track *track = victim->get_track(i)) track->remove_keypoint(victim);
    map->drop_front_frame();
    sh.counters.marginalizations++;
}

// `after` / `after_frame`: a solve that has been begun (the localisation) whose result for after_frame is this solve's starting
// point for that frame; both are ended here
void SlidingWindowTracker::refine_subwindow(BaBuilder *after, Frame *after_frame) {
    HostTimer host_timer__(sh.prof, 7);
    auto run = [&](BaBuilder &solver) {
        if (!after) {
            solver.solve();
            return;
        }
        solver.solve_begin(0, after, after_frame);
        after->solve_end();
        solver.solve_end();
    };
    Frame *frame = map->get_frame(map->frame_num() - 1);
    if (frame->subframes.empty()) {
        if (after) after->solve_end();
        return;
    }
    if (frame->subframes[0]->tag(FT_NO_TRANSLATION)) {
        if (frame->subframes.size() >= 9) {
            // compress a long rotation-only run: keep every third subframe, merging the IMU data (:353-371)
            for (size_t i = frame->subframes.size() / 3; i > 0; --i) {
                Frame *tgt_frame = frame->subframes[i * 3 - 1].get();
                std::vector<ImuData> imu_data;
                for (size_t j = i * 3 - 1; j > (i - 1) * 3; --j) {
                    Frame *src_frame = frame->subframes[j - 1].get();
                    imu_data.insert(imu_data.begin(), src_frame->preintegration.data.begin(), src_frame->preintegration.data.end());
                    map->untrack_frame(src_frame);
                    frame->subframes.erase(frame->subframes.begin() + (std::ptrdiff_t)(j - 1));
                }
                tgt_frame->preintegration.data.insert(tgt_frame->preintegration.data.begin(), imu_data.begin(), imu_data.end());
            }
        }
        BaBuilder solver(sh);
        frame->set_tag(FT_FIX_POSE, true);
        frame->set_tag(FT_FIX_MOTION, true);
        solver.add_frame_states(frame);
        for (size_t i = 0; i < frame->subframes.size(); ++i) {
            Frame *subframe = frame->subframes[i].get();
            solver.add_frame_states(subframe);
            Frame *prev_frame = (i == 0 ? frame : frame->subframes[i - 1].get());
            // every subframe interval re-integrated with the previous frame's biases (:379-384) -- inside the solve call
            if (!solver.add_integrated_preintegration(prev_frame, subframe, subframe->preintegration, subframe->image->t, prev_frame->motion.bg, prev_frame->motion.ba))
                solver.add_preintegration(prev_frame, subframe, subframe->preintegration, false);
        }
        Frame *last_subframe = frame->subframes.back().get();
        for (size_t k = 0; k < last_subframe->keypoint_num(); ++k)
            if (Track *track = last_subframe->get_track(k))
                if (track->tag(TT_VALID)) {
                    if (track->tag(TT_TRIANGULATED)) {
                        if (track->tag(TT_STATIC)) solver.add_reprojection_prior(last_subframe, track, k);
                    } else {
                        solver.add_rotation_prior(last_subframe, track);
                    }
                }
        solver.kind = 2;
        run(solver);
        frame->set_tag(FT_FIX_POSE, false);
        frame->set_tag(FT_FIX_MOTION, false);
    } else {
        BaBuilder solver(sh);
        frame->set_tag(FT_FIX_POSE, true);
        frame->set_tag(FT_FIX_MOTION, true);
        solver.add_frame_states(frame);
        for (size_t i = 0; i < frame->subframes.size(); ++i) {
            Frame *subframe = frame->subframes[i].get();
            solver.add_frame_states(subframe);
            Frame *prev_frame = (i == 0 ? frame : frame->subframes[i - 1].get());
            // (:416-421)
            if (!solver.add_integrated_preintegration(prev_frame, subframe, subframe->preintegration, subframe->image->t, prev_frame->motion.bg, prev_frame->motion.ba))
                solver.add_preintegration(prev_frame, subframe, subframe->preintegration, false);
            for (size_t k = 0; k < subframe->keypoint_num(); ++k)
                if (Track *track = subframe->get_track(k))
                    if (track->all_tagged({TT_VALID, TT_TRIANGULATED, TT_STATIC})) {
                        if (track->first_frame()->tag(FT_KEYFRAME)) solver.add_reprojection_prior(subframe, track, k);
                        // else: tracks anchored in a later subframe -- the reference indexes the KEYFRAME's factor list
                        // with the SUBFRAME's keypoint index here (:431-434), which reads an unrelated or out-of-range
                        // entry; that access is not reproduced ("deliberate deviations")
                    }
        }
        solver.kind = 2;
        run(solver);
        frame->set_tag(FT_FIX_POSE, false);
        frame->set_tag(FT_FIX_MOTION, false);
    }
    sh.counters.subwindow_solves++;
}


// =====================================================================================================================
// RD dynamic-outlier handling (sliding_window_tracker.cpp:461-769), parsac_flag only
// =====================================================================================================================
namespace {
// :557-583: relative camera motion between two frames
void predict_RT(const Frame *frame_i, const Frame *frame_j, M3 &R, V3 &t) {
    // P = Pwc^-1 PwI Pji PwI^-1 Pwc with Pji = Pwj^-1 Pwi (4 x 4 rigid transforms)
    struct T4 { M3 R; V3 t; };
    auto mul = [](const T4 &a, const T4 &b) { return T4{a.R * b.R, a.R * b.t + a.t}; };
    auto inv = [](const T4 &a) { const M3 Rt = transpose(a.R); return T4{Rt, -(Rt * a.t)}; };
    const T4 Pwc{to_mat(frame_i->camera.q_cs), frame_i->camera.p_cs}, PwI{to_mat(frame_i->imu.q_cs), frame_i->imu.p_cs};
    const T4 Pwi{to_mat(frame_i->pose.q), frame_i->pose.p}, Pwj{to_mat(frame_j->pose.q), frame_j->pose.p};
    const T4 Pji = mul(inv(Pwj), Pwi);
    const T4 P = mul(mul(mul(mul(inv(Pwc), PwI), Pji), inv(PwI)), Pwc);
    R = P.R;
    t = P.t;
}
// :469-474
double epipolar_dist(const M3 &F, const V2 &pt1, const V2 &pt2) {
    const V3 l = F * V3{pt1.x, pt1.y, 1.0};
    return std::fabs(pt2.x * l.x + pt2.y * l.y + l.z) / std::sqrt(l.x * l.x + l.y * l.y);
}
M3 k_matrix(const double *K) {
    M3 M;
    for (int i = 0; i < 9; ++i) M.m[i] = K[i];
    return M;
}
}  // namespace

bool SlidingWindowTracker::filter_parsac_2d2d(Frame *frame_i, Frame *frame_j, std::vector<char> &mask, std::vector<size_t> &pts_to_index) {
    std::vector<V2> pts1, pts2;
    for (size_t ki = 0; ki < frame_i->keypoint_num(); ++ki)
        if (Track *track = frame_i->get_track(ki)) {
            const size_t kj = track->get_keypoint_index(frame_j);
            // (`if (size_t kj = ...)` in the reference: an observation at keypoint index 0 is skipped, :506)
            if (kj != 0 && kj != nil) {
                pts1.push_back(hnormalized(frame_i->get_keypoint(ki)));
                pts2.push_back(hnormalized(frame_j->get_keypoint(kj)));
                pts_to_index.push_back(kj);
            }
        }
    if (pts1.size() < 10) return false;
    ParsacDeviceScorer dev{sh.backend.fn.parsac_score, sh.backend.fn.parsac_fetch, sh.backend.fn.user};
    dev.generate = sh.backend.fn.parsac_generate_score;
    (void)find_essential_matrix_parsac(pts1, pts2, mask, sh.essential_bin_confidences, m_th / frame_i->K[0], 0.999, 1000, 0,
                                       sh.backend.fn.parsac_score ? &dev : nullptr);
    return true;
}

bool SlidingWindowTracker::judge_track_status() {
    HostTimer host_timer__(sh.prof, 11);
    Frame *curr_frame = map->get_frame(map->frame_num() - 1);
    Frame *keyframe = map->get_frame(map->frame_num() - 2);
    Frame *last_frame = keyframe;
    if (!keyframe->subframes.empty()) last_frame = keyframe->subframes.back().get();
    curr_frame->preintegration.integrate(sh.backend, LANE_ESTIMATOR, curr_frame->image->t, last_frame->motion.bg, last_frame->motion.ba, true, true);
    curr_frame->preintegration.predict(last_frame, curr_frame);

    std::vector<V2> P2D;
    std::vector<V3> P3D;
    std::vector<size_t> lens;
    std::vector<int> indices_map(curr_frame->keypoint_num(), -1);
    for (size_t k = 0; k < curr_frame->keypoint_num(); ++k)
        if (Track *track = curr_frame->get_track(k))
            if (track->all_tagged({TT_VALID, TT_TRIANGULATED})) {
                P2D.push_back(hnormalized(curr_frame->get_keypoint(k)));
                P3D.push_back(track->get_landmark_point());
                lens.push_back(track->m_life);
                indices_map[k] = (int)P3D.size() - 1;
            }
    if (P2D.size() < 20) return false;
    const PoseState pose = curr_frame->get_pose(curr_frame->camera);
    std::vector<char> mask;
    const M3 Rcw = to_mat(conj(pose.q));
    const V3 tcw = -(Rcw * pose.p);
    ParsacDeviceScorer dev{sh.backend.fn.parsac_score, sh.backend.fn.parsac_fetch, sh.backend.fn.user};
    dev.generate = sh.backend.fn.parsac_generate_score;
    dev.iterations_hint = &sh.pnp_iterations_hint;
    (void)find_pnp_matrix_parsac_imu(P3D, P2D, lens, Rcw, tcw, 0.20, 1.0, mask, sh.pnp_bin_confidences, 1.0 / curr_frame->K[0], 0.999, 1000, 0,
                                     sh.backend.fn.parsac_score ? &dev : nullptr);
    mask.resize(P2D.size(), 0);
    sh.counters.parsac_judgements++;

    M3 R;
    V3 t;
    predict_RT(keyframe, curr_frame, R, t);
    // E = [t]x R, F = K^-T E K^-1 (:461-468, :617-619)
    const M3 tx{{0, -t.z, t.y, t.z, 0, -t.x, -t.y, t.x, 0}};
    const M3 E = tx * R;
    const M3 F = inverse3(transpose(k_matrix(keyframe->K))) * E * inverse3(k_matrix(curr_frame->K));
    std::vector<double> inliers_dist, outliers_dist;
    for (size_t i = 0; i < curr_frame->keypoint_num(); ++i) {
        if (indices_map[i] == -1) continue;
        const size_t j = curr_frame->get_track(i)->get_keypoint_index(keyframe);
        if (j == 0 || j == nil) continue;  // (`if (size_t j = ...; j != nil())` with the same index-0 quirk, :626-628)
        const V2 p1 = apply_k(keyframe->get_keypoint(j), keyframe->K), p2 = apply_k(curr_frame->get_keypoint(i), curr_frame->K);
        const double err = epipolar_dist(F, p1, p2) + epipolar_dist(transpose(F), p2, p1);
        (mask[(size_t)indices_map[i]] ? inliers_dist : outliers_dist).push_back(err);
    }
    const size_t min_num = 20;
    if (inliers_dist.size() < min_num || outliers_dist.size() < min_num) return false;
    std::sort(inliers_dist.begin(), inliers_dist.end());
    std::sort(outliers_dist.begin(), outliers_dist.end());
    const double th1 = inliers_dist[(size_t)(inliers_dist.size() * 0.5)], th2 = outliers_dist[(size_t)(outliers_dist.size() * 0.5)];
    if (th2 < th1 * 2) return false;  // ambiguous
    m_th = (th1 + th2) / 2;
    for (size_t k = 0; k < curr_frame->keypoint_num(); ++k)
        if (Track *track = curr_frame->get_track(k))
            if (indices_map[k] != -1) {
                const bool inlier = mask[(size_t)indices_map[k]] != 0;
                track->set_tag(TT_OUTLIER, !inlier);
                track->set_tag(TT_STATIC, inlier);
            }
    return true;
}

void SlidingWindowTracker::update_track_status(FrontendJob &job) {
    Frame *curr_frame = map->get_frame(map->frame_num() - 1);
    // the reference looks the new frame up in the feature-tracking map here (:700-704); its tracks' TT_STATIC bits were
    // taken at the hand-over (mirror_frame_maps) and the writes below are published at the next one
    if (!job.mirrored || job.old_track_flags.size() != curr_frame->keypoint_num()) return;
    std::vector<size_t> outlier_cnts(curr_frame->keypoint_num(), 0), matches_cnts(curr_frame->keypoint_num(), 0);
    const size_t last = map->frame_num() - 1;
    // std::min(last, std::max(last - check_size, size_t(0))) with unsigned wrap-around, as written (:706-709)
    const size_t start_idx = std::min(last, std::max(last - (size_t)sh.cfg.parsac_keyframe_check_size, size_t(0)));
    for (size_t i = start_idx; i < last; i++) {
        std::vector<char> mask;
        std::vector<size_t> pts_to_index;
        if (filter_parsac_2d2d(map->get_frame(i), curr_frame, mask, pts_to_index)) {
            mask.resize(pts_to_index.size(), 0);
            for (size_t j = 0; j < mask.size(); j++) {
                if (!mask[j]) outlier_cnts[pts_to_index[j]] += 1;
                matches_cnts[pts_to_index[j]] += 1;
            }
        }
    }
    for (size_t i = 0; i < curr_frame->keypoint_num(); i++)
        if (Track *curr_track = curr_frame->get_track(i)) {
            // the lookup goes through the frame id (compare<Frame *>), so the feature-tracker's frame finds the window's
            // clone: the keypoint index in old_frame is the one in curr_frame
            const size_t j = i;
            if (j == 0) continue;  // (`if (size_t j = ...)` index-0 quirk, :727)
            const bool old_exists = (job.old_track_flags[j] & 1u) != 0;
            bool old_static = (job.old_track_flags[j] & 2u) != 0;
            const size_t outlier_th = map->frame_num() / 2;
            if (outlier_cnts[i] > outlier_th / 2 && outlier_cnts[i] > 0.8 * matches_cnts[i]) curr_track->set_tag(TT_STATIC, false);
            if (old_exists && (!old_static || !curr_track->tag(TT_STATIC))) {
                if (curr_track->tag(TT_STATIC) || old_static) sh.counters.tracks_marked_dynamic++;
                curr_track->set_tag(TT_STATIC, false);
                job.old_tracks_nonstatic.push_back(j);
                old_static = false;
            }
        }
}

std::tuple<double, PoseState, MotionState> SlidingWindowTracker::get_latest_state() const {
    const Frame *frame = map->get_frame(map->frame_num() - 1);
    if (!frame->subframes.empty()) frame = frame->subframes.back().get();
    return {frame->image->t, frame->pose, frame->motion};
}

// =====================================================================================================================
// Handler (handler.cpp:15-227)
// =====================================================================================================================
namespace {
void propagate_state(double &state_time, PoseState &pose, MotionState &motion, double t, const V3 &w, const V3 &a) {
    const V3 gravity{0, 0, -GRAVITY_NOMINAL};
    const double dt = t - state_time;
    pose.p = pose.p + dt * motion.v + 0.5 * dt * dt * (gravity + rot(pose.q, a - motion.ba));
    motion.v = motion.v + dt * (gravity + rot(pose.q, a - motion.ba));
    pose.q = normalized(pose.q * expmap((w - motion.bg) * dt));
    state_time = t;
}
}  // namespace

Handler::Handler(Shared &sh) : feature_tracker(sh), frontend(&feature_tracker, sh), sh(sh) { feature_tracker.set_frontend(&frontend); }

PoseState Handler::track_gyroscope(double t, double x, double y, double z) {
    if (!accelerometers.empty()) {
        if (t < accelerometers.front().t) {
            gyroscopes.clear();
        } else {
            while (!accelerometers.empty() && t >= accelerometers.front().t) {
                const Acc acc = accelerometers.front();
                const double lambda = (acc.t - gyroscopes[0].t) / (t - gyroscopes[0].t);
                const V3 w = gyroscopes[0].w + lambda * (V3{x, y, z} - gyroscopes[0].w);
                track_imu({acc.t, w, acc.a});
                accelerometers.pop_front();
            }
            if (!accelerometers.empty())
                while (!gyroscopes.empty() && gyroscopes.front().t < t) gyroscopes.pop_front();
        }
    }
    gyroscopes.push_back({t, {x, y, z}});
    return predict_pose(t);
}

PoseState Handler::track_accelerometer(double t, double x, double y, double z) {
    if (!gyroscopes.empty() && t >= gyroscopes.front().t) {
        if (t > gyroscopes.back().t) {
            while (gyroscopes.size() > 1) gyroscopes.pop_front();
            accelerometers.push_back({t, {x, y, z}});
        } else if (t == gyroscopes.back().t) {
            while (gyroscopes.size() > 1) gyroscopes.pop_front();
            track_imu({t, gyroscopes.front().w, {x, y, z}});
        } else {
            while (t >= gyroscopes[1].t) gyroscopes.pop_front();
            const double lambda = (t - gyroscopes[0].t) / (gyroscopes[1].t - gyroscopes[0].t);
            const V3 w = gyroscopes[0].w + lambda * (gyroscopes[1].w - gyroscopes[0].w);
            track_imu({t, w, {x, y, z}});
        }
    }
    return predict_pose(t);
}

PoseState Handler::track_camera(std::shared_ptr<ImageRef> image) {
    const rdvio_pipeline_config &c = sh.cfg;
    std::unique_ptr<Frame> frame = std::make_unique<Frame>(sh.ids);
    std::copy(c.K, c.K + 9, frame->K);
    frame->image = image;
    frame->sqrt_inv_cov[0] = c.K[0] / std::sqrt(c.keypoint_noise_cov[0]);
    frame->sqrt_inv_cov[1] = c.K[1];
    frame->sqrt_inv_cov[2] = c.K[3];
    frame->sqrt_inv_cov[3] = c.K[4] / std::sqrt(c.keypoint_noise_cov[3]);
    frame->camera.q_cs = {c.q_bc[0], c.q_bc[1], c.q_bc[2], c.q_bc[3]};
    frame->camera.p_cs = {c.p_bc[0], c.p_bc[1], c.p_bc[2]};
    frame->imu.q_cs = {c.q_bi[0], c.q_bi[1], c.q_bi[2], c.q_bi[3]};
    frame->imu.p_cs = {c.p_bi[0], c.p_bi[1], c.p_bi[2]};
    double *nz = frame->preintegration.noise;
    std::copy(c.gyroscope_noise_cov, c.gyroscope_noise_cov + 9, nz);
    std::copy(c.accelerometer_noise_cov, c.accelerometer_noise_cov + 9, nz + 9);
    std::copy(c.gyroscope_bias_noise_cov, c.gyroscope_bias_noise_cov + 9, nz + 18);
    std::copy(c.accelerometer_bias_noise_cov, c.accelerometer_bias_noise_cov + 9, nz + 27);
    std::copy(nz, nz + 36, frame->keyframe_preintegration.noise);
    const double t = image->t;
    frames.emplace_back(std::move(frame));
    return predict_pose(t);
}

void Handler::track_imu(const ImuData &imu) {
    frontal_imus.push_back(imu);
    imus.push_back(imu);
    while (!imus.empty() && !frames.empty()) {
        if (imus.front().t <= frames.front()->image->t) {
            frames.front()->preintegration.data.push_back(imus.front());
            imus.pop_front();
        } else {
            feature_tracker.track_frame(std::move(frames.front()));
            frames.pop_front();
        }
    }
}

PoseState Handler::predict_pose(double t) {
    PoseState out;
    if (auto maybe = feature_tracker.get_latest_state()) {
        auto [state_time, pose, motion] = maybe.value();
        while (!frontal_imus.empty() && frontal_imus.front().t <= state_time) frontal_imus.pop_front();
        for (const ImuData &imu : frontal_imus)
            if (imu.t <= t) propagate_state(state_time, pose, motion, imu.t, imu.w, imu.a);
        const Q4 q_bo{sh.cfg.q_bo[0], sh.cfg.q_bo[1], sh.cfg.q_bo[2], sh.cfg.q_bo[3]};
        out.q = pose.q * q_bo;
        out.p = pose.p + rot(pose.q, V3{sh.cfg.p_bo[0], sh.cfg.p_bo[1], sh.cfg.p_bo[2]});
    } else {
        out.q = {0, 0, 0, 0};
        out.p = {0, 0, 0};
    }
    return out;
}

std::tuple<double, PoseState> Handler::get_latest_state() const {
    PoseState out;
    double timestamp = 0.0;
    if (auto maybe = feature_tracker.get_latest_state()) {
        auto [state_time, pose, motion] = maybe.value();
        (void)motion;
        out = pose;
        timestamp = state_time;
    } else {
        out.q = {0, 0, 0, 0};
        out.p = {0, 0, 0};
    }
    return {timestamp, out};
}

}  // namespace rdvio_pipe

// =====================================================================================================================
// C ABI
// =====================================================================================================================
using namespace rdvio_pipe;

extern "C" {

void rdvio_pipeline_config_default(rdvio_pipeline_config *c) {
    std::memset(c, 0, sizeof *c);
    c->q_bc[3] = c->q_bi[3] = c->q_bo[3] = 1.0;
    c->keypoint_noise_cov[0] = c->keypoint_noise_cov[3] = 1.0;
    // config.cpp:8-82
    c->sliding_window_size = 10;
    c->sliding_window_subframe_size = 3;
    c->sliding_window_force_keyframe_landmarks = 35;
    c->sliding_window_tracker_frequent = 1;
    c->feature_tracker_min_keypoint_distance = 20.0;
    c->feature_tracker_max_keypoint_detection = 150;
    c->feature_tracker_max_init_frames = 60;
    c->feature_tracker_max_frames = 200;
    c->feature_tracker_clahe_clip_limit = 6.0;
    c->feature_tracker_clahe_width = 8;
    c->feature_tracker_clahe_height = 8;
    c->feature_tracker_predict_keypoints = 1;
    c->initializer_keyframe_num = 8;
    c->initializer_keyframe_gap = 5;
    c->initializer_min_matches = 50;
    c->initializer_min_parallax = 10;
    c->initializer_min_triangulation = 50;
    c->initializer_min_landmarks = 30;
    c->solver_iteration_limit = 10;
    c->rotation_misalignment_threshold = 0.1;
    c->rotation_ransac_threshold = 10;
    c->random = 648;
    c->parsac_flag = 0;
    c->parsac_keyframe_check_size = 3;
    c->threading = 0;
    c->initializer_refine_imu = 1;
    c->tracker_gates_on_backend = 0;
}

int rdvio_pipeline_create(rdvio_pipeline **out, const rdvio_pipeline_config *cfg, const rdvio_backend *backend) {
    if (!out || !cfg || !backend) return RDVIO_ERR_INVALID;
    *out = nullptr;
    if (!backend->image_create || !backend->image_preprocess || !backend->image_detect || !backend->image_track || !backend->image_release ||
        !backend->image_destroy || !backend->preintegrate || !backend->ba_solve || !backend->marginalize)
        return RDVIO_ERR_INVALID;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->sliding_window_size < 2 || cfg->sliding_window_tracker_frequent < 1 ||
        cfg->initializer_keyframe_num < 2 || cfg->initializer_keyframe_gap < 1 || cfg->feature_tracker_max_keypoint_detection < 1 ||
        cfg->threading < 0 || cfg->threading > 2)
        return RDVIO_ERR_INVALID;
    auto *p = new rdvio_pipeline();
    p->shared.cfg = *cfg;
    p->shared.backend.fn = *backend;
    if (const char *e = std::getenv("RDVIO_PIPELINE_PROF")) p->shared.prof.on = e[0] == '1';
    parsac_prof() = ParsacProf{};
    parsac_prof().on = p->shared.prof.on;
    p->handler = std::make_unique<Handler>(p->shared);
    *out = p;
    return RDVIO_OK;
}

void rdvio_pipeline_destroy(rdvio_pipeline *p) {
    if (!p) return;
    const rdvio_backend fn = p->shared.backend.fn;
    try { p->handler->frontend.drain(); } catch (...) {}
    if (p->shared.prof.on) {   // diagnostic: inclusive stage times (backend calls inside) and the backend's own share
        const Counters &c = p->shared.counters;
        std::fprintf(stderr, "[rdvio pipeline] %ld frames; inclusive stage times, ms per frame:\n", (long)c.frames_tracked);
        for (int k = 0; k < HostProf::N; ++k)
            if (p->shared.prof.calls[k])
                std::fprintf(stderr, "  %-32s %8.4f  (%ld calls, %.1f us each)\n", HostProf::names[k], 1e3 * p->shared.prof.seconds[k] / std::max<long>(c.frames_tracked, 1),
                             p->shared.prof.calls[k], 1e6 * p->shared.prof.seconds[k] / p->shared.prof.calls[k]);
        static const char *bn[7] = {"preprocess", "detect", "track", "preintegrate", "ba_solve", "marginalize", "image_create"};
        static const char *kn[4] = {"localize_newframe", "refine_window", "refine_subwindow", "other"};
        for (int k = 0; k < 4; ++k)
            if (const SolveProf &sp = p->shared.solve_prof; sp.calls[k])
                std::fprintf(stderr, "  solves %-18s %5ld calls: %.1f iterations, %.0f factors, %.1f frames, %.1f us per call\n", kn[k], sp.calls[k],
                             (double)sp.iterations[k] / sp.calls[k], (double)sp.factors[k] / sp.calls[k], (double)sp.frames[k] / sp.calls[k],
                             1e6 * sp.seconds[k] / sp.calls[k]);
        const ParsacProf &pp = parsac_prof();
        if (pp.solves)
            std::fprintf(stderr, "  parsac: %ld solves, %ld batches, %ld iterations, %ld models, %ld fetches; ms per frame: setup %.4f models %.4f score %.4f fetch %.4f\n",
                         pp.solves, pp.batches, pp.iterations, pp.models, pp.fetches, 1e3 * pp.t_setup / std::max<long>(c.frames_tracked, 1),
                         1e3 * pp.t_models / std::max<long>(c.frames_tracked, 1), 1e3 * pp.t_score / std::max<long>(c.frames_tracked, 1),
                         1e3 * pp.t_fetch / std::max<long>(c.frames_tracked, 1));
        for (int k = 0; k < 7; ++k)
            std::fprintf(stderr, "  backend %-24s %8.4f  (%ld calls)\n", bn[k], 1e3 * c.backend_seconds[k] / std::max<long>(c.frames_tracked, 1), (long)c.backend_calls[k]);
    }
    p->handler.reset();  // frames release their images through the backend first
    delete p;
    if (fn.destroy) fn.destroy(fn.user);
}

const char *rdvio_pipeline_last_error(const rdvio_pipeline *p) { return p ? p->error.c_str() : "null pipeline"; }

int rdvio_pipeline_set_init_states(rdvio_pipeline *p, int n, const double *rows17) {
    if (!p || n < 0 || (n > 0 && !rows17)) return RDVIO_ERR_INVALID;
    p->shared.init_states.resize((size_t)n);
    for (int i = 0; i < n; ++i) std::copy(rows17 + 17 * (size_t)i, rows17 + 17 * (size_t)(i + 1), p->shared.init_states[(size_t)i].begin());
    return RDVIO_OK;
}

#define PIPE_GUARD(stmt)                          \
    try {                                         \
        stmt;                                     \
    } catch (const std::exception &e) {           \
        p->error = e.what();                      \
        return RDVIO_ERR_HIP;                     \
    }

static void store_pose(const PoseState &ps, double *pose7) {
    pose7[0] = ps.q.x; pose7[1] = ps.q.y; pose7[2] = ps.q.z; pose7[3] = ps.q.w;
    pose7[4] = ps.p.x; pose7[5] = ps.p.y; pose7[6] = ps.p.z;
}

int rdvio_pipeline_add_frame(rdvio_pipeline *p, double t, const uint8_t *gray, int width, int height, int stride, double *pose_out) {
    if (!p || !gray || width != p->shared.cfg.width || height != p->shared.cfg.height || stride < width) {
        if (p) p->error = "add_frame: image shape does not match the configured camera resolution";
        return RDVIO_ERR_INVALID;
    }
    PIPE_GUARD({
        Backend &be = p->shared.backend;
        auto image = std::make_shared<ImageRef>();
        image->backend = &be;
        image->t = t;
        image->width = width;
        image->height = height;
        {
            BackendTimer timer(p->shared.counters, 6);
            be.check(be.fn.image_create(be.fn.user, gray, width, height, stride, &image->handle), "image_create");
        }
        const PoseState ps = p->handler->track_camera(image);
        if (pose_out) store_pose(ps, pose_out);
    })
    return RDVIO_OK;
}

int rdvio_pipeline_add_gyro(rdvio_pipeline *p, double t, const double *gyro) {
    if (!p || !gyro) return RDVIO_ERR_INVALID;
    PIPE_GUARD(p->handler->track_gyroscope(t, gyro[0], gyro[1], gyro[2]))
    return RDVIO_OK;
}

int rdvio_pipeline_add_acc(rdvio_pipeline *p, double t, const double *acc) {
    if (!p || !acc) return RDVIO_ERR_INVALID;
    PIPE_GUARD(p->handler->track_accelerometer(t, acc[0], acc[1], acc[2]))
    return RDVIO_OK;
}

int rdvio_pipeline_add_motion(rdvio_pipeline *p, double t, const double *acc, const double *gyro) {
    if (int rc = rdvio_pipeline_add_gyro(p, t, gyro)) return rc;  // gyro first (rdvio.hpp:59-60)
    return rdvio_pipeline_add_acc(p, t, acc);
}

int rdvio_pipeline_state(const rdvio_pipeline *p) { return p ? p->handler->frontend.get_system_state() : 3; }

int rdvio_pipeline_latest_state(const rdvio_pipeline *p, double *t, double *pose7) {
    if (!p) return 0;
    auto [ts, ps] = p->handler->get_latest_state();
    if (t) *t = ts;
    if (pose7) store_pose(ps, pose7);
    return p->handler->feature_tracker.get_latest_state().has_value() ? 1 : 0;
}

int rdvio_pipeline_window_state(const rdvio_pipeline *p, double *t, double *state16) {
    if (!p) return 0;
    // Frontend::get_latest_state (frontend.cpp:79-83): the newest state the frontend has handed to the tracker -- in inline
    // mode SlidingWindowTracker::get_latest_state() of the step that just ran, in the pipelined modes the published one
    auto [ts, frame_id, pose, motion] = p->handler->frontend.get_latest_state();
    if (frame_id == nil) return 0;
    if (t) *t = ts;
    if (state16) {
        store_pose(pose, state16);
        const double m[9] = {motion.v.x, motion.v.y, motion.v.z, motion.bg.x, motion.bg.y, motion.bg.z, motion.ba.x, motion.ba.y, motion.ba.z};
        std::copy(m, m + 9, state16 + 7);
    }
    return 1;
}

int rdvio_pipeline_transform_world_cam(const rdvio_pipeline *p, double *T) {
    if (!p || !T) return RDVIO_ERR_INVALID;
    // rdvio.hpp:71-77: T_imu_to_cv * Twb * T_cam_to_body with T_imu_to_cv = [1 0 0; 0 0 -1; 0 1 0] (rdvio.hpp:16-23)
    auto [ts, ps] = p->handler->get_latest_state();
    (void)ts;
    const rdvio_pipeline_config &c = p->shared.cfg;
    auto make = [](const Q4 &q, const V3 &t, double *M) {
        const M3 R = (q.x == 0 && q.y == 0 && q.z == 0 && q.w == 0) ? M3{{0, 0, 0, 0, 0, 0, 0, 0, 0}} : to_mat(q);
        const double v[16] = {R.m[0], R.m[1], R.m[2], t.x, R.m[3], R.m[4], R.m[5], t.y, R.m[6], R.m[7], R.m[8], t.z, 0, 0, 0, 1};
        std::copy(v, v + 16, M);
    };
    auto mul = [](const double *A, const double *B, double *C) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0;
                for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
                C[4 * i + j] = s;
            }
    };
    double Twb[16], Tcb[16], tmp[16];
    make(ps.q, ps.p, Twb);
    make(Q4{c.q_bc[0], c.q_bc[1], c.q_bc[2], c.q_bc[3]}, V3{c.p_bc[0], c.p_bc[1], c.p_bc[2]}, Tcb);
    const double Ticv[16] = {1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, 1};
    mul(Ticv, Twb, tmp);
    mul(tmp, Tcb, T);
    return RDVIO_OK;
}

int rdvio_pipeline_local_map(const rdvio_pipeline *p, double *xyz, int capacity) {
    if (!p) return 0;
    p->handler->frontend.drain();  // the frontend's step in flight owns the sliding-window map
    if (!p->handler->frontend.sliding_window_tracker) return 0;
    const Map *map = p->handler->frontend.sliding_window_tracker->map.get();
    int n = 0;
    for (size_t i = 0; i < map->track_num(); ++i) {
        const Track *track = map->get_track(i);
        if (!track->all_tagged({TT_VALID, TT_TRIANGULATED})) continue;
        if (xyz && n < capacity) {
            const V3 x = track->get_landmark_point();
            xyz[3 * n] = x.x; xyz[3 * n + 1] = x.y; xyz[3 * n + 2] = x.z;
        }
        ++n;
    }
    return n;
}

int rdvio_pipeline_last_frame_keypoints(const rdvio_pipeline *p, int64_t *track_ids, double *xy, int capacity) {
    if (!p) return 0;
    const Map *map = p->handler->feature_tracker.map.get();
    if (map->frame_num() == 0) return 0;
    const Frame *frame = map->get_frame(map->frame_num() - 1);
    const int n = (int)frame->keypoint_num();
    for (int i = 0; i < n && i < capacity; ++i) {
        if (track_ids) track_ids[i] = frame->get_track((size_t)i) ? (int64_t)frame->get_track((size_t)i)->id() : -1;
        if (xy) {
            const V2 px = apply_k(frame->get_keypoint((size_t)i), frame->K);
            xy[2 * i] = px.x;
            xy[2 * i + 1] = px.y;
        }
    }
    return n;
}

int rdvio_pipeline_replay(rdvio_pipeline *p, rdvio_replay *r) {
    if (!p || !r || r->n_frames < 0 || r->n_imu < 0 || (r->n_frames > 0 && (!r->frames || !r->frame_t)) || (r->n_imu > 0 && !r->imu)) return RDVIO_ERR_INVALID;
    if ((r->kp_ids || r->kp_xy) && r->kp_capacity <= 0) return RDVIO_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    const double nan = std::nan("");
    int64_t seen = p->shared.counters.frames_tracked;
    // frames pushed before this call and not yet consumed are consumed (and recorded) here as well: count from what the
    // tracker HAS consumed plus what the handler still queues
    const int64_t seen0 = seen + (int64_t)p->handler->queued_frames();
    int rows = 0;
    auto record = [&]() {   // the feature tracker consumed another frame
        if (rows >= r->n_frames) return;
        const size_t k = (size_t)rows++;
        if (r->kp_n || r->kp_ids || r->kp_xy) {
            const int n = rdvio_pipeline_last_frame_keypoints(p, r->kp_ids ? r->kp_ids + k * (size_t)r->kp_capacity : nullptr,
                                                              r->kp_xy ? r->kp_xy + 2 * k * (size_t)r->kp_capacity : nullptr, r->kp_capacity);
            if (r->kp_n) r->kp_n[k] = n;
        }
        if (r->latest) {
            double *o = r->latest + 8 * k;
            if (!rdvio_pipeline_latest_state(p, o, o + 1)) std::fill(o, o + 8, nan);
        }
        if (r->window) {
            double *o = r->window + 17 * k;
            if (!rdvio_pipeline_window_state(p, o, o + 1)) std::fill(o, o + 17, nan);
        }
        if (r->sys_state) r->sys_state[k] = rdvio_pipeline_state(p);
        if (r->done_s) r->done_s[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    int ii = 0;
    auto push_imu = [&](const double *row) {
        const int rc = rdvio_pipeline_add_motion(p, row[0], row + 4, row + 1);
        if (rc == RDVIO_OK && p->shared.counters.frames_tracked != seen) {
            seen = p->shared.counters.frames_tracked;
            record();
        }
        return rc;
    };
    int rc = RDVIO_OK;
    for (int k = 0; k < r->n_frames && rc == RDVIO_OK; ++k) {
        while (ii < r->n_imu && r->imu[7 * (size_t)ii] <= r->frame_t[k] && rc == RDVIO_OK) rc = push_imu(r->imu + 7 * (size_t)ii++);
        if (rc == RDVIO_OK) rc = rdvio_pipeline_add_frame(p, r->frame_t[k], r->frames[k], r->width, r->height, r->stride, nullptr);
    }
    const int64_t pushed = seen0 + r->n_frames;
    while (ii < r->n_imu && rc == RDVIO_OK && (r->flush == 2 || (r->flush == 1 && p->shared.counters.frames_tracked < pushed)))
        rc = push_imu(r->imu + 7 * (size_t)ii++);
    r->imu_consumed = ii;
    r->frames_processed = rows;
    r->elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

int rdvio_pipeline_drain(rdvio_pipeline *p) {
    if (!p) return RDVIO_ERR_INVALID;
    PIPE_GUARD(p->handler->frontend.drain())
    return RDVIO_OK;
}

int rdvio_pipeline_counters(const rdvio_pipeline *p, int64_t *out) {
    if (!p || !out) return RDVIO_ERR_INVALID;
    p->handler->frontend.drain();
    const Counters &c = p->shared.counters;
    const Map *ft = p->handler->feature_tracker.map.get();
    out[0] = c.frames_tracked; out[1] = c.window_solves; out[2] = c.keyframes; out[3] = c.marginalizations;
    out[4] = c.localizations; out[5] = c.subwindow_solves;
    out[6] = ft->frame_num() ? (int64_t)ft->get_frame(ft->frame_num() - 1)->id() : -1;
    out[7] = p->handler->frontend.sliding_window_tracker ? (int64_t)p->handler->frontend.sliding_window_tracker->map->track_num() : 0;
    out[8] = c.max_problem_frames;
    out[9] = c.max_problem_factors;
    out[10] = c.solver_iterations;
    out[25] = c.no_translation_frames;
    out[26] = c.rotation_prior_factors;
    out[27] = c.parsac_judgements;
    out[28] = c.tracks_marked_dynamic;
    for (int k = 0; k < 7; ++k) {
        const bool pre = k == 3;  // microseconds and calls per backend call class
        const Backend &be = p->shared.backend;
        out[11 + 2 * k] = (int64_t)(1e6 * (pre ? be.preintegrate_seconds[0] + be.preintegrate_seconds[1] : c.backend_seconds[k]));
        out[12 + 2 * k] = pre ? be.preintegrate_calls[0] + be.preintegrate_calls[1] : c.backend_calls[k];
    }
    return RDVIO_OK;
}

}  // extern "C"
