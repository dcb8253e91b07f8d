// rdvio_hip.hpp -- dependency-free C++17 host mirror of the reference's plugin interfaces for the hot path,
// implemented on the C ABI of include/rdvio_hip.h (librdvio_hip.so).
//
// The reference's own host code is C++ with Eigen/OpenCV types; neither library exists in this build image, so
// this mirror uses std::array / std::vector where the reference uses Eigen::Vector / cv::Mat.  Method names,
// argument meaning and error behaviour follow the reference:
//   rdvio_hip::Image          <-> rdvio::Image / extra::OpenCvImage   (src/rdvio/include/rdvio/types.h:153-177,
//                                                                      src/rdvio_extra/src/opencv_image.cpp:38-161)
//   rdvio_hip::PreIntegrator  <-> rdvio::PreIntegrator                (src/rdvio_estimation/include/rdvio/estimation/preintegrator.h:10-47)
//   rdvio_hip::Solver         <-> rdvio::Solver                       (src/rdvio_estimation/include/rdvio/estimation/solver.h:15-70)
//   rdvio_hip::MarginalizationFactor <-> rdvio::MarginalizationFactor (.../marginalization_factor.h:9-40)
// INTEGRATION.md shows the Eigen/OpenCV-typed subclasses a maintainer adds inside the reference tree.
// Errors: configuration / capacity / HIP failures throw std::runtime_error (the reference throws only for
// config/IO, rdvio.hpp:47-48); estimation "failures" are reported through return values like Solver::solve.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rdvio_hip.h"

namespace rdvio_hip {

using vec2 = std::array<double, 2>;
using vec3 = std::array<double, 3>;
using State = std::array<double, RDVIO_STATE_SIZE>;  // q(x,y,z,w) p v bg ba

class Context {
  public:
    Context(int max_width, int max_height, int max_features = 1024, int max_window = 16, int max_factors = 16384,
            int device = 0, void *stream = nullptr) {
        int rc = rdvio_hip_ctx_create(&h_, device, max_width, max_height, max_features, max_window, max_factors, stream);
        if (rc != RDVIO_OK) {
            std::string msg = h_ ? rdvio_hip_last_error(h_) : "no HIP device or bad arguments";
            if (h_) rdvio_hip_ctx_destroy(h_);
            h_ = nullptr;
            throw std::runtime_error("rdvio_hip_ctx_create failed: " + msg);
        }
    }
    ~Context() { rdvio_hip_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    rdvio_hip_ctx *handle() const { return h_; }
    void check(int rc) const {
        if (rc != RDVIO_OK) throw std::runtime_error(std::string("rdvio_hip: ") + rdvio_hip_last_error(h_));
    }

  private:
    rdvio_hip_ctx *h_ = nullptr;
};

// rdvio::Image: one camera frame living in a context image slot (0/1, double-buffered like the tracker's two
// consecutive frames, feature_tracker.cpp:94).
class Image {
  public:
    Image(Context &ctx, int slot, const uint8_t *gray, int width, int height, int stride, double t = 0.0)
        : t(t), ctx_(ctx), slot_(slot), w_(width), h_(height), gray_(gray, gray + (size_t)stride * height), stride_(stride) {}

    double t;
    size_t width() const { return (size_t)w_; }
    size_t height() const { return (size_t)h_; }
    size_t level_num() const { return 3; }  // opencv_image.h:19
    const uint8_t *get_rawdata() const { return gray_.data(); }

    // Image::preprocess(clipLimit, width, height): CLAHE then 4-level pyramid with Scharr derivatives
    void preprocess(double clipLimit, int width, int height) {
        ctx_.check(rdvio_hip_image_preprocess(ctx_.handle(), slot_, gray_.data(), w_, h_, stride_, clipLimit, width, height));
    }
    // Image::detect_keypoints(keypoints in/out, max_points, keypoint_distance)
    void detect_keypoints(std::vector<vec2> &keypoints, size_t max_points = 1000, double keypoint_distance = 10) const {
        const size_t n0 = keypoints.size();
        keypoints.resize(n0 + max_points);
        int n_out = 0;
        ctx_.check(rdvio_hip_detect_keypoints(ctx_.handle(), slot_, keypoints.data()->data(), (int)n0, (int)keypoints.size(),
                                              (int)max_points, keypoint_distance, &n_out));
        keypoints.resize((size_t)n_out);
    }
    // Image::track_keypoints(next_image, curr, next in/out, status): an empty next_keypoints means "no initial
    // guess" exactly as in opencv_image.cpp:79-85
    void track_keypoints(const Image *next_image, const std::vector<vec2> &curr_keypoints, std::vector<vec2> &next_keypoints,
                         std::vector<char> &result_status) const {
        const bool has_guess = !next_keypoints.empty();
        if (!has_guess) next_keypoints.resize(curr_keypoints.size());
        if (next_keypoints.size() != curr_keypoints.size()) throw std::runtime_error("track_keypoints: size mismatch");
        result_status.assign(curr_keypoints.size(), 0);
        if (!next_image || curr_keypoints.empty()) return;
        std::vector<uint8_t> st(curr_keypoints.size());
        ctx_.check(rdvio_hip_track_keypoints(ctx_.handle(), slot_, next_image->slot_, (int)curr_keypoints.size(),
                                             curr_keypoints.data()->data(), next_keypoints.data()->data(), has_guess ? 1 : 0,
                                             st.data()));
        for (size_t i = 0; i < st.size(); ++i) result_status[i] = (char)st[i];
    }
    void release_image_buffer() { ctx_.check(rdvio_hip_image_release(ctx_.handle(), slot_)); }
    int slot() const { return slot_; }

  private:
    Context &ctx_;
    int slot_, w_, h_;
    std::vector<uint8_t> gray_;
    int stride_;
};

struct ImuData {
    double t;
    vec3 w, a;
};

// rdvio::PreIntegrator (preintegrator.h:10-47): same members, flat arrays instead of Eigen matrices.
struct PreIntegrator {
    explicit PreIntegrator(Context &ctx) : ctx(ctx) { reset(); }
    void reset() {
        record.assign(RDVIO_PREINT_SIZE, 0.0);
        record[4] = 1.0;  // delta.q = identity
    }
    // integrate(t, bg, ba, compute_jacobian, compute_covariance); false if there is no data (:80-81)
    bool integrate(double t, const vec3 &bg, const vec3 &ba, bool compute_jacobian, bool compute_covariance) {
        if (data.empty()) return false;
        std::vector<double> imu(data.size() * 7);
        for (size_t i = 0; i < data.size(); ++i) {
            imu[7 * i] = data[i].t;
            for (int k = 0; k < 3; ++k) {
                imu[7 * i + 1 + k] = data[i].w[k];
                imu[7 * i + 4 + k] = data[i].a[k];
            }
        }
        const int32_t off[2] = {0, (int32_t)data.size()};
        double noise[36];
        for (int i = 0; i < 9; ++i) {
            noise[i] = cov_w[i];
            noise[9 + i] = cov_a[i];
            noise[18 + i] = cov_bg[i];
            noise[27 + i] = cov_ba[i];
        }
        ctx.check(rdvio_hip_preintegrate(ctx.handle(), 1, off, imu.data(), &t, bg.data(), ba.data(), noise, compute_jacobian,
                                         compute_covariance, record.data()));
        return true;
    }
    double delta_t() const { return record[0]; }
    const double *delta_q() const { return &record[1]; }
    const double *delta_p() const { return &record[5]; }
    const double *delta_v() const { return &record[8]; }
    const double *cov() const { return &record[11]; }
    const double *sqrt_inv_cov() const { return &record[236]; }
    const double *jacobian() const { return &record[461]; }  // dq_dbg dp_dbg dp_dba dv_dbg dv_dba

    std::array<double, 9> cov_w{}, cov_a{}, cov_bg{}, cov_ba{};  // continuous noise covariances (3x3 row-major)
    std::vector<ImuData> data;
    std::vector<double> record;  // RDVIO_PREINT_SIZE
    Context &ctx;
};

// rdvio::MarginalizationFactor state: the sqrt prior and its linearisation frames.
struct MarginalizationFactor {
    std::vector<int32_t> frames;  // indices into the solver's / map's frame array
    std::vector<double> pose_motion_linearization_point;  // frames.size() x 16
    std::vector<double> sqrt_inv_cov, infovec;            // (15 n)^2, 15 n
    // MarginalizationFactor(map): prior over all frames but the newest, frame 0 pose pinned with 1e15
    // (marginalization_factor.h:14-31)
    static MarginalizationFactor initial(const std::vector<State> &map_frames) {
        MarginalizationFactor m;
        const size_t n = map_frames.size() - 1, D = 15 * n;
        m.frames.resize(n);
        m.pose_motion_linearization_point.resize(n * 16);
        for (size_t i = 0; i < n; ++i) {
            m.frames[i] = (int32_t)i;
            for (int k = 0; k < 16; ++k) m.pose_motion_linearization_point[16 * i + k] = map_frames[i][k];
        }
        m.sqrt_inv_cov.assign(D * D, 0.0);
        m.infovec.assign(D, 0.0);
        for (int k = 0; k < 6; ++k) m.sqrt_inv_cov[(size_t)k * D + k] = 1.0e15;
        return m;
    }
};

// rdvio::Solver: collects states and factors (indices instead of Frame*/Track* pointers), then solve().
class Solver {
  public:
    explicit Solver(Context &ctx, int iteration_limit = 10 /* Config::solver_iteration_limit, config.cpp:53 */)
        : ctx_(ctx), iteration_limit_(iteration_limit) {}

    // add_frame_states(frame): returns the frame index used by the factors; `fixed` = FT_FIX_POSE|FT_FIX_MOTION
    int add_frame_states(const State &s, bool fixed = false) {
        states_.push_back(s);
        frame_fixed_.push_back(fixed ? 1 : 0);
        return (int)states_.size() - 1;
    }
    // add_track_states(track): z_ref = bearing in the anchor frame, inv_depth; `fixed` for the prior flavours
    int add_track_states(const vec3 &z_ref, double inv_depth, bool fixed = false) {
        z_ref_.insert(z_ref_.end(), z_ref.begin(), z_ref.end());
        inv_depth_.push_back(inv_depth);
        lm_fixed_.push_back(fixed ? 1 : 0);
        return (int)inv_depth_.size() - 1;
    }
    // add_factor(ReprojectionErrorFactor): observation of landmark `track` in frame `frame`, anchored in `anchor`;
    // local_tangent = [b1 b2 z_obs] row-major (reprojection_factor.h:16-22).  Factors of one track must be added
    // consecutively (the reference iterates a track's keypoint_map the same way).
    void add_factor_reprojection(int frame, int anchor, int track, const std::array<double, 9> &local_tangent) {
        tgt_.push_back(frame);
        ref_.push_back(anchor);
        lm_.push_back(track);
        tangent_.insert(tangent_.end(), local_tangent.begin(), local_tangent.end());
    }
    void add_factor_rotation_prior(int frame, int anchor, const vec3 &z_ref, const std::array<double, 9> &local_tangent) {
        rot_tgt_.push_back(frame);
        rot_ref_.push_back(anchor);
        rot_zref_.insert(rot_zref_.end(), z_ref.begin(), z_ref.end());
        rot_tangent_.insert(rot_tangent_.end(), local_tangent.begin(), local_tangent.end());
    }
    void add_factor_preintegration(int frame_i, int frame_j, const PreIntegrator &pre) {
        pre_i_.push_back(frame_i);
        pre_j_.push_back(frame_j);
        preint_.insert(preint_.end(), pre.record.begin(), pre.record.end());
    }
    void add_factor(const MarginalizationFactor &m) { prior_ = &m; }
    void set_camera(const std::array<double, 14> &extrinsics, const std::array<double, 4> &sqrt_inv_cov) {
        extr_ = extrinsics;
        W_ = sqrt_inv_cov;
    }
    // bool solve(): like Solver::solve returns summary.IsSolutionUsable(); states/inverse depths are updated in place
    bool solve(rdvio_ba_summary *summary = nullptr) {
        rdvio_ba_problem pb{};
        pb.n_frames = (int32_t)states_.size();
        pb.states = states_.data()->data();
        pb.frame_fixed = frame_fixed_.data();
        pb.extr = extr_.data();
        pb.sqrt_inv_cov = W_.data();
        pb.n_landmarks = (int32_t)inv_depth_.size();
        pb.z_ref = z_ref_.data();
        pb.inv_depth = inv_depth_.data();
        pb.lm_fixed = lm_fixed_.data();
        pb.n_factors = (int32_t)tgt_.size();
        pb.tgt = tgt_.data(); pb.ref = ref_.data(); pb.lm = lm_.data(); pb.tangent = tangent_.data();
        pb.n_rot = (int32_t)rot_tgt_.size();
        pb.rot_tgt = rot_tgt_.data(); pb.rot_ref = rot_ref_.data(); pb.rot_zref = rot_zref_.data(); pb.rot_tangent = rot_tangent_.data();
        pb.n_preint = (int32_t)pre_i_.size();
        pb.pre_i = pre_i_.data(); pb.pre_j = pre_j_.data(); pb.preint = preint_.data();
        if (prior_) {
            pb.n_prior = (int32_t)prior_->frames.size();
            pb.prior_frames = prior_->frames.data();
            pb.prior_lin = prior_->pose_motion_linearization_point.data();
            pb.prior_S = prior_->sqrt_inv_cov.data();
            pb.prior_f = prior_->infovec.data();
        }
        rdvio_ba_summary sm{};
        std::vector<State> out(states_.size());
        std::vector<double> invd(inv_depth_.size());
        ctx_.check(rdvio_hip_ba_solve(ctx_.handle(), &pb, iteration_limit_, out.data()->data(), invd.data(), &sm));
        states_ = out;
        inv_depth_ = invd;
        if (summary) *summary = sm;
        return sm.termination != 2;
    }
    const State &frame_state(int i) const { return states_[(size_t)i]; }
    double inv_depth(int l) const { return inv_depth_[(size_t)l]; }

  private:
    Context &ctx_;
    int iteration_limit_;
    std::vector<State> states_;
    std::vector<uint8_t> frame_fixed_, lm_fixed_;
    std::vector<double> z_ref_, inv_depth_, tangent_, rot_zref_, rot_tangent_, preint_;
    std::vector<int32_t> tgt_, ref_, lm_, rot_tgt_, rot_ref_, pre_i_, pre_j_;
    std::array<double, 14> extr_{};
    std::array<double, 4> W_{};
    const MarginalizationFactor *prior_ = nullptr;
};

}  // namespace rdvio_hip
