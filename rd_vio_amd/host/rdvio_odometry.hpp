// rdvio_odometry.hpp -- dependency-free C++17 mirror of rdvio::Odometry
// (/root/reference/src/rdvio/include/rdvio/rdvio.hpp:25-115) on top of librdvio_pipeline.so + librdvio_hip.so.
//
// Same surface and behaviour as the reference class: addFrame(t, image), addMotion(t, acc, gyro) (gyro is pushed first,
// rdvio.hpp:59-60), addAcc, addGyro, transform_world_cam() (T_imu_to_cv * T_wb * T_cam_to_body, rdvio.hpp:71-77),
// state() (0 initialising / 1 tracking / 2 crash / 3 unknown), local_map() (R_imu_to_cv applied, rdvio.hpp:91-97),
// keypoints() (always empty in the reference: its loop bound is the never-filled keymap, feature_tracker.cpp:264).
// std::array / raw pointers stand where the reference uses Eigen / cv::Mat (neither exists in this build image); an
// Eigen/OpenCV-typed facade is the 20-line adapter shown in INTEGRATION.md.  addFrame throws std::runtime_error for a
// channel count other than 1, 3 or 4 (rdvio.hpp:47-48); configuration / HIP failures throw as well.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rdvio_pipeline.h"
#include "rdvio_yaml.hpp"

namespace rdvio_hip {

class Odometry {
  public:
    // rdvio::Odometry(calib, config) (rdvio.hpp:27-37): the sensor calibration file (configs/euroc_sensor.yaml) and the
    // SLAM settings file (configs/setting.yaml), read like rdvio::extra::YamlConfig(config, calib) reads them; throws the
    // Yaml*Exception kinds of rdvio_yaml.hpp for unreadable files, missing mandatory keys and wrongly typed values
    Odometry(const std::string &calib, const std::string &config, int device = 0, int max_features = 4096, int max_factors = 40000)
        : Odometry(load_yaml_config(config, calib), device, max_features, max_factors) {}
    // cfg: rdvio::Config values (rdvio_pipeline_config_default + calibration) already in the pipeline's struct
    explicit Odometry(const rdvio_pipeline_config &cfg, int device = 0, int max_features = 4096, int max_factors = 40000) : cfg_(cfg) {
        int rc = rdvio_hip_ctx_create(&ctx_, device, cfg.width, cfg.height, max_features, cfg.sliding_window_size > 16 ? cfg.sliding_window_size : 16,
                                      max_factors, nullptr);
        if (rc != RDVIO_OK) {
            const std::string msg = ctx_ ? rdvio_hip_last_error(ctx_) : "no HIP device";
            if (ctx_) rdvio_hip_ctx_destroy(ctx_);
            throw std::runtime_error("rdvio_hip_ctx_create failed: " + msg);
        }
        rc = rdvio_pipeline_create_hip(&vio_, &cfg_, ctx_);
        if (rc != RDVIO_OK) {
            rdvio_hip_ctx_destroy(ctx_);
            throw std::runtime_error("rdvio_pipeline_create_hip failed (unsupported configuration?)");
        }
    }
    ~Odometry() {
        rdvio_pipeline_destroy(vio_);
        rdvio_hip_ctx_destroy(ctx_);
    }
    Odometry(const Odometry &) = delete;
    Odometry &operator=(const Odometry &) = delete;

    // image: rows x cols x channels u8, row stride in bytes; 3 / 4 channels are BGR / BGRA like cv::cvtColor's
    // COLOR_BGR2GRAY (0.114 B + 0.587 G + 0.299 R, rounded)
    void addFrame(double t, const uint8_t *image, int cols, int rows, int channels, int stride_bytes) {
        if (channels != 1 && channels != 3 && channels != 4) throw std::runtime_error("Invalid image channel, must be 1, 3 or 4");
        const uint8_t *gray = image;
        int stride = stride_bytes;
        if (channels != 1) {
            gray_.resize((size_t)cols * rows);
            for (int y = 0; y < rows; ++y)
                for (int x = 0; x < cols; ++x) {
                    const uint8_t *p = image + (size_t)y * stride_bytes + (size_t)x * channels;
                    // OpenCV's fixed-point weights (B 1868, G 9617, R 4899, shift 14)
                    gray_[(size_t)y * cols + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
                }
            gray = gray_.data();
            stride = cols;
        }
        check(rdvio_pipeline_add_frame(vio_, t, gray, cols, rows, stride, nullptr));
    }
    void addMotion(double t, const std::array<double, 3> &acc, const std::array<double, 3> &gyro) {
        check(rdvio_pipeline_add_motion(vio_, t, acc.data(), gyro.data()));
    }
    void addAcc(double t, const std::array<double, 3> &acc) { check(rdvio_pipeline_add_acc(vio_, t, acc.data())); }
    void addGyro(double t, const std::array<double, 3> &gyro) { check(rdvio_pipeline_add_gyro(vio_, t, gyro.data())); }

    // 4x4 row-major
    std::array<double, 16> transform_world_cam() const {
        std::array<double, 16> T{};
        rdvio_pipeline_transform_world_cam(vio_, T.data());
        return T;
    }
    int state() const { return rdvio_pipeline_state(vio_); }
    std::vector<std::array<double, 3>> local_map() const {
        const int n = rdvio_pipeline_local_map(vio_, nullptr, 0);
        std::vector<double> xyz((size_t)n * 3);
        rdvio_pipeline_local_map(vio_, xyz.data(), n);
        std::vector<std::array<double, 3>> pts((size_t)n);
        for (int i = 0; i < n; ++i) pts[(size_t)i] = {xyz[3 * i], -xyz[3 * i + 2], xyz[3 * i + 1]};  // R_imu_to_cv = [1 0 0; 0 0 -1; 0 1 0]
        return pts;
    }
    std::vector<std::array<int, 2>> keypoints() const { return {}; }

    rdvio_pipeline *handle() const { return vio_; }  // bootstrap states, diagnostics

  private:
    void check(int rc) const {
        if (rc != RDVIO_OK) throw std::runtime_error(std::string("rdvio_pipeline: ") + rdvio_pipeline_last_error(vio_));
    }
    rdvio_pipeline_config cfg_;
    rdvio_hip_ctx *ctx_ = nullptr;
    rdvio_pipeline *vio_ = nullptr;
    std::vector<uint8_t> gray_;
};

}  // namespace rdvio_hip
