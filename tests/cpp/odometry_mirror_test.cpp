// Compile / run check of the rdvio::Odometry mirror (rd_vio_amd/host/rdvio_odometry.hpp).  Needs a GPU to run.
#include <cmath>
#include <cstdio>

#include "../../rd_vio_amd/host/rdvio_odometry.hpp"

int main() {
    rdvio_pipeline_config cfg;
    rdvio_pipeline_config_default(&cfg);
    cfg.width = 376;
    cfg.height = 240;
    const double K[9] = {229.327, 0, 183.6075, 0, 228.648, 124.1875, 0, 0, 1};
    for (int i = 0; i < 9; ++i) cfg.K[i] = K[i];
    for (int i = 0; i < 3; ++i) cfg.gyroscope_noise_cov[4 * i] = 2.88e-8, cfg.accelerometer_noise_cov[4 * i] = 4e-6,
                                cfg.gyroscope_bias_noise_cov[4 * i] = 3.76e-10, cfg.accelerometer_bias_noise_cov[4 * i] = 9e-6;
    rdvio_hip::Odometry vio(cfg);
    if (vio.state() != 0) return 1;                      // SYS_INITIALIZING
    std::vector<uint8_t> bgr(376 * 240 * 3);
    for (int y = 0; y < 240; ++y)
        for (int x = 0; x < 376; ++x)
            for (int c = 0; c < 3; ++c) bgr[(size_t)(y * 376 + x) * 3 + c] = (uint8_t)(110 + 60 * std::sin(0.13 * x + 0.2 * c) * std::cos(0.09 * y));
    for (int k = 0; k < 6; ++k) {
        for (int j = 0; j < 10; ++j) vio.addMotion(1.0 + 0.05 * k + 0.005 * j, {0, 0, 9.80665}, {0, 0, 0});
        vio.addFrame(1.0 + 0.05 * k + 0.05, bgr.data(), 376, 240, 3, 376 * 3);
    }
    bool threw = false;
    try {
        vio.addFrame(2.0, bgr.data(), 376, 240, 2, 376 * 2);
    } catch (const std::runtime_error &) {
        threw = true;                                      // rdvio.hpp:47-48
    }
    if (!threw || vio.state() != 0 || !vio.keypoints().empty() || !vio.local_map().empty()) return 2;
    const auto T = vio.transform_world_cam();
    if (T[15] != 1.0) return 3;
    std::printf("OK odometry mirror\n");
    return 0;
}
