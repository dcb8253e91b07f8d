// ThreadSanitizer driver of the host pipeline (test infrastructure): a pre-rendered synthetic stream (scripts/tsan_pipeline.sh writes it)
// through rdvio_pipeline over the CPU oracle backend with the frontend's step on its worker thread (threading = 2).  Built with
// -fsanitize=thread together with the pipeline sources; prints the last state and exits 0 -- the sanitizer reports to stderr and
// turns a race into a non-zero exit code.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rdvio_pipeline.h"

extern "C" void rdvio_oracle_backend_fill(rdvio_backend *b);

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[6];   // n_frames, w, h, n_imu, threading, parsac
    if (std::fread(hdr, sizeof hdr, 1, f) != 1) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2], ni = hdr[3];
    std::vector<double> K(9), extr(14), noise(36), ts(n), imu((size_t)ni * 7), gt((size_t)n * 17);
    std::vector<uint8_t> frames((size_t)n * w * h);
    if (std::fread(K.data(), 8, 9, f) != 9 || std::fread(extr.data(), 8, 14, f) != 14 || std::fread(noise.data(), 8, 36, f) != 36 ||
        std::fread(ts.data(), 8, n, f) != (size_t)n || std::fread(imu.data(), 8, (size_t)ni * 7, f) != (size_t)ni * 7 ||
        std::fread(gt.data(), 8, (size_t)n * 17, f) != (size_t)n * 17 || std::fread(frames.data(), 1, frames.size(), f) != frames.size())
        return 2;
    std::fclose(f);
    rdvio_pipeline_config cfg;
    rdvio_pipeline_config_default(&cfg);
    cfg.width = w; cfg.height = h;
    std::memcpy(cfg.K, K.data(), 72);
    std::memcpy(cfg.q_bc, &extr[0], 32); std::memcpy(cfg.p_bc, &extr[4], 24);
    std::memcpy(cfg.q_bi, &extr[7], 32); std::memcpy(cfg.p_bi, &extr[11], 24);
    cfg.keypoint_noise_cov[0] = cfg.keypoint_noise_cov[3] = 0.5; cfg.keypoint_noise_cov[1] = cfg.keypoint_noise_cov[2] = 0.0;
    std::memcpy(cfg.gyroscope_noise_cov, &noise[0], 72); std::memcpy(cfg.accelerometer_noise_cov, &noise[9], 72);
    std::memcpy(cfg.gyroscope_bias_noise_cov, &noise[18], 72); std::memcpy(cfg.accelerometer_bias_noise_cov, &noise[27], 72);
    cfg.sliding_window_size = 8; cfg.feature_tracker_max_keypoint_detection = 150; cfg.feature_tracker_min_keypoint_distance = 10.0;
    cfg.solver_iteration_limit = 30; cfg.initializer_keyframe_gap = 2; cfg.feature_tracker_max_frames = 20;
    cfg.sliding_window_force_keyframe_landmarks = 50; cfg.sliding_window_subframe_size = 3; cfg.rotation_misalignment_threshold = 0.02;
    cfg.threading = hdr[4];
    cfg.parsac_flag = hdr[5]; cfg.parsac_keyframe_check_size = 1;
    rdvio_backend be;
    std::memset(&be, 0, sizeof be);
    rdvio_oracle_backend_fill(&be);
    rdvio_pipeline *p = nullptr;
    if (rdvio_pipeline_create(&p, &cfg, &be) != 0) return 3;
    if (rdvio_pipeline_set_init_states(p, n, gt.data()) != 0) return 3;
    int k = 0;
    double last[8] = {0};
    for (int i = 0; i < n; ++i) {
        for (; k < ni && imu[7 * (size_t)k] <= ts[i]; ++k)
            if (rdvio_pipeline_add_motion(p, imu[7 * (size_t)k], &imu[7 * (size_t)k + 4], &imu[7 * (size_t)k + 1]) != 0) return 4;
        double pose[8];
        if (rdvio_pipeline_add_frame(p, ts[i], frames.data() + (size_t)i * w * h, w, h, w, pose) != 0) {
            std::fprintf(stderr, "add_frame %d: %s\n", i, rdvio_pipeline_last_error(p));
            return 4;
        }
        std::memcpy(last, pose, sizeof last);
    }
    const int state = rdvio_pipeline_state(p);
    rdvio_pipeline_destroy(p);
    std::printf("frames %d state %d last position %.4f %.4f %.4f\n", n, state, last[4], last[5], last[6]);
    return state == 1 ? 0 : 5;
}
