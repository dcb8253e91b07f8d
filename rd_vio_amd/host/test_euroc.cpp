// test_euroc.cpp -- the reference's examples/test_euroc.cpp (/root/reference/examples/test_euroc.cpp:46-95) headless on the
// HIP path: reads a EuRoC `mav0` tree like examples/dataset.hpp:454-624 does (cam0/data.csv + cam0/data/<ns>.png +
// cam0/sensor.yaml, imu0/data.csv; clips merged and ordered by timestamp; a clip with both delivers the IMU sample first,
// then the image; images undistorted with the raw K as the new camera matrix, dataset.hpp:232-236), feeds
// rdvio_hip::Odometry(calib, config) through addMotion / addFrame, and -- in place of the reference's viewer / SlimeVR
// output -- writes the trajectory in TUM format (t tx ty tz qx qy qz qw, one row per processed camera frame).
//
//   test_euroc MAV0_DIR CALIB.yaml CONFIG.yaml [--out traj.txt] [--max-frames N] [--bootstrap-from-groundtruth]
//
// --bootstrap-from-groundtruth hands the ground-truth states of the first keyframes to the pipeline instead of running the
// SfM / IMU-alignment stages (rdvio_pipeline_set_init_states); the Python harness scripts/run_euroc.py offers the same
// switch, and both produce the same trajectory to the file's 9 decimals (tests/test_euroc_harness.py).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "rdvio_odometry.hpp"
#include "rdvio_png.hpp"

namespace {

struct Clip {
    long long ns = 0;
    bool has_imu = false, has_image = false;
    double gyro[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
};

std::vector<std::vector<std::string>> read_csv(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("file not found: " + path);
    std::vector<std::vector<std::string>> rows;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> cols;
        std::stringstream ss(line);
        std::string cell;
        while (std::getline(ss, cell, ',')) cols.push_back(cell);
        rows.push_back(cols);
    }
    return rows;
}

// cv::undistort(img, K, dist) with the raw K as the new camera matrix: bilinear remap, constant 0 border
// (the arithmetic of rd_vio_amd/euroc.py: undistort_map + remap_bilinear)
struct Undistorter {
    int w = 0, h = 0;
    bool identity = true;
    std::vector<double> mx, my;
    void init(const double K[4], const std::vector<double> &dist, int w_, int h_) {
        w = w_;
        h = h_;
        identity = true;
        for (double d : dist) identity = identity && d == 0.0;
        if (identity) return;
        const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist.size() > 4 ? dist[4] : 0.0;
        mx.resize((size_t)w * h);
        my.resize((size_t)w * h);
        for (int v = 0; v < h; ++v)
            for (int u = 0; u < w; ++u) {
                const double x = ((double)u - K[2]) / K[0], y = ((double)v - K[3]) / K[1];
                const double r2 = x * x + y * y;
                const double radial = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
                const double xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
                const double yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
                mx[(size_t)v * w + u] = K[0] * xd + K[2];
                my[(size_t)v * w + u] = K[1] * yd + K[3];
            }
    }
    std::vector<uint8_t> apply(const std::vector<uint8_t> &img) const {
        if (identity) return img;
        std::vector<uint8_t> out((size_t)w * h);
        auto at = [&](long yy, long xx) -> double { return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? (double)img[(size_t)yy * w + xx] : 0.0; };
        for (size_t i = 0; i < out.size(); ++i) {
            const double fx0 = std::floor(mx[i]), fy0 = std::floor(my[i]);
            const long x0 = (long)fx0, y0 = (long)fy0;
            const double fx = mx[i] - fx0, fy = my[i] - fy0;
            const double val = (at(y0, x0) * (1 - fx) + at(y0, x0 + 1) * fx) * (1 - fy) + (at(y0 + 1, x0) * (1 - fx) + at(y0 + 1, x0 + 1) * fx) * fy;
            out[i] = (uint8_t)std::min(255.0, std::max(0.0, std::nearbyint(val)));
        }
        return out;
    }
};

}  // namespace

int main(int argc, char **argv) {
    using namespace rdvio_hip;
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s MAV0_DIR CALIB.yaml CONFIG.yaml [--out traj.txt] [--max-frames N] [--bootstrap-from-groundtruth]\n", argv[0]);
        return 2;
    }
    const std::string mav = argv[1], calib = argv[2], config = argv[3];
    std::string out_path = "trajectory_tum.txt";
    long max_frames = -1;
    bool bootstrap = false;
    for (int i = 4; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out_path = argv[++i];
        else if (!std::strcmp(argv[i], "--max-frames") && i + 1 < argc) max_frames = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--bootstrap-from-groundtruth")) bootstrap = true;
        else {
            std::fprintf(stderr, "unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    try {
        // ---- dataset (examples/dataset.hpp:454-624)
        const YamlNode cam_yaml = load_yaml_file(mav + "/cam0/sensor.yaml");
        const YamlNode *intr = cam_yaml.path("intrinsics"), *res = cam_yaml.path("resolution");
        if (!intr || intr->seq.size() != 4 || !res || res->seq.size() != 2) throw std::runtime_error("cam0/sensor.yaml: intrinsics / resolution missing");
        double K4[4];
        for (int i = 0; i < 4; ++i) K4[i] = std::strtod(intr->seq[(size_t)i].scalar.c_str(), nullptr);
        const int W = std::atoi(res->seq[0].scalar.c_str()), H = std::atoi(res->seq[1].scalar.c_str());
        std::vector<double> dist;
        if (const YamlNode *d = cam_yaml.path("distortion_coefficients"))
            for (const YamlNode &e : d->seq) dist.push_back(std::strtod(e.scalar.c_str(), nullptr));
        while (dist.size() < 4) dist.push_back(0.0);
        Undistorter und;
        und.init(K4, dist, W, H);
        std::map<long long, Clip> clips;
        for (const auto &row : read_csv(mav + "/cam0/data.csv")) {
            const long long ns = std::atoll(row.at(0).c_str());
            clips[ns].ns = ns;
            clips[ns].has_image = true;
        }
        for (const auto &row : read_csv(mav + "/imu0/data.csv")) {
            if (row.size() < 7) continue;
            const long long ns = std::atoll(row[0].c_str());
            Clip &c = clips[ns];
            c.ns = ns;
            c.has_imu = true;
            for (int k = 0; k < 3; ++k) {
                c.gyro[k] = std::strtod(row[(size_t)(1 + k)].c_str(), nullptr);   // gyro first (dataset.hpp:565-571)
                c.acc[k] = std::strtod(row[(size_t)(4 + k)].c_str(), nullptr);
            }
        }
        std::vector<double> frame_times;
        for (const auto &kv : clips)
            if (kv.second.has_image) frame_times.push_back((double)kv.first / 1e9);

        Odometry vio(calib, config);
        const rdvio_pipeline_config cfg = load_yaml_config(config, calib);

        if (bootstrap) {
            // ground truth rows: t[ns], p(3), q(w,x,y,z), v(3), bw(3), ba(3) -> (t, q(x,y,z,w), p, v, bg, ba), linearly interpolated
            std::vector<std::array<double, 17>> g;
            for (const auto &row : read_csv(mav + "/state_groundtruth_estimate0/data.csv")) {
                if (row.size() < 17) continue;
                double v[17];
                for (int k = 0; k < 17; ++k) v[k] = std::strtod(row[(size_t)k].c_str(), nullptr);
                std::array<double, 17> r{};
                r[0] = v[0] / 1e9;
                r[1] = v[5]; r[2] = v[6]; r[3] = v[7]; r[4] = v[4];
                r[5] = v[1]; r[6] = v[2]; r[7] = v[3];
                for (int k = 0; k < 9; ++k) r[(size_t)(8 + k)] = v[8 + k];
                g.push_back(r);
            }
            if (g.size() < 2) throw std::runtime_error("--bootstrap-from-groundtruth needs mav0/state_groundtruth_estimate0/data.csv");
            const size_t want = std::min(frame_times.size(), (size_t)cfg.initializer_keyframe_num * cfg.initializer_keyframe_gap * 4);
            std::vector<double> rows(want * 17);
            for (size_t i = 0; i < want; ++i) {
                const double t = frame_times[i];
                size_t j = (size_t)(std::lower_bound(g.begin(), g.end(), t, [](const std::array<double, 17> &r, double tt) { return r[0] < tt; }) - g.begin());
                j = std::min(std::max<size_t>(j, 1), g.size() - 1);
                const double a = (t - g[j - 1][0]) / (g[j][0] - g[j - 1][0]);
                double row[17];
                for (int k = 0; k < 17; ++k) row[k] = (1 - a) * g[j - 1][(size_t)k] + a * g[j][(size_t)k];
                double q1[4] = {g[j][1], g[j][2], g[j][3], g[j][4]};
                const double *q0 = &g[j - 1][1];
                if (q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3] < 0)
                    for (double &x : q1) x = -x;
                double q[4], n2 = 0.0;
                for (int k = 0; k < 4; ++k) {
                    q[k] = (1 - a) * q0[k] + a * q1[k];
                    n2 += q[k] * q[k];
                }
                const double nrm = std::sqrt(n2);
                for (int k = 0; k < 4; ++k) row[1 + k] = q[k] / nrm;
                row[0] = t;
                std::copy(row, row + 17, rows.begin() + (long)(17 * i));
            }
            if (rdvio_pipeline_set_init_states(vio.handle(), (int)want, rows.data()) != RDVIO_OK) throw std::runtime_error("set_init_states failed");
        }

        // ---- replay (examples/test_euroc.cpp:46-95)
        std::vector<std::array<double, 8>> traj;
        long n_img = 0;
        long long seen = 0;
        double spent = 0.0;
        int64_t cnt[29];
        for (const auto &kv : clips) {
            const Clip &c = kv.second;
            const double t = (double)c.ns / 1e9;
            if (c.has_imu) {
                const auto t0 = std::chrono::steady_clock::now();
                vio.addMotion(t, {c.acc[0], c.acc[1], c.acc[2]}, {c.gyro[0], c.gyro[1], c.gyro[2]});
                spent += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            if (c.has_image) {
                if (max_frames >= 0 && n_img >= max_frames) break;
                PngImage png = read_png(mav + "/cam0/data/" + std::to_string(c.ns) + ".png");
                if (png.width != W || png.height != H) throw std::runtime_error("image size differs from cam0/sensor.yaml");
                if (png.channels == 3) {  // PNG stores RGB; Odometry::addFrame expects OpenCV's BGR
                    // the reference undistorts every image, colour or gray (dataset.hpp:585-592); this harness remaps single-channel
                    // images only, so a colour stream from a distorted camera is refused rather than fed through raw
                    if (!und.identity) throw std::runtime_error("colour images with lens distortion are not supported by this harness (undistort them first)");
                    for (size_t i = 0; i < (size_t)W * H; ++i) std::swap(png.pixels[3 * i], png.pixels[3 * i + 2]);
                    const auto t0 = std::chrono::steady_clock::now();
                    vio.addFrame(t, png.pixels.data(), W, H, 3, 3 * W);
                    spent += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                } else {
                    const std::vector<uint8_t> img = und.apply(png.pixels);
                    const auto t0 = std::chrono::steady_clock::now();
                    vio.addFrame(t, img.data(), W, H, 1, W);
                    spent += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                }
                ++n_img;
            }
            rdvio_pipeline_counters(vio.handle(), cnt);
            if (cnt[0] != seen) {
                seen = cnt[0];
                double tt = 0.0, pose[7];
                if (rdvio_pipeline_latest_state(vio.handle(), &tt, pose)) traj.push_back({tt, pose[4], pose[5], pose[6], pose[0], pose[1], pose[2], pose[3]});
            }
        }
        std::FILE *fo = std::fopen(out_path.c_str(), "w");
        if (!fo) throw std::runtime_error("cannot write " + out_path);
        for (const auto &r : traj) {
            for (int k = 0; k < 8; ++k) std::fprintf(fo, k ? " %.9f" : "%.9f", r[(size_t)k]);
            std::fprintf(fo, "\n");
        }
        std::fclose(fo);
        // Odometry::local_map (rdvio.hpp:91-97): the window's valid triangulated landmarks with R_imu_to_cv = [1 0 0; 0 0 -1; 0 1 0]
        // applied; the C entry point returns them in the world frame, the mirror class applies the swap
        const std::vector<std::array<double, 3>> lm = vio.local_map();
        std::vector<double> raw(3 * lm.size() + 3);
        const int n_raw = rdvio_pipeline_local_map(vio.handle(), raw.data(), (int)lm.size());
        bool swap_ok = n_raw == (int)lm.size();
        for (size_t k = 0; swap_ok && k < lm.size(); ++k)
            swap_ok = lm[k][0] == raw[3 * k] && lm[k][1] == -raw[3 * k + 2] && lm[k][2] == raw[3 * k + 1];
        const std::array<double, 16> Twc = vio.transform_world_cam();
        std::printf("{\"frames\": %ld, \"poses\": %zu, \"pipeline_seconds\": %.3f, \"state\": %d, \"local_map_points\": %zu, "
                    "\"local_map_axis_swap_ok\": %s, \"transform_world_cam_last_row_ok\": %s, \"trajectory\": \"%s\"}\n",
                    n_img, traj.size(), spent, vio.state(), lm.size(), swap_ok ? "true" : "false",
                    (Twc[12] == 0 && Twc[13] == 0 && Twc[14] == 0 && Twc[15] == 1) ? "true" : "false", out_path.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "test_euroc: %s\n", e.what());
        return 1;
    }
    return 0;
}
