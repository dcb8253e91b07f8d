// C++ exercise of the host mirror (rd_vio_amd/host/rdvio_hip.hpp) written the way a caller of the reference's
// rdvio::Image / PreIntegrator / Solver would use them (cf. src/rdvio_map/src/frame.cpp:55-172,
// src/rdvio/src/sliding_window_tracker.cpp:101-125).  Needs a GPU; prints "OK" and exits 0 on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../rd_vio_amd/host/rdvio_hip.hpp"

using namespace rdvio_hip;

#define REQUIRE(c)                                                     \
    do {                                                               \
        if (!(c)) {                                                    \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                  \
        }                                                              \
    } while (0)

static std::vector<uint8_t> scene(int w, int h, double ox, double oy) {
    std::vector<uint8_t> img((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const double X = x + ox, Y = y + oy;
            double v = 110 + 40 * std::sin(X * 0.11) * std::cos(Y * 0.07) + 30 * std::sin((X + 2 * Y) * 0.05) +
                       25 * std::cos((3 * X - Y) * 0.031);
            img[(size_t)y * w + x] = (uint8_t)std::fmin(255.0, std::fmax(0.0, std::round(v)));
        }
    return img;
}

int main() {
    const int w = 752, h = 480;
    Context ctx(w, h, 512, 8, 4096);
    // --- seam 1: Image
    auto a = scene(w, h, 0, 0), b = scene(w, h, 2.5, -1.5);
    Image img0(ctx, 0, a.data(), w, h, w, 0.0), img1(ctx, 1, b.data(), w, h, w, 0.05);
    img0.preprocess(6.0, 8, 8);
    img1.preprocess(6.0, 8, 8);
    std::vector<vec2> kps;
    img0.detect_keypoints(kps, 150, 10.0);
    REQUIRE(kps.size() > 20 && kps.size() <= 150);
    for (auto &p : kps) REQUIRE(p[0] >= 20 && p[0] < w - 20 && p[1] >= 20 && p[1] < h - 20);
    std::vector<vec2> next;  // empty: no initial guess
    std::vector<char> status;
    img0.track_keypoints(&img1, kps, next, status);
    REQUIRE(next.size() == kps.size() && status.size() == kps.size());
    int ok = 0;
    double mx = 0, my = 0;
    for (size_t i = 0; i < kps.size(); ++i)
        if (status[i]) {
            ++ok;
            mx += next[i][0] - kps[i][0];
            my += next[i][1] - kps[i][1];
        }
    REQUIRE(ok > (int)kps.size() / 2);
    REQUIRE(std::fabs(mx / ok + 2.5) < 0.3 && std::fabs(my / ok - 1.5) < 0.3);
    img0.release_image_buffer();
    bool threw = false;
    try {
        img0.track_keypoints(&img1, kps, next, status);  // released slot: must fail loudly
    } catch (const std::runtime_error &) {
        threw = true;
    }
    REQUIRE(threw);
    // --- seam 2: PreIntegrator (constant rate about z: dq = exp(w T))
    PreIntegrator pre(ctx);
    pre.cov_w = {2.88e-8, 0, 0, 0, 2.88e-8, 0, 0, 0, 2.88e-8};
    pre.cov_a = {4e-6, 0, 0, 0, 4e-6, 0, 0, 0, 4e-6};
    pre.cov_bg = {3.76e-10, 0, 0, 0, 3.76e-10, 0, 0, 0, 3.76e-10};
    pre.cov_ba = {9e-6, 0, 0, 0, 9e-6, 0, 0, 0, 9e-6};
    REQUIRE(!pre.integrate(1.0, {0, 0, 0}, {0, 0, 0}, true, true));  // no data -> false
    for (int i = 0; i < 20; ++i) pre.data.push_back({i * 0.005, {0, 0, 0.4}, {0.1, -0.2, 9.8}});
    REQUIRE(pre.integrate(0.1, {0, 0, 0}, {0, 0, 0}, true, true));
    REQUIRE(std::fabs(pre.delta_t() - 0.1) < 1e-15);
    REQUIRE(std::fabs(pre.delta_q()[2] - std::sin(0.02)) < 1e-12 && std::fabs(pre.delta_q()[3] - std::cos(0.02)) < 1e-12);
    // --- seam 2: Solver, localize_newframe shape: one free frame, fixed anchor frame and landmarks
    Solver solver(ctx, 10);
    std::array<double, 14> extr = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0};
    solver.set_camera(extr, {458.654 / std::sqrt(0.5), 0, 0, 457.296 / std::sqrt(0.5)});
    State s0{}, s1{};
    s0[3] = 1; s1[3] = 1;
    s1[4] = 0.30;  // true x translation 0.30; start from a perturbed guess
    State s1_init = s1;
    s1_init[4] += 0.02; s1_init[5] -= 0.015;
    const int f0 = solver.add_frame_states(s0, true), f1 = solver.add_frame_states(s1_init, false);
    // without IMU factors only the pose is observable: velocity/bias rows stay at their values
    for (int i = 0; i < 40; ++i) {
        const double X = -1.5 + 0.08 * i, Y = -0.8 + 0.04 * (i % 7), Z = 4.0 + 0.1 * (i % 5);
        const double n0 = std::sqrt(X * X + Y * Y + Z * Z);
        const int l = solver.add_track_states({X / n0, Y / n0, Z / n0}, 1.0 / n0, true);
        const double x1 = X - 0.30, n1 = std::sqrt(x1 * x1 + Y * Y + Z * Z);
        const vec3 z = {x1 / n1, Y / n1, Z / n1};
        // local_tangent = [b1 b2 z] (lie_algebra.cpp:47-56): z is dominated by its z component here -> e_x axis
        vec3 b1 = {z[1] * 0 - z[2] * 0, z[2] * 1 - z[0] * 0, z[0] * 0 - z[1] * 1};  // z x e_x
        double nb = std::sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
        for (auto &v : b1) v /= nb;
        vec3 b2 = {z[1] * b1[2] - z[2] * b1[1], z[2] * b1[0] - z[0] * b1[2], z[0] * b1[1] - z[1] * b1[0]};
        nb = std::sqrt(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
        for (auto &v : b2) v /= nb;
        solver.add_factor_reprojection(f1, f0, l, {b1[0], b2[0], z[0], b1[1], b2[1], z[1], b1[2], b2[2], z[2]});
    }
    rdvio_ba_summary sm{};
    REQUIRE(solver.solve(&sm));
    REQUIRE(sm.final_cost < 1e-10 && sm.final_cost < sm.initial_cost);
    REQUIRE(std::fabs(solver.frame_state(f1)[4] - 0.30) < 1e-6 && std::fabs(solver.frame_state(f1)[5]) < 1e-6);
    REQUIRE(solver.frame_state(f0)[4] == 0.0);  // fixed frame untouched
    std::printf("OK %d/%zu tracked, solver iterations %d\n", ok, kps.size(), sm.iterations);
    return 0;
}
