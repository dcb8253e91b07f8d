// Row N3 parity: the tracker's two-view gates (5-point essential RANSAC, 2-point rotation RANSAC) with hypothesis generation and
// scoring on the device (rdvio_hip_ransac_generate_score / _ransac_fetch) and the track-length Poisson-disk thinning on the device
// (rdvio_hip_thin_tracks) against the host road of geom.hpp / pipeline.cpp: models, inlier masks and keep flags bit-identical.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../include/rdvio_hip.h"
#include "../../rd_vio_amd/host/pipeline/geom.hpp"

using namespace rdvio_pipe;

static int fails = 0;
#define CHECK(c, ...)                     \
    do {                                  \
        if (!(c)) {                       \
            std::printf("FAIL: ");        \
            std::printf(__VA_ARGS__);     \
            std::printf("\n");            \
            ++fails;                      \
        }                                 \
    } while (0)

static int gen(void *user, int kind, int n, int changed, const double *pa, const double *pb, double thr, int n_iter, const int32_t *samples, int32_t *per_iter,
               double *models, int32_t *counts) {
    return rdvio_hip_ransac_generate_score((rdvio_hip_ctx *)user, kind, n, changed, pa, pb, thr, n_iter, samples, per_iter, models, counts);
}
static int fetch(void *user, int m, uint8_t *mask) { return rdvio_hip_ransac_fetch((rdvio_hip_ctx *)user, m, mask); }

int main() {
    rdvio_hip_ctx *ctx = nullptr;
    if (rdvio_hip_ctx_create(&ctx, 0, 752, 480, 1024, 10, 4096, nullptr) != RDVIO_OK) {
        std::printf("FAIL: no HIP context\n");
        return 1;
    }
    std::mt19937 rng(11);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    int gates = 0;
    for (int scene = 0; scene < 8; ++scene) {
        const int n = scene % 2 ? 1000 : 150;
        const bool pure_rotation = scene >= 6;
        const M3 R = to_mat(expmap(V3{0.02 * U(rng), 0.03 * U(rng), 0.02 * U(rng)}));
        const V3 t = pure_rotation ? V3{0, 0, 0} : V3{0.08 * U(rng), 0.05 * U(rng), 0.03 * U(rng)};
        const int n_out = scene == 2 ? n / 2 : n / 8;   // outliers
        std::vector<V2> h1, h2;
        std::vector<V3> b1, b2;
        for (int i = 0; i < n; ++i) {
            const V3 X{2.4 * U(rng), 1.6 * U(rng), 4.0 + U(rng)};
            V3 Y = R * X + t;
            if (i < n_out) Y = Y + V3{0.3 * U(rng), 0.3 * U(rng), 0.0};
            const V3 z1 = normalized(X), z2 = normalized(Y);
            b1.push_back(z1);
            b2.push_back(z2);
            h1.push_back(hnormalized(z1));
            V2 o = hnormalized(z2);
            o.x += 2e-4 * U(rng);
            o.y += 2e-4 * U(rng);
            h2.push_back(o);
        }
        std::vector<char> mh, md;
        const M3 Eh = find_essential_matrix(h1, h2, mh, 1.0 / 458.0);
        RansacDevice dev{gen, fetch, ctx};
        const M3 Ed = find_essential_matrix(h1, h2, md, 1.0 / 458.0, 0.999, 1000, 0, &dev);
        CHECK(mh == md, "scene %d: essential gate masks differ", scene);
        CHECK(std::memcmp(&Eh, &Ed, sizeof Eh) == 0, "scene %d: essential matrices differ", scene);
        std::vector<char> rh, rd;
        const M3 Rh = find_rotation_matrix(b1, b2, rh, (M_PI / 180.0) * 10.0);
        RansacDevice dev2{gen, fetch, ctx};
        const M3 Rd = find_rotation_matrix(b1, b2, rd, (M_PI / 180.0) * 10.0, 0.999, 1000, 0, &dev2);
        CHECK(rh == rd, "scene %d: rotation gate masks differ", scene);
        CHECK(std::memcmp(&Rh, &Rd, sizeof Rh) == 0, "scene %d: rotation matrices differ", scene);
        size_t kept = 0;
        for (char c : mh) kept += c != 0;
        CHECK(kept >= (size_t)(n - n_out) * 8 / 10, "scene %d: essential gate kept %zu of %d", scene, kept, n - n_out);
        gates += 2;
    }
    // track-length thinning: random keypoints (dense enough to collide), random processing orders, some tracks trash
    int thinnings = 0;
    for (int trial = 0; trial < 12; ++trial) {
        const int w = trial % 3 == 2 ? 1280 : 752, h = trial % 3 == 2 ? 720 : 480;
        const int n = trial % 2 ? 1000 : 150;
        const double radius = trial % 4 == 3 ? 20.0 : 10.0;
        std::vector<double> xy(2 * (size_t)n);
        std::vector<uint8_t> trash(n, 0);
        for (int i = 0; i < n; ++i) {
            xy[2 * i] = 20.0 + (w - 41) * 0.5 * (U(rng) + 1.0) * (trial % 5 == 4 ? 0.3 : 1.0);
            xy[2 * i + 1] = 20.0 + (h - 41) * 0.5 * (U(rng) + 1.0) * (trial % 5 == 4 ? 0.3 : 1.0);
            trash[i] = (rng() % 11) == 0;
        }
        std::vector<int32_t> order;
        for (int i = 0; i < n; ++i)
            if (rng() % 7) order.push_back(i);
        std::shuffle(order.begin(), order.end(), rng);
        std::vector<uint8_t> keep_h(order.size()), keep_d(order.size(), 2);
        PoissonDisk2 filter(radius);
        for (size_t k = 0; k < order.size(); ++k) {
            const V2 pt{xy[2 * order[k]], xy[2 * order[k] + 1]};
            const bool ok = filter.permit_point(pt) && !trash[order[k]];
            if (ok) filter.preset_point(pt);
            keep_h[k] = ok ? 1 : 0;
        }
        const int rc = rdvio_hip_thin_tracks(ctx, w, h, radius, n, xy.data(), (int)order.size(), order.data(), trash.data(), keep_d.data());
        CHECK(rc == RDVIO_OK, "trial %d: rdvio_hip_thin_tracks returned %d (%s)", trial, rc, rdvio_hip_last_error(ctx));
        CHECK(keep_h == keep_d, "trial %d: keep flags differ", trial);
        size_t kept = 0;
        for (uint8_t c : keep_h) kept += c;
        CHECK(kept > 0 && kept < order.size(), "trial %d: degenerate thinning (%zu of %zu kept)", trial, kept, order.size());
        ++thinnings;
    }
    int32_t bad_order[1] = {7};
    uint8_t tr[4] = {0, 0, 0, 0}, kp[1];
    double pts[8] = {30, 30, 40, 40, 50, 50, 60, 60};
    CHECK(rdvio_hip_thin_tracks(ctx, 752, 480, 10.0, 4, pts, 1, bad_order, tr, kp) != RDVIO_OK, "an order entry outside the points must be refused");
    rdvio_hip_ctx_destroy(ctx);
    if (fails) return 1;
    std::printf("OK two-view gates on the device == host on %d gates, thinning on %d orders\n", gates, thinnings);
    return 0;
}
