// Row N2 parity: PARSAC / IMU-PARSAC with hypothesis scoring on the device (rdvio_hip_parsac_score / _fetch behind the
// ParsacDeviceScorer hook) against the same loop scored on the host (parsac.hpp): models, inlier masks and the 400 bin
// confidences must be bit-identical.  Scenes: 300 / 1000 correspondences with a consistently moving subset, more than 20
// occupied bins (weighted bin sampler) and fewer (lot box), the IMU-prior rejection exit.
#include <cstdio>
#include <cstring>
#include <random>

#include "../../include/rdvio_hip.h"
#include "../../rd_vio_amd/host/pipeline/parsac.hpp"

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

static int dev_score(void *user, const rdvio_parsac_batch *b, rdvio_parsac_result *r) { return rdvio_hip_parsac_score((rdvio_hip_ctx *)user, b, r); }
static int dev_fetch(void *user, int m, uint8_t *mask, int32_t *bins) { return rdvio_hip_parsac_fetch((rdvio_hip_ctx *)user, m, mask, bins); }

int main() {
    rdvio_hip_ctx *ctx = nullptr;
    if (rdvio_hip_ctx_create(&ctx, 0, 752, 480, 1024, 10, 4096, nullptr) != RDVIO_OK) {
        std::printf("FAIL: no HIP context\n");
        return 1;
    }
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    const M3 R = to_mat(expmap(V3{0.03, -0.05, 0.02}));
    const V3 t{0.12, -0.04, 0.06};
    int solves = 0;
    for (int scene = 0; scene < 6; ++scene) {
        const int n = scene % 2 ? 1000 : 300;
        const double spread = scene >= 4 ? 0.15 : 1.0;   // a narrow field: < 20 occupied bins -> lot-box sampling
        const int n_static = scene == 3 ? n / 10 : (7 * n) / 10;  // scene 3: the prior explains < 15 % -> prior rejection exit
        std::vector<V3> P3;
        std::vector<V2> p2, a;
        std::vector<size_t> lens;
        for (int i = 0; i < n; ++i) {
            const V3 X{2.4 * spread * U(rng), 1.6 * spread * U(rng), 4.0 + U(rng)};
            const V3 Xobs = i < n_static ? X : X + V3{0.4, -0.25, 0.05 * U(rng)};
            P3.push_back(X);
            V2 obs = hnormalized(R * Xobs + t);
            obs.x += 2e-4 * U(rng);
            obs.y += 2e-4 * U(rng);
            p2.push_back(obs);
            a.push_back(hnormalized(X));
            lens.push_back(3 + (size_t)(i % 11));
        }
        for (int pass = 0; pass < 2; ++pass) {   // second pass: the bin confidences of the first feed the sampler
            static std::vector<float> bins_h(400, 0.5f), bins_d(400, 0.5f), ebins_h(400, 0.5f), ebins_d(400, 0.5f);
            std::vector<char> mh, md;
            const Pose4 Th = find_pnp_matrix_parsac_imu(P3, p2, lens, R, t, 0.20, 1.0, mh, bins_h, 1.0 / 458.0);
            ParsacDeviceScorer dev{dev_score, dev_fetch, ctx};
            const Pose4 Td = find_pnp_matrix_parsac_imu(P3, p2, lens, R, t, 0.20, 1.0, md, bins_d, 1.0 / 458.0, 0.999, 1000, 0, &dev);
            CHECK(mh == md, "scene %d pass %d: IMU-PARSAC masks differ", scene, pass);
            CHECK(std::memcmp(&Th, &Td, sizeof Th) == 0, "scene %d pass %d: IMU-PARSAC models differ", scene, pass);
            CHECK(std::memcmp(bins_h.data(), bins_d.data(), 400 * sizeof(float)) == 0, "scene %d pass %d: PnP bin confidences differ", scene, pass);
            std::vector<char> eh, ed;
            const M3 Eh = find_essential_matrix_parsac(a, p2, eh, ebins_h, 1.0 / 458.0);
            ParsacDeviceScorer dev2{dev_score, dev_fetch, ctx};
            const M3 Ed = find_essential_matrix_parsac(a, p2, ed, ebins_d, 1.0 / 458.0, 0.999, 1000, 0, &dev2);
            CHECK(eh == ed, "scene %d pass %d: PARSAC-essential masks differ", scene, pass);
            CHECK(std::memcmp(&Eh, &Ed, sizeof Eh) == 0, "scene %d pass %d: essential matrices differ", scene, pass);
            CHECK(std::memcmp(ebins_h.data(), ebins_d.data(), 400 * sizeof(float)) == 0, "scene %d pass %d: essential bin confidences differ", scene, pass);
            size_t kept = 0;
            for (char c : mh) kept += c != 0;
            if (scene != 3) CHECK(kept >= (size_t)(n_static * 9 / 10) && kept <= (size_t)n_static + (size_t)n / 20, "scene %d: %zu of %d kept", scene, kept, n_static);
            solves += 2;
        }
    }
    // argument checks of the C entry points
    rdvio_parsac_batch bad{};
    rdvio_parsac_result rr;
    CHECK(rdvio_hip_parsac_score(ctx, &bad, &rr) != RDVIO_OK, "an empty batch must be refused");
    CHECK(rdvio_hip_parsac_fetch(ctx, 100000, nullptr, nullptr) != RDVIO_OK, "a model outside the last batch must be refused");
    rdvio_hip_ctx_destroy(ctx);
    if (fails) return 1;
    std::printf("OK parsac device scoring == host scoring on %d solves\n", solves);
    return 0;
}
