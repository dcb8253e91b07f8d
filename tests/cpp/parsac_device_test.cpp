// Row N2 parity: PARSAC / IMU-PARSAC with hypothesis scoring on the device (rdvio_hip_parsac_score / _fetch behind the
// ParsacDeviceScorer hook), and with hypothesis GENERATION on the device as well (rdvio_hip_parsac_generate_score: EPnP from six
// points, the five-point essential solver, one wavefront per hypothesis), against the same loop solved and scored on the host
// (parsac.hpp): models, inlier masks and the 400 bin confidences must be bit-identical on all three roads.  Scenes: 300 / 1000 correspondences with a consistently moving subset, more than 20
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
static int dev_generate(void *user, const rdvio_parsac_batch *b, int n_iter, const int32_t *samples, int32_t *per_iter, double *models, rdvio_parsac_result *r) {
    return rdvio_hip_parsac_generate_score((rdvio_hip_ctx *)user, b, n_iter, samples, per_iter, models, r);
}

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
            static std::vector<float> bins_h(400, 0.5f), bins_d(400, 0.5f), ebins_h(400, 0.5f), ebins_d(400, 0.5f), bins_g(400, 0.5f), ebins_g(400, 0.5f);
            std::vector<char> mh, md, mg, eg;
            const Pose4 Th = find_pnp_matrix_parsac_imu(P3, p2, lens, R, t, 0.20, 1.0, mh, bins_h, 1.0 / 458.0);
            ParsacDeviceScorer dev{dev_score, dev_fetch, ctx};
            const Pose4 Td = find_pnp_matrix_parsac_imu(P3, p2, lens, R, t, 0.20, 1.0, md, bins_d, 1.0 / 458.0, 0.999, 1000, 0, &dev);
            CHECK(mh == md, "scene %d pass %d: IMU-PARSAC masks differ", scene, pass);
            CHECK(std::memcmp(&Th, &Td, sizeof Th) == 0, "scene %d pass %d: IMU-PARSAC models differ", scene, pass);
            CHECK(std::memcmp(bins_h.data(), bins_d.data(), 400 * sizeof(float)) == 0, "scene %d pass %d: PnP bin confidences differ", scene, pass);
            ParsacDeviceScorer gen{dev_score, dev_fetch, ctx};
            gen.generate = dev_generate;
            const Pose4 Tg = find_pnp_matrix_parsac_imu(P3, p2, lens, R, t, 0.20, 1.0, mg, bins_g, 1.0 / 458.0, 0.999, 1000, 0, &gen);
            CHECK(mh == mg, "scene %d pass %d: IMU-PARSAC masks differ (device-generated hypotheses)", scene, pass);
            CHECK(std::memcmp(&Th, &Tg, sizeof Th) == 0, "scene %d pass %d: IMU-PARSAC models differ (device-generated hypotheses)", scene, pass);
            CHECK(std::memcmp(bins_h.data(), bins_g.data(), 400 * sizeof(float)) == 0, "scene %d pass %d: PnP bin confidences differ (device-generated hypotheses)", scene, pass);
            std::vector<char> eh, ed;
            const M3 Eh = find_essential_matrix_parsac(a, p2, eh, ebins_h, 1.0 / 458.0);
            ParsacDeviceScorer dev2{dev_score, dev_fetch, ctx};
            const M3 Ed = find_essential_matrix_parsac(a, p2, ed, ebins_d, 1.0 / 458.0, 0.999, 1000, 0, &dev2);
            CHECK(eh == ed, "scene %d pass %d: PARSAC-essential masks differ", scene, pass);
            CHECK(std::memcmp(&Eh, &Ed, sizeof Eh) == 0, "scene %d pass %d: essential matrices differ", scene, pass);
            CHECK(std::memcmp(ebins_h.data(), ebins_d.data(), 400 * sizeof(float)) == 0, "scene %d pass %d: essential bin confidences differ", scene, pass);
            ParsacDeviceScorer gen2{dev_score, dev_fetch, ctx};
            gen2.generate = dev_generate;
            const M3 Eg = find_essential_matrix_parsac(a, p2, eg, ebins_g, 1.0 / 458.0, 0.999, 1000, 0, &gen2);
            CHECK(eh == eg, "scene %d pass %d: PARSAC-essential masks differ (device-generated hypotheses)", scene, pass);
            CHECK(std::memcmp(&Eh, &Eg, sizeof Eh) == 0, "scene %d pass %d: essential matrices differ (device-generated hypotheses)", scene, pass);
            CHECK(std::memcmp(ebins_h.data(), ebins_g.data(), 400 * sizeof(float)) == 0, "scene %d pass %d: essential bin confidences differ (device-generated hypotheses)", scene, pass);
            size_t kept = 0;
            for (char c : mh) kept += c != 0;
            if (scene != 3) CHECK(kept >= (size_t)(n_static * 9 / 10) && kept <= (size_t)n_static + (size_t)n / 20, "scene %d: %zu of %d kept", scene, kept, n_static);
            solves += 2;
        }
    }
    // every generated hypothesis against the host solver, bit for bit (not only the winning one): 64 EPnP samples, 12 five-point samples
    {
        const int n = 400;
        std::vector<double> X(3 * n), u(2 * n), a2(2 * n);
        std::vector<int32_t> d2v(n, 0), vs(1, n);
        const double bxy[2] = {0.0, 0.0};
        for (int i = 0; i < n; ++i) {
            const V3 P{2.4 * U(rng), 1.6 * U(rng), 4.0 + U(rng)};
            const V2 o = hnormalized(R * P + t), o1 = hnormalized(P);
            X[3 * i] = P.x; X[3 * i + 1] = P.y; X[3 * i + 2] = P.z;
            u[2 * i] = o.x + 1e-3 * U(rng); u[2 * i + 1] = o.y + 1e-3 * U(rng);
            a2[2 * i] = o1.x; a2[2 * i + 1] = o1.y;
        }
        std::uniform_int_distribution<int> pick(0, n - 1);
        for (int kind = 1; kind >= 0; --kind) {
            const int n_iter = kind == 1 ? 64 : 12, dof = kind == 1 ? 6 : 5, md = kind == 1 ? 12 : 9, per = kind == 1 ? 1 : 10;
            std::vector<int32_t> smp(n_iter * dof), per_iter(n_iter);
            for (int32_t &v : smp) v = pick(rng);
            if (kind == 1) for (int k = 0; k < 6; ++k) smp[k] = 7;   // a degenerate sample: six times the same point
            std::vector<double> models((size_t)n_iter * per * md);
            std::vector<rdvio_parsac_result> res((size_t)n_iter * per);
            rdvio_parsac_batch pb{};
            pb.kind = kind; pb.n_points = n; pb.points_changed = 1; pb.pa = kind == 1 ? X.data() : a2.data(); pb.pb = u.data(); pb.threshold = 1e-4;
            pb.n_valid = 1; pb.data_to_valid = d2v.data(); pb.valid_sizes = vs.data(); pb.bin_xy = bxy;
            CHECK(rdvio_hip_parsac_generate_score(ctx, &pb, n_iter, smp.data(), per_iter.data(), models.data(), res.data()) == RDVIO_OK, "generate_score failed: %s",
                  rdvio_hip_last_error(ctx));
            int packed = 0, differing = 0;
            for (int it = 0; it < n_iter; ++it) {
                std::vector<double> host;
                if (kind == 1) {
                    std::array<V3, 6> Xs; std::array<V2, 6> xs;
                    for (int k = 0; k < 6; ++k) { const int i = smp[6 * it + k]; Xs[k] = V3{X[3 * i], X[3 * i + 1], X[3 * i + 2]}; xs[k] = V2{u[2 * i], u[2 * i + 1]}; }
                    const Pose4 P = solve_pnp_6pt(Xs, xs)[0];
                    host.resize(12);
                    parsac_flatten(P, host.data());
                } else {
                    std::array<V2, 5> s1, s2;
                    for (int k = 0; k < 5; ++k) { const int i = smp[5 * it + k]; s1[k] = V2{a2[2 * i], a2[2 * i + 1]}; s2[k] = V2{u[2 * i], u[2 * i + 1]}; }
                    for (const M3 &E : solve_essential_5pt(s1, s2)) host.insert(host.end(), E.m, E.m + 9);
                }
                CHECK((int)host.size() == per_iter[it] * md, "kind %d iteration %d: %d hypotheses on the device, %zu on the host", kind, it, per_iter[it], host.size() / md);
                if ((int)host.size() == per_iter[it] * md && std::memcmp(host.data(), &models[(size_t)packed * md], host.size() * sizeof(double)) != 0) ++differing;
                packed += per_iter[it];
            }
            CHECK(differing == 0, "kind %d: %d of %d iterations have hypotheses that differ from the host solver's bits", kind, differing, n_iter);
            std::printf("kind %d: %d hypotheses of %d samples identical on host and device\n", kind, packed, n_iter);
        }
    }
    // argument checks of the C entry points
    rdvio_parsac_batch bad{};
    rdvio_parsac_result rr;
    CHECK(rdvio_hip_parsac_score(ctx, &bad, &rr) != RDVIO_OK, "an empty batch must be refused");
    CHECK(rdvio_hip_parsac_fetch(ctx, 100000, nullptr, nullptr) != RDVIO_OK, "a model outside the last batch must be refused");
    rdvio_hip_ctx_destroy(ctx);
    if (fails) return 1;
    std::printf("OK parsac device scoring / generation == host on %d solves\n", solves);
    return 0;
}
