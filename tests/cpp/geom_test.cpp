// Self-consistency checks of the host geometry (rd_vio_amd/host/pipeline/geom.hpp): the two-view solvers recover a
// known configuration.  PARITY UNPINNED: the reference has no tests or fixtures for these functions.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../rd_vio_amd/host/pipeline/geom.hpp"
#include "../../rd_vio_amd/host/pipeline/parsac.hpp"

using namespace rdvio_pipe;

static int fails = 0;
#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);             \
            std::printf("\n");                    \
            ++fails;                              \
        }                                         \
    } while (0)

int main() {
    std::mt19937 rng(648);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    // --- the PARSAC sampler's private generator is the C library's rand() after srand(seed), draw for draw
    for (unsigned seed : {0u, 1u, 648u, 123456789u}) {
        std::srand(seed);
        GlibcRand g(seed);
        int bad = 0;
        for (int i = 0; i < 20000; ++i) bad += (uint32_t)std::rand() != g.next();
        CHECK(bad == 0, "GlibcRand(%u) differs from rand() in %d of 20000 draws", seed, bad);
        CHECK(GlibcRand::max == (uint32_t)RAND_MAX, "RAND_MAX is not 2^31 - 1 here");
    }
    // --- real eigenvalues / eigenvectors of a non-symmetric matrix with a known spectrum
    {
        const int n = 6;
        std::vector<double> D(n * n, 0.0), P(n * n), Pinv(n * n), A(n * n, 0.0);
        const double ev[n] = {3.0, -1.5, 0.25, 7.0, 1.0, -4.0};
        for (int i = 0; i < n; ++i) D[i * n + i] = ev[i];
        for (double &v : P) v = U(rng);
        for (int i = 0; i < n; ++i) P[i * n + i] += 3.0;
        // invert P by Gauss-Jordan
        std::vector<double> M(n * 2 * n, 0.0);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) M[i * 2 * n + j] = P[i * n + j];
            M[i * 2 * n + n + i] = 1.0;
        }
        for (int c = 0; c < n; ++c) {
            int p = c;
            for (int r = c + 1; r < n; ++r)
                if (std::fabs(M[r * 2 * n + c]) > std::fabs(M[p * 2 * n + c])) p = r;
            for (int j = 0; j < 2 * n; ++j) std::swap(M[c * 2 * n + j], M[p * 2 * n + j]);
            const double d = M[c * 2 * n + c];
            for (int j = 0; j < 2 * n; ++j) M[c * 2 * n + j] /= d;
            for (int r = 0; r < n; ++r)
                if (r != c) {
                    const double f = M[r * 2 * n + c];
                    for (int j = 0; j < 2 * n; ++j) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
                }
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) Pinv[i * n + j] = M[i * 2 * n + n + j];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < n; ++k) A[i * n + j] += P[i * n + k] * ev[k] * Pinv[k * n + j];
        std::vector<double> wr, wi;
        CHECK(real_eigenvalues(n, A, wr, wi), "hqr did not converge");
        for (int k = 0; k < n; ++k) {
            double best = 1e9;
            for (int i = 0; i < n; ++i) best = std::min(best, std::fabs(wr[i] - ev[k]) + std::fabs(wi[i]));
            CHECK(best < 1e-9, "eigenvalue %g missing (err %g)", ev[k], best);
            const std::vector<double> x = eigenvector_for(n, A, ev[k]);
            double res = 0;
            for (int i = 0; i < n; ++i) {
                double s = -ev[k] * x[i];
                for (int j = 0; j < n; ++j) s += A[i * n + j] * x[j];
                res = std::max(res, std::fabs(s));
            }
            CHECK(res < 1e-8, "eigenvector residual %g", res);
        }
    }
    // --- two views of random points: 5-point solver, rotation solver, triangulation
    const Q4 q = normalized(Q4{0.05, -0.08, 0.03, 1.0});
    const M3 R = to_mat(q);
    const V3 t{0.3, -0.1, 0.05};
    std::vector<V3> X;
    std::vector<V2> p1, p2;
    std::vector<V3> b1, b2;
    for (int i = 0; i < 60; ++i) {
        const V3 x{2.0 * U(rng), 1.5 * U(rng), 4.0 + U(rng)};
        const V3 y = R * x + t;
        X.push_back(x);
        p1.push_back(hnormalized(x));
        p2.push_back(hnormalized(y));
        b1.push_back(normalized(x));
        b2.push_back(normalized(R * x));  // pure rotation pair for the Wahba solver
    }
    {
        // true essential matrix E = [t]x R: p2^T E p1 = 0
        const M3 tx{{0, -t.z, t.y, t.z, 0, -t.x, -t.y, t.x, 0}};
        const M3 Et = tx * R;
        std::array<V2, 5> s1{p1[0], p1[1], p1[2], p1[3], p1[4]}, s2{p2[0], p2[1], p2[2], p2[3], p2[4]};
        const std::vector<M3> sols = solve_essential_5pt(s1, s2);
        CHECK(!sols.empty(), "5-point solver returned no solution");
        double best = 1e9;
        for (const M3 &E : sols) {
            double n1 = 0, n2 = 0, d = 0;
            for (int i = 0; i < 9; ++i) { n1 += E.m[i] * E.m[i]; n2 += Et.m[i] * Et.m[i]; d += E.m[i] * Et.m[i]; }
            best = std::min(best, 1.0 - std::fabs(d) / std::sqrt(n1 * n2));
            for (int i = 0; i < 5; ++i) {
                const V3 Ep = E * V3{s1[i].x, s1[i].y, 1.0};
                const double r = s2[i].x * Ep.x + s2[i].y * Ep.y + Ep.z;
                CHECK(std::fabs(r) < 1e-8 * std::sqrt(n1), "solution violates an epipolar constraint: %g", r);
            }
        }
        CHECK(best < 1e-8, "no 5-point solution matches the true E (best 1-cos %g)", best);
        std::vector<char> mask;
        const M3 E = find_essential_matrix(p1, p2, mask, 1.0);
        (void)E;
        size_t inl = 0;
        for (char c : mask) inl += c;
        CHECK(mask.size() == p1.size() && inl == p1.size(), "RANSAC inliers %zu of %zu", inl, p1.size());
    }
    {
        std::vector<char> mask;
        const M3 Rr = find_rotation_matrix(b1, b2, mask, (M_PI / 180.0) * 10.0);
        double err = 0;
        for (int i = 0; i < 9; ++i) err = std::max(err, std::fabs(Rr.m[i] - R.m[i]));
        CHECK(err < 1e-9, "rotation error %g", err);
        CHECK(std::fabs(det(Rr) - 1.0) < 1e-12, "det %g", det(Rr));
    }
    {
        std::vector<std::array<double, 12>> Ps = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0},
                                                  {R.m[0], R.m[1], R.m[2], t.x, R.m[3], R.m[4], R.m[5], t.y, R.m[6], R.m[7], R.m[8], t.z}};
        for (int i = 0; i < 10; ++i) {
            const V3 y = R * X[i] + t;
            const auto h = triangulate_point(Ps, {normalized(X[i]), normalized(y)});
            const V3 x{h[0] / h[3], h[1] / h[3], h[2] / h[3]};
            CHECK(norm(x - X[i]) < 1e-9, "triangulation error %g", norm(x - X[i]));
        }
    }
    {
        // points on a plane n.X = d seen from two poses: H = R + t n^T / d (normalised coordinates)
        const V3 n = normalized(V3{0.1, -0.2, 1.0});
        const double d = 4.0;
        std::vector<V2> a, b;
        for (int i = 0; i < 40; ++i) {
            const double x = 2.0 * U(rng), y = 1.5 * U(rng);
            const double z = (d - n.x * x - n.y * y) / n.z;
            const V3 X{x, y, z}, Y = R * X + t;
            a.push_back(hnormalized(X));
            b.push_back(hnormalized(Y));
        }
        std::vector<char> mask;
        const M3 H = find_homography_matrix(a, b, mask, 1e-3, 0.999, 1000, 648);
        size_t inl = 0;
        for (char c : mask) inl += c;
        CHECK(inl == a.size(), "homography inliers %zu of %zu", inl, a.size());
        for (size_t i = 0; i < a.size(); ++i) CHECK(homography_geometric_error(H, a[i], b[i]) < 1e-16, "homography transfer error");
        M3 R1, R2;
        V3 T1, T2, n1, n2;
        CHECK(decompose_homography(H, R1, R2, T1, T2, n1, n2), "plane-induced homography reported as pure rotation");
        double best = 1e9;
        for (const M3 *Rc : {&R1, &R2}) {
            double e = 0;
            for (int i = 0; i < 9; ++i) e = std::max(e, std::fabs(Rc->m[i] - R.m[i]));
            best = std::min(best, e);
        }
        CHECK(best < 1e-8, "homography decomposition: rotation error %g", best);
        const V3 tn = normalized(t);
        const double cs = std::max(std::fabs(dot(normalized(T1), tn)), std::fabs(dot(normalized(T2), tn)));
        CHECK(cs > 1.0 - 1e-8, "homography decomposition: translation direction cos %g", cs);
        // a rotation-only homography is recognised
        M3 Ra, Rb;
        CHECK(!decompose_homography(R, Ra, Rb, T1, T2, n1, n2), "pure rotation not recognised");
        // essential decomposition contains the true rotation and +-t
        const M3 tx{{0, -t.z, t.y, t.z, 0, -t.x, -t.y, t.x, 0}};
        M3 E1, E2;
        V3 TE;
        decompose_essential(tx * R, E1, E2, TE);
        double be = 1e9;
        for (const M3 *Rc : {&E1, &E2}) {
            double e = 0;
            for (int i = 0; i < 9; ++i) e = std::max(e, std::fabs(Rc->m[i] - R.m[i]));
            be = std::min(be, e);
        }
        CHECK(be < 1e-9 && std::fabs(std::fabs(dot(TE, tn)) - 1.0) < 1e-9, "essential decomposition error %g", be);
        // least squares: exact solution of a consistent overdetermined system; quaternion helpers
        std::vector<double> A(12 * 4), xs{1.5, -2.0, 0.25, 3.0}, rhs(12, 0.0);
        for (double &v : A) v = U(rng);
        for (int r = 0; r < 12; ++r)
            for (int c = 0; c < 4; ++c) rhs[r] += A[r * 4 + c] * xs[c];
        const std::vector<double> xr = least_squares(12, 4, A, rhs);
        for (int c = 0; c < 4; ++c) CHECK(std::fabs(xr[c] - xs[c]) < 1e-9, "least squares x[%d] = %g", c, xr[c]);
        const Q4 qr = from_mat(R);
        CHECK(std::fabs(std::fabs(qr.x * q.x + qr.y * q.y + qr.z * q.z + qr.w * q.w) - 1.0) < 1e-12, "from_mat");
        const V3 w{0.3, -0.2, 0.5};
        const V3 wl = logmap(expmap(w));
        CHECK(norm(wl - w) < 1e-12, "logmap(expmap(w))");
        const V3 va = normalized(V3{0.2, -1.0, 0.4}), vb = normalized(V3{-0.5, 0.1, 0.9});
        CHECK(norm(rot(from_two_vectors(va, vb), va) - vb) < 1e-12, "from_two_vectors");
    }
    {
        // EPnP: six points, known pose; the float32 round trips of pnp.h:11-48 bound the accuracy
        std::array<V3, 6> Xs;
        std::array<V2, 6> xs;
        for (int i = 0; i < 6; ++i) {
            Xs[i] = V3{2.0 * U(rng), 1.5 * U(rng), 4.0 + U(rng)};
            xs[i] = hnormalized(R * Xs[i] + t);
        }
        const std::vector<Pose4> sol = solve_pnp_6pt(Xs, xs);
        CHECK(sol.size() == 1, "solve_pnp_6pt returned %zu poses", sol.size());
        double er = 0;
        for (int i = 0; i < 9; ++i) er = std::max(er, std::fabs(sol[0].R.m[i] - R.m[i]));
        CHECK(er < 1e-4 && norm(sol[0].t - t) < 1e-3, "EPnP pose error: R %g t %g", er, norm(sol[0].t - t));
        // IMU-PARSAC: 70 static points + 30 points on an object that moved by 0.4 m between mapping and observation;
        // the prior is the true pose.  Inliers = the static set.
        std::vector<V3> P3;
        std::vector<V2> p2;
        std::vector<size_t> lens;
        for (int i = 0; i < 100; ++i) {
            const V3 X{2.4 * U(rng), 1.6 * U(rng), 4.0 + U(rng)};
            const V3 Xobs = i < 70 ? X : X + V3{0.4, -0.25, 0.0};
            P3.push_back(X);
            p2.push_back(hnormalized(R * Xobs + t));
            lens.push_back(5 + (size_t)(i % 7));
        }
        std::vector<char> mask;
        std::vector<float> bins(400, 0.5f);
        const Pose4 T = find_pnp_matrix_parsac_imu(P3, p2, lens, R, t, 0.20, 1.0, mask, bins, 1.0 / 458.0);
        size_t good = 0, bad = 0;
        for (int i = 0; i < 100; ++i) (i < 70 ? good : bad) += (mask[(size_t)i] != 0);
        CHECK(good >= 66 && bad <= 2, "IMU-PARSAC: %zu of 70 static kept, %zu of 30 dynamic kept", good, bad);
        double eT = 0;
        for (int i = 0; i < 9; ++i) eT = std::max(eT, std::fabs(T.R.m[i] - R.m[i]));
        CHECK(eT < 5e-3, "IMU-PARSAC pose rotation error %g", eT);
        // PARSAC essential: the same split seen in two views
        std::vector<V2> a, b;
        for (int i = 0; i < 100; ++i) {
            const V3 X = P3[(size_t)i];
            a.push_back(hnormalized(X));
            b.push_back(p2[(size_t)i]);
        }
        std::vector<float> ebins(400, 0.5f);
        std::vector<char> emask;
        (void)find_essential_matrix_parsac(a, b, emask, ebins, 1.0 / 458.0);
        good = bad = 0;
        for (int i = 0; i < 100; ++i) (i < 70 ? good : bad) += (emask[(size_t)i] != 0);
        CHECK(good >= 60 && bad <= 6, "PARSAC essential: %zu of 70 static kept, %zu of 30 dynamic kept", good, bad);
        float mx = 0;
        for (float c : ebins) mx = std::max(mx, c);
        CHECK(mx > 0.5f && mx <= 1.0f, "bin confidences not updated (max %g)", (double)mx);
    }
    {
        // LotBox draws are a permutation prefix
        LotBox box(10);
        box.seed(0);
        bool seen[10] = {false};
        for (int i = 0; i < 10; ++i) {
            const size_t k = box.draw_without_replacement();
            CHECK(k < 10 && !seen[k], "LotBox repeated %zu", k);
            seen[k] = true;
        }
        CHECK(box.draw_without_replacement() == size_t(-1), "LotBox should be empty");
    }
    if (fails) return 1;
    std::printf("OK geometry\n");
    return 0;
}
