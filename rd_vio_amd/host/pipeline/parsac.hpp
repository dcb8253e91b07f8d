// RD-VIO's dynamic-outlier machinery (row A19 of SURVEY.md section 8): PARSAC / IMU-PARSAC hypothesis scoring over a
// 20 x 20 image grid and the EPnP solver their PnP hypotheses come from.
//
// Reference behaviour restated (file:line under /root/reference):
//   Sampler, Parsac<>      src/rdvio_util/include/rdvio/util/parsac.h:9-284
//   IMU_Parsac<>           src/rdvio_util/include/rdvio/util/imu_parsac.h:11-410
//   solve_pnp_6pt, find_pnp_matrix_parsac_imu   src/rdvio_geometry/include/rdvio/geometry/pnp.h:11-48, 167-206
//   find_essential_matrix_parsac                src/rdvio_geometry/src/stereo.cpp:126-157
// solve_pnp_6pt calls cv::solvePnP(..., CV_EPNP) on float32 points with an identity camera matrix and round-trips the
// pose through a float32 Rodrigues vector; OpenCV is not available, so EPnP (Lepetit, Moreno-Noguer, Fua 2009 -- control
// points, 12-dim null space, three beta initialisations + Gauss-Newton, Arun alignment, lowest reprojection error wins) is
// restated here with the same float32 rounding points.  PARITY UNPINNED (no reference fixtures; OpenCV version unpinned).
//
// Kept quirks: the weighted sampler returns a BIN index that the caller uses as a DATA index (parsac.h:120-126,
// imu_parsac.h:83-91); it draws from the C library's rand() re-seeded with srand(0) per solve (parsac.h:10-13); scores are
// float.  One deviation: a point outside [-norm_scale, norm_scale) would index past the 400 bins in the reference
// (undefined behaviour); its bin coordinate is clamped here.  The reference's function-local static bin confidences
// (process-global, pnp.h:195, stereo.cpp:147) are passed in by the caller (per pipeline).
#pragma once

#include <cfloat>
#include <cstdlib>

#include <stdexcept>

#include "../../../include/rdvio_pipeline.h"
#include "geom.hpp"

namespace rdvio_pipe {

struct Pose4 {  // the 3 x 4 part of a 4 x 4 rigid transform
    M3 R;
    V3 t;
};

// pnp.h:89-93
inline double pnp_reproject_error(const Pose4 &T, const V3 &P1, const V2 &p2) {
    const V3 q = T.R * P1 + T.t;
    return sqnorm(p2 - V2{q.x / q.z, q.y / q.z});
}

// ---------------------------------------------------------------------------------------------------------------------
// EPnP for n >= 4 points, camera matrix = identity
// ---------------------------------------------------------------------------------------------------------------------
namespace epnp {

inline void control_points(const std::vector<V3> &pw, V3 cws[4]) {
    const int n = (int)pw.size();
    cws[0] = V3{0, 0, 0};
    for (const V3 &p : pw) cws[0] = cws[0] + p;
    cws[0] = cws[0] / (double)n;
    double C[9] = {0}, V[9], lam[3];
    for (const V3 &p : pw) {
        const double d[3] = {p.x - cws[0].x, p.y - cws[0].y, p.z - cws[0].z};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) C[3 * i + j] += d[i] * d[j];
    }
    sym_eigen(3, C, V, lam);
    std::vector<int> ord = ascending_order(3, lam);
    std::reverse(ord.begin(), ord.end());
    for (int i = 1; i < 4; ++i) {
        const int c = ord[i - 1];
        const double k = std::sqrt(std::max(lam[c], 0.0) / n);
        cws[i] = cws[0] + k * V3{V[0 * 3 + c], V[1 * 3 + c], V[2 * 3 + c]};
    }
}

inline bool barycentric(const std::vector<V3> &pw, const V3 cws[4], std::vector<double> &alphas) {
    M3 CC;
    for (int i = 0; i < 3; ++i) {
        const double c0 = i == 0 ? cws[0].x : (i == 1 ? cws[0].y : cws[0].z);
        for (int j = 1; j < 4; ++j) CC.m[3 * i + j - 1] = (i == 0 ? cws[j].x : (i == 1 ? cws[j].y : cws[j].z)) - c0;
    }
    if (!(std::fabs(det(CC)) > 0.0)) return false;
    const M3 Ci = inverse3(CC);
    alphas.resize(4 * pw.size());
    for (size_t i = 0; i < pw.size(); ++i) {
        const V3 d = pw[i] - cws[0];
        const V3 a = Ci * d;
        alphas[4 * i + 1] = a.x; alphas[4 * i + 2] = a.y; alphas[4 * i + 3] = a.z;
        alphas[4 * i] = 1.0 - a.x - a.y - a.z;
    }
    return true;
}

// rows of L (6 x 10) from the four null-space vectors v[0..3] (each 12 = 4 control points x 3)
inline void compute_L(const double *const v[4], double L[6][10]) {
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double dv[4][6][3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j)
            for (int k = 0; k < 3; ++k) dv[i][j][k] = v[i][3 * pa[j] + k] - v[i][3 * pb[j] + k];
    auto d = [&](int a, int b, int j) { return dv[a][j][0] * dv[b][j][0] + dv[a][j][1] * dv[b][j][1] + dv[a][j][2] * dv[b][j][2]; };
    for (int j = 0; j < 6; ++j) {
        L[j][0] = d(0, 0, j); L[j][1] = 2 * d(0, 1, j); L[j][2] = d(1, 1, j); L[j][3] = 2 * d(0, 2, j); L[j][4] = 2 * d(1, 2, j);
        L[j][5] = d(2, 2, j); L[j][6] = 2 * d(0, 3, j); L[j][7] = 2 * d(1, 3, j); L[j][8] = 2 * d(2, 3, j); L[j][9] = d(3, 3, j);
    }
}

inline void gauss_newton(const double L[6][10], const double rho[6], double b[4]) {
    for (int it = 0; it < 5; ++it) {
        std::vector<double> A(24), r(6);
        for (int i = 0; i < 6; ++i) {
            const double *l = L[i];
            A[4 * i + 0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
            A[4 * i + 1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
            A[4 * i + 2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
            A[4 * i + 3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
            r[i] = rho[i] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                             l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
        }
        const std::vector<double> x = least_squares(6, 4, A, r);
        for (int k = 0; k < 4; ++k) b[k] += x[k];
    }
}

// camera-frame points from the betas, sign fix, Arun alignment; returns the mean reprojection error
inline double pose_from_betas(const double *const v[4], const double b[4], const std::vector<double> &alphas, const std::vector<V3> &pw,
                              const std::vector<V2> &us, Pose4 &out) {
    const int n = (int)pw.size();
    V3 ccs[4];
    for (int i = 0; i < 4; ++i) {
        ccs[i] = V3{0, 0, 0};
        for (int k = 0; k < 4; ++k) ccs[i] = ccs[i] + b[k] * V3{v[k][3 * i], v[k][3 * i + 1], v[k][3 * i + 2]};
    }
    std::vector<V3> pc(n);
    for (int i = 0; i < n; ++i) {
        pc[i] = V3{0, 0, 0};
        for (int j = 0; j < 4; ++j) pc[i] = pc[i] + alphas[4 * i + j] * ccs[j];
    }
    if (pc[0].z < 0.0)
        for (V3 &p : pc) p = -p;
    V3 pc0{0, 0, 0}, pw0{0, 0, 0};
    for (int i = 0; i < n; ++i) { pc0 = pc0 + pc[i]; pw0 = pw0 + pw[i]; }
    pc0 = pc0 / (double)n;
    pw0 = pw0 / (double)n;
    M3 ABt;
    for (double &x : ABt.m) x = 0.0;
    for (int i = 0; i < n; ++i) {
        const double a[3] = {pc[i].x - pc0.x, pc[i].y - pc0.y, pc[i].z - pc0.z}, c[3] = {pw[i].x - pw0.x, pw[i].y - pw0.y, pw[i].z - pw0.z};
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) ABt.m[3 * r + q] += a[r] * c[q];
    }
    M3 U, V;
    double sv[3];
    svd3(ABt, U, sv, V);
    M3 R = U * transpose(V);
    if (det(R) < 0) {
        R.m[6] = -R.m[6]; R.m[7] = -R.m[7]; R.m[8] = -R.m[8];
    }
    out.R = R;
    out.t = pc0 - R * pw0;
    double err = 0.0;
    for (int i = 0; i < n; ++i) {
        const V3 q = R * pw[i] + out.t;
        const double du = us[i].x - q.x / q.z, dv = us[i].y - q.y / q.z;
        err += std::sqrt(du * du + dv * dv);
    }
    return err / n;
}

inline bool solve(const std::vector<V3> &pw, const std::vector<V2> &us, Pose4 &best) {
    const int n = (int)pw.size();
    if (n < 4) return false;
    V3 cws[4];
    control_points(pw, cws);
    std::vector<double> alphas;
    if (!barycentric(pw, cws, alphas)) return false;
    std::vector<double> MtM(144, 0.0), Vv(144), lam(12);
    for (int i = 0; i < n; ++i) {
        double r1[12], r2[12];
        for (int j = 0; j < 4; ++j) {
            const double a = alphas[4 * i + j];
            r1[3 * j] = a; r1[3 * j + 1] = 0.0; r1[3 * j + 2] = a * (0.0 - us[i].x);
            r2[3 * j] = 0.0; r2[3 * j + 1] = a; r2[3 * j + 2] = a * (0.0 - us[i].y);
        }
        for (int p = 0; p < 12; ++p)
            for (int q = 0; q < 12; ++q) MtM[12 * p + q] += r1[p] * r1[q] + r2[p] * r2[q];
    }
    sym_eigen(12, MtM.data(), Vv.data(), lam.data());
    const std::vector<int> ord = ascending_order(12, lam.data());
    double vbuf[4][12];
    const double *v[4];
    for (int k = 0; k < 4; ++k) {  // v[0] = smallest eigenvalue's vector ... v[3] = fourth smallest
        for (int i = 0; i < 12; ++i) vbuf[k][i] = Vv[12 * i + ord[k]];
        v[k] = vbuf[k];
    }
    double L[6][10], rho[6];
    compute_L(v, L);
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    for (int j = 0; j < 6; ++j) {
        const V3 d = cws[pa[j]] - cws[pb[j]];
        rho[j] = dot(d, d);
    }
    auto sub_solve = [&](std::initializer_list<int> cols) {
        const int nc = (int)cols.size();
        std::vector<double> A(6 * nc), r(rho, rho + 6);
        int c = 0;
        for (int col : cols) {
            for (int j = 0; j < 6; ++j) A[nc * j + c] = L[j][col];
            ++c;
        }
        return least_squares(6, nc, A, r);
    };
    double betas[3][4];
    {   // N = 4 approximation: betas from [B11 B12 B13 B14]
        const std::vector<double> b4 = sub_solve({0, 1, 3, 6});
        double *b = betas[0];
        if (b4[0] < 0) { b[0] = std::sqrt(-b4[0]); b[1] = -b4[1] / b[0]; b[2] = -b4[2] / b[0]; b[3] = -b4[3] / b[0]; }
        else { b[0] = std::sqrt(b4[0]); b[1] = b4[1] / b[0]; b[2] = b4[2] / b[0]; b[3] = b4[3] / b[0]; }
    }
    {   // N = 2: [B11 B12 B22]
        const std::vector<double> b3 = sub_solve({0, 1, 2});
        double *b = betas[1];
        if (b3[0] < 0) { b[0] = std::sqrt(-b3[0]); b[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0; }
        else { b[0] = std::sqrt(b3[0]); b[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0; }
        if (b3[1] < 0) b[0] = -b[0];
        b[2] = b[3] = 0.0;
    }
    {   // N = 3: [B11 B12 B22 B13 B23]
        const std::vector<double> b5 = sub_solve({0, 1, 2, 3, 4});
        double *b = betas[2];
        if (b5[0] < 0) { b[0] = std::sqrt(-b5[0]); b[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0; }
        else { b[0] = std::sqrt(b5[0]); b[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) b[0] = -b[0];
        b[2] = b5[3] / b[0];
        b[3] = 0.0;
    }
    double best_err = DBL_MAX;
    bool ok = false;
    for (int c = 0; c < 3; ++c) {
        if (!std::isfinite(betas[c][0]) || !std::isfinite(betas[c][1]) || !std::isfinite(betas[c][2]) || !std::isfinite(betas[c][3])) continue;
        gauss_newton(L, rho, betas[c]);
        Pose4 cand;
        const double err = pose_from_betas(v, betas[c], alphas, pw, us, cand);
        if (std::isfinite(err) && err < best_err) {
            best_err = err;
            best = cand;
            ok = true;
        }
    }
    return ok;
}

}  // namespace epnp

// pnp.h:11-48: EPnP on float32 copies of the points, pose round-tripped through a float32 Rodrigues vector
inline std::vector<Pose4> solve_pnp_6pt(const std::array<V3, 6> &Xs, const std::array<V2, 6> &xs) {
    std::vector<V3> pw(6);
    std::vector<V2> us(6);
    for (int i = 0; i < 6; ++i) {
        pw[i] = V3{(double)(float)Xs[i].x, (double)(float)Xs[i].y, (double)(float)Xs[i].z};
        us[i] = V2{(double)(float)xs[i].x, (double)(float)xs[i].y};
    }
    Pose4 P;
    if (!epnp::solve(pw, us, P)) {
        // cv::solvePnP leaves rvec / tvec at whatever EPnP produced; a degenerate sample gives an unusable pose, which the
        // inlier test then rejects.  An identity pose plays that role here.
        P = Pose4{};
    }
    const V3 rv = logmap(from_mat(P.R));
    const V3 rvf{(double)(float)rv.x, (double)(float)rv.y, (double)(float)rv.z};
    Pose4 out;
    out.R = to_mat(expmap(rvf));
    for (double &m : out.R.m) m = (double)(float)m;
    out.t = V3{(double)(float)P.t.x, (double)(float)P.t.y, (double)(float)P.t.z};
    return {out};
}

// ---------------------------------------------------------------------------------------------------------------------
// PARSAC / IMU-PARSAC
// ---------------------------------------------------------------------------------------------------------------------
class WeightedBinSampler {  // parsac.h:9-52
  public:
    explicit WeightedBinSampler(const std::vector<float> &acc) : acc_(acc) { std::srand(0); }
    size_t draw_by_weight() {
        size_t index;
        do {
            const float r = std::rand() / (float)RAND_MAX;
            index = (size_t)(std::upper_bound(acc_.begin() + 1, acc_.end(), r) - acc_.begin() - 1);
        } while (std::find(sampled_.begin(), sampled_.end(), index) != sampled_.end());
        sampled_.push_back(index);
        return index;
    }
    void refill_all() { sampled_.clear(); }

  private:
    const std::vector<float> &acc_;
    std::vector<size_t> sampled_;
};

// Shared state of both variants: the 20 x 20 grid over [-norm_scale, norm_scale)^2 and its occupied ("valid") bins
struct ParsacGrid {
    size_t nBinsX = 20, nBinsY = 20, nBins = 400, nValidBins = 0;
    float binW = 0.1f, binH = 0.1f;
    double norm_scale = 1.0, dynamic_probability = 0.0;
    bool use_lens = false;
    std::vector<V2> binLocations;
    std::vector<size_t> mapBinToValid, mapValidToBin, mapDataToValid, validSizes;
    std::vector<float> validLens, validConf, validConfPrior, validConfAccPrior;

    void setup(const std::vector<V2> &pts, const std::vector<size_t> *lens, const std::vector<float> &binConfidences) {
        binH = (float)(2 * norm_scale / nBinsY);
        binW = (float)(2 * norm_scale / nBinsX);
        binLocations.clear();
        float y = binH * 0.5f;
        for (size_t i = 0; i < nBinsY; ++i, y += binH) {
            float x = binW * 0.5f;
            for (size_t j = 0; j < nBinsX; ++j, x += binW) binLocations.push_back(V2{x - norm_scale, y - norm_scale});
        }
        const size_t N = pts.size();
        mapDataToValid.assign(N, 0);
        mapBinToValid.assign(nBins, SIZE_MAX);
        mapValidToBin.clear();
        validSizes.clear();
        validLens.clear();
        auto coord = [&](double v, float step, size_t n) {
            const double c = std::floor((v + norm_scale) / step);
            return (size_t)std::min<double>(std::max(c, 0.0), (double)n - 1);  // (clamped; see the header comment)
        };
        for (size_t i = 0; i < N; ++i) {
            const size_t iBin = coord(pts[i].x, binW, nBinsX) + nBinsX * coord(pts[i].y, binH, nBinsY);
            const size_t iv = mapBinToValid[iBin];
            if (iv == SIZE_MAX) {
                mapBinToValid[iBin] = mapValidToBin.size();
                mapDataToValid[i] = mapValidToBin.size();
                mapValidToBin.push_back(iBin);
                validSizes.push_back(1);
                if (use_lens) validLens.push_back((float)(*lens)[i]);
            } else {
                mapDataToValid[i] = iv;
                ++validSizes[iv];
                if (use_lens) validLens[iv] += (float)(*lens)[i];
            }
        }
        nValidBins = validSizes.size();
        if (use_lens)
            for (size_t i = 0; i < nValidBins; ++i) validLens[i] /= validSizes[i];
        // prior confidences of the occupied bins, floored at 0.5, normalised, accumulated (parsac.h:101-109)
        validConfPrior.resize(nValidBins);
        float sum = 0;
        for (size_t i = 0; i < nValidBins; ++i) {
            validConfPrior[i] = std::max(0.5f, binConfidences[mapValidToBin[i]]);
            sum += validConfPrior[i];
        }
        const float norm = 1.0f / sum;
        for (float &c : validConfPrior) c *= norm;
        validConfAccPrior.assign(nValidBins + 1, 0.0f);
        for (size_t i = 0; i < nValidBins; ++i) validConfAccPrior[i + 1] = validConfAccPrior[i] + validConfPrior[i];
        const float n2 = 1.f / validConfAccPrior[nValidBins];
        for (size_t i = 0; i < nValidBins; ++i) validConfAccPrior[i] *= n2;
    }

    std::vector<size_t> inliers_per_valid_bin(const std::vector<char> &mask) const {
        std::vector<size_t> cnt(nValidBins, 0);
        for (size_t i = 0; i < mask.size(); ++i)
            if (mask[i] == 1) cnt[mapDataToValid[i]]++;
        return cnt;
    }

    // parsac.h:215-262 / imu_parsac.h:233-280: coverage-weighted score in float arithmetic
    float score(const std::vector<size_t> &binInliers) {
        validConf.resize(nValidBins);
        float cs = 0, cs2 = 0;
        V2 sum{0.0, 0.0};
        for (size_t iv = 0; iv < nValidBins; ++iv) {
            float c = float(binInliers[iv]) / validSizes[iv];
            if (use_lens) {
                const float t = (float)(1 - std::pow(dynamic_probability, 0.10 * validLens[iv]));
                c = t * float(binInliers[iv]) / validSizes[iv];
            }
            validConf[iv] = c;
            const V2 x = binLocations[mapValidToBin[iv]];
            sum.x += x.x * c;
            sum.y += x.y * c;
            cs += c;
            cs2 += c * c;
        }
        float norm = 1.f / cs;
        const V2 mean{sum.x * norm, sum.y * norm};
        float Cxx = 0, Cxy = 0, Cyy = 0;
        for (size_t iv = 0; iv < nValidBins; ++iv) {
            const float c = validConf[iv];
            const V2 x = binLocations[mapValidToBin[iv]];
            const double dx = x.x - mean.x, dy = x.y - mean.y;
            Cxx += (float)((dx * dx) * c);
            Cxy += (float)((dx * dy) * c);
            Cyy += (float)((dy * dy) * c);
        }
        norm = cs / (cs * cs - cs2);
        const float imgRatio = norm * std::sqrt(Cxx * Cyy - Cxy * Cxy);
        return imgRatio * cs;
    }

    void write_back(std::vector<float> &binConfidences) const {
        binConfidences.resize(nBins);
        for (size_t iBin = 0; iBin < nBins; ++iBin)
            binConfidences[iBin] = mapBinToValid[iBin] == SIZE_MAX ? 0.0f : validConf[mapBinToValid[iBin]];
    }
};

// Hypothesis scoring behind the backend (rdvio_backend::parsac_score / parsac_fetch; the HIP product implements them,
// rdvio_hip_parsac_score).  kind / pa / pb are the flattened correspondences of the call site; flatten(model, out) writes
// the model as 9 (essential) or 12 ([R | t]) doubles.  fn == nullptr: the scoring below runs on the host.
struct ParsacDeviceScorer {
    int (*score)(void *user, const rdvio_parsac_batch *batch, rdvio_parsac_result *results) = nullptr;
    int (*fetch)(void *user, int model, uint8_t *mask, int32_t *bin_inliers) = nullptr;
    void *user = nullptr;
    int kind = 0;
    const double *pa = nullptr, *pb = nullptr;
};
inline void parsac_flatten(const M3 &E, double *out) {
    for (int q = 0; q < 9; ++q) out[q] = E.m[q];
}
inline void parsac_flatten(const Pose4 &T, double *out) {
    for (int q = 0; q < 9; ++q) out[q] = T.R.m[q];
    out[9] = T.t.x; out[10] = T.t.y; out[11] = T.t.z;
}

// Parsac<DoF>::solve (parsac.h:74-171) / IMU_Parsac<DoF>::solve (imu_parsac.h:28-163).
// solve(sample indices) -> models, error(model, i) -> double, pts2 = the image points that are bucketed.
// imu != nullptr selects the IMU variant: prior model inliers (error <= 2 threshold), overlap counting, lens weighting.
//
// The loop is evaluated a batch of iterations at a time: sampling and model generation do not depend on the scores (the
// samplers are seeded per solve and draw in a fixed order), only the NUMBER of iterations does, through the adaptive
// iter_max.  So the hypotheses of the next PARSAC_BATCH iterations are generated up front, scored together -- on the device
// when the backend offers the hook, on the host otherwise -- and the reference's accept / early-exit decisions are then
// replayed on the results in iteration order; hypotheses beyond the iteration at which the loop ends are simply ignored.
// Both roads run this same control flow and produce bit-identical scores, masks and bin confidences.
constexpr size_t PARSAC_BATCH = 8;

template <size_t DoF, class Model, class SolveFn, class ErrorFn>
struct ParsacResult {
    Model model;
    std::vector<char> inlier_mask;
    bool prior_rejected = false;
};

template <size_t DoF, class Model, class SolveFn, class ErrorFn>
ParsacResult<DoF, Model, SolveFn, ErrorFn> parsac_solve(size_t size, const std::vector<V2> &pts2, double threshold, double confidence, size_t max_iteration,
                                                       int seed, std::vector<float> &binConfidences, SolveFn solve, ErrorFn error, Model identity,
                                                       double norm_scale = 1.0, const Model *imu_prior = nullptr,
                                                       const std::vector<size_t> *lens = nullptr, double dynamic_probability = 0.0,
                                                       const ParsacDeviceScorer *dev = nullptr) {
    ParsacResult<DoF, Model, SolveFn, ErrorFn> out;
    out.model = identity;
    LotBox lotbox(size);
    lotbox.seed((unsigned)seed);
    const double K = std::log(std::max(1 - confidence, 1.0e-5));
    size_t inlier_count = 0;
    if (size < DoF) {
        out.inlier_mask.assign(size, 0);
        return out;
    }
    ParsacGrid grid;
    grid.norm_scale = norm_scale;
    grid.use_lens = imu_prior != nullptr;
    grid.dynamic_probability = dynamic_probability;
    grid.setup(pts2, lens, binConfidences);
    WeightedBinSampler sampler(grid.validConfAccPrior);
    std::vector<char> prior_mask;
    if (imu_prior) {  // ComputePriorDistribution (imu_parsac.h:176-201)
        size_t prior_inliers = 0;
        prior_mask.assign(size, 0);
        for (size_t i = 0; i < size; ++i)
            if (error(*imu_prior, i) <= threshold * 2.0) {
                prior_inliers++;
                prior_mask[i] = 1;
            }
        if ((double)prior_inliers / size < 0.15 || prior_inliers < 20) {
            out.inlier_mask.assign(size, 1);
            out.model = identity;
            out.prior_rejected = true;
            return out;
        }
    }
    const bool on_device = dev && dev->score && dev->fetch;
    // flattened grid for the device road (uploaded with the first batch)
    std::vector<int32_t> d2v, vsizes;
    std::vector<double> bin_xy;
    std::vector<float> lens_w;
    std::vector<uint8_t> prior_u8;
    constexpr int MD = sizeof(Model) == sizeof(M3) ? 9 : 12;
    if (on_device) {
        d2v.assign(grid.mapDataToValid.begin(), grid.mapDataToValid.end());
        vsizes.assign(grid.validSizes.begin(), grid.validSizes.end());
        bin_xy.resize(2 * grid.nValidBins);
        for (size_t iv = 0; iv < grid.nValidBins; ++iv) {
            bin_xy[2 * iv] = grid.binLocations[grid.mapValidToBin[iv]].x;
            bin_xy[2 * iv + 1] = grid.binLocations[grid.mapValidToBin[iv]].y;
        }
        if (grid.use_lens) {  // the track-length factor of imu_parsac.h:243-246 is a per-bin constant of the solve
            lens_w.resize(grid.nValidBins);
            for (size_t iv = 0; iv < grid.nValidBins; ++iv) lens_w[iv] = (float)(1 - std::pow(dynamic_probability, 0.10 * grid.validLens[iv]));
        }
        if (imu_prior) prior_u8.assign(prior_mask.begin(), prior_mask.end());
    }
    std::vector<size_t> bestBinInliers(grid.nValidBins, 0);
    size_t iter_max = max_iteration;
    float scoreMax = imu_prior ? -FLT_MAX : 0.0f;
    bool first_batch = true;
    for (size_t iter0 = 0; iter0 < iter_max; iter0 += PARSAC_BATCH) {
        // ---- hypotheses of iterations iter0 .. iter0 + B - 1
        const size_t B = std::min(PARSAC_BATCH, iter_max - iter0);
        std::vector<Model> models;
        std::vector<size_t> first_of(B + 1, 0);
        for (size_t b = 0; b < B; ++b) {
            std::array<size_t, DoF> sample;
            lotbox.refill_all();
            sampler.refill_all();
            for (size_t si = 0; si < DoF; ++si)
                sample[si] = grid.nValidBins > 20 ? sampler.draw_by_weight() : lotbox.draw_without_replacement();  // (bin index used as data index)
            const std::vector<Model> ms = solve(sample);
            models.insert(models.end(), ms.begin(), ms.end());
            first_of[b + 1] = models.size();
        }
        const size_t nm = models.size();
        // ---- scores
        std::vector<size_t> counts(nm, 0), effs(nm, 0);
        std::vector<float> scores(nm, 0.0f);
        std::vector<std::vector<char>> masks;           // host road only
        std::vector<std::vector<size_t>> bin_inl;       // host road only
        if (on_device && nm > 0) {
            std::vector<double> flat(nm * MD);
            for (size_t k = 0; k < nm; ++k) parsac_flatten(models[k], &flat[k * MD]);
            // the device kernel takes at most RDVIO_PARSAC_MAX_MODELS hypotheses per launch (8 iterations x 10 essential
            // matrices = 80 at most)
            rdvio_parsac_batch pb{};
            pb.kind = dev->kind;
            pb.n_points = (int32_t)size;
            pb.points_changed = first_batch ? 1 : 0;
            pb.pa = dev->pa;
            pb.pb = dev->pb;
            pb.threshold = threshold;
            pb.n_valid = (int32_t)grid.nValidBins;
            pb.data_to_valid = d2v.data();
            pb.valid_sizes = vsizes.data();
            pb.bin_xy = bin_xy.data();
            pb.lens_weight = grid.use_lens ? lens_w.data() : nullptr;
            pb.prior_mask = imu_prior ? prior_u8.data() : nullptr;
            pb.n_models = (int32_t)nm;
            pb.models = flat.data();
            std::vector<rdvio_parsac_result> res(nm);
            if (dev->score(dev->user, &pb, res.data()) != RDVIO_OK) throw std::runtime_error("backend parsac_score failed");
            first_batch = false;
            for (size_t k = 0; k < nm; ++k) {
                counts[k] = (size_t)res[k].count;
                effs[k] = (size_t)res[k].effective;
                scores[k] = res[k].score;
            }
        } else {
            masks.resize(nm);
            bin_inl.resize(nm);
            for (size_t k = 0; k < nm; ++k) {
                std::vector<char> &mask = masks[k];
                mask.assign(size, 0);
                size_t count = 0;
                for (size_t i = 0; i < size; ++i)
                    if (error(models[k], i) <= threshold) {
                        count++;
                        mask[i] = 1;
                    }
                size_t effective = count;
                if (imu_prior) {
                    effective = 0;
                    for (size_t i = 0; i < size; ++i)
                        if (prior_mask[i] && mask[i]) effective++;
                }
                counts[k] = count;
                effs[k] = effective;
                bin_inl[k] = grid.inliers_per_valid_bin(mask);
                scores[k] = grid.score(bin_inl[k]);
            }
        }
        // ---- replay of the reference's loop body on the results, in iteration order
        long best_in_batch = -1;
        for (size_t b = 0; b < B && iter0 + b < iter_max; ++b)
            for (size_t k = first_of[b]; k < first_of[b + 1]; ++k) {
                const size_t effective = effs[k];
                if (imu_prior && effective < DoF) continue;
                const float score = scores[k];
                if (score > scoreMax || (score == scoreMax && effective > inlier_count)) {
                    scoreMax = score;
                    out.model = models[k];
                    inlier_count = effective;
                    best_in_batch = (long)k;
                    const double ratio = inlier_count / (double)size;
                    const double N = K / std::log(1 - std::pow(ratio, 5));
                    if (N < (double)iter_max) iter_max = (size_t)std::ceil(N);
                }
            }
        if (best_in_batch >= 0) {
            if (on_device) {
                std::vector<uint8_t> m8(size);
                std::vector<int32_t> bi(grid.nValidBins);
                if (dev->fetch(dev->user, (int)best_in_batch, m8.data(), bi.data()) != RDVIO_OK) throw std::runtime_error("backend parsac_fetch failed");
                out.inlier_mask.assign(m8.begin(), m8.end());
                bestBinInliers.assign(bi.begin(), bi.end());
            } else {
                out.inlier_mask.swap(masks[(size_t)best_in_batch]);
                bestBinInliers = bin_inl[(size_t)best_in_batch];
            }
        }
    }
    if (imu_prior && inlier_count < DoF) {
        out.inlier_mask.assign(size, 1);
        out.model = identity;
        return out;
    }
    (void)grid.score(bestBinInliers);
    grid.write_back(binConfidences);
    out.inlier_mask.resize(size, 0);  // (the reference leaves the mask empty when no hypothesis scored above zero)
    return out;
}

// stereo.cpp:126-157
inline M3 find_essential_matrix_parsac(const std::vector<V2> &p1, const std::vector<V2> &p2, std::vector<char> &mask, std::vector<float> &binConfidences,
                                       double threshold = 1.0, double confidence = 0.999, size_t max_iteration = 1000, int seed = 0,
                                       ParsacDeviceScorer *dev = nullptr) {
    const double t1 = 3.84;
    auto solve = [&](const std::array<size_t, 5> &s) {
        std::array<V2, 5> a, b;
        for (int i = 0; i < 5; ++i) { a[i] = p1[s[i]]; b[i] = p2[s[i]]; }
        return solve_essential_5pt(a, b);
    };
    auto err = [&](const M3 &E, size_t i) {
        return essential_geometric_error(E, p1[i], p2[i]) + essential_geometric_error(transpose(E), p2[i], p1[i]);
    };
    std::vector<double> fa, fb;
    if (dev) {
        fa.resize(2 * p1.size());
        fb.resize(2 * p2.size());
        for (size_t i = 0; i < p1.size(); ++i) { fa[2 * i] = p1[i].x; fa[2 * i + 1] = p1[i].y; fb[2 * i] = p2[i].x; fb[2 * i + 1] = p2[i].y; }
        dev->kind = 0; dev->pa = fa.data(); dev->pb = fb.data();
    }
    auto res = parsac_solve<5, M3>(p1.size(), p2, 2.0 * t1 * threshold * threshold, confidence, max_iteration, seed, binConfidences, solve, err, M3{},
                                   1.0, (const M3 *)nullptr, nullptr, 0.0, dev);
    mask.swap(res.inlier_mask);
    return res.model;
}

// pnp.h:167-206
inline Pose4 find_pnp_matrix_parsac_imu(const std::vector<V3> &Xs, const std::vector<V2> &xs, const std::vector<size_t> &lens, const M3 &R, const V3 &t,
                                        double dynamic_prob, double scale, std::vector<char> &mask, std::vector<float> &binConfidences,
                                        double threshold = 1.0, double confidence = 0.999, size_t max_iteration = 1000, int seed = 0,
                                        ParsacDeviceScorer *dev = nullptr) {
    const double t2 = 5.99;
    auto solve = [&](const std::array<size_t, 6> &s) {
        std::array<V3, 6> a;
        std::array<V2, 6> b;
        for (int i = 0; i < 6; ++i) { a[i] = Xs[s[i]]; b[i] = xs[s[i]]; }
        return solve_pnp_6pt(a, b);
    };
    auto err = [&](const Pose4 &T, size_t i) { return pnp_reproject_error(T, Xs[i], xs[i]); };
    const Pose4 prior{R, t};
    std::vector<double> fa, fb;
    if (dev) {
        fa.resize(3 * Xs.size());
        fb.resize(2 * xs.size());
        for (size_t i = 0; i < Xs.size(); ++i) { fa[3 * i] = Xs[i].x; fa[3 * i + 1] = Xs[i].y; fa[3 * i + 2] = Xs[i].z; fb[2 * i] = xs[i].x; fb[2 * i + 1] = xs[i].y; }
        dev->kind = 1; dev->pa = fa.data(); dev->pb = fb.data();
    }
    auto res = parsac_solve<6, Pose4>(Xs.size(), xs, 2.0 * t2 * threshold * threshold, confidence, max_iteration, seed, binConfidences, solve, err,
                                      Pose4{}, scale, &prior, &lens, dynamic_prob, dev);
    mask.swap(res.inlier_mask);
    return res.model;
}

}  // namespace rdvio_pipe
