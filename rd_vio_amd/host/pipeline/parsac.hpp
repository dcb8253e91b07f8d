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
// restated (csrc/hypo_solvers.hpp, shared with the device) with the same float32 rounding points.  PARITY UNPINNED (no
// reference fixtures; OpenCV version unpinned).
//
// Kept quirks: the weighted sampler returns a BIN index that the caller uses as a DATA index (parsac.h:120-126,
// imu_parsac.h:83-91); it draws rand()'s sequence after srand(0) per solve (parsac.h:10-13; GlibcRand below); scores are
// float.  One deviation: a point outside [-norm_scale, norm_scale) would index past the 400 bins in the reference
// (undefined behaviour); its bin coordinate is clamped here.  The reference's function-local static bin confidences
// (process-global, pnp.h:195, stereo.cpp:147) are passed in by the caller (per pipeline).
#pragma once

#include <cfloat>
#include <chrono>
#include <cstdlib>

#include <stdexcept>

#include "../../../include/rdvio_pipeline.h"
#include "../../csrc/hypo_solvers.hpp"
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

// pnp.h:11-48: EPnP on float32 copies of the six correspondences, the pose round-tripped through a float32 Rodrigues vector.
// The solver is csrc/hypo_solvers.hpp (hypo::epnp6), the same source the device kernel runs -- here with the serial executor.
inline std::vector<Pose4> solve_pnp_6pt(const std::array<V3, 6> &Xs, const std::array<V2, 6> &xs) {
    double X[18], u[12], model[12];
    for (int i = 0; i < 6; ++i) {
        X[3 * i] = Xs[i].x; X[3 * i + 1] = Xs[i].y; X[3 * i + 2] = Xs[i].z;
        u[2 * i] = xs[i].x; u[2 * i + 1] = xs[i].y;
    }
    hypo::EpnpWork work;
    hypo::epnp6(hypo::SerialExec{}, &work, X, u, model);
    Pose4 out;
    for (int k = 0; k < 9; ++k) out.R.m[k] = model[k];
    out.t = V3{model[9], model[10], model[11]};
    return {out};
}

// ---------------------------------------------------------------------------------------------------------------------
// PARSAC / IMU-PARSAC
// ---------------------------------------------------------------------------------------------------------------------
// The reference's sampler draws from the C library's rand() after srand(0) (parsac.h:10-13).  rand()'s state is process-global:
// another thread that draws from it (a second pipeline, a runtime library) would silently change the hypotheses.  This is glibc's
// generator (random_r.c, TYPE_3: the additive feedback x_i = x_{i-31} + x_{i-3} over 32-bit words, seeded by the Lehmer
// sequence 16807 x mod 2^31 - 1, the first 310 outputs discarded, an output is the word shifted right by one) as a private
// object: the same sequence as srand(seed); rand(); ... draw for draw (tests/cpp/geom_test.cpp compares them), owned by one solve.
class GlibcRand {
  public:
    static constexpr uint32_t max = 2147483647u;  // RAND_MAX
    explicit GlibcRand(unsigned seed) {
        int32_t word = seed == 0 ? 1 : (int32_t)seed;
        r_[0] = (uint32_t)word;
        for (int i = 1; i < 31; ++i) {
            const long hi = word / 127773, lo = word % 127773;
            long w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            word = (int32_t)w;
            r_[i] = (uint32_t)word;
        }
        f_ = 3;
        b_ = 0;
        for (int i = 0; i < 310; ++i) (void)next();
    }
    uint32_t next() {
        r_[f_] += r_[b_];
        const uint32_t out = r_[f_] >> 1;
        f_ = f_ == 30 ? 0 : f_ + 1;
        b_ = b_ == 30 ? 0 : b_ + 1;
        return out;
    }

  private:
    uint32_t r_[31];
    int f_, b_;
};

class WeightedBinSampler {  // parsac.h:9-52
  public:
    explicit WeightedBinSampler(const std::vector<float> &acc) : acc_(acc), rand_(0) {}
    size_t draw_by_weight() {
        size_t index;
        do {
            const float r = rand_.next() / (float)GlibcRand::max;
            index = (size_t)(std::upper_bound(acc_.begin() + 1, acc_.end(), r) - acc_.begin() - 1);
        } while (std::find(sampled_.begin(), sampled_.end(), index) != sampled_.end());
        sampled_.push_back(index);
        return index;
    }
    void refill_all() { sampled_.clear(); }

  private:
    const std::vector<float> &acc_;
    GlibcRand rand_;  // srand(0) per solve, as the reference's Sampler constructor does
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
    // optional: hypotheses GENERATED on the device from the sample indices (rdvio_backend::parsac_generate_score) and scored
    // in the same call; NULL = models come from the host solvers
    int (*generate)(void *user, const rdvio_parsac_batch *batch, int n_iterations, const int32_t *samples, int32_t *models_per_iteration,
                    double *models, rdvio_parsac_result *results) = nullptr;
    // optional: where the caller keeps how many iterations its last solve of this kind replayed -- the first batch of the next one
    // is sized to that (a solve over 1000 points needs ~100 iterations every frame: one round trip instead of two).  A batch's size
    // never shows in the result, so neither does the hint.
    size_t *iterations_hint = nullptr;
};
inline void parsac_flatten(const M3 &E, double *out) {
    for (int q = 0; q < 9; ++q) out[q] = E.m[q];
}
inline void parsac_flatten(const Pose4 &T, double *out) {
    for (int q = 0; q < 9; ++q) out[q] = T.R.m[q];
    out[9] = T.t.x; out[10] = T.t.y; out[11] = T.t.z;
}
inline void parsac_unflatten(const double *in, M3 &E) {
    for (int q = 0; q < 9; ++q) E.m[q] = in[q];
}
inline void parsac_unflatten(const double *in, Pose4 &T) {
    for (int q = 0; q < 9; ++q) T.R.m[q] = in[q];
    T.t = V3{in[9], in[10], in[11]};
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
// with device generation a batch is one launch whatever its size (one workgroup per hypothesis): the typical solve ends within
// its first batch
constexpr size_t PARSAC_BATCH_GENERATED_PNP = 32, PARSAC_BATCH_GENERATED_ESSENTIAL = 24;

// diagnostic accumulators (RDVIO_PIPELINE_PROF): where a PARSAC solve spends its time
struct ParsacProf {
    bool on = false;
    double t_setup = 0, t_models = 0, t_score = 0, t_fetch = 0;
    long solves = 0, batches = 0, iterations = 0, models = 0, fetches = 0;
};
inline ParsacProf &parsac_prof() {
    static ParsacProf p;
    return p;
}
struct ParsacTick {
    double &acc;
    bool on;
    std::chrono::steady_clock::time_point t0;
    explicit ParsacTick(double &a) : acc(a), on(parsac_prof().on) {
        if (on) t0 = std::chrono::steady_clock::now();
    }
    ~ParsacTick() {
        if (on) acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};

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
    ParsacProf &prof = parsac_prof();
    prof.solves++;
    ParsacGrid grid;
    grid.norm_scale = norm_scale;
    grid.use_lens = imu_prior != nullptr;
    grid.dynamic_probability = dynamic_probability;
    {
        ParsacTick tick(prof.t_setup);
        grid.setup(pts2, lens, binConfidences);
    }
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
    const bool generated = on_device && dev->generate;
    const size_t batch = !generated ? PARSAC_BATCH : (MD == 12 ? PARSAC_BATCH_GENERATED_PNP : PARSAC_BATCH_GENERATED_ESSENTIAL);
    // A batch behind the first one takes everything the adaptive iteration count still asks for (it has been cut down by the
    // first batch's best model), up to the backend's model capacity and -- so that the masks still ride back with the results --
    // 256 KB of masks + bin counts: a solve that needs a hundred iterations (1000 points, a low inlier ratio) is two round trips
    // instead of four.  The replay stops where the sequential loop stops; a batch's size never shows in the result.
    const size_t per_iteration = MD == 12 ? 1 : 10;
    const size_t batch_cap = !generated ? batch
                                        : std::max(batch, std::min<size_t>(RDVIO_PARSAC_MAX_MODELS / per_iteration,
                                                                           (size_t)(256 * 1024) / (size + 4 * grid.nValidBins + 1) / per_iteration));
    size_t B = 0, replayed = 0;
    for (size_t iter0 = 0; iter0 < iter_max; iter0 += B) {
        // ---- hypotheses of iterations iter0 .. iter0 + B - 1
        const size_t first = (generated && dev->iterations_hint) ? std::min(batch_cap, std::max(batch, (*dev->iterations_hint + 7) / 8 * 8)) : batch;
        B = std::min(iter0 == 0 ? first : batch_cap, iter_max - iter0);
        std::vector<Model> models;
        std::vector<size_t> first_of(B + 1, 0);
        std::vector<int32_t> samples(generated ? B * DoF : 0);
        prof.batches++;
        ParsacTick *tick_models = new ParsacTick(prof.t_models);
        for (size_t b = 0; b < B; ++b) {
            std::array<size_t, DoF> sample;
            lotbox.refill_all();
            sampler.refill_all();
            for (size_t si = 0; si < DoF; ++si)
                sample[si] = grid.nValidBins > 20 ? sampler.draw_by_weight() : lotbox.draw_without_replacement();  // (bin index used as data index)
            if (generated) {
                for (size_t si = 0; si < DoF; ++si) samples[b * DoF + si] = (int32_t)sample[si];
                continue;
            }
            const std::vector<Model> ms = solve(sample);
            models.insert(models.end(), ms.begin(), ms.end());
            first_of[b + 1] = models.size();
        }
        delete tick_models;
        // the batch descriptor of the device road
        rdvio_parsac_batch pb{};
        if (on_device) {
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
        }
        std::vector<size_t> counts, effs;
        std::vector<float> scores;
        if (generated) {
            // one call: the device solves the minimal problems of the B samples (hypo_solvers.hpp, the code solve() runs on the
            // host) and scores what it found; models come back for the replay below
            ParsacTick tick(prof.t_score);
            const size_t per = MD == 12 ? 1 : 10;
            std::vector<int32_t> per_iter(B);
            std::vector<double> flat(B * per * MD);
            std::vector<rdvio_parsac_result> res(B * per);
            if (dev->generate(dev->user, &pb, (int)B, samples.data(), per_iter.data(), flat.data(), res.data()) != RDVIO_OK)
                throw std::runtime_error("backend parsac_generate_score failed");
            first_batch = false;
            for (size_t b = 0; b < B; ++b) first_of[b + 1] = first_of[b] + (size_t)per_iter[b];
            const size_t nmg = first_of[B];
            models.resize(nmg);
            counts.resize(nmg); effs.resize(nmg); scores.resize(nmg);
            for (size_t k = 0; k < nmg; ++k) {
                parsac_unflatten(&flat[k * MD], models[k]);
                counts[k] = (size_t)res[k].count;
                effs[k] = (size_t)res[k].effective;
                scores[k] = res[k].score;
            }
        }
        const size_t nm = models.size();
        prof.models += (long)nm;
        ParsacTick *tick_score = new ParsacTick(prof.t_score);
        // ---- scores
        if (!generated) {
            counts.assign(nm, 0);
            effs.assign(nm, 0);
            scores.assign(nm, 0.0f);
        }
        std::vector<std::vector<char>> masks;           // host road only
        std::vector<std::vector<size_t>> bin_inl;       // host road only
        if (generated) {
            // (scored above)
        } else if (on_device && nm > 0) {
            std::vector<double> flat(nm * MD);
            for (size_t k = 0; k < nm; ++k) parsac_flatten(models[k], &flat[k * MD]);
            // the device kernel takes at most RDVIO_PARSAC_MAX_MODELS hypotheses per launch (8 iterations x 10 essential
            // matrices = 80 at most)
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
        delete tick_score;
        // ---- replay of the reference's loop body on the results, in iteration order
        long best_in_batch = -1;
        for (size_t b = 0; b < B && iter0 + b < iter_max; ++b, prof.iterations++, ++replayed)
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
            ParsacTick tick(prof.t_fetch);
            prof.fetches++;
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
    if (generated && dev->iterations_hint) *dev->iterations_hint = replayed;
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
