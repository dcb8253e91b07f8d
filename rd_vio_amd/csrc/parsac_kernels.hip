// PARSAC / IMU-PARSAC hypothesis scoring on gfx950.
//
// Replaces the inner loop of Parsac<DoF>::solve and IMU_Parsac<DoF>::solve
//   /root/reference/src/rdvio_util/include/rdvio/util/parsac.h:128-160 (error test, mask, bin counts), :215-262 (score)
//   /root/reference/src/rdvio_util/include/rdvio/util/imu_parsac.h:93-150, :233-280 (the same with the prior-overlap count
//   and the track-length weighting)
// with the two error functions of its call sites: essential.h:14-19 used symmetrically at stereo.cpp:137-141, and
// pnp.h:89-93.  One wavefront per hypothesis: the correspondences are spread over the lanes (double arithmetic, no
// contraction: an error compares with the threshold exactly as on the host), bin counts are integer LDS atomics (order
// independent), and the coverage score -- a float recurrence over the occupied bins in bin order -- runs on lane 0 in the
// reference's operation order, so scores, masks and counts are bit-identical to the host restatement
// (host/pipeline/parsac.hpp), which is what the orchestration runs over a backend without this hook.
#include <vector>

#include "ctx.hpp"
#include "hypo_solvers.hpp"

namespace {

struct PsArgs {
    int kind, n, n_valid, n_models, has_prior, has_lens;
    // generated batches: slot m belongs to iteration m / per_iter and is a hypothesis only if m % per_iter < counts[m / per_iter]
    int per_iter;
    const int32_t *counts;
    double threshold;
    const double *pa, *pb, *bin_xy, *models;
    const int32_t *d2v, *valid_sizes;
    const float *lens_w;
    const uint8_t *prior;
    uint8_t *masks;
    int32_t *bin_inliers;
    rdvio_parsac_result *results;
};

__device__ __forceinline__ double ess_err(const double *E, bool transposed, double ax, double ay, double bx, double by) {
    // essential_geometric_error(E, a, b) with E or E^T: Ea = E (ax, ay, 1); r = b . Ea; r^2 / (Ea.x^2 + Ea.y^2)
    const double e0 = transposed ? E[0] : E[0], e1 = transposed ? E[3] : E[1], e2 = transposed ? E[6] : E[2];
    const double e3 = transposed ? E[1] : E[3], e4 = transposed ? E[4] : E[4], e5 = transposed ? E[7] : E[5];
    const double e6 = transposed ? E[2] : E[6], e7 = transposed ? E[5] : E[7], e8 = transposed ? E[8] : E[8];
    const double x = e0 * ax + e1 * ay + e2 * 1.0, y = e3 * ax + e4 * ay + e5 * 1.0, z = e6 * ax + e7 * ay + e8 * 1.0;
    const double r = bx * x + by * y + z;
    return r * r / (x * x + y * y);
}

__global__ __launch_bounds__(64) void parsac_score_kernel(PsArgs a) {
    __shared__ int bins[RDVIO_PARSAC_MAX_BINS];
    const int m = blockIdx.x, lane = threadIdx.x;
    if (a.counts && m % a.per_iter >= a.counts[m / a.per_iter]) {   // an empty slot of a generated batch (workgroup-uniform)
        if (lane == 0) {
            rdvio_parsac_result r;
            r.count = -1; r.effective = -1; r.score = 0.f; r.pad_ = 0;
            a.results[m] = r;
        }
        return;
    }
    for (int i = lane; i < a.n_valid; i += 64) bins[i] = 0;
    __syncthreads();
    const double *M = a.models + (a.kind == 1 ? 12 : 9) * (size_t)m;   // (kinds 0 and 2: nine doubles)
    double Mr[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) Mr[q] = (q < 9 || a.kind == 1) ? M[q] : 0.0;
    int count = 0, eff = 0;
    for (int i = lane; i < a.n; i += 64) {
        double err;
        if (a.kind == 2) {   // rotation gate: cos(threshold) <= (R p1) . p2 <= 1  (hypo::rotation_inlier, shared with the host road)
            const double p1[3] = {a.pa[3 * i], a.pa[3 * i + 1], a.pa[3 * i + 2]}, p2[3] = {a.pb[3 * i], a.pb[3 * i + 1], a.pb[3 * i + 2]};
            const bool in2 = hypo::rotation_inlier(Mr, p1, p2, a.threshold);
            a.masks[(size_t)m * a.n + i] = in2 ? 1 : 0;
            if (in2) {
                count++;
                atomicAdd(&bins[a.d2v[i]], 1);
            }
            continue;
        }
        const double bx = a.pb[2 * i], by = a.pb[2 * i + 1];
        if (a.kind == 1) {
            const double X = a.pa[3 * i], Y = a.pa[3 * i + 1], Z = a.pa[3 * i + 2];
            const double qx = (Mr[0] * X + Mr[1] * Y + Mr[2] * Z) + Mr[9], qy = (Mr[3] * X + Mr[4] * Y + Mr[5] * Z) + Mr[10],
                         qz = (Mr[6] * X + Mr[7] * Y + Mr[8] * Z) + Mr[11];
            const double dx = bx - qx / qz, dy = by - qy / qz;
            err = dx * dx + dy * dy;
        } else {
            const double ax = a.pa[2 * i], ay = a.pa[2 * i + 1];
            err = ess_err(Mr, false, ax, ay, bx, by) + ess_err(Mr, true, bx, by, ax, ay);
        }
        const bool in = err <= a.threshold;
        a.masks[(size_t)m * a.n + i] = in ? 1 : 0;
        if (in) {
            count++;
            if (a.has_prior && a.prior[i]) eff++;
            atomicAdd(&bins[a.d2v[i]], 1);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        count += __shfl_xor(count, off);
        eff += __shfl_xor(eff, off);
    }
    __syncthreads();
    for (int i = lane; i < a.n_valid; i += 64) a.bin_inliers[(size_t)m * a.n_valid + i] = bins[i];
    if (lane != 0) return;
    // score (parsac.h:215-262 / imu_parsac.h:233-280), float recurrences in bin order
    float cs = 0.f, cs2 = 0.f;
    double sx = 0.0, sy = 0.0;
    for (int iv = 0; iv < a.n_valid; ++iv) {
        float c = (float)bins[iv] / (float)a.valid_sizes[iv];
        if (a.has_lens) c = a.lens_w[iv] * (float)bins[iv] / (float)a.valid_sizes[iv];
        sx += a.bin_xy[2 * iv] * (double)c;
        sy += a.bin_xy[2 * iv + 1] * (double)c;
        cs += c;
        cs2 += c * c;
    }
    float norm = 1.f / cs;
    const double mx = sx * (double)norm, my = sy * (double)norm;
    float Cxx = 0.f, Cxy = 0.f, Cyy = 0.f;
    for (int iv = 0; iv < a.n_valid; ++iv) {
        float c = (float)bins[iv] / (float)a.valid_sizes[iv];
        if (a.has_lens) c = a.lens_w[iv] * (float)bins[iv] / (float)a.valid_sizes[iv];
        const double dx = a.bin_xy[2 * iv] - mx, dy = a.bin_xy[2 * iv + 1] - my;
        Cxx += (float)((dx * dx) * (double)c);
        Cxy += (float)((dx * dy) * (double)c);
        Cyy += (float)((dy * dy) * (double)c);
    }
    norm = cs / (cs * cs - cs2);
    const float img_ratio = norm * sqrtf(Cxx * Cyy - Cxy * Cxy);
    rdvio_parsac_result r;
    r.count = count;
    r.effective = a.has_prior ? eff : count;
    r.score = img_ratio * cs;
    r.pad_ = 0;
    a.results[m] = r;
}

// Hypothesis generation (row N2): one 64-lane workgroup per PARSAC iteration solves the minimal problem of that iteration's
// sample -- EPnP from six 3-D / 2-D correspondences (solve_pnp_6pt, pnp.h:11-48) or the five-point essential solver
// (essential.cpp:286-298) -- with the shared solvers of hypo_solvers.hpp, i.e. the very source the host road runs, step by
// step over the lanes; the models land in the slots parsac_score_kernel reads next (no host hop).
struct GenArgs {
    int kind, n_iter;
    const double *pa, *pb;
    const int32_t *samples;   // n_iter x (6 | 5) point indices (range-checked on the host)
    double *models;           // n_iter x 12, or n_iter x 10 x 9
    int32_t *counts;          // n_iter: hypotheses found (1 for EPnP, 0..10 for the five-point solver)
};

__global__ __launch_bounds__(64) void parsac_generate_kernel(GenArgs a) {
    __shared__ union {
        hypo::EpnpWork pnp;
        hypo::Ess5Work ess;
    } work;
    __shared__ double pts_a[18], pts_b[12];
    __shared__ int n_found;
    const int b = blockIdx.x;
    const hypo::WaveExec x{(int)threadIdx.x};
    if (a.kind == 1) {
        const int32_t *smp = a.samples + 6 * (size_t)b;
        x.each(30, [=](int e) {
            if (e < 18) pts_a[e] = a.pa[3 * (size_t)smp[e / 3] + e % 3];
            else pts_b[e - 18] = a.pb[2 * (size_t)smp[(e - 18) / 2] + (e - 18) % 2];
        });
        hypo::epnp6(x, &work.pnp, pts_a, pts_b, a.models + 12 * (size_t)b);
        if (threadIdx.x == 0) a.counts[b] = 1;
    } else if (a.kind == 2) {
        const int32_t *smp = a.samples + 2 * (size_t)b;
        x.each(12, [=](int e) {
            if (e < 6) pts_a[e] = a.pa[3 * (size_t)smp[e / 3] + e % 3];
            else pts_b[e - 6] = a.pb[3 * (size_t)smp[(e - 6) / 3] + (e - 6) % 3];
        });
        x.one([=]() { hypo::rotation2(pts_a, pts_b, a.models + 9 * (size_t)b); });
        if (threadIdx.x == 0) a.counts[b] = 1;
    } else {
        const int32_t *smp = a.samples + 5 * (size_t)b;
        x.each(20, [=](int e) {
            if (e < 10) pts_a[e] = a.pa[2 * (size_t)smp[e / 2] + e % 2];
            else pts_b[e - 10] = a.pb[2 * (size_t)smp[(e - 10) / 2] + (e - 10) % 2];
        });
        hypo::essential5(x, &work.ess, pts_a, pts_b, a.models + 90 * (size_t)b, &n_found);
        if (threadIdx.x == 0) a.counts[b] = n_found;
    }
}

template <class Tp>
size_t put(uint8_t *base, size_t &off, const Tp *src, size_t n) {
    off = (off + 15) & ~(size_t)15;
    const size_t at = off;
    if (src && n) memcpy(base + at, src, n * sizeof(Tp));
    off += n * sizeof(Tp);
    return at;
}

}  // namespace

extern "C" {

static int parsac_fetch(rdvio_hip_ctx *ctx, int which, int model, uint8_t *mask, int32_t *bin_inliers);

// one scoring launch; n_iter > 0: the hypotheses are generated on the device first (samples: n_iter x dof point indices)
static int parsac_run(rdvio_hip_ctx *ctx, int which, const rdvio_parsac_batch *b, rdvio_parsac_result *results, int n_iter, const int32_t *samples,
                      int32_t *models_per_iteration, double *models_out) {
    if (!ctx || !b || !results) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::PsState &P = ctx->ps[which];
    const bool gen = n_iter > 0;
    const int n = b->n_points, nv = b->n_valid;
    // kind 0: essential matrix, five correspondences, up to ten models; 1: pose, six; 2: rotation, two bearing pairs
    const int per_iter = b->kind == 0 ? 10 : 1, dof = b->kind == 0 ? 5 : (b->kind == 1 ? 6 : 2);
    const int nm = gen ? n_iter * per_iter : b->n_models;
    if ((b->kind != 0 && b->kind != 1 && b->kind != 2) || n <= 0 || nv <= 0 || nm < 0 || !b->pa || !b->pb || !b->data_to_valid || !b->valid_sizes || !b->bin_xy ||
        (!gen && nm > 0 && !b->models) || (gen && (!samples || !models_per_iteration || !models_out)))
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad PARSAC batch");
    if (n > P.max_points || nv > RDVIO_PARSAC_MAX_BINS || nm > RDVIO_PARSAC_MAX_MODELS)
        return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "PARSAC batch of %d points / %d bins / %d models exceeds capacity (%d / %d / %d)", n, nv, nm,
                          P.max_points, RDVIO_PARSAC_MAX_BINS, RDVIO_PARSAC_MAX_MODELS);
    if (!b->points_changed && (P.n != n || P.kind != b->kind || P.nv != nv))
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "PARSAC batch reuses points that were never uploaded");
    for (int i = 0; b->points_changed && i < n; ++i)
        if (b->data_to_valid[i] < 0 || b->data_to_valid[i] >= nv) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "PARSAC point %d maps outside the occupied bins", i);
    // shapes are checked on the host so that the generation kernel can never gather out of bounds
    for (int i = 0; gen && i < n_iter * dof; ++i)
        if (samples[i] < 0 || samples[i] >= n) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "PARSAC sample index %d outside the %d points", samples[i], n);
    if (nm == 0) return RDVIO_OK;
    hipStream_t st = ctx->lane[P.lane];
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));  // the pinned blob may still be in flight
    uint8_t *hb = (uint8_t *)P.host, *db = (uint8_t *)P.dev;
    const int pdim = b->kind == 0 ? 2 : 3, pbdim = b->kind == 2 ? 3 : 2, mdim = b->kind == 1 ? 12 : 9;
    // (a batch that reuses the uploaded points keeps their layout, whatever optional pointers it passes)
    const bool has_lens = b->points_changed ? b->lens_weight != nullptr : P.has_lens;
    const bool has_prior = b->points_changed ? b->prior_mask != nullptr : P.has_prior;
    // static part (points, grid) at fixed offsets, models (or samples + model slots + counts) behind it; one copy each
    size_t off = 0;
    const size_t o_pa = put(hb, off, b->points_changed ? b->pa : (const double *)nullptr, (size_t)n * pdim);
    const size_t o_pb = put(hb, off, b->points_changed ? b->pb : (const double *)nullptr, (size_t)n * pbdim);
    const size_t o_xy = put(hb, off, b->points_changed ? b->bin_xy : (const double *)nullptr, (size_t)nv * 2);
    const size_t o_d2v = put(hb, off, b->points_changed ? b->data_to_valid : (const int32_t *)nullptr, (size_t)n);
    const size_t o_vs = put(hb, off, b->points_changed ? b->valid_sizes : (const int32_t *)nullptr, (size_t)nv);
    const size_t o_lw = put(hb, off, b->points_changed ? b->lens_weight : (const float *)nullptr, has_lens ? (size_t)nv : 0);
    const size_t o_pm = put(hb, off, b->points_changed ? b->prior_mask : (const uint8_t *)nullptr, has_prior ? (size_t)n : 0);
    const size_t o_dyn = (off + 15) & ~(size_t)15;
    off = o_dyn;
    size_t o_models = 0, o_samples = 0, up_end;
    if (gen) {
        o_samples = put(hb, off, samples, (size_t)n_iter * dof);
        up_end = off;   // only the samples travel up (counts and models are written into the result region on the device)
    } else {
        o_models = put(hb, off, b->models, (size_t)nm * mdim);
        up_end = off;
    }
    if (off > P.in_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "PARSAC staging buffer too small");
    if (b->points_changed) {
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(db, hb, up_end, hipMemcpyHostToDevice, st));
        P.n = n;
        P.kind = b->kind;
        P.nv = nv;
        P.has_prior = has_prior;
        P.has_lens = has_lens;
    } else {
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(db + o_dyn, hb + o_dyn, up_end - o_dyn, hipMemcpyHostToDevice, st));
    }
    // Everything the batch brings back -- per-model records, (generated) counts and models, and, when small enough to ride along,
    // every model's inlier mask and bin counts (the winner's are what the caller asks for next: rdvio_hip_parsac_fetch) -- is
    // written into ONE device region that mirrors the result part of the pinned blob: one copy down instead of four.
    const size_t mask_bytes = (size_t)nm * n, bins_bytes = (size_t)nm * nv * sizeof(int32_t);
    const bool inline_masks = mask_bytes + bins_bytes <= RDVIO_PARSAC_MASKS_INLINE;
    const size_t r_gen = ((size_t)nm * sizeof(rdvio_parsac_result) + 15) & ~(size_t)15;
    const size_t r_models = r_gen + (gen ? (((size_t)n_iter * sizeof(int32_t) + 15) & ~(size_t)15) : 0);
    const size_t gen_end = r_models + (gen ? (size_t)nm * mdim * sizeof(double) : 0);
    const size_t r_masks = (gen_end + 15) & ~(size_t)15, r_bins = r_masks + ((mask_bytes + 15) & ~(size_t)15);
    const size_t down_total = inline_masks ? r_bins + bins_bytes : gen_end;
    if (down_total > P.down_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "PARSAC result region too small");
    uint8_t *dd = (uint8_t *)P.down_dev;
    if (gen) {
        GenArgs g;
        g.kind = b->kind; g.n_iter = n_iter;
        g.pa = (const double *)(db + o_pa); g.pb = (const double *)(db + o_pb);
        g.samples = (const int32_t *)(db + o_samples);
        g.models = (double *)(dd + r_models);
        g.counts = (int32_t *)(dd + r_gen);
        hipLaunchKernelGGL(parsac_generate_kernel, dim3(n_iter), dim3(64), 0, st, g);
        RDVIO_HIP_CHECK(ctx, hipGetLastError());
    }
    PsArgs a;
    a.kind = b->kind; a.n = n; a.n_valid = nv; a.n_models = nm; a.has_prior = P.has_prior; a.has_lens = P.has_lens;
    a.per_iter = per_iter;
    a.counts = gen ? (const int32_t *)(dd + r_gen) : nullptr;
    a.threshold = b->threshold;
    a.pa = (const double *)(db + o_pa); a.pb = (const double *)(db + o_pb); a.bin_xy = (const double *)(db + o_xy);
    a.models = gen ? (const double *)(dd + r_models) : (const double *)(db + o_models);
    a.d2v = (const int32_t *)(db + o_d2v); a.valid_sizes = (const int32_t *)(db + o_vs);
    a.lens_w = (const float *)(db + o_lw); a.prior = db + o_pm;
    a.masks = inline_masks ? dd + r_masks : P.masks;
    a.bin_inliers = inline_masks ? (int32_t *)(dd + r_bins) : P.bins;
    a.results = (rdvio_parsac_result *)dd;
    hipLaunchKernelGGL(parsac_score_kernel, dim3(nm), dim3(64), 0, st, a);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    uint8_t *down0 = (uint8_t *)P.host + P.in_bytes;
    rdvio_parsac_result *down = (rdvio_parsac_result *)down0;
    uint8_t *down_gen = down0 + r_gen;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down0, dd, down_total, hipMemcpyDeviceToHost, st));
    P.masks_host = P.bins_host = 0;
    if (inline_masks) {
        P.masks_host = P.in_bytes + r_masks;
        P.bins_host = P.in_bytes + r_bins;
    }
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    if (!gen) {
        memcpy(results, down, (size_t)nm * sizeof(rdvio_parsac_result));
        for (int k = 0; k < nm; ++k) P.slot_of[k] = k;
        P.nm = nm;
        return RDVIO_OK;
    }
    // pack the occupied slots in iteration order (the order Parsac<>::solve meets the models in)
    const int32_t *cnt = (const int32_t *)down_gen;
    const double *mod = (const double *)(down0 + r_models);
    int packed = 0;
    for (int it = 0; it < n_iter; ++it) {
        const int c = cnt[it] < 0 ? 0 : (cnt[it] > per_iter ? per_iter : cnt[it]);
        models_per_iteration[it] = c;
        for (int k = 0; k < c; ++k, ++packed) {
            const int slot = it * per_iter + k;
            memcpy(models_out + (size_t)packed * mdim, mod + (size_t)slot * mdim, (size_t)mdim * sizeof(double));
            results[packed] = down[slot];
            P.slot_of[packed] = slot;
        }
    }
    P.nm = packed;
    return RDVIO_OK;
}

int rdvio_hip_parsac_score(rdvio_hip_ctx *ctx, const rdvio_parsac_batch *b, rdvio_parsac_result *results) {
    return parsac_run(ctx, 0, b, results, 0, nullptr, nullptr, nullptr);
}

int rdvio_hip_parsac_generate_score(rdvio_hip_ctx *ctx, const rdvio_parsac_batch *b, int n_iterations, const int32_t *samples,
                                    int32_t *models_per_iteration, double *models, rdvio_parsac_result *results) {
    if (n_iterations <= 0) return ctx ? rdvio_fail(ctx, RDVIO_ERR_INVALID, "no PARSAC iterations to generate") : RDVIO_ERR_INVALID;
    return parsac_run(ctx, 0, b, results, n_iterations, samples, models_per_iteration, models);
}

int rdvio_hip_parsac_fetch(rdvio_hip_ctx *ctx, int model, uint8_t *mask, int32_t *bin_inliers) { return parsac_fetch(ctx, 0, model, mask, bin_inliers); }

// Frame::track_keypoints' two RANSAC gates (frame.cpp:108-118; ransac.h:31-76): the same machinery on the FRONTEND lane with the
// tracker's own staging, scored by inlier count alone (one bin holding every point stands in for the grid)
int rdvio_hip_ransac_generate_score(rdvio_hip_ctx *ctx, int kind, int n_points, int points_changed, const double *pa, const double *pb, double threshold,
                                    int n_iterations, const int32_t *samples, int32_t *models_per_iteration, double *models, int32_t *inlier_counts) {
    if (!ctx || n_points <= 0 || n_iterations <= 0 || !inlier_counts) return RDVIO_ERR_INVALID;
    if (kind != 0 && kind != 2) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "RANSAC gate kind must be 0 (essential) or 2 (rotation)");
    rdvio_hip_ctx::PsState &P = ctx->ps[1];
    if (n_points > P.max_points) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d points exceed capacity %d", n_points, P.max_points);
    static thread_local std::vector<int32_t> d2v;
    if ((int)d2v.size() < n_points) d2v.assign((size_t)n_points, 0);
    const int32_t vs = n_points;
    const double bxy[2] = {0.0, 0.0};
    rdvio_parsac_batch b{};
    b.kind = kind; b.n_points = n_points; b.points_changed = points_changed; b.pa = pa; b.pb = pb; b.threshold = threshold;
    b.n_valid = 1; b.data_to_valid = d2v.data(); b.valid_sizes = &vs; b.bin_xy = bxy;
    const int per = kind == 0 ? 10 : 1;
    static thread_local std::vector<rdvio_parsac_result> res;
    res.resize((size_t)n_iterations * per);
    if (int rc = parsac_run(ctx, 1, &b, res.data(), n_iterations, samples, models_per_iteration, models)) return rc;
    int packed = 0;
    for (int it = 0; it < n_iterations; ++it)
        for (int k = 0; k < models_per_iteration[it]; ++k, ++packed) inlier_counts[packed] = res[(size_t)packed].count;
    return RDVIO_OK;
}

int rdvio_hip_ransac_fetch(rdvio_hip_ctx *ctx, int model, uint8_t *mask) { return parsac_fetch(ctx, 1, model, mask, nullptr); }


static int parsac_fetch(rdvio_hip_ctx *ctx, int which, int model, uint8_t *mask, int32_t *bin_inliers) {
    if (!ctx) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::PsState &P = ctx->ps[which];
    if (model < 0 || model >= P.nm) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "PARSAC model %d was not in the last scored batch", model);
    model = P.slot_of[model];   // a generated batch leaves unoccupied slots between its models
    if (P.masks_host) {         // already on the host (parsac_run)
        if (mask) memcpy(mask, (const uint8_t *)P.host + P.masks_host + (size_t)model * P.n, (size_t)P.n);
        if (bin_inliers)
            memcpy(bin_inliers, (const uint8_t *)P.host + P.bins_host + (size_t)model * P.nv * sizeof(int32_t), (size_t)P.nv * sizeof(int32_t));
        return RDVIO_OK;
    }
    hipStream_t st = ctx->lane[P.lane];
    uint8_t *down = (uint8_t *)P.host + P.in_bytes;
    const size_t mb = ((size_t)P.n + 15) & ~(size_t)15;
    if (mask) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down, P.masks + (size_t)model * P.n, (size_t)P.n, hipMemcpyDeviceToHost, st));
    if (bin_inliers)
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down + mb, P.bins + (size_t)model * P.nv, (size_t)P.nv * sizeof(int32_t),
                                            hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    if (mask) memcpy(mask, down, (size_t)P.n);
    if (bin_inliers) memcpy(bin_inliers, down + mb, (size_t)P.nv * sizeof(int32_t));
    return RDVIO_OK;
}

}  // extern "C"
