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
#include "ctx.hpp"

namespace {

struct PsArgs {
    int kind, n, n_valid, n_models, has_prior, has_lens;
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
    for (int i = lane; i < a.n_valid; i += 64) bins[i] = 0;
    __syncthreads();
    const double *M = a.models + (a.kind == 1 ? 12 : 9) * (size_t)m;
    double Mr[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) Mr[q] = (q < 9 || a.kind == 1) ? M[q] : 0.0;
    int count = 0, eff = 0;
    for (int i = lane; i < a.n; i += 64) {
        double err;
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
    for (int i = lane; i < a.n_valid; i += 64) a.bin_inliers[(size_t)m * RDVIO_PARSAC_MAX_BINS + i] = bins[i];
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

int rdvio_hip_parsac_score(rdvio_hip_ctx *ctx, const rdvio_parsac_batch *b, rdvio_parsac_result *results) {
    if (!ctx || !b || !results) return RDVIO_ERR_INVALID;
    const int n = b->n_points, nv = b->n_valid, nm = b->n_models;
    if ((b->kind != 0 && b->kind != 1) || n <= 0 || nv <= 0 || nm < 0 || !b->pa || !b->pb || !b->data_to_valid || !b->valid_sizes || !b->bin_xy ||
        (nm > 0 && !b->models))
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad PARSAC batch");
    if (n > ctx->ps_max_points || nv > RDVIO_PARSAC_MAX_BINS || nm > RDVIO_PARSAC_MAX_MODELS)
        return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "PARSAC batch of %d points / %d bins / %d models exceeds capacity (%d / %d / %d)", n, nv, nm,
                          ctx->ps_max_points, RDVIO_PARSAC_MAX_BINS, RDVIO_PARSAC_MAX_MODELS);
    if (!b->points_changed && (ctx->ps_n != n || ctx->ps_kind != b->kind || ctx->ps_nv != nv))
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "PARSAC batch reuses points that were never uploaded");
    for (int i = 0; b->points_changed && i < n; ++i)
        if (b->data_to_valid[i] < 0 || b->data_to_valid[i] >= nv) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "PARSAC point %d maps outside the occupied bins", i);
    if (nm == 0) return RDVIO_OK;
    hipStream_t st = ctx->lane[RDVIO_LANE_SOLVER];
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));  // the pinned blob may still be in flight
    uint8_t *hb = (uint8_t *)ctx->ps_host, *db = (uint8_t *)ctx->ps_dev;
    const int pdim = b->kind == 1 ? 3 : 2, mdim = b->kind == 1 ? 12 : 9;
    // (a batch that reuses the uploaded points keeps their layout, whatever optional pointers it passes)
    const bool has_lens = b->points_changed ? b->lens_weight != nullptr : ctx->ps_has_lens;
    const bool has_prior = b->points_changed ? b->prior_mask != nullptr : ctx->ps_has_prior;
    // static part (points, grid) at fixed offsets, models behind it; one copy each
    size_t off = 0;
    const size_t o_pa = put(hb, off, b->points_changed ? b->pa : (const double *)nullptr, (size_t)n * pdim);
    const size_t o_pb = put(hb, off, b->points_changed ? b->pb : (const double *)nullptr, (size_t)n * 2);
    const size_t o_xy = put(hb, off, b->points_changed ? b->bin_xy : (const double *)nullptr, (size_t)nv * 2);
    const size_t o_d2v = put(hb, off, b->points_changed ? b->data_to_valid : (const int32_t *)nullptr, (size_t)n);
    const size_t o_vs = put(hb, off, b->points_changed ? b->valid_sizes : (const int32_t *)nullptr, (size_t)nv);
    const size_t o_lw = put(hb, off, b->points_changed ? b->lens_weight : (const float *)nullptr, has_lens ? (size_t)nv : 0);
    const size_t o_pm = put(hb, off, b->points_changed ? b->prior_mask : (const uint8_t *)nullptr, has_prior ? (size_t)n : 0);
    const size_t static_bytes = (off + 15) & ~(size_t)15;
    const size_t o_models = put(hb, off, b->models, (size_t)nm * mdim);
    if (off > ctx->ps_in_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "PARSAC staging buffer too small");
    if (b->points_changed) {
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(db, hb, off, hipMemcpyHostToDevice, st));
        ctx->ps_n = n;
        ctx->ps_kind = b->kind;
        ctx->ps_nv = nv;
        ctx->ps_has_prior = has_prior;
        ctx->ps_has_lens = has_lens;
    } else {
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(db + o_models, hb + o_models, off - o_models, hipMemcpyHostToDevice, st));
    }
    (void)static_bytes;
    PsArgs a;
    a.kind = b->kind; a.n = n; a.n_valid = nv; a.n_models = nm; a.has_prior = ctx->ps_has_prior; a.has_lens = ctx->ps_has_lens;
    a.threshold = b->threshold;
    a.pa = (const double *)(db + o_pa); a.pb = (const double *)(db + o_pb); a.bin_xy = (const double *)(db + o_xy);
    a.models = (const double *)(db + o_models);
    a.d2v = (const int32_t *)(db + o_d2v); a.valid_sizes = (const int32_t *)(db + o_vs);
    a.lens_w = (const float *)(db + o_lw); a.prior = db + o_pm;
    a.masks = ctx->ps_masks; a.bin_inliers = ctx->ps_bins; a.results = ctx->ps_results;
    hipLaunchKernelGGL(parsac_score_kernel, dim3(nm), dim3(64), 0, st, a);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    rdvio_parsac_result *down = (rdvio_parsac_result *)((uint8_t *)ctx->ps_host + ctx->ps_in_bytes);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down, ctx->ps_results, (size_t)nm * sizeof(rdvio_parsac_result), hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    memcpy(results, down, (size_t)nm * sizeof(rdvio_parsac_result));
    ctx->ps_nm = nm;
    return RDVIO_OK;
}

int rdvio_hip_parsac_fetch(rdvio_hip_ctx *ctx, int model, uint8_t *mask, int32_t *bin_inliers) {
    if (!ctx) return RDVIO_ERR_INVALID;
    if (model < 0 || model >= ctx->ps_nm) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "PARSAC model %d was not in the last scored batch", model);
    hipStream_t st = ctx->lane[RDVIO_LANE_SOLVER];
    uint8_t *down = (uint8_t *)ctx->ps_host + ctx->ps_in_bytes;
    const size_t mb = ((size_t)ctx->ps_n + 15) & ~(size_t)15;
    if (mask) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down, ctx->ps_masks + (size_t)model * ctx->ps_n, (size_t)ctx->ps_n, hipMemcpyDeviceToHost, st));
    if (bin_inliers)
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down + mb, ctx->ps_bins + (size_t)model * RDVIO_PARSAC_MAX_BINS, (size_t)ctx->ps_nv * sizeof(int32_t),
                                            hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    if (mask) memcpy(mask, down, (size_t)ctx->ps_n);
    if (bin_inliers) memcpy(bin_inliers, down + mb, (size_t)ctx->ps_nv * sizeof(int32_t));
    return RDVIO_OK;
}

}  // extern "C"
