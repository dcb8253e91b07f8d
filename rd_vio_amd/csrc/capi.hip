// C ABI of librdvio_hip.so (include/rdvio_hip.h): argument checking on the host, staging copies,
// kernel launches.  There is NO CPU fallback: every entry point either runs the HIP kernels or fails.
#include <algorithm>
#include <climits>
#include <cmath>
#include <new>
#include <vector>

#include <cstdlib>

#include "ctx.hpp"
#include "host_select.hpp"
#include "select_caps.hpp"

#include <atomic>
static std::atomic<int> g_live_contexts[64];
int rdvio_live_contexts(int device) { return (device >= 0 && device < 64) ? g_live_contexts[device].load() : 1; }

extern "C" {

const char *rdvio_hip_version(void) { return "rdvio_hip 0.1 (gfx950)"; }

int rdvio_hip_pyr_layout_init(int w, int h, int max_level, rdvio_pyr_layout *L) {
    if (!L || w <= 0 || h <= 0 || max_level < 0) return RDVIO_ERR_INVALID;
    memset(L, 0, sizeof *L);
    L->border = RDVIO_PYR_BORDER;
    int lw = w, lh = h;
    int64_t ioff = 0, doff = 0;
    for (int lv = 0; lv <= max_level && lv < RDVIO_MAX_LEVELS; ++lv) {
        L->w[lv] = lw;
        L->h[lv] = lh;
        const int stride = (lw + 2 * L->border + 63) / 64 * 64;
        L->stride[lv] = stride;
        L->img_off[lv] = ioff;
        L->deriv_off[lv] = doff;
        const int64_t rows = lh + 2 * L->border;
        ioff += (int64_t)stride * rows;
        doff += (int64_t)stride * rows * 2;
        L->levels = lv + 1;
        // cv::buildOpticalFlowPyramid stops before a level that would not exceed the LK window
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= RDVIO_LK_WIN || lh <= RDVIO_LK_WIN) break;
    }
    L->img_bytes = ioff;
    L->deriv_elems = doff;
    return RDVIO_OK;
}

#define CTX_ALLOC(ptr, bytes)                                                                         \
    do {                                                                                              \
        hipError_t e__ = hipMalloc((void **)&(ptr), (bytes));                                         \
        if (e__ != hipSuccess) {                                                                      \
            rdvio_fail(ctx, RDVIO_ERR_HIP, "hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e__)); \
            *out = ctx;                                                                               \
            return RDVIO_ERR_HIP;                                                                     \
        }                                                                                             \
    } while (0)

int rdvio_hip_ctx_create(rdvio_hip_ctx **out, int device, int max_w, int max_h, int max_feat, int max_window,
                         int max_factors, void *stream) {
    if (!out) return RDVIO_ERR_INVALID;
    *out = nullptr;
    if (max_w < 32 || max_h < 32 || max_feat <= 0 || max_window <= 0 || max_factors <= 0) return RDVIO_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RDVIO_ERR_HIP;  // fail loudly: no GPU, no product
    if (device < 0 || device >= ndev) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx *ctx = new (std::nothrow) rdvio_hip_ctx();
    if (!ctx) return RDVIO_ERR_HIP;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        return RDVIO_ERR_HIP;
    }
    ctx->max_w = max_w;
    ctx->max_h = max_h;
    ctx->max_feat = max_feat;
    ctx->max_window = max_window;
    ctx->max_factors = max_factors;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return RDVIO_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    for (int l = 0; l < 3; ++l) ctx->lane[l] = ctx->stream;
    rdvio_hip_pyr_layout_init(max_w, max_h, RDVIO_MAX_LEVELS - 1, &ctx->maxL);
    for (int s = 0; s < RDVIO_NUM_SLOTS; ++s) {
        CTX_ALLOC(ctx->slots[s].pyr_img, (size_t)ctx->maxL.img_bytes);
        CTX_ALLOC(ctx->slots[s].pyr_deriv, (size_t)ctx->maxL.deriv_elems * sizeof(int16_t));
    }
    CTX_ALLOC(ctx->gray, (size_t)max_w * max_h);
    for (int sl = 0; sl < RDVIO_NUM_SLOTS; ++sl) {
        CTX_ALLOC(ctx->gray_slot[sl], (size_t)max_w * max_h);
        if (hipHostMalloc((void **)&ctx->gray_pinned[sl], (size_t)max_w * max_h, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->gray_ev[sl], hipEventDisableTiming) != hipSuccess) {
            rdvio_fail(ctx, RDVIO_ERR_HIP, "image staging allocation failed");
            *out = ctx;
            return RDVIO_ERR_HIP;
        }
    }
    CTX_ALLOC(ctx->clahe_lut, (size_t)RDVIO_MAX_TILES * 256);
    CTX_ALLOC(ctx->harris, (size_t)max_w * max_h * sizeof(float));
    CTX_ALLOC(ctx->harris_scalars, 4 * sizeof(uint32_t));
    ctx->harris_cand_cap = (max_w * max_h) / 4 + 1024;  // 3x3 strict local maxima cannot exceed 1/4 of the pixels
    CTX_ALLOC(ctx->harris_cand, (size_t)ctx->harris_cand_cap * sizeof(HarrisCand));
    CTX_ALLOC(ctx->sel_hdr, 64 + (size_t)RDVIO_SEL_CORNERS_MAX * 2 * sizeof(double));
    ctx->sel_new = (double *)(ctx->sel_hdr + 16);
    CTX_ALLOC(ctx->sel_corners, (size_t)RDVIO_SEL_CORNERS_MAX * 2 * sizeof(float));
    CTX_ALLOC(ctx->sel_existing, (size_t)RDVIO_SEL_PTS_MAX * 2 * sizeof(double));
    CTX_ALLOC(ctx->lk_curr, (size_t)max_feat * (4 * sizeof(double) + 1) + 64);  // curr | next | status for the host entry point
    CTX_ALLOC(ctx->lk_next, (size_t)max_feat * 2 * sizeof(double));
    CTX_ALLOC(ctx->lk_prevf, (size_t)max_feat * 2 * sizeof(float));
    CTX_ALLOC(ctx->lk_nextf, (size_t)max_feat * 2 * sizeof(float));
    CTX_ALLOC(ctx->lk_status, (size_t)max_feat);
    if (const char *e = getenv("RDVIO_HOST_SELECT")) ctx->force_host_select = e[0] == '1';  // diagnostic: keypoint selection on the host road
    if (const char *e = getenv("RDVIO_HELPER_MIN_FACTORS")) ctx->helper_min_factors = std::max(atoi(e), 1);
    if (const char *e = getenv("RDVIO_SOLVER_WGS")) ctx->solver_wgs = std::min(std::max(atoi(e), 1), RDVIO_MAX_SOLVER_WGS);
    const int nfr = max_window + 2;
    const int max_lm = max_factors;  // every factor could belong to its own landmark
    CTX_ALLOC(ctx->ba_states, (size_t)nfr * 16 * sizeof(double));
    CTX_ALLOC(ctx->ba_extr, 18 * sizeof(double));
    CTX_ALLOC(ctx->ba_zref, (size_t)max_lm * 3 * sizeof(double));
    CTX_ALLOC(ctx->ba_invd, (size_t)max_lm * sizeof(double));
    CTX_ALLOC(ctx->ba_tangent, (size_t)max_factors * 9 * sizeof(double));
    CTX_ALLOC(ctx->ba_idx, (size_t)max_factors * 3 * sizeof(int32_t));
    CTX_ALLOC(ctx->ba_r, (size_t)max_factors * 2 * sizeof(double));
    CTX_ALLOC(ctx->ba_Jt, (size_t)max_factors * 12 * sizeof(double));
    CTX_ALLOC(ctx->ba_Jr, (size_t)max_factors * 12 * sizeof(double));
    CTX_ALLOC(ctx->ba_Jd, (size_t)max_factors * 2 * sizeof(double));
    ctx->pre_max_seg = nfr + 8;
    ctx->pre_max_samples = ctx->pre_max_seg * 256;
    CTX_ALLOC(ctx->pre_out, (size_t)ctx->pre_max_seg * RDVIO_PREINT_SIZE * sizeof(double));
    CTX_ALLOC(ctx->pre_blob, ((size_t)ctx->pre_max_seg * 7 + 36 + (size_t)ctx->pre_max_samples * 7 + (size_t)ctx->pre_max_seg + 8) * sizeof(double));
    CTX_ALLOC(ctx->pre2_out, (size_t)ctx->pre_max_seg * RDVIO_PREINT_SIZE * sizeof(double));
    CTX_ALLOC(ctx->pre2_blob, ((size_t)ctx->pre_max_seg * 7 + 36 + (size_t)ctx->pre_max_samples * 7 + (size_t)ctx->pre_max_seg + 8) * sizeof(double));
    {
        const size_t F = (size_t)max_factors, Lm = (size_t)max_factors, Nmax = 15 * (size_t)nfr, npre = (size_t)nfr + 8;
        size_t bytes = (1 << 16) + F * (520 + 3 * 26 * 8 + 12) + Lm * (192 + 48 * (size_t)nfr) + Nmax * Nmax * 8 * 7 + npre * 930 * 8 +
                       npre * (RDVIO_PREINT_SIZE + 1400) * 8 + Nmax * 8 * 32 +
                       (size_t)RDVIO_MAX_SOLVER_WGS * (6 * (size_t)nfr + 2) * (6 * (size_t)nfr + 2) * 8;  // (per-workgroup partial Schur terms)
        bytes += bytes / 4;
        ctx->ba_arena_bytes = ctx->ba_host_bytes = bytes;
        for (int s = 0; s < RDVIO_BA_SLOTS; ++s) {
            CTX_ALLOC(ctx->ba[s].arena, bytes);
            if (hipHostMalloc(&ctx->ba[s].host, bytes, hipHostMallocDefault) != hipSuccess) {
                rdvio_fail(ctx, RDVIO_ERR_HIP, "hipHostMalloc(ba host blob) failed");
                *out = ctx;
                return RDVIO_ERR_HIP;
            }
        }
    }
    {
        // marginalisation slot: a BA problem with every frame free + the tail's dense scratch (8 N x N matrices)
        const size_t Nmax = 15 * (size_t)nfr;
        const size_t bytes = ctx->ba_arena_bytes + Nmax * Nmax * 8 * 8 + (1 << 16);
        ctx->marg_bytes = bytes;
        CTX_ALLOC(ctx->marg.arena, bytes);
        if (hipHostMalloc(&ctx->marg.host, bytes, hipHostMallocDefault) != hipSuccess) {
            rdvio_fail(ctx, RDVIO_ERR_HIP, "hipHostMalloc(marg host blob) failed");
            *out = ctx;
            return RDVIO_ERR_HIP;
        }
    }
    for (int which = 0; which < 2; ++which) {
        rdvio_hip_ctx::PsState &P = ctx->ps[which];
        P.lane = which == 0 ? RDVIO_LANE_SOLVER : RDVIO_LANE_FRONTEND;
        P.max_points = std::max(4096, 4 * max_feat);
        const size_t n = (size_t)P.max_points;
        P.in_bytes = (n * (6 * 8 + 4 + 1) + RDVIO_PARSAC_MAX_BINS * 24 + (size_t)RDVIO_PARSAC_MAX_MODELS * 12 * 8 + 4096 + 15) & ~(size_t)15;
        CTX_ALLOC(P.dev, P.in_bytes);
        CTX_ALLOC(P.masks, (size_t)RDVIO_PARSAC_MAX_MODELS * n);
        CTX_ALLOC(P.bins, (size_t)RDVIO_PARSAC_MAX_MODELS * RDVIO_PARSAC_MAX_BINS * sizeof(int32_t));
        CTX_ALLOC(P.results, (size_t)RDVIO_PARSAC_MAX_MODELS * sizeof(rdvio_parsac_result));
        // behind the inputs: the results of a batch -- per-model records, generated models, and (when small enough to ride along)
        // every model's inlier mask and bin counts
        P.down_bytes = (size_t)RDVIO_PARSAC_MAX_MODELS * (sizeof(rdvio_parsac_result) + 12 * 8 + 4) + n + (size_t)RDVIO_PARSAC_MASKS_INLINE + 65536;
        CTX_ALLOC(P.down_dev, P.down_bytes);
        if (hipHostMalloc(&P.host, P.in_bytes + P.down_bytes, hipHostMallocDefault) != hipSuccess) {
            rdvio_fail(ctx, RDVIO_ERR_HIP, "hipHostMalloc(parsac blob) failed");
            *out = ctx;
            return RDVIO_ERR_HIP;
        }
    }
    {
        ctx->thin_bytes = (size_t)std::max(4096, 4 * max_feat) * (2 * 8 + 4 + 1 + 1) + 4096;
        CTX_ALLOC(ctx->thin_dev, ctx->thin_bytes);
        if (hipHostMalloc(&ctx->thin_host, ctx->thin_bytes, hipHostMallocDefault) != hipSuccess) {
            rdvio_fail(ctx, RDVIO_ERR_HIP, "hipHostMalloc(thinning blob) failed");
            *out = ctx;
            return RDVIO_ERR_HIP;
        }
    }
    ctx->pinned_bytes = std::max<size_t>((size_t)ctx->harris_cand_cap * sizeof(HarrisCand) + 64, 1 << 20);
    ctx->pinned_bytes = std::max<size_t>(ctx->pinned_bytes, ((size_t)ctx->pre_max_seg * (7 + RDVIO_PREINT_SIZE + 1) + 64 + (size_t)ctx->pre_max_samples * 7) * sizeof(double));
    if (hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        rdvio_fail(ctx, RDVIO_ERR_HIP, "hipHostMalloc failed");
        *out = ctx;
        return RDVIO_ERR_HIP;
    }
    ctx->pre2_pinned_bytes = ((size_t)ctx->pre_max_seg * (7 + RDVIO_PREINT_SIZE + 1) + 64 + (size_t)ctx->pre_max_samples * 7) * sizeof(double);
    if (hipHostMalloc(&ctx->pre2_pinned, ctx->pre2_pinned_bytes, hipHostMallocDefault) != hipSuccess) {
        rdvio_fail(ctx, RDVIO_ERR_HIP, "hipHostMalloc failed");
        *out = ctx;
        return RDVIO_ERR_HIP;
    }
    if (device < 64) {
        g_live_contexts[device].fetch_add(1);
        ctx->counted = true;
    }
    *out = ctx;
    return RDVIO_OK;
}

void rdvio_hip_ctx_destroy(rdvio_hip_ctx *ctx) {
    if (!ctx) return;
    if (ctx->counted) g_live_contexts[ctx->device].fetch_sub(1);
    (void)hipSetDevice(ctx->device);
    for (int l = 0; l < 3; ++l) (void)rdvio_wait(ctx, ctx->lane[l]);
    for (int l = 0; l < 3; ++l) {
        if (ctx->wait_ev[l]) (void)hipEventDestroy(ctx->wait_ev[l]);
        ctx->wait_ev[l] = nullptr;
    }
    for (int s = 0; s < RDVIO_NUM_SLOTS; ++s) {
        (void)hipFree(ctx->slots[s].pyr_img);
        (void)hipFree(ctx->slots[s].pyr_deriv);
        (void)hipFree(ctx->gray_slot[s]);
        if (ctx->gray_pinned[s]) (void)hipHostFree(ctx->gray_pinned[s]);
        if (ctx->gray_ev[s]) (void)hipEventDestroy(ctx->gray_ev[s]);
    }
    void *bufs[] = {ctx->gray, ctx->clahe_lut, ctx->harris, ctx->harris_scalars, ctx->harris_cand, ctx->sel_hdr, ctx->sel_corners, ctx->sel_existing, ctx->ps[0].dev, ctx->ps[0].down_dev, ctx->ps[1].down_dev, ctx->ps[0].masks, ctx->ps[0].bins, ctx->ps[0].results, ctx->ps[1].dev, ctx->ps[1].masks, ctx->ps[1].bins,
                    ctx->ps[1].results, ctx->thin_dev,
                    ctx->lk_curr,
                    ctx->lk_next, ctx->lk_prevf, ctx->lk_nextf, ctx->lk_status, ctx->ba_states, ctx->ba_extr,
                    ctx->ba_zref, ctx->ba_invd, ctx->ba_tangent, ctx->ba_idx, ctx->ba_r, ctx->ba_Jt, ctx->ba_Jr,
                    ctx->ba_Jd, ctx->pre_out, ctx->pre_blob, ctx->pre2_out, ctx->pre2_blob};
    for (void *b : bufs) (void)hipFree(b);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pre2_pinned) (void)hipHostFree(ctx->pre2_pinned);
    for (int which = 0; which < 2; ++which)
        if (ctx->ps[which].host) (void)hipHostFree(ctx->ps[which].host);
    if (ctx->thin_host) (void)hipHostFree(ctx->thin_host);
    if (ctx->marg.host) (void)hipHostFree(ctx->marg.host);
    (void)hipFree(ctx->marg.arena);
    for (int s = 0; s < RDVIO_BA_SLOTS; ++s) {
        if (ctx->ba[s].ev0) (void)hipEventDestroy(ctx->ba[s].ev0);
        if (ctx->ba[s].ev1) (void)hipEventDestroy(ctx->ba[s].ev1);
        if (ctx->ba[s].up_ev) (void)hipEventDestroy(ctx->ba[s].up_ev);
        if (s == 0 && ctx->chain_ev) (void)hipEventDestroy(ctx->chain_ev);
        if (s == 0 && ctx->chain_stream) (void)hipStreamDestroy(ctx->chain_stream);
        if (ctx->ba[s].host) (void)hipHostFree(ctx->ba[s].host);
        (void)hipFree(ctx->ba[s].arena);
    }
    for (int l = 0; l < 3; ++l) {
        if (ctx->lane_ev[l]) (void)hipEventDestroy(ctx->lane_ev[l]);
        if (ctx->own_lane[l]) (void)hipStreamDestroy(ctx->lane[l]);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rdvio_hip_last_error(const rdvio_hip_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int rdvio_hip_sync(rdvio_hip_ctx *ctx) {
    if (!ctx) return RDVIO_ERR_INVALID;
    for (int l = 0; l < 3; ++l)
        if (l == 0 || ctx->lane[l] != ctx->lane[0]) RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[l]));
    return RDVIO_OK;
}

static bool bad_lane(int lane) { return lane < 0 || lane > 2; }

int rdvio_hip_ctx_set_wait_mode(rdvio_hip_ctx *ctx, int blocking) {
    if (!ctx) return RDVIO_ERR_INVALID;
    RDVIO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (int l = 2; l >= 0; --l)   // [0] last: rdvio_wait tests it
        if (blocking && !ctx->wait_ev[l]) RDVIO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->wait_ev[l], hipEventBlockingSync | hipEventDisableTiming));
    ctx->blocking_wait = blocking != 0;
    return RDVIO_OK;
}

int rdvio_hip_ctx_set_lane_stream(rdvio_hip_ctx *ctx, int lane, void *stream) {
    if (!ctx || bad_lane(lane)) return RDVIO_ERR_INVALID;
    if (lane == RDVIO_LANE_FRONTEND) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "the frontend lane is the context's stream (rdvio_hip_ctx_create)");
    RDVIO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[lane]));
    for (int l = 0; l < 3; ++l)   // the lanes' dependency events belong to the context's device, whichever thread waits later
        if (!ctx->lane_ev[l]) RDVIO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->lane_ev[l], hipEventDisableTiming));
    if (ctx->own_lane[lane]) {
        RDVIO_HIP_CHECK(ctx, hipStreamDestroy(ctx->lane[lane]));
        ctx->own_lane[lane] = false;
    }
    if (stream) {
        ctx->lane[lane] = (hipStream_t)stream;
    } else {
        hipStream_t s = nullptr;
        RDVIO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        ctx->lane[lane] = s;
        ctx->own_lane[lane] = true;
    }
    return RDVIO_OK;
}

int rdvio_hip_ctx_ensure_lane_streams(rdvio_hip_ctx *ctx) {
    if (!ctx) return RDVIO_ERR_INVALID;
    for (int lane = RDVIO_LANE_SOLVER; lane <= RDVIO_LANE_MARG; ++lane)
        if (ctx->lane[lane] == ctx->lane[RDVIO_LANE_FRONTEND])
            if (int rc = rdvio_hip_ctx_set_lane_stream(ctx, lane, nullptr)) return rc;
    return RDVIO_OK;
}

int rdvio_hip_lane_wait(rdvio_hip_ctx *ctx, int lane, int on_lane) {
    if (!ctx || bad_lane(lane) || bad_lane(on_lane)) return RDVIO_ERR_INVALID;
    if (ctx->lane[lane] == ctx->lane[on_lane]) return RDVIO_OK;  // same stream: already ordered
    if (!ctx->lane_ev[on_lane]) RDVIO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->lane_ev[on_lane], hipEventDisableTiming));
    RDVIO_HIP_CHECK(ctx, hipEventRecord(ctx->lane_ev[on_lane], ctx->lane[on_lane]));
    RDVIO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->lane[lane], ctx->lane_ev[on_lane], 0));
    return RDVIO_OK;
}

int rdvio_hip_lane_sync(rdvio_hip_ctx *ctx, int lane) {
    if (!ctx || bad_lane(lane)) return RDVIO_ERR_INVALID;
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[lane]));
    return RDVIO_OK;
}

// ------------------------------------------------------------------------------------------ seam 1
static int check_image_args(rdvio_hip_ctx *ctx, int slot, const void *gray, int w, int h, int stride, int tx, int ty) {
    if (!ctx) return RDVIO_ERR_INVALID;
    if (slot < 0 || slot >= RDVIO_NUM_SLOTS) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "slot %d out of range", slot);
    if (!gray || w < 32 || h < 32 || stride < w) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad image %dx%d stride %d", w, h, stride);
    if (w > ctx->max_w || h > ctx->max_h) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "image %dx%d exceeds context %dx%d", w, h, ctx->max_w, ctx->max_h);
    if (tx <= 0 || ty <= 0 || tx * ty > RDVIO_MAX_TILES) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad CLAHE tile grid %dx%d", tx, ty);
    return RDVIO_OK;
}

int rdvio_hip_image_preprocess_dev(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray_dev, int w, int h, int stride,
                                   double clip, int tiles_x, int tiles_y) {
    if (int rc = check_image_args(ctx, slot, gray_dev, w, h, stride, tiles_x, tiles_y)) return rc;
    return rdvio_launch_preprocess(ctx, slot, gray_dev, w, h, stride, clip, tiles_x, tiles_y);
}

int rdvio_hip_image_upload(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray, int w, int h, int stride) {
    if (int rc = check_image_args(ctx, slot, gray, w, h, stride, 1, 1)) return rc;
    // the pinned buffer of this slot may still feed the previous upload
    RDVIO_HIP_CHECK(ctx, hipEventSynchronize(ctx->gray_ev[slot]));
    uint8_t *pin = ctx->gray_pinned[slot];
    if (stride == w) memcpy(pin, gray, (size_t)w * h);
    else
        for (int y = 0; y < h; ++y) memcpy(pin + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->gray_slot[slot], pin, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream));
    RDVIO_HIP_CHECK(ctx, hipEventRecord(ctx->gray_ev[slot], ctx->stream));
    ctx->gray_w[slot] = w;
    ctx->gray_h[slot] = h;
    return RDVIO_OK;
}

int rdvio_hip_image_preprocess_uploaded(rdvio_hip_ctx *ctx, int slot, double clip, int tiles_x, int tiles_y) {
    if (!ctx || slot < 0 || slot >= RDVIO_NUM_SLOTS) return RDVIO_ERR_INVALID;
    if (ctx->gray_w[slot] <= 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no image uploaded for slot %d", slot);
    if (int rc = check_image_args(ctx, slot, ctx->gray_slot[slot], ctx->gray_w[slot], ctx->gray_h[slot], ctx->gray_w[slot], tiles_x, tiles_y)) return rc;
    return rdvio_launch_preprocess(ctx, slot, ctx->gray_slot[slot], ctx->gray_w[slot], ctx->gray_h[slot], ctx->gray_w[slot], clip, tiles_x, tiles_y);
}

int rdvio_hip_image_preprocess(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray, int w, int h, int stride, double clip,
                               int tiles_x, int tiles_y) {
    if (int rc = check_image_args(ctx, slot, gray, w, h, stride, tiles_x, tiles_y)) return rc;
    // through the slot's pinned staging buffer: when this returns the caller's pixels have been read (an asynchronous copy
    // straight out of pageable memory could still be reading them)
    if (int rc = rdvio_hip_image_upload(ctx, slot, gray, w, h, stride)) return rc;
    return rdvio_launch_preprocess(ctx, slot, ctx->gray_slot[slot], w, h, w, clip, tiles_x, tiles_y);
}

int rdvio_hip_image_download(rdvio_hip_ctx *ctx, int slot, uint8_t *pyr_img, int16_t *pyr_deriv) {
    if (!ctx || slot < 0 || slot >= RDVIO_NUM_SLOTS) return RDVIO_ERR_INVALID;
    ImageSlot &S = ctx->slots[slot];
    if (!S.valid) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "slot %d not preprocessed", slot);
    if (pyr_img) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(pyr_img, S.pyr_img, (size_t)S.L.img_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (pyr_deriv)
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(pyr_deriv, S.pyr_deriv, (size_t)S.L.deriv_elems * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->stream));
    return RDVIO_OK;
}

int rdvio_hip_image_release(rdvio_hip_ctx *ctx, int slot) {
    if (!ctx || slot < 0 || slot >= RDVIO_NUM_SLOTS) return RDVIO_ERR_INVALID;
    ctx->slots[slot].valid = false;  // buffers are context-owned and reused by the next preprocess
    return RDVIO_OK;
}

int rdvio_hip_track_keypoints_dev(rdvio_hip_ctx *ctx, int slot_curr, int slot_next, int n, const double *curr_dev,
                                  double *next_dev, int has_guess, uint8_t *status_dev) {
    if (!ctx || n < 0 || (n > 0 && (!curr_dev || !next_dev || !status_dev))) return RDVIO_ERR_INVALID;
    return rdvio_launch_track(ctx, slot_curr, slot_next, n, curr_dev, next_dev, has_guess, status_dev);
}

int rdvio_hip_track_keypoints(rdvio_hip_ctx *ctx, int slot_curr, int slot_next, int n, const double *curr_xy,
                              double *next_xy, int has_guess, uint8_t *status) {
    if (!ctx || n < 0 || (n > 0 && (!curr_xy || !next_xy || !status))) return RDVIO_ERR_INVALID;
    if (n > ctx->max_feat) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d features exceed context capacity %d", n, ctx->max_feat);
    if (n == 0) return RDVIO_OK;
    // one pinned blob up (curr | guess), one kernel, one pinned blob down (next | status): lk_curr / lk_next / lk_status are
    // carved from one allocation, so both directions are single copies
    const size_t nd = (size_t)n * 2;
    double *up = (double *)ctx->pinned;
    memcpy(up, curr_xy, nd * sizeof(double));
    memcpy(up + nd, next_xy, nd * sizeof(double));
    double *d_curr = ctx->lk_curr, *d_next = ctx->lk_curr + nd;
    uint8_t *d_status = (uint8_t *)(ctx->lk_curr + 2 * nd);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(d_curr, up, 2 * nd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = rdvio_launch_track(ctx, slot_curr, slot_next, n, d_curr, d_next, has_guess, d_status)) return rc;
    uint8_t *down = (uint8_t *)(up + 2 * nd);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down, d_next, nd * sizeof(double) + (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->stream));
    memcpy(next_xy, down, nd * sizeof(double));
    memcpy(status, down + nd * sizeof(double), (size_t)n);
    return RDVIO_OK;
}

int rdvio_hip_lk_flow(rdvio_hip_ctx *ctx, int slot_prev, int slot_next, int n, const float *prev_xy, float *next_xy,
                      uint8_t *status, int max_iter, double eps) {
    if (!ctx || n < 0 || (n > 0 && (!prev_xy || !next_xy || !status))) return RDVIO_ERR_INVALID;
    if (n > ctx->max_feat) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d features exceed context capacity %d", n, ctx->max_feat);
    if (n == 0) return RDVIO_OK;
    const size_t bytes = (size_t)n * 2 * sizeof(float);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->lk_prevf, prev_xy, bytes, hipMemcpyHostToDevice, ctx->stream));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->lk_nextf, next_xy, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = rdvio_launch_lk_flow(ctx, slot_prev, slot_next, n, ctx->lk_prevf, ctx->lk_nextf, ctx->lk_status, max_iter, eps)) return rc;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(next_xy, ctx->lk_nextf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(status, ctx->lk_status, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->stream));
    return RDVIO_OK;
}

int rdvio_hip_harris_response(rdvio_hip_ctx *ctx, int slot, float *resp) {
    if (!ctx || !resp || slot < 0 || slot >= RDVIO_NUM_SLOTS) return RDVIO_ERR_INVALID;
    ImageSlot &S = ctx->slots[slot];
    if (!S.valid) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "slot %d not preprocessed", slot);
    if (int rc = rdvio_launch_harris(ctx, slot)) return rc;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(resp, ctx->harris, (size_t)S.w * S.h * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->stream));
    return RDVIO_OK;
}

int rdvio_hip_detect_keypoints(rdvio_hip_ctx *ctx, int slot, double *keypoints, int n_existing, int capacity,
                               int max_points, double min_distance, int *n_out) {
    if (!ctx || !keypoints || !n_out || slot < 0 || slot >= RDVIO_NUM_SLOTS || n_existing < 0 || max_points <= 0 ||
        capacity < n_existing)
        return RDVIO_ERR_INVALID;
    ImageSlot &S = ctx->slots[slot];
    if (!S.valid) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "slot %d not preprocessed", slot);
    // GFTTDetector::create(max_points, 1.0e-3, 20, 3, true): opencv_image.cpp:184-188
    if (int rc = rdvio_launch_harris(ctx, slot)) return rc;
    if (int rc = rdvio_launch_harris_candidates(ctx, slot, 1.0e-3)) return rc;
    // Selection on the device (select_kernels.hip): sort, greedy minDistance, Poisson-disk thinning against the existing
    // keypoints, border test.  Up: the existing keypoints; down: a 64-byte header + the accepted new keypoints, one copy.
    const double gftt_min_dist = 20.0;
    const bool device_road = !ctx->force_host_select && n_existing <= RDVIO_SEL_PTS_MAX && max_points <= RDVIO_SEL_CORNERS_MAX;
    if (device_road) {
        if (n_existing > 0)
            RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->sel_existing, keypoints, (size_t)n_existing * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (int rc = rdvio_launch_select(ctx, slot, max_points, gftt_min_dist, min_distance, n_existing)) return rc;
        int32_t *hdr = (int32_t *)ctx->pinned;
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(hdr, ctx->sel_hdr, 64 + (size_t)max_points * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->stream));
        ctx->last_select_path = hdr[1] == 0 ? hdr[4] : -1;
        for (int q = 0; q < 4; ++q) ctx->last_select_stamps[q] = hdr[8 + q];
        ctx->last_select_stamps[4] = hdr[0];
        if (getenv("RDVIO_DEBUG_SELECT")) fprintf(stderr, "[select] first chunk: neighbours collected at %.1f us, decided at %.1f us, %d polling trips (wave 0)\n", hdr[12] / 100.0, hdr[13] / 100.0, hdr[14]);
        if (hdr[1] == 0) {
            const int n_new = hdr[3];
            if (n_existing + n_new > capacity) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "keypoint capacity %d too small", capacity);
            memcpy(keypoints + 2 * (size_t)n_existing, (const uint8_t *)hdr + 64, (size_t)n_new * 2 * sizeof(double));
            *n_out = n_existing + n_new;
            return RDVIO_OK;
        }
        // beyond the kernels' LDS capacities (e.g. > 8192 local maxima on a noise image): the host road below
    }
    uint32_t *scalars = (uint32_t *)ctx->pinned;
    HarrisCand *cand = (HarrisCand *)((uint8_t *)ctx->pinned + 64);
    const int prefix = std::min(ctx->harris_cand_cap, 4096);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(scalars, ctx->harris_scalars, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(cand, ctx->harris_cand, (size_t)prefix * sizeof(HarrisCand), hipMemcpyDeviceToHost, ctx->stream));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->stream));
    const int nc = (int)std::min<uint32_t>(scalars[1], (uint32_t)ctx->harris_cand_cap);
    if (nc > prefix) {
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(cand + prefix, ctx->harris_cand + prefix, (size_t)(nc - prefix) * sizeof(HarrisCand), hipMemcpyDeviceToHost,
                                            ctx->stream));
        RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->stream));
    }
    int total = rdvio_host_select_keypoints(cand, nc, S.w, S.h, max_points, 20.0, min_distance, keypoints, n_existing, capacity);
    if (total < 0) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "keypoint capacity %d too small", capacity);
    *n_out = total;
    return RDVIO_OK;
}

// ------------------------------------------------------------------------------------------ seam 2
// staging of one caller of the host entry point: the frontend lane's (Image / tracker side) or the solver lane's (the
// estimator's integrations, concurrent with the tracker's in a threaded pipeline)
struct PreStage {
    hipStream_t st;
    double *pinned;
    size_t pinned_bytes;
    double *blob, *out;
    size_t *pending = nullptr;   // non-null: enqueue only; [0] offset, [1] length (doubles) of the records in the pinned blob
};

static int preintegrate_host(rdvio_hip_ctx *ctx, const PreStage &S, int nseg, const int32_t *seg_off, const double *imu, const double *t_end,
                             const double *bg, const double *ba, const double *noise, int cj, int cc, double *out) {
    if (!ctx || nseg < 0 || (nseg > 0 && (!seg_off || !imu || !t_end || !bg || !ba || !noise || (!out && !S.pending)))) return RDVIO_ERR_INVALID;
    if (nseg == 0) return RDVIO_OK;
    if (nseg > ctx->pre_max_seg) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d segments exceed capacity %d", nseg, ctx->pre_max_seg);
    if (seg_off[0] != 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "seg_off[0] must be 0");
    for (int i = 0; i < nseg; ++i)
        if (seg_off[i + 1] < seg_off[i]) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "seg_off must be non-decreasing");
    const int ns = seg_off[nseg];
    if (ns > ctx->pre_max_samples) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d samples exceed capacity %d", ns, ctx->pre_max_samples);
    // one pinned blob up (par | noise | samples | offsets), one kernel, one pinned blob down: a frame makes several of
    // these calls, so the per-call fixed cost matters more than the 4 KB of payload
    const size_t o_par = 0, o_noise = (size_t)ctx->pre_max_seg * 7, o_imu = o_noise + 36, o_off = o_imu + (size_t)ctx->pre_max_samples * 7;
    const size_t in_doubles = o_off + ((size_t)ctx->pre_max_seg + 2) / 2 + 1, out_doubles = (size_t)nseg * RDVIO_PREINT_SIZE;
    if ((in_doubles + out_doubles) * sizeof(double) > S.pinned_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "preintegration staging buffer too small");
    double *blob = S.pinned;
    for (int i = 0; i < nseg; ++i) {
        blob[o_par + 7 * i] = t_end[i];
        for (int k = 0; k < 3; ++k) {
            blob[o_par + 7 * i + 1 + k] = bg[3 * i + k];
            blob[o_par + 7 * i + 4 + k] = ba[3 * i + k];
        }
    }
    memcpy(blob + o_noise, noise, 36 * sizeof(double));
    if (ns > 0) memcpy(blob + o_imu, imu, (size_t)ns * 7 * sizeof(double));
    memcpy(blob + o_off, seg_off, (size_t)(nseg + 1) * sizeof(int32_t));
    // only the used prefix of each region travels: par, noise and the samples are contiguous up to the last sample
    const size_t head = (o_imu + (size_t)ns * 7) * sizeof(double);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(S.blob, blob, head, hipMemcpyHostToDevice, S.st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(S.blob + o_off, blob + o_off, (size_t)(nseg + 1) * sizeof(int32_t), hipMemcpyHostToDevice, S.st));
    if (int rc = rdvio_launch_preintegrate(ctx, S.st, nseg, (const int32_t *)(S.blob + o_off), S.blob + o_imu, S.blob + o_par, S.blob + o_noise, cj, cc, S.out))
        return rc;
    double *down = blob + in_doubles;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down, S.out, out_doubles * sizeof(double), hipMemcpyDeviceToHost, S.st));
    if (S.pending) {   // begin / end form: the caller collects the records later
        S.pending[0] = in_doubles;
        S.pending[1] = out_doubles;
        return RDVIO_OK;
    }
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, S.st));
    memcpy(out, down, out_doubles * sizeof(double));
    return RDVIO_OK;
}

int rdvio_hip_preintegrate(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off, const double *imu, const double *t_end,
                           const double *bg, const double *ba, const double *noise, int cj, int cc, double *out) {
    if (!ctx) return RDVIO_ERR_INVALID;
    return preintegrate_host(ctx, PreStage{ctx->stream, (double *)ctx->pinned, ctx->pinned_bytes, ctx->pre_blob, ctx->pre_out}, nseg, seg_off, imu,
                             t_end, bg, ba, noise, cj, cc, out);
}

int rdvio_hip_preintegrate_estimator(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off, const double *imu, const double *t_end,
                                     const double *bg, const double *ba, const double *noise, int cj, int cc, double *out) {
    if (!ctx) return RDVIO_ERR_INVALID;
    return preintegrate_host(ctx, PreStage{ctx->lane[RDVIO_LANE_SOLVER], (double *)ctx->pre2_pinned, ctx->pre2_pinned_bytes, ctx->pre2_blob, ctx->pre2_out},
                             nseg, seg_off, imu, t_end, bg, ba, noise, cj, cc, out);
}

int rdvio_hip_preintegrate_estimator_begin(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off, const double *imu, const double *t_end,
                                           const double *bg, const double *ba, const double *noise, int cj, int cc) {
    if (!ctx) return RDVIO_ERR_INVALID;
    if (ctx->pre2_pending[1] != 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "an estimator preintegration is already in flight");
    ctx->pre2_pending[0] = ctx->pre2_pending[1] = 0;
    PreStage S{ctx->lane[RDVIO_LANE_SOLVER], (double *)ctx->pre2_pinned, ctx->pre2_pinned_bytes, ctx->pre2_blob, ctx->pre2_out};
    S.pending = ctx->pre2_pending;
    return preintegrate_host(ctx, S, nseg, seg_off, imu, t_end, bg, ba, noise, cj, cc, nullptr);
}

int rdvio_hip_preintegrate_estimator_end(rdvio_hip_ctx *ctx, double *out) {
    if (!ctx) return RDVIO_ERR_INVALID;
    if (ctx->pre2_pending[1] == 0) return RDVIO_OK;   // nothing in flight (begin with no segments)
    if (!out) return RDVIO_ERR_INVALID;
    const size_t off = ctx->pre2_pending[0], n = ctx->pre2_pending[1];
    ctx->pre2_pending[0] = ctx->pre2_pending[1] = 0;
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[RDVIO_LANE_SOLVER]));
    memcpy(out, (const double *)ctx->pre2_pinned + off, n * sizeof(double));
    return RDVIO_OK;
}

int rdvio_hip_debug_last_select_path(const rdvio_hip_ctx *ctx) { return ctx ? ctx->last_select_path : -1; }
int rdvio_hip_debug_last_select_stamps(const rdvio_hip_ctx *ctx, int32_t *out5) {
    if (!ctx || !out5) return RDVIO_ERR_INVALID;
    for (int q = 0; q < 5; ++q) out5[q] = ctx->last_select_stamps[q];
    return RDVIO_OK;
}

int rdvio_hip_ctx_attach_thread(rdvio_hip_ctx *ctx) {
    if (!ctx) return RDVIO_ERR_INVALID;
    RDVIO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return RDVIO_OK;
}

int rdvio_hip_preintegrate_dev(rdvio_hip_ctx *ctx, int nseg, const int32_t *seg_off_dev, const double *imu_dev,
                               const double *par_dev, const double *noise_dev, int cj, int cc, double *out_dev) {
    if (!ctx || nseg < 0 || (nseg > 0 && (!seg_off_dev || !imu_dev || !par_dev || !noise_dev || !out_dev))) return RDVIO_ERR_INVALID;
    return rdvio_launch_preintegrate(ctx, ctx->stream, nseg, seg_off_dev, imu_dev, par_dev, noise_dev, cj, cc, out_dev);
}

static int upload_ba_problem(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb) {
    if (!pb || pb->n_frames <= 0 || pb->n_factors < 0 || pb->n_landmarks < 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad BA problem");
    if (pb->n_frames > ctx->max_window + 2) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d frames exceed window capacity", pb->n_frames);
    if (pb->n_factors > ctx->max_factors || pb->n_landmarks > ctx->max_factors)
        return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d factors / %d landmarks exceed capacity %d", pb->n_factors, pb->n_landmarks, ctx->max_factors);
    if (!pb->states || !pb->extr || !pb->sqrt_inv_cov) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null BA arrays");
    if (pb->n_factors > 0 && (!pb->tgt || !pb->ref || !pb->lm || !pb->tangent || !pb->z_ref || !pb->inv_depth))
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null BA factor arrays");
    // shapes are checked on the host so a kernel can never index out of bounds
    for (int k = 0; k < pb->n_factors; ++k) {
        if (pb->tgt[k] < 0 || pb->tgt[k] >= pb->n_frames || pb->ref[k] < 0 || pb->ref[k] >= pb->n_frames || pb->lm[k] < 0 ||
            pb->lm[k] >= pb->n_landmarks)
            return rdvio_fail(ctx, RDVIO_ERR_INVALID, "factor %d indexes out of range", k);
    }
    hipStream_t st = ctx->stream;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_states, pb->states, (size_t)pb->n_frames * 16 * sizeof(double), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_extr, pb->extr, 14 * sizeof(double), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_extr + 14, pb->sqrt_inv_cov, 4 * sizeof(double), hipMemcpyHostToDevice, st));
    if (pb->n_landmarks > 0) {
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_zref, pb->z_ref, (size_t)pb->n_landmarks * 3 * sizeof(double), hipMemcpyHostToDevice, st));
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_invd, pb->inv_depth, (size_t)pb->n_landmarks * sizeof(double), hipMemcpyHostToDevice, st));
    }
    if (pb->n_factors > 0) {
        const size_t nb = (size_t)pb->n_factors * sizeof(int32_t);
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_idx, pb->tgt, nb, hipMemcpyHostToDevice, st));
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_idx + ctx->max_factors, pb->ref, nb, hipMemcpyHostToDevice, st));
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_idx + 2 * (size_t)ctx->max_factors, pb->lm, nb, hipMemcpyHostToDevice, st));
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_tangent, pb->tangent, (size_t)pb->n_factors * 9 * sizeof(double), hipMemcpyHostToDevice, st));
    }
    return RDVIO_OK;
}

int rdvio_hip_reprojection_eval(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, double *r, double *Jt, double *Jr,
                                double *Jd) {
    if (!ctx || !r) return RDVIO_ERR_INVALID;
    if ((Jt || Jr || Jd) && !(Jt && Jr && Jd)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "pass all of Jt/Jr/Jd or none");
    if (int rc = upload_ba_problem(ctx, pb)) return rc;
    const int nf = pb->n_factors;
    if (nf == 0) return RDVIO_OK;
    if (int rc = rdvio_launch_reprojection(ctx, nf, Jt != nullptr)) return rc;
    hipStream_t st = ctx->stream;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(r, ctx->ba_r, (size_t)nf * 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Jt) {
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(Jt, ctx->ba_Jt, (size_t)nf * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(Jr, ctx->ba_Jr, (size_t)nf * 12 * sizeof(double), hipMemcpyDeviceToHost, st));
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(Jd, ctx->ba_Jd, (size_t)nf * 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    return RDVIO_OK;
}

int rdvio_hip_rotation_prior_eval(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, double *r, double *J) {
    if (!ctx || !r) return RDVIO_ERR_INVALID;
    if (!pb || pb->n_frames <= 0 || pb->n_rot < 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad BA problem");
    if (pb->n_frames > ctx->max_window + 2) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d frames exceed window capacity", pb->n_frames);
    if (pb->n_rot > ctx->max_factors) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d rotation priors exceed capacity %d", pb->n_rot, ctx->max_factors);
    if (!pb->states || !pb->extr || !pb->sqrt_inv_cov) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null BA arrays");
    const int n = pb->n_rot;
    if (n == 0) return RDVIO_OK;
    if (!pb->rot_tgt || !pb->rot_ref || !pb->rot_zref || !pb->rot_tangent) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null rotation-prior arrays");
    for (int k = 0; k < n; ++k)
        if (pb->rot_tgt[k] < 0 || pb->rot_tgt[k] >= pb->n_frames || pb->rot_ref[k] < 0 || pb->rot_ref[k] >= pb->n_frames)
            return rdvio_fail(ctx, RDVIO_ERR_INVALID, "rotation prior %d indexes out of range", k);
    hipStream_t st = ctx->stream;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_states, pb->states, (size_t)pb->n_frames * 16 * sizeof(double), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_extr, pb->extr, 14 * sizeof(double), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_extr + 14, pb->sqrt_inv_cov, 4 * sizeof(double), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_idx, pb->rot_tgt, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_idx + ctx->max_factors, pb->rot_ref, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_zref, pb->rot_zref, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ba_tangent, pb->rot_tangent, (size_t)n * 9 * sizeof(double), hipMemcpyHostToDevice, st));
    if (int rc = rdvio_launch_rotation_prior(ctx, n, J != nullptr)) return rc;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(r, ctx->ba_r, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (J) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(J, ctx->ba_Jt, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    return RDVIO_OK;
}

}  // extern "C"
