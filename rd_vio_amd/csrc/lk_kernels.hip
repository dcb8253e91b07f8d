// Pyramidal Lucas-Kanade feature tracking on gfx950, one 64-lane wavefront per feature.
//
// Replaces OpenCvImage::track_keypoints (/root/reference/src/rdvio_extra/src/opencv_image.cpp:75-154):
// two cv::calcOpticalFlowPyrLK calls (forward, backward; win 21x21, maxLevel 3, <=30 iterations or
// |delta|^2 <= 0.01^2, OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4) with the 20-px border,
// rows/4 max-flow and 0.5-px forward-backward rejections in between -- fused into ONE launch.
//
// Mapping to the hardware:
//  * 441 window pixels = 21 rows x 3 segments of 7 pixels -> 63 lanes, 7 pixels per lane.  The
//    template (I, dI/dx, dI/dy as int16, 21 values per lane) lives in VGPRs for the whole level.
//  * the search image J is staged once per level into a 36x32-byte LDS tile around the current
//    estimate (dword-aligned, coalesced row segments) and re-staged only if the window walks out of it;
//    every Newton iteration then reads LDS only.
//  * the 2x2 normal matrix and the mismatch vector are EXACT int64 sums (per-lane int32 partials,
//    64-lane xor-shuffle reduction), converted to float once: order-independent, hence bit-identical
//    to the oracle; all float steps are single-rounded (-ffp-contract=off).
// Roofline: HBM/L2-bound on paper -- per feature.level.direction 22x22 B template + 22x22x4 B
// gradients + one 36x32 B tile (~2.9 KB); in practice latency-bound (serial levels
// and iterations), so the figure of merit is time per frame.
#include "ctx.hpp"

namespace {

constexpr int WIN = RDVIO_LK_WIN;       // 21
constexpr int W_BITS = 14;
constexpr int TILE_W = 36;              // bytes per LDS tile row (9 dwords)
constexpr int TILE_H = 32;
constexpr int TILE_STRIDE = 40;         // padded LDS row stride in bytes

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

struct LkLevelCtx {
    const uint8_t *I;      // interior origin of level (template image)
    const short2 *dI;      // interior origin of derivative level
    const uint8_t *J;      // interior origin of level (search image)
    int w, h, s;
};

// One cv::calcOpticalFlowPyrLK for one point (all levels), executed by one wavefront.
// Returns the status byte; next point through nx_io/ny_io (in: initial guess at level 0 scale).
__device__ uint8_t lk_flow_one(const rdvio_pyr_layout &L, const uint8_t *__restrict__ imgI,
                               const int16_t *__restrict__ derI, const uint8_t *__restrict__ imgJ, float prev_x,
                               float prev_y, float &nx_io, float &ny_io, int max_iter, double eps_sq,
                               uint8_t *tile /* LDS, TILE_H*TILE_STRIDE */) {
    const int lane = threadIdx.x & 63;
    const int row = lane / 3, seg = lane - row * 3;  // lane 63 -> row 21: idle
    const bool active = lane < 63;
    const int x0 = seg * 7;
    const float half = (WIN - 1) * 0.5f;
    const int max_level = L.levels - 1;
    uint8_t status = 1;
    float out_x = nx_io, out_y = ny_io;

    for (int lv = max_level; lv >= 0; --lv) {
        const int w = L.w[lv], h = L.h[lv], s = L.stride[lv], B = L.border;
        const uint8_t *I = imgI + L.img_off[lv] + (size_t)B * s + B;
        const uint8_t *J = imgJ + L.img_off[lv] + (size_t)B * s + B;
        const short2 *dI = reinterpret_cast<const short2 *>(derI + L.deriv_off[lv]) + (size_t)B * s + B;

        const float scale = (float)(1.0 / (double)(1 << lv));
        float px = prev_x * scale, py = prev_y * scale;
        float nx, ny;
        if (lv == max_level) {
            nx = out_x * scale;
            ny = out_y * scale;
        } else {
            nx = out_x * 2.f;
            ny = out_y * 2.f;
        }
        out_x = nx;
        out_y = ny;

        px -= half;
        py -= half;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -WIN || ipx >= w || ipy < -WIN || ipy >= h) {
            if (lv == 0) status = 0;
            continue;
        }
        float a = px - (float)ipx, b = py - (float)ipy;
        int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
        int iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
        int iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        // ---- template: 7 pixels per lane, taps shared along the row (8 columns x 2 rows) ----
        short Iv[7], Ix[7], Iy[7];
        int pA11 = 0, pA12 = 0, pA22 = 0;
        if (active) {
            const uint8_t *r0 = I + (ptrdiff_t)(ipy + row) * s + ipx + x0;
            const uint8_t *r1 = r0 + s;
            const short2 *d0 = dI + (ptrdiff_t)(ipy + row) * s + ipx + x0;
            const short2 *d1 = d0 + s;
            int t0 = r0[0], t1 = r1[0];
            short2 g0 = d0[0], g1 = d1[0];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                int u0 = r0[k + 1], u1 = r1[k + 1];
                short2 q0 = d0[k + 1], q1 = d1[k + 1];
                int ival = (t0 * iw00 + u0 * iw01 + t1 * iw10 + u1 * iw11 + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                int ixval = (g0.x * iw00 + q0.x * iw01 + g1.x * iw10 + q1.x * iw11 + (1 << (W_BITS - 1))) >> W_BITS;
                int iyval = (g0.y * iw00 + q0.y * iw01 + g1.y * iw10 + q1.y * iw11 + (1 << (W_BITS - 1))) >> W_BITS;
                Iv[k] = (short)ival;
                Ix[k] = (short)ixval;
                Iy[k] = (short)iyval;
                pA11 += ixval * ixval;
                pA12 += ixval * iyval;
                pA22 += iyval * iyval;
                t0 = u0; t1 = u1; g0 = q0; g1 = q1;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) { Iv[k] = 0; Ix[k] = 0; Iy[k] = 0; }
        }
        const long long iA11 = wave_sum_i64(pA11), iA12 = wave_sum_i64(pA12), iA22 = wave_sum_i64(pA22);
        const float FLT_SCALE = 1.f / (float)(1 << 20);
        const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig =
            (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if ((double)minEig < 1e-4 || D < 1.1920928955078125e-07f) {
            if (lv == 0) status = 0;
            continue;
        }
        D = 1.f / D;

        nx -= half;
        ny -= half;
        float pdx = 0.f, pdy = 0.f;
        int tox = INT_MIN, toy = 0;  // LDS tile origin (level coords); INT_MIN = nothing staged
        for (int j = 0; j < max_iter; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (lv == 0) status = 0;
                break;
            }
            // (re)stage the search tile if the 22x22 tap footprint is not inside it
            if (tox == INT_MIN || inx < tox || inx + WIN + 1 > tox + TILE_W || iny < toy || iny + WIN + 1 > toy + TILE_H) {
                tox = (inx - 4) & ~3;
                toy = iny - 5;
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < TILE_H * (TILE_W / 4); i += 64) {
                    int ty = i / (TILE_W / 4), tx = i - ty * (TILE_W / 4);
                    const uint32_t *src =
                        reinterpret_cast<const uint32_t *>(J + (ptrdiff_t)(toy + ty) * s + tox) + tx;
                    *reinterpret_cast<uint32_t *>(tile + ty * TILE_STRIDE + tx * 4) = *src;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            a = nx - (float)inx;
            b = ny - (float)iny;
            iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
            iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
            iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int pb1 = 0, pb2 = 0;
            if (active) {
                const uint8_t *r0 = tile + (iny - toy + row) * TILE_STRIDE + (inx - tox) + x0;
                const uint8_t *r1 = r0 + TILE_STRIDE;
                int t0 = r0[0], t1 = r1[0];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    int u0 = r0[k + 1], u1 = r1[k + 1];
                    int jv = (t0 * iw00 + u0 * iw01 + t1 * iw10 + u1 * iw11 + (1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                    int diff = jv - Iv[k];
                    pb1 += diff * Ix[k];
                    pb2 += diff * Iy[k];
                    t0 = u0; t1 = u1;
                }
            }
            const long long ib1 = wave_sum_i64(pb1), ib2 = wave_sum_i64(pb2);
            const float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx;
            ny += dy;
            out_x = nx + half;
            out_y = ny + half;
            if ((double)dx * (double)dx + (double)dy * (double)dy <= eps_sq) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                out_x -= dx * 0.5f;
                out_y -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }
    }
    nx_io = out_x;
    ny_io = out_y;
    return status;
}

// OpenCvImage::track_keypoints, fused: forward flow, rejections, backward flow, forward-backward check.
__global__ __launch_bounds__(64) void lk_track_kernel(rdvio_pyr_layout L, const uint8_t *__restrict__ img_c,
                                                      const int16_t *__restrict__ der_c,
                                                      const uint8_t *__restrict__ img_n,
                                                      const int16_t *__restrict__ der_n, int n,
                                                      const double *__restrict__ curr, double *__restrict__ next,
                                                      int has_guess, uint8_t *__restrict__ status_out) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE_H * TILE_STRIDE];
    const int i = blockIdx.x;
    if (i >= n) return;
    const int cols = L.w[0], rows = L.h[0];
    // to_opencv(): double -> float (opencv_image.cpp:8-16)
    const float cx = (float)curr[2 * i], cy = (float)curr[2 * i + 1];
    float nx = has_guess ? (float)next[2 * i] : cx;
    float ny = has_guess ? (float)next[2 * i + 1] : cy;
    uint8_t st = lk_flow_one(L, img_c, der_c, img_n, cx, cy, nx, ny, 30, 1e-4, tile);
    if (nx < 20.f || nx >= (float)(cols - 20) || ny < 20.f || ny >= (float)(rows - 20)) st = 0;
    if (st) {
        float dx = nx - cx, dy = ny - cy;
        double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        if (nrm > (double)(rows / 4)) st = 0;
    }
    if (st) {  // wave-uniform: the backward result is only consulted for forward survivors (:128-134)
        float rx = cx, ry = cy;
        uint8_t rst = lk_flow_one(L, img_n, der_n, img_c, nx, ny, rx, ry, 30, 1e-4, tile);
        float dx = cx - rx, dy = cy - ry;
        double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        if (!rst || nrm > 0.5) st = 0;
    }
    if ((threadIdx.x & 63) == 0) {
        status_out[i] = st;
        if (st) {
            next[2 * i] = (double)nx;
            next[2 * i + 1] = (double)ny;
        }
    }
}

// a single calcOpticalFlowPyrLK (unit-parity entry point)
__global__ __launch_bounds__(64) void lk_flow_kernel(rdvio_pyr_layout L, const uint8_t *__restrict__ img_p,
                                                     const int16_t *__restrict__ der_p,
                                                     const uint8_t *__restrict__ img_n, int n,
                                                     const float *__restrict__ prev, float *__restrict__ next,
                                                     uint8_t *__restrict__ status_out, int max_iter, double eps_sq) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE_H * TILE_STRIDE];
    const int i = blockIdx.x;
    if (i >= n) return;
    float nx = next[2 * i], ny = next[2 * i + 1];
    uint8_t st = lk_flow_one(L, img_p, der_p, img_n, prev[2 * i], prev[2 * i + 1], nx, ny, max_iter, eps_sq, tile);
    if ((threadIdx.x & 63) == 0) {
        status_out[i] = st;
        next[2 * i] = nx;
        next[2 * i + 1] = ny;
    }
}

}  // namespace

static int check_slots(rdvio_hip_ctx *ctx, int a, int b) {
    if (a < 0 || a >= RDVIO_NUM_SLOTS || b < 0 || b >= RDVIO_NUM_SLOTS)
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "image slot out of range");
    if (!ctx->slots[a].valid || !ctx->slots[b].valid)
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "image slot not preprocessed");
    if (ctx->slots[a].w != ctx->slots[b].w || ctx->slots[a].h != ctx->slots[b].h)
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "image slots differ in size");
    return RDVIO_OK;
}

int rdvio_launch_track(rdvio_hip_ctx *ctx, int slot_curr, int slot_next, int n, const double *curr_dev,
                       double *next_dev, int has_guess, uint8_t *status_dev) {
    if (int rc = check_slots(ctx, slot_curr, slot_next)) return rc;
    if (n <= 0) return RDVIO_OK;
    ImageSlot &C = ctx->slots[slot_curr], &N = ctx->slots[slot_next];
    hipLaunchKernelGGL(lk_track_kernel, dim3(n), dim3(64), 0, ctx->stream, C.L, C.pyr_img, C.pyr_deriv, N.pyr_img,
                       N.pyr_deriv, n, curr_dev, next_dev, has_guess, status_dev);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}

int rdvio_launch_lk_flow(rdvio_hip_ctx *ctx, int slot_prev, int slot_next, int n, const float *prev_dev,
                         float *next_dev, uint8_t *status_dev, int max_iter, double eps) {
    if (int rc = check_slots(ctx, slot_prev, slot_next)) return rc;
    if (n <= 0) return RDVIO_OK;
    if (max_iter < 0) max_iter = 0;
    if (max_iter > 100) max_iter = 100;
    if (eps < 0) eps = 0;
    if (eps > 10) eps = 10;
    ImageSlot &P = ctx->slots[slot_prev], &N = ctx->slots[slot_next];
    hipLaunchKernelGGL(lk_flow_kernel, dim3(n), dim3(64), 0, ctx->stream, P.L, P.pyr_img, P.pyr_deriv, N.pyr_img, n,
                       prev_dev, next_dev, status_dev, max_iter, eps * eps);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}
