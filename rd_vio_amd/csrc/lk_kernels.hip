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
constexpr int NLV = RDVIO_MAX_LEVELS;
// search-image tile of one level in LDS: 48 x 40 bytes around the expected position (the 22 x 22 tap footprint has 13 px of
// slack on either side, 9 above and below), padded row stride of 13 dwords
constexpr int TILE_W = 48;
constexpr int TILE_H = 40;
constexpr int TILE_STRIDE = 52;
constexpr int TILE_BYTES = TILE_H * TILE_STRIDE;
constexpr int TILE_DW = TILE_W / 4;
constexpr int TILE_LOADS = (TILE_H * TILE_DW + 63) / 64;   // dwords per lane

// 64-lane integer sums on the DPP network (row shifts and row broadcasts inside the VALU: a step costs four instructions, where a
// ds_bpermute shuffle of a 64-bit value is two LDS-crossbar round trips).  Integer addition: the order cannot change the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_add_i64(long long v) {
    const int lo = (int)(unsigned long long)v, hi = (int)((unsigned long long)v >> 32);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + (long long)(((unsigned long long)(unsigned)phi << 32) | (unsigned long long)(unsigned)plo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    v = dpp_add_i64<0xb1, 0xf>(v);    // quad_perm [1 0 3 2]
    v = dpp_add_i64<0x4e, 0xf>(v);    // quad_perm [2 3 0 1]
    v = dpp_add_i64<0x114, 0xf>(v);   // row_shr 4
    v = dpp_add_i64<0x118, 0xf>(v);   // row_shr 8: lane 15 of every row holds the row's sum
    v = dpp_add_i64<0x142, 0xa>(v);   // row_bcast 15 into rows 1 and 3
    v = dpp_add_i64<0x143, 0xc>(v);   // row_bcast 31 into rows 2 and 3: lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane((int)(unsigned long long)v, 63), hi = __builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), 63);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

struct TileOrg {
    int x, y;   // level coordinates of the tile's first byte
};
// the tile that holds the tap footprint of a window whose top-left tap is (inx, iny), kept inside the level's padded arena
__device__ __forceinline__ TileOrg tile_origin(int inx, int iny, int w, int h, int B) {
    int tx = (inx - 13) & ~3, ty = iny - 9;
    const int xhi = (w + B - TILE_W) & ~3, yhi = h + B - TILE_H;
    tx = tx < -B ? -B : (tx > xhi ? xhi : tx);
    ty = ty < -B ? -B : (ty > yhi ? yhi : ty);
    return TileOrg{tx, ty};
}
__device__ __forceinline__ bool tile_holds(const TileOrg &o, int inx, int iny) {
    return inx >= o.x && inx + WIN + 1 <= o.x + TILE_W && iny >= o.y && iny + WIN + 1 <= o.y + TILE_H;
}
// global -> registers (issue) and registers -> LDS (commit) are separate so that the loads of several tiles and of the templates
// are in flight together
__device__ __forceinline__ void tile_issue(const uint8_t *__restrict__ J, int s, const TileOrg &o, uint32_t (&v)[TILE_LOADS]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < TILE_LOADS; ++q) {
        const int i = lane + 64 * q;
        const int ty = i / TILE_DW, tx = i - ty * TILE_DW;
        v[q] = (i < TILE_H * TILE_DW) ? *reinterpret_cast<const uint32_t *>(J + (ptrdiff_t)(o.y + ty) * s + o.x + 4 * tx) : 0u;
    }
}
__device__ __forceinline__ void tile_commit(uint8_t *tile, const uint32_t (&v)[TILE_LOADS]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < TILE_LOADS; ++q) {
        const int i = lane + 64 * q;
        const int ty = i / TILE_DW, tx = i - ty * TILE_DW;
        if (i < TILE_H * TILE_DW) *reinterpret_cast<uint32_t *>(tile + ty * TILE_STRIDE + tx * 4) = v[q];
    }
}

// where one level's window sits in the template image, and the raw taps of this lane's seven pixels (two rows of eight)
struct LkTaps {
    int ipx, ipy, iw00, iw01, iw10, iw11;
    bool inside;
    int t[2][8];
    short2 g[2][8];
};
// the position of level lv's search window before the first iteration, if the levels above change nothing
__device__ __forceinline__ void expected_window(float gx, float gy, int lv, int &inx, int &iny) {
    const float scale = (float)(1.0 / (double)(1 << lv)), half = (WIN - 1) * 0.5f;
    inx = (int)floorf(gx * scale - half);
    iny = (int)floorf(gy * scale - half);
}

__device__ __forceinline__ void taps_issue(const rdvio_pyr_layout &L, const uint8_t *__restrict__ imgI, const int16_t *__restrict__ derI, float prev_x,
                                           float prev_y, int lv, LkTaps &T) {
    const int lane = threadIdx.x & 63;
    const int row = lane / 3, seg = lane - row * 3, x0 = seg * 7;
    const int w = L.w[lv], h = L.h[lv], s = L.stride[lv], B = L.border;
    const float half = (WIN - 1) * 0.5f;
    const float scale = (float)(1.0 / (double)(1 << lv));
    const float px = prev_x * scale - half, py = prev_y * scale - half;
    T.ipx = (int)floorf(px);
    T.ipy = (int)floorf(py);
    T.inside = !(T.ipx < -WIN || T.ipx >= w || T.ipy < -WIN || T.ipy >= h);
    const float a = px - (float)T.ipx, b = py - (float)T.ipy;
    T.iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    T.iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
    T.iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
    T.iw11 = (1 << W_BITS) - T.iw00 - T.iw01 - T.iw10;
    if (T.inside && lane < 63) {
        const uint8_t *I = imgI + L.img_off[lv] + (size_t)B * s + B;
        const short2 *dI = reinterpret_cast<const short2 *>(derI + L.deriv_off[lv]) + (size_t)B * s + B;
        const uint8_t *r0 = I + (ptrdiff_t)(T.ipy + row) * s + T.ipx + x0;
        const short2 *d0 = dI + (ptrdiff_t)(T.ipy + row) * s + T.ipx + x0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            T.t[0][k] = r0[k];
            T.t[1][k] = r0[s + k];
            T.g[0][k] = d0[k];
            T.g[1][k] = d0[s + k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            T.t[0][k] = 0;
            T.t[1][k] = 0;
            T.g[0][k] = short2{0, 0};
            T.g[1][k] = short2{0, 0};
        }
    }
}

// One cv::calcOpticalFlowPyrLK for one point (all levels), executed by one wavefront.  Returns the status byte; next point through
// nx_io / ny_io (in: initial guess at level 0 scale).
//   tiles: NLV LDS tiles, one per level.  PRESTAGED: the caller has issued the tile loads (vt, origins org) around the positions
//   expected from the initial guess; otherwise they are issued here.  Either way the template taps of ALL levels are requested
//   before the first tile is committed: the whole flow waits for global memory once, then works from registers and LDS (a window
//   that leaves its tile -- the levels above moved it by more than the slack -- stages a new one).
//   COMMIT: 0 the tiles are in LDS already; 1 commit vt to tiles; 2 also commit v2 to the NLV tiles behind them (the tiles of the
//   flow that follows: requested together with this flow's, written once this flow's template taps have been requested).
template <bool PRESTAGED, int COMMIT = 1>
__device__ uint8_t lk_flow_one(const rdvio_pyr_layout &L, const uint8_t *__restrict__ imgI, const int16_t *__restrict__ derI,
                               const uint8_t *__restrict__ imgJ, float prev_x, float prev_y, float &nx_io, float &ny_io, int max_iter,
                               double eps_sq, uint8_t *tiles, TileOrg (&org)[NLV], uint32_t (&vt)[NLV][TILE_LOADS],
                               uint32_t (&v2)[NLV][TILE_LOADS]) {
    const int lane = threadIdx.x & 63;
    const int row = lane / 3, seg = lane - row * 3;  // lane 63 -> row 21: idle
    const bool active = lane < 63;
    const int x0 = seg * 7;
    const float half = (WIN - 1) * 0.5f;
    const int max_level = L.levels - 1;
    uint8_t status = 1;
    float out_x = nx_io, out_y = ny_io;

    if (!PRESTAGED) {
#pragma unroll
        for (int lv = 0; lv < NLV; ++lv)
            if (lv <= max_level) {
                int ex, ey;
                expected_window(nx_io, ny_io, lv, ex, ey);
                // (a guess outside the level is rejected before anything is read from the tile: any origin inside the arena will do)
                ex = ex < -WIN ? -WIN : (ex >= L.w[lv] ? L.w[lv] - 1 : ex);
                ey = ey < -WIN ? -WIN : (ey >= L.h[lv] ? L.h[lv] - 1 : ey);
                org[lv] = tile_origin(ex, ey, L.w[lv], L.h[lv], L.border);
                tile_issue(imgJ + L.img_off[lv] + (size_t)L.border * L.stride[lv] + L.border, L.stride[lv], org[lv], vt[lv]);
            }
    }
    LkTaps taps[NLV];
#pragma unroll
    for (int lv = 0; lv < NLV; ++lv)
        if (lv <= max_level) taps_issue(L, imgI, derI, prev_x, prev_y, lv, taps[lv]);
    if (COMMIT) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int lv = 0; lv < NLV; ++lv)
            if (lv <= max_level) {
                tile_commit(tiles + lv * TILE_BYTES, vt[lv]);
                if (COMMIT == 2) tile_commit(tiles + (NLV + lv) * TILE_BYTES, v2[lv]);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

#pragma unroll
    for (int lq = 0; lq < NLV; ++lq) {
        const int lv = NLV - 1 - lq;
        if (lv > max_level) continue;
        const int w = L.w[lv], h = L.h[lv], s = L.stride[lv], B = L.border;
        const uint8_t *J = imgJ + L.img_off[lv] + (size_t)B * s + B;
        uint8_t *tile = tiles + lv * TILE_BYTES;
        const LkTaps &T = taps[lv];

        const float scale = (float)(1.0 / (double)(1 << lv));
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

        if (!T.inside) {
            if (lv == 0) status = 0;
            continue;
        }
        int iw00 = T.iw00, iw01 = T.iw01, iw10 = T.iw10, iw11 = T.iw11;

        // ---- template: 7 pixels per lane, taps shared along the row (8 columns x 2 rows) ----
        short Iv[7], Ix[7], Iy[7];
        int pA11 = 0, pA12 = 0, pA22 = 0;
        if (active) {
            int t0 = T.t[0][0], t1 = T.t[1][0];
            short2 g0 = T.g[0][0], g1 = T.g[1][0];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                int u0 = T.t[0][k + 1], u1 = T.t[1][k + 1];
                short2 q0 = T.g[0][k + 1], q1 = T.g[1][k + 1];
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
        TileOrg o = org[lv];
        for (int j = 0; j < max_iter; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -WIN || inx >= w || iny < -WIN || iny >= h) {
                if (lv == 0) status = 0;
                break;
            }
            // stage the search tile again if the 22x22 tap footprint is not inside it
            if (!tile_holds(o, inx, iny)) {
                o = tile_origin(inx, iny, w, h, B);
                uint32_t v[TILE_LOADS];
                tile_issue(J, s, o, v);
                __builtin_amdgcn_wave_barrier();
                tile_commit(tile, v);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const float a = nx - (float)inx;
            const float b = ny - (float)iny;
            iw00 = __float2int_rn((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
            iw01 = __float2int_rn(a * (1.f - b) * (float)(1 << W_BITS));
            iw10 = __float2int_rn((1.f - a) * b * (float)(1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int pb1 = 0, pb2 = 0;
            if (active) {
                const uint8_t *r0 = tile + (iny - o.y + row) * TILE_STRIDE + (inx - o.x) + x0;
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
    // [0, NLV): the next image around the guess (forward flow); [NLV, 2 NLV): the current image around the point (backward flow)
    __shared__ __attribute__((aligned(16))) uint8_t tiles[2 * NLV * TILE_BYTES];
    const int i = blockIdx.x;
    if (i >= n) return;
    const int cols = L.w[0], rows = L.h[0];
    // to_opencv(): double -> float (opencv_image.cpp:8-16)
    const float cx = (float)curr[2 * i], cy = (float)curr[2 * i + 1];
    float nx = has_guess ? (float)next[2 * i] : cx;
    float ny = has_guess ? (float)next[2 * i + 1] : cy;
    // every tile either flow can be expected to need is requested before anything is waited for: the backward flow starts from the
    // point itself (opencv_image.cpp:128-131), the forward flow from the guess
    TileOrg of[NLV], ob[NLV];
    uint32_t vf[NLV][TILE_LOADS], vb[NLV][TILE_LOADS];
#pragma unroll
    for (int lv = 0; lv < NLV; ++lv)
        if (lv < L.levels) {
            const int w = L.w[lv], h = L.h[lv], s = L.stride[lv], B = L.border;
            int ex, ey;
            expected_window(nx, ny, lv, ex, ey);
            ex = ex < -WIN ? -WIN : (ex >= w ? w - 1 : ex);
            ey = ey < -WIN ? -WIN : (ey >= h ? h - 1 : ey);
            of[lv] = tile_origin(ex, ey, w, h, B);
            tile_issue(img_n + L.img_off[lv] + (size_t)B * s + B, s, of[lv], vf[lv]);
            expected_window(cx, cy, lv, ex, ey);
            ex = ex < -WIN ? -WIN : (ex >= w ? w - 1 : ex);
            ey = ey < -WIN ? -WIN : (ey >= h ? h - 1 : ey);
            ob[lv] = tile_origin(ex, ey, w, h, B);
            tile_issue(img_c + L.img_off[lv] + (size_t)B * s + B, s, ob[lv], vb[lv]);
        }
    uint8_t st = lk_flow_one<true, 2>(L, img_c, der_c, img_n, cx, cy, nx, ny, 30, 1e-4, tiles, of, vf, vb);
    if (nx < 20.f || nx >= (float)(cols - 20) || ny < 20.f || ny >= (float)(rows - 20)) st = 0;
    if (st) {
        float dx = nx - cx, dy = ny - cy;
        double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
        if (nrm > (double)(rows / 4)) st = 0;
    }
    if (st) {  // wave-uniform: the backward result is only consulted for forward survivors (:128-134)
        float rx = cx, ry = cy;
        uint8_t rst = lk_flow_one<true, 0>(L, img_n, der_n, img_c, nx, ny, rx, ry, 30, 1e-4, tiles + NLV * TILE_BYTES, ob, vb, vb);
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
    __shared__ __attribute__((aligned(16))) uint8_t tiles[NLV * TILE_BYTES];
    const int i = blockIdx.x;
    if (i >= n) return;
    float nx = next[2 * i], ny = next[2 * i + 1];
    TileOrg org[NLV];
    uint32_t vt[NLV][TILE_LOADS];
    uint8_t st = lk_flow_one<false, 1>(L, img_p, der_p, img_n, prev[2 * i], prev[2 * i + 1], nx, ny, max_iter, eps_sq, tiles, org, vt, vt);
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
