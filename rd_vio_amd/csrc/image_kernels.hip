// Image side of the feature tracker on gfx950: CLAHE, 4-level pyramid + Scharr derivatives, Harris.
//
// Replaces the OpenCV calls of OpenCvImage::preprocess / detect_keypoints
// (/root/reference/src/rdvio_extra/src/opencv_image.cpp:156-161, :38-73, :179-188):
// cv::CLAHE::apply, cv::buildOpticalFlowPyramid(win 21x21, maxLevel 3, withDerivatives),
// cv::GFTTDetector (Harris, block 3, k 0.04).  All arithmetic is integer or a fixed sequence of
// single-rounded float/double operations (this file is compiled with -ffp-contract=off), so results
// are bit-identical to the CPU oracle regardless of how the work is split over lanes.
//
// Data layout: one padded arena per frame (rdvio_pyr_layout): u8 levels with a 32-px
// BORDER_REFLECT_101 frame and 64-byte-multiple row strides (coalesced 64-lane row segments),
// derivatives as interleaved int16 (dx,dy) so one dword load fetches both.
#include "ctx.hpp"

namespace {

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * (len - 1) - p;
    return p;
}
__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---------------------------------------------------------------------------------------------
// CLAHE pass 1: one workgroup per tile -> clipped histogram -> LUT (cv::CLAHE, 8-bit path).
// HBM-bound on paper (1 B read per pixel) but tiny: 64 tiles x 5640 px for EuRoC.
// ---------------------------------------------------------------------------------------------
// 1024 threads per tile (a tile is ~5.6 k pixels at EuRoC size: five or six per thread, all loads of a thread in flight together),
// one private histogram per wavefront (sixteen-fold less contention on the LDS atomics), wave-level scans instead of sixteen
// workgroup barriers.  Integer counts: the split over wavefronts cannot change them.
constexpr int CL_T = 1024;
constexpr int CL_PX = 8;   // pixels per thread and pass
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int n = __shfl_up(v, off);
        if (lane >= off) v += n;
    }
    return v;
}
__global__ __launch_bounds__(CL_T) void clahe_lut_kernel(const uint8_t *__restrict__ src, int w, int h, int stride,
                                                        int tiles_x, int tw, int th, int clip, float lut_scale,
                                                        uint8_t *__restrict__ lut) {
    __shared__ int hist[CL_T / 64][256];
    __shared__ int part[8];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
#pragma unroll
    for (int q = 0; q < 4; ++q) hist[wave][lane + 64 * q] = 0;
    __syncthreads();
    const int area = tw * th;
    for (int base = 0; base < area; base += CL_T * CL_PX) {
        int v[CL_PX];
#pragma unroll
        for (int q = 0; q < CL_PX; ++q) {
            const int i = base + t + CL_T * q;
            v[q] = -1;
            if (i < area) {
                const int y = i / tw, x = i - y * tw;
                const int sy = reflect101(ty * th + y, h), sx = reflect101(tx * tw + x, w);
                v[q] = src[(size_t)sy * stride + sx];
            }
        }
#pragma unroll
        for (int q = 0; q < CL_PX; ++q)
            if (v[q] >= 0) atomicAdd(&hist[wave][v[q]], 1);
    }
    __syncthreads();
    const bool bin = t < 256;   // one bin per thread of the first four wavefronts (the others only keep the barriers company)
    int v = 0;
    if (bin) {
#pragma unroll
        for (int q = 0; q < CL_T / 64; ++q) v += hist[q][t];
    }
    if (clip > 0 && bin) {
        int excess = v > clip ? v - clip : 0;
        if (v > clip) v = clip;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) excess += __shfl_xor(excess, off);
        if (lane == 0) part[wave] = excess;
    }
    __syncthreads();
    if (clip > 0 && bin) {
        const int clipped = part[0] + part[1] + part[2] + part[3];
        int batch = clipped / 256;
        int residual = clipped - batch * 256;
        v += batch;
        if (residual != 0) {
            int step = 256 / residual;
            if (step < 1) step = 1;
            if (t % step == 0 && t / step < residual) v++;
        }
    }
    // inclusive prefix sum over the 256 bins
    const int inc = wave_incl_scan_i32(v);
    if (bin && lane == 63) part[4 + wave] = inc;
    __syncthreads();
    if (!bin) return;
    int run = inc;
    for (int q = 0; q < wave; ++q) run += part[4 + q];
    float f = (float)run * lut_scale;
    lut[tile * 256 + t] = sat_u8(__float2int_rn(f));
}

// CLAHE pass 2 for one pixel: bilinear blend of the four neighbouring tile LUTs.
__device__ __forceinline__ uint8_t clahe_apply(const uint8_t *__restrict__ src, int stride, int x, int y,
                                               const uint8_t *__restrict__ lut, int tiles_x, int tiles_y, float inv_tw,
                                               float inv_th) {
    float tyf = (float)y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0);
    ty2 = min(ty2, tiles_y - 1);
    float txf = (float)x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    float xa = txf - (float)tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0);
    tx2 = min(tx2, tiles_x - 1);
    int v = src[(size_t)y * stride + x];
    float l11 = lut[(ty1 * tiles_x + tx1) * 256 + v];
    float l12 = lut[(ty1 * tiles_x + tx2) * 256 + v];
    float l21 = lut[(ty2 * tiles_x + tx1) * 256 + v];
    float l22 = lut[(ty2 * tiles_x + tx2) * 256 + v];
    float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    return sat_u8(__float2int_rn(res));
}

// ---------------------------------------------------------------------------------------------
// One pyramid level per launch: level value (CLAHE for level 0, pyrDown of level lv-1 otherwise)
// for every pixel of the PADDED domain (interior + reflect-101 border), staged through an LDS tile
// with a 1-px halo so the Scharr derivative of interior pixels comes from LDS.
// Algorithmic traffic: level 0: 1 B read + 1 B write + 4 B deriv write per pixel;
// level l>0: 1 B write + 4 B deriv write (+ L2-resident re-reads of level l-1).
// ---------------------------------------------------------------------------------------------
#define PT_W 32
#define PT_H 8
// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md), each with its own L2:
// with the plain blockIdx -> tile map, horizontally adjacent 32-px tiles -- which share every 128-B line of an image row -- run
// on different XCDs and each XCD fetches the line for itself (round-2 PMC: the level-0 pyramid kernel fetched 1.21 MB for a
// 0.36 MB image, the Harris kernel 1.11 MB for 0.43 MB).  Here the workgroups of one XCD take a contiguous run of tiles
// (row-major: a band of tile rows), so a line is fetched by one L2.  A pure relabelling: results are unchanged.
__device__ __forceinline__ void xcd_tile(int &bx, int &by) {
    const int gx = gridDim.x, nb = gx * gridDim.y, b = blockIdx.y * gx + blockIdx.x;
    const int per = nb / 8;
    const int t = (b < per * 8) ? (b % 8) * per + b / 8 : b;   // (the last nb % 8 blocks keep their tiles)
    by = t / gx;
    bx = t - by * gx;
}
template <bool LEVEL0>
__global__ __launch_bounds__(256) void pyr_level_kernel(rdvio_pyr_layout L, int lv, uint8_t *__restrict__ pyr_img,
                                                       int16_t *__restrict__ pyr_deriv,
                                                       const uint8_t *__restrict__ gray, int gray_stride,
                                                       const uint8_t *__restrict__ lut, int tiles_x, int tiles_y,
                                                       float inv_tw, float inv_th) {
    __shared__ int tile[(PT_H + 2)][(PT_W + 2) + 1];
    const int B = L.border, w = L.w[lv], h = L.h[lv], s = L.stride[lv];
    int tbx, tby;
    xcd_tile(tbx, tby);
    const int bx0 = tbx * PT_W - B, by0 = tby * PT_H - B;  // image coords of the tile origin
    const uint8_t *prev = nullptr;
    int ps = 0;
    if (!LEVEL0) {
        ps = L.stride[lv - 1];
        prev = pyr_img + L.img_off[lv - 1] + (size_t)B * ps + B;  // interior origin of level lv-1 (border is valid)
    }
    for (int i = threadIdx.x; i < (PT_H + 2) * (PT_W + 2); i += 256) {
        int ly = i / (PT_W + 2), lx = i - ly * (PT_W + 2);
        int x = bx0 + lx - 1, y = by0 + ly - 1;
        int val = 0;
        if (x >= -B - 1 && x <= w + B && y >= -B - 1 && y <= h + B) {
            int rx = reflect101(x, w), ry = reflect101(y, h);
            if (LEVEL0) {
                val = clahe_apply(gray, gray_stride, rx, ry, lut, tiles_x, tiles_y, inv_tw, inv_th);
            } else {
                // cv::pyrDown: separable [1 4 6 4 1], (sum + 128) >> 8; taps 2r-2..2r+2 fall inside level lv-1's border
                int acc = 0;
#pragma unroll
                for (int j = -2; j <= 2; ++j) {
                    const uint8_t *row = prev + (ptrdiff_t)(2 * ry + j) * ps + 2 * rx;
                    int r = row[-2] + row[2] + 4 * (row[-1] + row[1]) + 6 * row[0];
                    int kj = (j == 0) ? 6 : ((j == -1 || j == 1) ? 4 : 1);
                    acc += kj * r;
                }
                val = (acc + 128) >> 8;
            }
        }
        tile[ly][lx] = val;
    }
    __syncthreads();
    const int lx = threadIdx.x % PT_W, ly = threadIdx.x / PT_W;
    const int x = bx0 + lx, y = by0 + ly;
    if (x >= w + B || y >= h + B) return;
    uint8_t *img = pyr_img + L.img_off[lv];
    img[(size_t)(y + B) * s + (x + B)] = (uint8_t)tile[ly + 1][lx + 1];
    if (x >= 0 && x < w && y >= 0 && y < h) {
        // calcScharrDeriv: dx = [3 10 3]^T x [-1 0 1], dy = [-1 0 1]^T x [3 10 3]
        int p00 = tile[ly][lx], p01 = tile[ly][lx + 1], p02 = tile[ly][lx + 2];
        int p10 = tile[ly + 1][lx], p12 = tile[ly + 1][lx + 2];
        int p20 = tile[ly + 2][lx], p21 = tile[ly + 2][lx + 1], p22 = tile[ly + 2][lx + 2];
        int dx = ((p02 + p22) * 3 + p12 * 10) - ((p00 + p20) * 3 + p10 * 10);
        int dy = ((p20 - p00) + (p22 - p02)) * 3 + (p21 - p01) * 10;
        short2 d;
        d.x = (short)dx;
        d.y = (short)dy;
        short2 *der = reinterpret_cast<short2 *>(pyr_deriv + L.deriv_off[lv]);
        der[(size_t)(y + B) * s + (x + B)] = d;
    }
}

// ---------------------------------------------------------------------------------------------
// Harris response (cornerHarris block 3, Sobel 3, k) -- exact-integer formulation ("Image-side arithmetic").
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}

// One workgroup = 64 x 16 output pixels, four per thread.
//   1. the u8 tile (+2 px all round, +4 on the left so that rows start on a dword) goes to LDS with coalesced dword loads -- one
//      load per four pixels (the one-pixel-per-thread version issued nine byte loads per gradient);
//   2. Sobel gradients of the 66 x 18 positions the 3 x 3 box sums need, from LDS, four per thread with dword reads, stored as one
//      packed (gx, gy) int16 pair per position (|g| <= 1020);
//   3. per thread the three column sums of products are formed once per column and shared by its four outputs.
// The box filter's border is REFLECT_101 on the gradient PRODUCTS: the gradient "at" x = -1 is the gradient at x = 1 (not the
// Sobel of the reflected image there, whose x-derivative has the other sign) -- positions on the outer ring read the reflected
// position's taps.  Exact integer sums, one double expression per pixel: bit-identical to the one-pixel-per-thread kernel.
#define HT_W 64
#define HT_H 16
#define HT_IMG_STRIDE 76   // bytes per LDS image row: 72 used (4 left, 64, 4 right), 19 dwords
#define HT_G_STRIDE 67     // words per LDS gradient row: 66 used
__device__ __forceinline__ int harris_sobel_packed(const uint8_t *simg, int lrow, int lcol) {
    // taps around LDS image position (lrow, lcol): rows lrow-1..lrow+1, bytes lcol-1..lcol+1
    const uint8_t *r0 = simg + (lrow - 1) * HT_IMG_STRIDE + lcol, *r1 = r0 + HT_IMG_STRIDE, *r2 = r1 + HT_IMG_STRIDE;
    const int gx = (r0[1] + 2 * r1[1] + r2[1]) - (r0[-1] + 2 * r1[-1] + r2[-1]);
    const int gy = (r2[-1] + 2 * r2[0] + r2[1]) - (r0[-1] + 2 * r0[0] + r0[1]);
    return (gx & 0xffff) | (gy << 16);
}
__global__ __launch_bounds__(256) void harris_kernel(rdvio_pyr_layout L, const uint8_t *__restrict__ pyr_img, double k,
                                                    float *__restrict__ resp, uint32_t *__restrict__ max_out) {
    __shared__ __attribute__((aligned(16))) uint8_t simg[(HT_H + 4) * HT_IMG_STRIDE];
    __shared__ int sg[(HT_H + 2) * HT_G_STRIDE];
    __shared__ uint32_t smax[4];
    const int B = L.border, w = L.w[0], h = L.h[0], s = L.stride[0];
    const uint8_t *img = pyr_img + L.img_off[0] + (size_t)B * s + B;
    int tbx, tby;
    xcd_tile(tbx, tby);
    const int bx0 = tbx * HT_W, by0 = tby * HT_H;
    const int t = threadIdx.x;
    // ---- 1. image rows by0-2 .. by0+17, bytes bx0-4 .. bx0+67 (inside the arena's reflect-101 frame or, beyond x = w + 31, in
    // row padding that only feeds masked outputs)
    for (int i = t; i < (HT_H + 4) * 18; i += 256) {
        const int r = i / 18, c = i - 18 * r;
        int y = by0 - 2 + r;
        y = y > h + B - 1 ? h + B - 1 : y;   // (rows below the frame feed masked outputs only)
        *reinterpret_cast<uint32_t *>(simg + r * HT_IMG_STRIDE + 4 * c) = *reinterpret_cast<const uint32_t *>(img + (ptrdiff_t)y * s + bx0 - 4 + 4 * c);
    }
    __syncthreads();
    // ---- 2. gradients at x in [bx0-1, bx0+64], y in [by0-1, by0+16]; word (row gy, col gx) of sg is position (by0-1+gy, bx0-1+gx)
    const int tx = t & 15, ty = t >> 4;
    for (int grow = ty; grow < HT_H + 2; grow += 16) {
        const int y = by0 - 1 + grow;
        const int ry = (y < 0) ? -y : (y >= h ? 2 * (h - 1) - y : y);   // y = -1 -> 1, y = h -> h - 2 (deeper rows feed masked outputs)
        const int lrow = ry - (by0 - 2);
        const int x0 = bx0 + 4 * tx;
        int *dst = sg + grow * HT_G_STRIDE + 1 + 4 * tx;
        if (lrow >= 1 && lrow <= HT_H + 2) {
            if (x0 + 3 < w) {
                // dwords tx, tx+1, tx+2 of three rows: bytes x0-4 .. x0+7, of which x0-1 .. x0+4 are used
                const uint32_t *q0 = reinterpret_cast<const uint32_t *>(simg + (lrow - 1) * HT_IMG_STRIDE) + tx;
                const uint32_t *q1 = reinterpret_cast<const uint32_t *>(simg + lrow * HT_IMG_STRIDE) + tx;
                const uint32_t *q2 = reinterpret_cast<const uint32_t *>(simg + (lrow + 1) * HT_IMG_STRIDE) + tx;
                int p0[6], p1[6], p2[6];
                {
                    const uint32_t a0 = q0[0], b0 = q0[1], c0 = q0[2], a1 = q1[0], b1 = q1[1], c1 = q1[2], a2 = q2[0], b2 = q2[1], c2 = q2[2];
                    p0[0] = a0 >> 24; p1[0] = a1 >> 24; p2[0] = a2 >> 24;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p0[1 + j] = (b0 >> (8 * j)) & 255;
                        p1[1 + j] = (b1 >> (8 * j)) & 255;
                        p2[1 + j] = (b2 >> (8 * j)) & 255;
                    }
                    p0[5] = c0 & 255; p1[5] = c1 & 255; p2[5] = c2 & 255;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = (p0[j + 2] + 2 * p1[j + 2] + p2[j + 2]) - (p0[j] + 2 * p1[j] + p2[j]);
                    const int gy = (p2[j] + 2 * p2[j + 1] + p2[j + 2]) - (p0[j] + 2 * p0[j + 1] + p0[j + 2]);
                    dst[j] = (gx & 0xffff) | (gy << 16);
                }
            } else {
                // the image's right edge runs through this group: x = w reads the taps of w - 2, beyond it nothing is needed
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int x = x0 + j;
                    const int rx = x >= w ? 2 * (w - 1) - x : x;
                    dst[j] = (x <= w && rx >= bx0 - 3) ? harris_sobel_packed(simg, lrow, rx - (bx0 - 4)) : 0;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = 0;
        }
    }
    if (t < 2 * (HT_H + 2)) {   // the two outer columns: x = bx0 - 1 and x = bx0 + 64
        const int grow = t >> 1, right = t & 1;
        const int y = by0 - 1 + grow, x = right ? bx0 + HT_W : bx0 - 1;
        const int ry = (y < 0) ? -y : (y >= h ? 2 * (h - 1) - y : y);
        const int rx = (x < 0) ? -x : (x >= w ? 2 * (w - 1) - x : x);
        const int lrow = ry - (by0 - 2), lcol = rx - (bx0 - 4);
        const bool ok = x <= w && lrow >= 1 && lrow <= HT_H + 2 && lcol >= 1 && lcol <= 70;
        sg[grow * HT_G_STRIDE + (right ? HT_W + 1 : 0)] = ok ? harris_sobel_packed(simg, lrow, lcol) : 0;
    }
    __syncthreads();
    // ---- 3. four outputs per thread: pixels (by0 + ty, bx0 + 4 tx + j)
    const int x0 = bx0 + 4 * tx, y = by0 + ty;
    int cxx[6], cxy[6], cyy[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        cxx[c] = 0; cxy[c] = 0; cyy[c] = 0;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int *row = sg + (ty + r) * HT_G_STRIDE + 4 * tx;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int v = row[c];
            const int gx = (int)(short)(v & 0xffff), gy = v >> 16;
            cxx[c] += gx * gx;
            cxy[c] += gx * gy;
            cyy[c] += gy * gy;
        }
    }
    float rr[4];
    uint32_t o = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int sxx = cxx[j] + cxx[j + 1] + cxx[j + 2], sxy = cxy[j] + cxy[j + 1] + cxy[j + 2], syy = cyy[j] + cyy[j + 1] + cyy[j + 2];
        const double sc = 1.0 / (4.0 * 3.0 * 255.0);
        const double s2 = sc * sc;
        double a = s2 * (double)sxx, b = s2 * (double)sxy, c = s2 * (double)syy;
        rr[j] = (float)(a * c - b * b - k * (a + c) * (a + c));
        if (x0 + j < w && y < h) o = max(o, float_to_ordered(rr[j]));
    }
    if (y < h) {
        if (x0 + 3 < w && (w & 3) == 0) {
            *reinterpret_cast<float4 *>(resp + (size_t)y * w + x0) = float4{rr[0], rr[1], rr[2], rr[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < w) resp[(size_t)y * w + x0 + j] = rr[j];
        }
    }
    // block max -> one atomic per block
    for (int off = 32; off > 0; off >>= 1) o = max(o, (uint32_t)__shfl_xor((int)o, off));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = o;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(max_out, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
}

// goodFeaturesToTrack: threshold at quality*max (THRESH_TOZERO), 3x3 local maximum on the thresholded map,
// rows/cols 1..n-2 only.  Survivors are appended unordered; the host sorts them by (value desc, index desc),
// which is cv::greaterThanPtr's order, so the candidate ORDER is deterministic.
__global__ __launch_bounds__(256) void harris_candidates_kernel(const float *__restrict__ resp, int w, int h,
                                                               double quality, uint32_t *__restrict__ scalars,
                                                               HarrisCand *__restrict__ cand, int cap) {
    int tbx, tby;
    xcd_tile(tbx, tby);
    const int x = tbx * PT_W + threadIdx.x % PT_W, y = tby * PT_H + threadIdx.x / PT_W;
    if (x < 1 || y < 1 || x >= w - 1 || y >= h - 1) return;
    const float maxv = ordered_to_float(scalars[0]);
    const float thr = (float)((double)maxv * quality);
    float v = resp[(size_t)y * w + x];
    float tv = v > thr ? v : 0.f;
    if (tv == 0.f) return;
    float m = tv;
#pragma unroll
    for (int j = -1; j <= 1; ++j)
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
            float n = resp[(size_t)(y + j) * w + (x + i)];
            n = n > thr ? n : 0.f;
            m = fmaxf(m, n);
        }
    if (tv == m) {
        uint32_t slot = atomicAdd(&scalars[1], 1u);
        if ((int)slot < cap) {
            cand[slot].v = tv;
            cand[slot].idx = y * w + x;
        }
    }
}

}  // namespace

int rdvio_launch_preprocess(rdvio_hip_ctx *ctx, int slot, const uint8_t *gray_dev, int w, int h, int stride,
                            double clip_limit, int tiles_x, int tiles_y) {
    ImageSlot &S = ctx->slots[slot];
    if (S.w != w || S.h != h) {
        rdvio_hip_pyr_layout_init(w, h, RDVIO_MAX_LEVELS - 1, &S.L);
        // derivative borders are BORDER_CONSTANT(0) and never written by the kernels
        RDVIO_HIP_CHECK(ctx, hipMemsetAsync(S.pyr_deriv, 0, (size_t)S.L.deriv_elems * sizeof(int16_t), ctx->stream));
        S.w = w;
        S.h = h;
    }
    // cv::CLAHE::apply tile geometry (pads right/bottom by reflect-101 to a tile multiple)
    int ew = w, eh = h;
    if (w % tiles_x != 0 || h % tiles_y != 0) {
        ew = w + (tiles_x - (w % tiles_x));
        eh = h + (tiles_y - (h % tiles_y));
    }
    const int tw = ew / tiles_x, th = eh / tiles_y;
    const int area = tw * th;
    const float lut_scale = (float)255 / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * area / 256);
        if (clip < 1) clip = 1;
    }
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles_x * tiles_y), dim3(CL_T), 0, ctx->stream, gray_dev, w, h, stride,
                       tiles_x, tw, th, clip, lut_scale, ctx->clahe_lut);
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    const int B = S.L.border;
    for (int lv = 0; lv < S.L.levels; ++lv) {
        dim3 grid((S.L.w[lv] + 2 * B + PT_W - 1) / PT_W, (S.L.h[lv] + 2 * B + PT_H - 1) / PT_H);
        if (lv == 0)
            hipLaunchKernelGGL(pyr_level_kernel<true>, grid, dim3(256), 0, ctx->stream, S.L, lv, S.pyr_img, S.pyr_deriv,
                               gray_dev, stride, ctx->clahe_lut, tiles_x, tiles_y, inv_tw, inv_th);
        else
            hipLaunchKernelGGL(pyr_level_kernel<false>, grid, dim3(256), 0, ctx->stream, S.L, lv, S.pyr_img, S.pyr_deriv,
                               nullptr, 0, nullptr, 0, 0, 0.f, 0.f);
    }
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    S.valid = true;
    return RDVIO_OK;
}

int rdvio_launch_harris(rdvio_hip_ctx *ctx, int slot) {
    ImageSlot &S = ctx->slots[slot];
    RDVIO_HIP_CHECK(ctx, hipMemsetAsync(ctx->harris_scalars, 0, 2 * sizeof(uint32_t), ctx->stream));
    dim3 grid((S.w + HT_W - 1) / HT_W, (S.h + HT_H - 1) / HT_H);
    hipLaunchKernelGGL(harris_kernel, grid, dim3(256), 0, ctx->stream, S.L, S.pyr_img, 0.04, ctx->harris,
                       ctx->harris_scalars);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}

int rdvio_launch_harris_candidates(rdvio_hip_ctx *ctx, int slot, double quality) {
    ImageSlot &S = ctx->slots[slot];
    dim3 grid((S.w + PT_W - 1) / PT_W, (S.h + PT_H - 1) / PT_H);
    hipLaunchKernelGGL(harris_candidates_kernel, grid, dim3(256), 0, ctx->stream, ctx->harris, S.w, S.h, quality,
                       ctx->harris_scalars, ctx->harris_cand, ctx->harris_cand_cap);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}
