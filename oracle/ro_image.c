/*
 * ORACLE (test infrastructure, NOT product code) -- CPU restatement of the image side of
 * rd_vio's per-frame feature tracker (rows A1-A3 of SURVEY.md section 8a).
 *
 * The reference delegates this arithmetic to OpenCV 4.x (unpinned version, absent from
 * /root/reference and from this image): cv::CLAHE, cv::buildOpticalFlowPyramid,
 * cv::calcOpticalFlowPyrLK, cv::GFTTDetector -- call sites
 * src/rdvio_extra/src/opencv_image.cpp:44,94,120,157,159,179-188.  What follows restates
 * OpenCV's published algorithms (modules/imgproc/src/clahe.cpp, pyramids.cpp,
 * featureselect.cpp, corner.cpp; modules/video/src/lkpyramid.cpp) from knowledge.
 * PARITY UNPINNED: no OpenCV build or fixture is available to check against.
 *
 * Deliberate, documented deviations (DESIGN.md "Image-side arithmetic"):
 *  - LK sums (A11,A12,A22,b1,b2) are accumulated EXACTLY in int64 and converted to float
 *    once; OpenCV accumulates in float in an order that depends on its SIMD build, so it
 *    has no single canonical result.  Exact accumulation is order-independent, which is
 *    what lets a parallel GPU reduction be bit-identical to this oracle.
 *  - Harris: Sobel/box sums are exact integers, the response is evaluated in double from
 *    them and rounded to float once (OpenCV: float filters with build-dependent order).
 */
#include "rdvio_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */
static inline int border_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}
static inline int cv_round_f(float v) { return (int)lrintf(v); } /* round-half-even, as cvRound (SSE cvtss2si) */
static inline int cv_floor_f(float v) { return (int)floorf(v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ------------------------------------------------------------------ pyramid layout */
void ro_pyr_layout_init(int w, int h, int max_level, ro_pyr_layout *L) {
    memset(L, 0, sizeof *L);
    L->border = RO_PYR_BORDER;
    int lw = w, lh = h, lv = 0;
    int64_t ioff = 0, doff = 0;
    for (lv = 0; lv <= max_level && lv < RO_MAX_LEVELS; ++lv) {
        L->w[lv] = lw;
        L->h[lv] = lh;
        int stride = (lw + 2 * L->border + 63) / 64 * 64;
        L->stride[lv] = stride;
        L->img_off[lv] = ioff;
        L->deriv_off[lv] = doff;
        int64_t rows = lh + 2 * L->border;
        ioff += (int64_t)stride * rows;
        doff += (int64_t)stride * rows * 2;
        L->levels = lv + 1;
        /* buildOpticalFlowPyramid stops when the next level would not exceed the window */
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
        if (lw <= RO_LK_WIN || lh <= RO_LK_WIN) break;
    }
    L->img_bytes = ioff;
    L->deriv_elems = doff;
}

static inline uint8_t *img_at(uint8_t *base, const ro_pyr_layout *L, int lv) {
    return base + L->img_off[lv] + (int64_t)L->border * L->stride[lv] + L->border;
}
static inline int16_t *der_at(int16_t *base, const ro_pyr_layout *L, int lv) {
    return base + L->deriv_off[lv] + ((int64_t)L->border * L->stride[lv] + L->border) * 2;
}

/* ------------------------------------------------------------------ A1a: CLAHE (cv::CLAHE::apply, 8-bit) */
void ro_clahe(const uint8_t *src, int w, int h, int src_stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t *dst, int dst_stride) {
    /* pad to a tile multiple by BORDER_REFLECT_101 on the right/bottom (clahe.cpp apply()) */
    int ew = w, eh = h;
    if (w % tiles_x != 0 || h % tiles_y != 0) {
        ew = w + (tiles_x - (w % tiles_x));
        eh = h + (tiles_y - (h % tiles_y));
    }
    int tw = ew / tiles_x, th = eh / tiles_y;
    int tile_area = tw * th;
    float lut_scale = (float)(255) / (float)tile_area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_area / 256);
        if (clip < 1) clip = 1;
    }
    uint8_t *lut = (uint8_t *)malloc((size_t)tiles_x * tiles_y * 256);
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            int hist[256];
            memset(hist, 0, sizeof hist);
            for (int y = 0; y < th; ++y) {
                int sy = border_reflect101(ty * th + y, h);
                for (int x = 0; x < tw; ++x) {
                    int sx = border_reflect101(tx * tw + x, w);
                    hist[src[sy * src_stride + sx]]++;
                }
            }
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; ++i)
                    if (hist[i] > clip) {
                        clipped += hist[i] - clip;
                        hist[i] = clip;
                    }
                int batch = clipped / 256;
                int residual = clipped - batch * 256;
                for (int i = 0; i < 256; ++i) hist[i] += batch;
                if (residual != 0) {
                    int step = 256 / residual;
                    if (step < 1) step = 1;
                    for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            uint8_t *tl = lut + (size_t)(ty * tiles_x + tx) * 256;
            int sum = 0;
            for (int i = 0; i < 256; ++i) {
                sum += hist[i];
                tl[i] = sat_u8(cv_round_f((float)sum * lut_scale));
            }
        }
    /* bilinear blend of the four neighbouring tile LUTs (CLAHE_Interpolation_Body) */
    float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < h; ++y) {
        float tyf = (float)y * inv_th - 0.5f;
        int ty1 = cv_floor_f(tyf), ty2 = ty1 + 1;
        float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        for (int x = 0; x < w; ++x) {
            float txf = (float)x * inv_tw - 0.5f;
            int tx1 = cv_floor_f(txf), tx2 = tx1 + 1;
            float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            int v = src[y * src_stride + x];
            float l11 = lut[(size_t)(ty1 * tiles_x + tx1) * 256 + v];
            float l12 = lut[(size_t)(ty1 * tiles_x + tx2) * 256 + v];
            float l21 = lut[(size_t)(ty2 * tiles_x + tx1) * 256 + v];
            float l22 = lut[(size_t)(ty2 * tiles_x + tx2) * 256 + v];
            float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            dst[y * dst_stride + x] = sat_u8(cv_round_f(res));
        }
    }
    free(lut);
}

/* ------------------------------------------------------------------ A1b: buildOpticalFlowPyramid */
static void fill_border_reflect101(uint8_t *base, const ro_pyr_layout *L, int lv) {
    int B = L->border, w = L->w[lv], h = L->h[lv], s = L->stride[lv];
    uint8_t *p0 = base + L->img_off[lv];
    for (int y = -B; y < h + B; ++y) {
        int sy = border_reflect101(y, h);
        for (int x = -B; x < w + B; ++x) {
            if (x >= 0 && x < w && y >= 0 && y < h) continue;
            int sx = border_reflect101(x, w);
            p0[(int64_t)(y + B) * s + (x + B)] = p0[(int64_t)(sy + B) * s + (sx + B)];
        }
    }
}

/* cv::pyrDown, 8-bit: separable [1 4 6 4 1], (sum + 128) >> 8, BORDER_REFLECT_101 */
static void pyr_down(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride) {
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            int acc = 0;
            static const int k[5] = {1, 4, 6, 4, 1};
            for (int j = -2; j <= 2; ++j) {
                int sy = border_reflect101(2 * y + j, sh);
                int row = 0;
                for (int i = -2; i <= 2; ++i) {
                    int sx = border_reflect101(2 * x + i, sw);
                    row += k[i + 2] * src[(int64_t)sy * sstride + sx];
                }
                acc += k[j + 2] * row;
            }
            dst[(int64_t)y * dstride + x] = (uint8_t)((acc + 128) >> 8);
        }
}

/* calcScharrDeriv (lkpyramid.cpp): dx = [3 10 3]^T x [-1 0 1], dy = [-1 0 1]^T x [3 10 3], reflect-101 */
static void scharr_deriv(const uint8_t *src, int w, int h, int sstride, int16_t *dst, int dstride) {
    for (int y = 0; y < h; ++y) {
        int y0 = border_reflect101(y - 1, h), y2 = border_reflect101(y + 1, h);
        for (int x = 0; x < w; ++x) {
            int x0 = border_reflect101(x - 1, w), x2 = border_reflect101(x + 1, w);
#define PX(yy, xx) ((int)src[(int64_t)(yy) * sstride + (xx)])
            int s0 = (PX(y0, x0) + PX(y2, x0)) * 3 + PX(y, x0) * 10;
            int s2 = (PX(y0, x2) + PX(y2, x2)) * 3 + PX(y, x2) * 10;
            int d0 = PX(y2, x0) - PX(y0, x0);
            int d1 = PX(y2, x) - PX(y0, x);
            int d2 = PX(y2, x2) - PX(y0, x2);
#undef PX
            dst[((int64_t)y * dstride + x) * 2 + 0] = (int16_t)(s2 - s0);
            dst[((int64_t)y * dstride + x) * 2 + 1] = (int16_t)((d0 + d2) * 3 + d1 * 10);
        }
    }
}

/* Build all levels from a (CLAHE'd) level-0 image.  img arena: L->img_bytes; deriv arena: L->deriv_elems int16.
 * Image borders are BORDER_REFLECT_101, derivative borders BORDER_CONSTANT(0) (buildOpticalFlowPyramid defaults). */
void ro_build_pyramid(const uint8_t *img, int w, int h, int img_stride, const ro_pyr_layout *L, uint8_t *pyr_img,
                      int16_t *pyr_deriv) {
    memset(pyr_img, 0, (size_t)L->img_bytes);
    memset(pyr_deriv, 0, (size_t)L->deriv_elems * sizeof(int16_t));
    uint8_t *l0 = img_at(pyr_img, L, 0);
    for (int y = 0; y < h; ++y) memcpy(l0 + (int64_t)y * L->stride[0], img + (int64_t)y * img_stride, (size_t)w);
    for (int lv = 0; lv < L->levels; ++lv) {
        if (lv > 0)
            pyr_down(img_at(pyr_img, L, lv - 1), L->w[lv - 1], L->h[lv - 1], L->stride[lv - 1], img_at(pyr_img, L, lv),
                     L->w[lv], L->h[lv], L->stride[lv]);
        fill_border_reflect101(pyr_img, L, lv);
        scharr_deriv(img_at(pyr_img, L, lv), L->w[lv], L->h[lv], L->stride[lv], der_at(pyr_deriv, L, lv), L->stride[lv]);
    }
}

/* OpenCvImage::preprocess, opencv_image.cpp:156-161 */
void ro_preprocess(const uint8_t *gray, int w, int h, int stride, double clip, int tiles_x, int tiles_y,
                   const ro_pyr_layout *L, uint8_t *pyr_img, int16_t *pyr_deriv) {
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * h);
    ro_clahe(gray, w, h, stride, clip, tiles_x, tiles_y, tmp, w);
    ro_build_pyramid(tmp, w, h, w, L, pyr_img, pyr_deriv);
    free(tmp);
}

/* ------------------------------------------------------------------ A2: pyramidal LK */
#define W_BITS 14
/* one level of LKTrackerInvoker for one point.  Returns 0 if the point was dropped at this level. */
static void lk_level(const ro_pyr_layout *L, int lv, int max_level, const uint8_t *Ibase, const int16_t *dIbase,
                     const uint8_t *Jbase, float prev_x, float prev_y, float *next_xy, uint8_t *status, int max_iter,
                     double eps_sq, double min_eig_thr, int use_initial) {
    const int win = RO_LK_WIN;
    const float half = (win - 1) * 0.5f;
    int w = L->w[lv], h = L->h[lv], s = L->stride[lv];
    const uint8_t *I = Ibase + L->img_off[lv] + (int64_t)L->border * s + L->border;
    const uint8_t *J = Jbase + L->img_off[lv] + (int64_t)L->border * s + L->border;
    const int16_t *dI = dIbase + L->deriv_off[lv] + ((int64_t)L->border * s + L->border) * 2;

    float scale = (float)(1.0 / (double)(1 << lv));
    float px = prev_x * scale, py = prev_y * scale;
    float nx, ny;
    if (lv == max_level) {
        if (use_initial) { nx = next_xy[0] * scale; ny = next_xy[1] * scale; }
        else { nx = px; ny = py; }
    } else {
        nx = next_xy[0] * 2.f;
        ny = next_xy[1] * 2.f;
    }
    next_xy[0] = nx;
    next_xy[1] = ny;

    px -= half; py -= half;
    int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
    if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
        if (lv == 0) *status = 0;
        return;
    }
    float a = px - (float)ipx, b = py - (float)ipy;
    int iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    int iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
    int iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    int16_t Iw[RO_LK_WIN * RO_LK_WIN], dIw[RO_LK_WIN * RO_LK_WIN * 2];
    int64_t iA11 = 0, iA12 = 0, iA22 = 0;
    for (int y = 0; y < win; ++y) {
        const uint8_t *src = I + (int64_t)(y + ipy) * s + ipx;
        const int16_t *dsrc = dI + ((int64_t)(y + ipy) * s + ipx) * 2;
        for (int x = 0; x < win; ++x) {
#define DESCALE(v, n) (((v) + (1 << ((n)-1))) >> (n))
            int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + s] * iw10 + src[x + s + 1] * iw11, W_BITS - 5);
            int ixval = DESCALE(dsrc[2 * x] * iw00 + dsrc[2 * x + 2] * iw01 + dsrc[2 * (x + s)] * iw10 +
                                    dsrc[2 * (x + s) + 2] * iw11, W_BITS);
            int iyval = DESCALE(dsrc[2 * x + 1] * iw00 + dsrc[2 * x + 3] * iw01 + dsrc[2 * (x + s) + 1] * iw10 +
                                    dsrc[2 * (x + s) + 3] * iw11, W_BITS);
            Iw[y * win + x] = (int16_t)ival;
            dIw[(y * win + x) * 2] = (int16_t)ixval;
            dIw[(y * win + x) * 2 + 1] = (int16_t)iyval;
            iA11 += (int64_t)ixval * ixval;
            iA12 += (int64_t)ixval * iyval;
            iA22 += (int64_t)iyval * iyval;
        }
    }
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
    if ((double)minEig < min_eig_thr || D < FLT_EPSILON) {
        if (lv == 0) *status = 0;
        return;
    }
    D = 1.f / D;

    nx -= half; ny -= half;
    float pdx = 0, pdy = 0;
    for (int j = 0; j < max_iter; ++j) {
        int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
        if (inx < -win || inx >= w || iny < -win || iny >= h) {
            if (lv == 0) *status = 0;
            break;
        }
        a = nx - (float)inx;
        b = ny - (float)iny;
        iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
        iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
        iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t ib1 = 0, ib2 = 0;
        for (int y = 0; y < win; ++y) {
            const uint8_t *Jp = J + (int64_t)(y + iny) * s + inx;
            for (int x = 0; x < win; ++x) {
                int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + s] * iw10 + Jp[x + s + 1] * iw11, W_BITS - 5) -
                           Iw[y * win + x];
                ib1 += (int64_t)diff * dIw[(y * win + x) * 2];
                ib2 += (int64_t)diff * dIw[(y * win + x) * 2 + 1];
            }
        }
#undef DESCALE
        float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
        float dx = (A12 * b2 - A22 * b1) * D;
        float dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        next_xy[0] = nx + half;
        next_xy[1] = ny + half;
        if ((double)dx * (double)dx + (double)dy * (double)dy <= eps_sq) break;
        if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
            next_xy[0] -= dx * 0.5f;
            next_xy[1] -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }
}

/* cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, prevPts, nextPts(inout), status, err, Size(21,21), maxLevel,
 *   TermCriteria(COUNT+EPS, max_iter, eps), OPTFLOW_USE_INITIAL_FLOW, minEigThreshold=1e-4) */
void ro_lk_flow(const ro_pyr_layout *L, const uint8_t *prev_img, const int16_t *prev_deriv, const uint8_t *next_img,
                int n, const float *prev_xy, float *next_xy, uint8_t *status, int max_iter, double eps) {
    if (max_iter < 0) max_iter = 0;
    if (max_iter > 100) max_iter = 100;
    if (eps < 0) eps = 0;
    if (eps > 10) eps = 10;
    double eps_sq = eps * eps;
    int max_level = L->levels - 1;
    for (int i = 0; i < n; ++i) status[i] = 1;
    for (int lv = max_level; lv >= 0; --lv)
        for (int i = 0; i < n; ++i)
            lk_level(L, lv, max_level, prev_img, prev_deriv, next_img, prev_xy[2 * i], prev_xy[2 * i + 1], next_xy + 2 * i,
                     status + i, max_iter, eps_sq, 1e-4, 1);
}

/* OpenCvImage::track_keypoints, opencv_image.cpp:75-154.
 * curr/next are double pixel coordinates; has_guess==0 -> next starts at curr (:79-85). */
void ro_track_keypoints(const ro_pyr_layout *L, const uint8_t *cur_img, const int16_t *cur_deriv,
                        const uint8_t *nxt_img, const int16_t *nxt_deriv, int n, const double *curr,
                        double *next, int has_guess, uint8_t *status) {
    if (n == 0) return;
    int cols = L->w[0], rows = L->h[0];
    float *cp = (float *)malloc(sizeof(float) * 2 * n), *np = (float *)malloc(sizeof(float) * 2 * n);
    float *rp = (float *)malloc(sizeof(float) * 2 * n);
    uint8_t *st = (uint8_t *)malloc(n), *rst = (uint8_t *)malloc(n);
    for (int i = 0; i < 2 * n; ++i) {
        cp[i] = (float)curr[i];
        np[i] = has_guess ? (float)next[i] : cp[i];
    }
    ro_lk_flow(L, cur_img, cur_deriv, nxt_img, n, cp, np, st, 30, 0.01);
    for (int i = 0; i < n; ++i) {
        status[i] = st[i];
        if (np[2 * i] < 20 || np[2 * i] >= (float)(cols - 20) || np[2 * i + 1] < 20 || np[2 * i + 1] >= (float)(rows - 20))
            status[i] = 0;
        if (status[i]) {
            float dx = np[2 * i] - cp[2 * i], dy = np[2 * i + 1] - cp[2 * i + 1];
            double nrm = sqrt((double)dx * (double)dx + (double)dy * (double)dy);
            if (nrm > (double)(rows / 4)) status[i] = 0;
        }
    }
    memcpy(rp, cp, sizeof(float) * 2 * n);
    ro_lk_flow(L, nxt_img, nxt_deriv, cur_img, n, np, rp, rst, 30, 0.01);
    for (int i = 0; i < n; ++i) {
        if (status[i]) {
            float dx = cp[2 * i] - rp[2 * i], dy = cp[2 * i + 1] - rp[2 * i + 1];
            double nrm = sqrt((double)dx * dx + (double)dy * dy);
            if (!rst[i] || nrm > 0.5) status[i] = 0;
        }
    }
    for (int i = 0; i < n; ++i)
        if (status[i]) {
            next[2 * i] = np[2 * i];
            next[2 * i + 1] = np[2 * i + 1];
        }
    free(cp); free(np); free(rp); free(st); free(rst);
}

/* ------------------------------------------------------------------ A3: GFTT-Harris */
/* cornerHarris(blockSize 3, ksize 3, k) response as float.  Exact-integer restatement: Sobel sums are
 * integers, the 3x3 box sums of their products are exact in int64, response evaluated once in double:
 *   s = 1/(4*3*255); a = s^2 Sxx, b = s^2 Sxy, c = s^2 Syy; R = a c - b^2 - k (a+c)^2
 * Borders: BORDER_REFLECT_101 for Sobel and for the box filter (BORDER_DEFAULT). */
void ro_harris_response(const uint8_t *img, int w, int h, int stride, double k, float *resp) {
    int *dx = (int *)malloc(sizeof(int) * w * h), *dy = (int *)malloc(sizeof(int) * w * h);
    for (int y = 0; y < h; ++y) {
        int y0 = border_reflect101(y - 1, h), y2 = border_reflect101(y + 1, h);
        for (int x = 0; x < w; ++x) {
            int x0 = border_reflect101(x - 1, w), x2 = border_reflect101(x + 1, w);
#define PX(yy, xx) ((int)img[(int64_t)(yy) * stride + (xx)])
            dx[y * w + x] = (PX(y0, x2) + 2 * PX(y, x2) + PX(y2, x2)) - (PX(y0, x0) + 2 * PX(y, x0) + PX(y2, x0));
            dy[y * w + x] = (PX(y2, x0) + 2 * PX(y2, x) + PX(y2, x2)) - (PX(y0, x0) + 2 * PX(y0, x) + PX(y0, x2));
#undef PX
        }
    }
    const double s = 1.0 / (4.0 * 3.0 * 255.0);
    const double s2 = s * s;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int64_t sxx = 0, sxy = 0, syy = 0;
            for (int j = -1; j <= 1; ++j) {
                int yy = border_reflect101(y + j, h);
                for (int i = -1; i <= 1; ++i) {
                    int xx = border_reflect101(x + i, w);
                    int64_t gx = dx[yy * w + xx], gy = dy[yy * w + xx];
                    sxx += gx * gx; sxy += gx * gy; syy += gy * gy;
                }
            }
            double a = s2 * (double)sxx, b = s2 * (double)sxy, c = s2 * (double)syy;
            resp[y * w + x] = (float)(a * c - b * b - k * (a + c) * (a + c));
        }
    free(dx); free(dy);
}

typedef struct { float v; int idx; } cand_t;
static int cand_cmp(const void *pa, const void *pb) {
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->idx > b->idx) ? -1 : (a->idx < b->idx ? 1 : 0); /* greaterThanPtr: ties by higher address */
}

/* cv::goodFeaturesToTrack(useHarris) as driven by GFTTDetector(maxCorners, 1e-3, minDist, 3, true, 0.04):
 * threshold at quality*max, 3x3 local maxima (dilate), sort desc, greedy min-distance grid, cap.
 * Output: corners (x,y) float + responses, in acceptance order.  Returns count. */
int ro_good_features(const uint8_t *img, int w, int h, int stride, int max_corners, double quality, double min_dist,
                     double k, float *out_xy, float *out_resp) {
    float *eig = (float *)malloc(sizeof(float) * w * h);
    ro_harris_response(img, w, h, stride, k, eig);
    double maxv = -DBL_MAX;
    for (int i = 0; i < w * h; ++i)
        if (eig[i] > maxv) maxv = eig[i];
    /* threshold(eig, eig, maxVal*quality, 0, THRESH_TOZERO): keep src if src > thresh (thresh cast to float) */
    float thr = (float)(maxv * quality);
    for (int i = 0; i < w * h; ++i)
        if (!(eig[i] > thr)) eig[i] = 0.f;
    cand_t *cands = (cand_t *)malloc(sizeof(cand_t) * w * h);
    int nc = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float v = eig[y * w + x];
            if (v == 0.f) continue;
            float m = v;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) {
                    float t = eig[(y + j) * w + x + i];
                    if (t > m) m = t;
                }
            if (v == m) {
                cands[nc].v = v;
                cands[nc].idx = y * w + x;
                nc++;
            }
        }
    qsort(cands, nc, sizeof(cand_t), cand_cmp);
    int ncorners = 0;
    if (min_dist >= 1) {
        int cell = (int)lrint(min_dist);
        int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        double md2 = min_dist * min_dist;
        /* grid cells as linked lists over accepted corners */
        int *head = (int *)malloc(sizeof(int) * gw * gh), *next = (int *)malloc(sizeof(int) * (nc > 0 ? nc : 1));
        for (int i = 0; i < gw * gh; ++i) head[i] = -1;
        for (int i = 0; i < nc; ++i) {
            int y = cands[i].idx / w, x = cands[i].idx % w;
            int xc = x / cell, yc = y / cell;
            int x1 = xc - 1 < 0 ? 0 : xc - 1, y1 = yc - 1 < 0 ? 0 : yc - 1;
            int x2 = xc + 1 > gw - 1 ? gw - 1 : xc + 1, y2 = yc + 1 > gh - 1 ? gh - 1 : yc + 1;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (int c = head[yy * gw + xx]; c >= 0; c = next[c]) {
                        float ddx = (float)x - out_xy[2 * c], ddy = (float)y - out_xy[2 * c + 1];
                        if ((double)(ddx * ddx + ddy * ddy) < md2) { good = 0; break; }
                    }
            if (good) {
                out_xy[2 * ncorners] = (float)x;
                out_xy[2 * ncorners + 1] = (float)y;
                if (out_resp) out_resp[ncorners] = cands[i].v;
                next[ncorners] = head[yc * gw + xc];
                head[yc * gw + xc] = ncorners;
                ++ncorners;
                if (max_corners > 0 && ncorners == max_corners) break;
            }
        }
        free(head); free(next);
    } else {
        for (int i = 0; i < nc; ++i) {
            out_xy[2 * ncorners] = (float)(cands[i].idx % w);
            out_xy[2 * ncorners + 1] = (float)(cands[i].idx / w);
            if (out_resp) out_resp[ncorners] = cands[i].v;
            ++ncorners;
            if (max_corners > 0 && ncorners == max_corners) break;
        }
    }
    free(cands); free(eig);
    return ncorners;
}

/* ------------------------------------------------------------------ PoissonDiskFilter<2> (util/poisson_disk_filter.h) */
/* The reference keeps a sparse hash grid holding at most ONE point per cell (later presets overwrite the
 * cell, :20-24) and scans cells in a quirky order: the first cell (ibegin) is skipped and one cell past
 * the end is visited (:77-92).  Restated with a dense grid + explicit handling of both quirks. */
typedef struct {
    double radius, r2, gsize;
    int span, gx0, gy0, gw, gh;
    int *cell; /* index into pts or -1 */
    double *pts;
    int npts, cap;
} pdf_t;

static void pdf_init(pdf_t *f, double radius, int w, int h, int cap) {
    f->radius = radius;
    f->r2 = radius * radius;
    f->gsize = radius / sqrt(2.0);
    f->span = (int)ceil(sqrt(2.0));
    /* cover coordinates in [-4r, max+4r] */
    f->gx0 = (int)floor(-4.0 * radius / f->gsize) - f->span - 2;
    f->gy0 = f->gx0;
    f->gw = (int)floor((w + 4.0 * radius) / f->gsize) + f->span + 3 - f->gx0;
    f->gh = (int)floor((h + 4.0 * radius) / f->gsize) + f->span + 3 - f->gy0;
    f->cell = (int *)malloc(sizeof(int) * f->gw * f->gh);
    for (int i = 0; i < f->gw * f->gh; ++i) f->cell[i] = -1;
    f->pts = (double *)malloc(sizeof(double) * 2 * cap);
    f->npts = 0;
    f->cap = cap;
}
static void pdf_free(pdf_t *f) { free(f->cell); free(f->pts); }
static int pdf_lookup(const pdf_t *f, int ix, int iy) {
    ix -= f->gx0; iy -= f->gy0;
    if (ix < 0 || iy < 0 || ix >= f->gw || iy >= f->gh) return -1;
    return f->cell[iy * f->gw + ix];
}
static int pdf_test(const pdf_t *f, double x, double y, int *oix, int *oiy) {
    int ix = (int)floor(x / f->gsize), iy = (int)floor(y / f->gsize);
    *oix = ix; *oiy = iy;
    int bx = ix - f->span, by = iy - f->span, ex = ix + f->span, ey = iy + f->span;
    int cx = bx, cy = by;
    while (cy <= ey) {
        cx++;
        if (cx > ex) { cx = bx; cy++; }
        int p = pdf_lookup(f, cx, cy);
        if (p >= 0) {
            double dx = x - f->pts[2 * p], dy = y - f->pts[2 * p + 1];
            if (dx * dx + dy * dy < f->r2) return 0;
        }
    }
    return 1;
}
static void pdf_put(pdf_t *f, double x, double y, int ix, int iy) {
    int cx = ix - f->gx0, cy = iy - f->gy0;
    if (cx >= 0 && cy >= 0 && cx < f->gw && cy < f->gh) f->cell[cy * f->gw + cx] = f->npts;
    f->pts[2 * f->npts] = x;
    f->pts[2 * f->npts + 1] = y;
    f->npts++;
}

/* OpenCvImage::detect_keypoints, opencv_image.cpp:38-73: GFTT -> sort by response (desc; GFTT output is
 * already in that order) -> PoissonDiskFilter seeded with the existing keypoints -> drop < 20 px from border.
 * keypoints: in/out double (x,y), n_existing on entry; returns new total (capacity must be n_existing+max_corners). */
int ro_detect_keypoints(const uint8_t *img, int w, int h, int stride, int max_corners, double min_dist_poisson,
                        double *keypoints, int n_existing) {
    float *xy = (float *)malloc(sizeof(float) * 2 * (max_corners > 0 ? max_corners : w * h));
    /* GFTTDetector::create(max_points, 1.0e-3, 20, 3, true) -> k = 0.04 (opencv_image.cpp:184-188) */
    int nc = ro_good_features(img, w, h, stride, max_corners, 1.0e-3, 20.0, 0.04, xy, NULL);
    int total = n_existing;
    if (nc > 0) {
        pdf_t f;
        pdf_init(&f, min_dist_poisson, w, h, n_existing + nc);
        int ix, iy;
        for (int i = 0; i < n_existing; ++i) {
            double x = keypoints[2 * i], y = keypoints[2 * i + 1];
            ix = (int)floor(x / f.gsize); iy = (int)floor(y / f.gsize);
            pdf_put(&f, x, y, ix, iy);
        }
        for (int i = 0; i < nc; ++i) {
            double x = xy[2 * i], y = xy[2 * i + 1];
            if (pdf_test(&f, x, y, &ix, &iy)) {
                pdf_put(&f, x, y, ix, iy);
                if (x < 20 || y < 20 || x >= w - 20 || y >= h - 20) continue;
                keypoints[2 * total] = x;
                keypoints[2 * total + 1] = y;
                total++;
            }
        }
        pdf_free(&f);
    }
    free(xy);
    return total;
}
