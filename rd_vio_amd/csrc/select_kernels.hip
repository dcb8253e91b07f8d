// Keypoint selection on gfx950: what cv::goodFeaturesToTrack does after the response map (sort by response, greedy
// minDistance selection, maxCorners cut) and what OpenCvImage::detect_keypoints does with the result
// (/root/reference/src/rdvio_extra/src/opencv_image.cpp:44-72: response order, PoissonDiskFilter<2> seeded with the existing
// keypoints, 20-px border test) -- order-exact with the host restatement in host_select.cpp, which stays as the road for
// inputs beyond this file's LDS capacities (it is product host code, not the oracle).
//
// Two single-workgroup kernels (everything lives in LDS; the candidate list is a few thousand entries):
//   gftt_select_kernel    bitonic sort of the Harris local maxima by (response desc, pixel index desc) = cv::greaterThanPtr;
//                         then the greedy minDistance pass.  The greedy pass is sequential as written ("accept a
//                         candidate unless an ALREADY ACCEPTED corner is closer than minDistance"), but its result is the
//                         lexicographically-first maximal independent set of the "closer than minDistance" graph in
//                         priority order, which has a parallel fixed-point form: a candidate is rejected as soon as one
//                         higher-priority neighbour is accepted and accepted as soon as all of them are rejected.  Every
//                         candidate of a chunk of 1024 (in priority order; a chunk only depends on earlier ones) polls its
//                         handful of higher-priority neighbours until that rule fires: the sequential loop's result exactly,
//                         including the stop at maxCorners accepted.
//   poisson_filter_kernel PoissonDiskFilter<2>::preset_points / insert_points, kept literally: ONE point index per grid
//                         cell (a later preset overwrites an earlier one in the same cell), the reference's neighbourhood
//                         walk that skips the first cell of the 5x5 block and visits one cell past its end
//                         (poisson_disk_filter.h:77-92), points compared in double.  The inserts are sequential in the
//                         reference and stay sequential here: one wavefront, one corner per trip, the 25 visited cells on
//                         25 lanes, one ballot.
#include "ctx.hpp"
#include "select_caps.hpp"

namespace {

constexpr int ST = 1024;

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(ST) void gftt_select_kernel(const HarrisCand *__restrict__ cand, const uint32_t *__restrict__ scalars, int cap,
                                                        int w, int h, int max_corners, int cell, float md2, float *__restrict__ corners_out,
                                                        int32_t *__restrict__ hdr) {
    __shared__ unsigned long long keys[RDVIO_SEL_NC_MAX];   // 64 KB: (ordered response << 32) | pixel index, sorted descending
    __shared__ unsigned int clist[RDVIO_SEL_NC_MAX];        // 32 KB: candidates grouped by grid cell: rank | x in cell << 13 | y in cell << 19
    __shared__ unsigned char state[RDVIO_SEL_NC_MAX];       //  8 KB: 0 undecided, 1 accepted, 2 rejected
    __shared__ int coff[RDVIO_SEL_GCELLS_MAX + 1];          // 16 KB: first clist entry of a cell
    __shared__ int ccur[RDVIO_SEL_GCELLS_MAX];              // 16 KB: per-cell counters / fill cursors
    __shared__ int wsum[ST / 64];
    __shared__ int s_total, s_fail;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int nc = (int)min(scalars[1], (uint32_t)cap);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell, ncell = gw * gh;
    const unsigned long long clk0 = wall_clock64();   // diagnostic stamps (10 ns units) -> hdr[8..12]
#define SEL_STAMP(k) do { if (t == 0) hdr[8 + (k)] = (int)(wall_clock64() - clk0); } while (0)
    if (t == 0) {
        hdr[0] = nc;
        hdr[1] = (nc > RDVIO_SEL_NC_MAX || ncell > RDVIO_SEL_GCELLS_MAX || max_corners > RDVIO_SEL_CORNERS_MAX || w > 65535 || h > 65535 || cell > 64) ? 1 : 0;  // host road
        hdr[2] = 0;
        hdr[14] = 0;
    }
    if (nc > RDVIO_SEL_NC_MAX || ncell > RDVIO_SEL_GCELLS_MAX || max_corners > RDVIO_SEL_CORNERS_MAX || w > 65535 || h > 65535 || cell > 64) return;
    // ---- which candidates take part.  The greedy pass stops at maxCorners accepted corners, as a rule long before the end of the
    // list: sorting all of it (4096-8192 keys: 78-91 compare-exchange stages) to walk the first few hundred is the kernel's largest
    // piece.  First attempt: only the candidates of the top response bins -- a histogram over 8 exponent + 4 mantissa bits of the
    // response, bins taken from the top until they hold `target` candidates; those ARE the head of the sorted list (everything
    // left out has a smaller key), so the greedy pass over them is the head of the full pass.  If it ends short of maxCorners
    // the kernel starts over with every candidate.
    __shared__ int s_bstar, s_n1, s_cnt;
    const int target = max(1024, 6 * max_corners);
    for (int attempt = 0; attempt < 2; ++attempt) {
    bool partial = attempt == 0 && nc > 2 * target && (scalars[0] & 0x80000000u) != 0;   // (workgroup-uniform)
    if (partial) {
        for (int i = t; i < 4096; i += ST) coff[i] = 0;
        if (t == 0) {
            s_bstar = -1;
            s_n1 = 0;
            s_cnt = 0;
        }
        __syncthreads();
        for (int i = t; i < nc; i += ST) {
            const uint32_t o = f2ord(cand[i].v);
            atomicAdd(&coff[(o & 0x80000000u) ? (int)((o >> 19) & 0xfffu) : 0], 1);
        }
        __syncthreads();
        // counts from the top bin downwards: thread t owns bins 4095 - 4 t - q
        int v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = coff[4095 - (4 * t + q)];
            sum += v[q];
        }
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int nn_ = __shfl_up(inc, off);
            if (lane >= off) inc += nn_;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int q = 0; q < wave; ++q) run += wsum[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int after = run + v[q];
            if (run < target && after >= target) {   // (one bin crosses the target, if the list is long enough)
                s_bstar = 4095 - (4 * t + q);
                s_n1 = after;
            }
            run = after;
        }
        __syncthreads();
        if (s_bstar < 0 || s_n1 > 2048) partial = false;
        __syncthreads();
    }
    const int n = partial ? s_n1 : nc;   // candidates of this attempt
    // ---- sort (bitonic, descending).  Keys are unique (the pixel index is part of them), padding keys (0) sink to the end.
    int P = 2;
    while (P < n) P <<= 1;
    if (partial) {
        const int bstar = s_bstar;
        for (int i = t; i < nc; i += ST) {
            const uint32_t o = f2ord(cand[i].v);
            const int bin = (o & 0x80000000u) ? (int)((o >> 19) & 0xfffu) : 0;
            if (bin >= bstar) keys[atomicAdd(&s_cnt, 1)] = ((unsigned long long)o << 32) | (uint32_t)cand[i].idx;
        }
        for (int i = n + t; i < P; i += ST) keys[i] = 0;
        for (int i = t; i < P; i += ST) state[i] = 0;
    } else {
        for (int i = t; i < P; i += ST) {
            unsigned long long k = 0;
            if (i < nc) k = ((unsigned long long)f2ord(cand[i].v) << 32) | (uint32_t)cand[i].idx;
            keys[i] = k;
            if (i < RDVIO_SEL_NC_MAX) state[i] = 0;
        }
    }
    for (int i = t; i < ncell; i += ST) ccur[i] = 0;
    __syncthreads();
    SEL_STAMP(0);
    // Thread t owns the elements t, t + ST, ...: a compare-exchange at distance j < 64 stays inside one wavefront (same element
    // row, lanes t and t ^ j), whose LDS operations execute in order -- such a stage needs no workgroup barrier, and 57 of the 78
    // stages of a 4096-key sort are of that kind.  A barrier stands before and behind every stage that crosses wavefronts.
    // Up to two keys per thread (P <= 2 ST: every frame of the reference's sizes, and the top-bin attempt always) the keys are
    // sorted IN REGISTERS: wave-local stages exchange through the cross-lane network, the two keys of a thread meet in the
    // thread, only the stages at distance 64 ... ST/2 go through LDS.
    if (P <= 2 * ST) {
        const bool two = P > ST;
        const int PP = two ? 2 * ST : ST;   // (threads beyond P hold zero keys: they stay at the end of a descending sort)
        unsigned long long r[2];
        r[0] = t < P ? keys[t] : 0ull;
        r[1] = (two && t + ST < P) ? keys[t + ST] : 0ull;
        __syncthreads();
        for (int k = 2; k <= PP; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= ST) {   // the thread's own two keys (indices t and t + ST), k = 2 ST: descending
                    if (r[0] < r[1]) {
                        const unsigned long long tmp = r[0];
                        r[0] = r[1];
                        r[1] = tmp;
                    }
                    continue;
                }
                unsigned long long o[2];
                if (j >= 64) {
                    keys[t] = r[0];
                    if (two) keys[t + ST] = r[1];
                    __syncthreads();
                    o[0] = keys[t ^ j];
                    o[1] = two ? keys[(t ^ j) + ST] : 0ull;
                    __syncthreads();
                } else {
                    o[0] = __shfl_xor(r[0], j);
                    o[1] = two ? __shfl_xor(r[1], j) : 0ull;
                }
                const bool lower = (t & j) == 0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (e == 1 && !two) break;
                    const bool desc = ((t + ST * e) & k) == 0;
                    const bool take_max = lower == desc;
                    const unsigned long long mx = r[e] > o[e] ? r[e] : o[e], mn = r[e] > o[e] ? o[e] : r[e];
                    r[e] = take_max ? mx : mn;
                }
            }
        if (t < P) keys[t] = r[0];
        if (two && t + ST < P) keys[t + ST] = r[1];
        __syncthreads();
    } else {
    bool synced = true;
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool cross = j >= 64;
            if (cross && !synced) __syncthreads();
            for (int i = t; i < P; i += ST) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < b : a > b) {
                        keys[i] = b;
                        keys[ixj] = a;
                    }
                }
            }
            if (cross) {
                __syncthreads();
                synced = true;
            } else {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                synced = false;
            }
        }
    if (!synced) __syncthreads();
    }
    SEL_STAMP(1);
    // ---- candidates grouped by grid cell (counting sort; the order inside a cell does not matter)
    for (int i = t; i < n; i += ST) {
        const int idx = (int)(uint32_t)keys[i];
        const int y = idx / w, x = idx - y * w;
        atomicAdd(&ccur[(y / cell) * gw + x / cell], 1);
    }
    __syncthreads();
    {
        // exclusive scan over the cells: four consecutive cells per thread, wave scan, wave totals through LDS
        int v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * t + q;
            v[q] = c < ncell ? ccur[c] : 0;
            sum += v[q];
        }
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(inc, off);
            if (lane >= off) inc += n;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < wave; ++q) base += wsum[q];
        int run = base + inc - sum;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * t + q;
            if (c < ncell) coff[c] = run;
            run += v[q];
        }
        if (t == ST - 1) coff[ncell] = run;
    }
    __syncthreads();
    for (int i = t; i < ncell; i += ST) ccur[i] = 0;
    if (t == 0) {
        s_total = 0;
        s_fail = 0;
    }
    __syncthreads();
    for (int i = t; i < n; i += ST) {
        const int idx = (int)(uint32_t)keys[i];
        const int y = idx / w, x = idx - y * w, xc = x / cell, yc = y / cell, c = yc * gw + xc;
        clist[coff[c] + atomicAdd(&ccur[c], 1)] = (unsigned)i | ((unsigned)(x - xc * cell) << 13) | ((unsigned)(y - yc * cell) << 19);
    }
    __syncthreads();
    SEL_STAMP(2);
    // ---- greedy minDistance selection, a chunk of 1024 ranks at a time
    for (int base = 0; base < n; base += ST) {
        if (s_total >= max_corners) break;  // (uniform: s_total is only written between barriers)
        const int i = base + t;
        const bool mine = i < n;
        int x = 0, y = 0;
        if (mine) {
            const int idx = (int)(uint32_t)keys[i];
            y = idx / w;
            x = idx - y * w;
        }
        const int xc = x / cell, yc = y / cell;
        const int x1 = max(0, xc - 1), y1 = max(0, yc - 1), x2 = min(gw - 1, xc + 1), y2 = min(gh - 1, yc + 1);
        // the cells x1..x2 of a grid row are one contiguous run of the cell-ordered list: three runs per candidate, their
        // boundaries (also the cell boundaries inside a run) read up front; an entry carries its rank and its position inside its
        // cell, so nothing else is read per entry
        auto for_each_close_higher = [&](auto &&fn) {
            for (int yy = y1; yy <= y2; ++yy) {
                const int c0 = yy * gw + x1;
                const int b0 = coff[c0], b1 = coff[c0 + 1], b2 = (x1 + 1 <= x2) ? coff[c0 + 2] : b1, b3 = (x1 + 2 <= x2) ? coff[c0 + 3] : b2;
                for (int e = b0; e < b3; ++e) {
                    const unsigned v = clist[e];
                    const int j = (int)(v & 8191u);
                    if (j >= i) continue;
                    const int xx = x1 + (e >= b1 ? 1 : 0) + (e >= b2 ? 1 : 0);
                    const float dx = (float)x - (float)(xx * cell + (int)((v >> 13) & 63u)), dy = (float)y - (float)(yy * cell + (int)((v >> 19) & 63u));
                    if (!(dx * dx + dy * dy < md2)) continue;
                    fn(j);
                }
            }
        };
        // higher-priority candidates closer than minDistance, collected once (typically a handful)
        constexpr int NBMAX = 12;
        int nb[NBMAX], nn = 0;
        bool overflow = false;
        if (mine)
            for_each_close_higher([&](int j) {
                if (nn < NBMAX) {
#pragma unroll
                    for (int q = 0; q < NBMAX; ++q)
                        if (q == nn) nb[q] = j;   // (static indices: the list stays in registers)
                    ++nn;
                } else {
                    overflow = true;
                }
            });
        // Resolution by polling: a candidate only ever waits for higher-priority ones, so the highest-priority undecided
        // candidate can always decide.  Every thread polls its own neighbours (states only move from undecided to final).
        // The loop is WAVE-UNIFORM (it runs until every lane of the wavefront is done) and the state is stored inside its
        // body: a divergent `store; break` would be moved behind the loop by the compiler's control-flow structurisation,
        // i.e. published only when the whole wavefront has left the loop -- a lane waiting for another lane of its own
        // wavefront would then wait forever.  The trip count is bounded; running out of trips hands the frame to the host road.
        if (base == 0) SEL_STAMP(4);
        typedef volatile __attribute__((address_space(3))) unsigned char lds_vu8;  // (volatile AND LDS-typed: ds_read / ds_write, re-read every trip)
        lds_vu8 *vstate = (lds_vu8 *)state;
        bool done = !mine;
        int trips = 0;
        while (__ballot(!done) != 0ull) {
            if (!done) {
                bool any_acc = false, any_und = false;
#pragma unroll
                for (int q = 0; q < NBMAX; ++q)
                    if (q < nn) {
                        const int sj = vstate[nb[q]];
                        any_acc |= sj == 1;
                        any_und |= sj == 0;
                    }
                if (overflow && !any_acc)   // (rare: more than NBMAX close neighbours) the full walk
                    for_each_close_higher([&](int j) {
                        const int sj = vstate[j];
                        any_acc |= sj == 1;
                        any_und |= sj == 0;
                    });
                if (any_acc || !any_und) {
                    vstate[i] = any_acc ? 2 : 1;
                    done = true;
                }
            }
            if (++trips > 200000) {  // (never reached by a consistent candidate list: ~0.1 s of polling)
                if (!done) s_fail = 1;
                done = true;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (base == 0 && lane == 0) atomicMax(&hdr[14], trips);
        __syncthreads();
        if (base == 0) SEL_STAMP(5);
        if (s_fail) {
            if (t == 0) hdr[1] = 4;
            return;
        }
        // accepted candidates of the chunk in rank order -> their positions in the corner list
        const bool acc = mine && state[i] == 1;
        const unsigned long long bal = __ballot(acc);
        const int before_me = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = s_total, chunk = 0;
        for (int q = 0; q < ST / 64; ++q) {
            if (q < wave) off += wsum[q];
            chunk += wsum[q];
        }
        if (acc && off + before_me < max_corners) {
            corners_out[2 * (off + before_me)] = (float)x;
            corners_out[2 * (off + before_me) + 1] = (float)y;
        }
        __syncthreads();
        if (t == 0) s_total += chunk;
        __syncthreads();
    }
    __syncthreads();
    SEL_STAMP(3);
    if (t == 0) hdr[4] = partial ? 1 : (attempt == 0 ? 0 : 2);   // diagnostic: 1 the top bins sufficed, 2 they did not (second pass over everything), 0 one pass over everything
    if (!(partial && s_total < max_corners)) break;   // (uniform: s_total is final behind the barrier)
    }
    if (t == 0) hdr[2] = min(s_total, max_corners);
}

// PoissonDiskFilter<2> (poisson_disk_filter.h) + the 20-px border test of opencv_image.cpp:61-68.
// existing: n_existing x 2 doubles; corners: hdr[2] x 2 floats (response order); new_out: accepted corners inside the
// border, in order; hdr[3] = their count; hdr[1] |= 2 when the grid or the point table would not fit (host road).
__global__ __launch_bounds__(256) void poisson_filter_kernel(const double *__restrict__ existing, int n_existing, const float *__restrict__ corners,
                                                            int w, int h, double radius, double gftt_md2, double *__restrict__ new_out,
                                                            int32_t *__restrict__ hdr) {
    __shared__ int grid[RDVIO_SEL_PGRID_MAX];                                  // 88 KB: point index per cell or -1
    __shared__ __attribute__((aligned(16))) double pts[2 * RDVIO_SEL_PTS_MAX];  // 64 KB
    const int t = threadIdx.x;
    if (hdr[1] != 0) return;
    const int ncorn = hdr[2];
    const double cellp = radius / sqrt(2.0), r2 = radius * radius;
    // cells touched by any corner inside the image, widened by the walk's reach (2 left / up, 2 right, 3 down) -> 3 all round
    constexpr int M = 3;
    const int gx = (int)floor((double)(w - 1) / cellp) + 1 + 2 * M, gy = (int)floor((double)(h - 1) / cellp) + 1 + 2 * M;
    if ((long long)gx * gy > RDVIO_SEL_PGRID_MAX || n_existing + ncorn > RDVIO_SEL_PTS_MAX) {
        if (t == 0) hdr[1] |= 2;
        return;
    }
    for (int i = t; i < gx * gy; i += 256) grid[i] = -1;
    __syncthreads();
    // presets: "grid_[cell] = index" in order, i.e. the LAST preset of a cell stays (atomicMax of the index)
    for (int i = t; i < n_existing; i += 256) {
        const double x = existing[2 * i], y = existing[2 * i + 1];
        pts[2 * i] = x;
        pts[2 * i + 1] = y;
        const double fx = floor(x / cellp), fy = floor(y / cellp);
        // a preset further out than the margin can never be visited from a corner inside the image
        if (fx >= -M && fy >= -M && fx < gx - M && fy < gy - M) atomicMax(&grid[((int)fy + M) * gx + (int)fx + M], i);
    }
    // cell coordinates of every corner (two double divisions each) in parallel, ahead of the sequential pass
    __shared__ short ccx[RDVIO_SEL_CORNERS_MAX], ccy[RDVIO_SEL_CORNERS_MAX];
    __shared__ unsigned char keep[RDVIO_SEL_CORNERS_MAX];
    for (int k = t; k < ncorn; k += 256) {
        ccx[k] = (short)(int)floor((double)corners[2 * k] / cellp);
        ccy[k] = (short)(int)floor((double)corners[2 * k + 1] / cellp);
    }
    __syncthreads();
    // does corner (x, y) in cell (cx, cy) conflict with a resident point of the grid?  the reference's walk over the block
    // [cx-2, cx+2] x [cy-2, cy+2]: it steps BEFORE it looks, so it never visits (cx-2, cy-2) and ends on (cx-2, cy+3)
    auto conflict_at = [&](double x, double y, int cx, int cy, int pos) {
        const int vx = cx - 2 + pos % 5, vy = cy - 2 + pos / 5;
        const int ax = vx + M, ay = vy + M;
        if (ax < 0 || ay < 0 || ax >= gx || ay >= gy) return false;
        const int p = grid[ay * gx + ax];
        if (p < 0) return false;
        const double dx = x - pts[2 * p], dy = y - pts[2 * p + 1];
        return dx * dx + dy * dy < r2;
    };
    if (gftt_md2 >= r2) {
        // The corners are at least minDistance apart, so with radius <= minDistance no accepted corner can reject another
        // one (their distance is never < radius) and an accepted corner never lands in an occupied cell: every insert
        // only depends on the presets -- decided in parallel, one corner per thread, order restored by a prefix count.
        for (int k = t; k < ncorn; k += 256) {
            const double x = (double)corners[2 * k], y = (double)corners[2 * k + 1];
            bool c = false;
            for (int pos = 1; pos <= 25 && !c; ++pos) c = conflict_at(x, y, ccx[k], ccy[k], pos);
            keep[k] = (!c && !(x < 20 || y < 20 || x >= w - 20 || y >= h - 20)) ? 1 : 0;
        }
        __syncthreads();
        if (t >= 64) return;
        int total = 0;
        for (int base = 0; base < ncorn; base += 64) {
            const int k = base + t;
            const bool kp = k < ncorn && keep[k];
            const unsigned long long bal = __ballot(kp);
            if (kp) {
                const int o = total + __popcll(bal & ((1ull << t) - 1ull));
                new_out[2 * o] = (double)corners[2 * k];
                new_out[2 * o + 1] = (double)corners[2 * k + 1];
            }
            total += __popcll(bal);
        }
        if (t == 0) hdr[3] = total;
        return;
    }
    if (t >= 64) return;
    int npts = n_existing, total = 0;
    for (int k = 0; k < ncorn; ++k) {
        const double x = (double)corners[2 * k], y = (double)corners[2 * k + 1];
        const int cx = ccx[k], cy = ccy[k];
        const bool conflict = t < 25 && conflict_at(x, y, cx, cy, t + 1);
        if (__ballot(conflict) != 0ull) continue;
        if (t == 0) {
            grid[(cy + M) * gx + cx + M] = npts;
            pts[2 * npts] = x;
            pts[2 * npts + 1] = y;
        }
        npts++;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (x < 20 || y < 20 || x >= w - 20 || y >= h - 20) continue;
        if (t == 0) {
            new_out[2 * total] = x;
            new_out[2 * total + 1] = y;
        }
        total++;
    }
    if (t == 0) hdr[3] = total;
}

// Track-length thinning of Frame::track_keypoints (frame.cpp:134-161): the surviving keypoints, longest track first (the ORDER comes
// from the host: the reference sorts with std::sort, whose order among equal lengths is the library's), pass through a
// PoissonDiskFilter<2> in the next image -- a keypoint is kept if no resident point is closer than the radius AND its track is not
// TT_TRASH, and a kept keypoint becomes resident.  Sequential by nature; one wavefront walks the list, the 25 visited cells of a
// candidate on 25 lanes, one ballot per candidate (the walk of poisson_filter_kernel's ordered mode).
__global__ __launch_bounds__(64) void thin_tracks_kernel(const double *__restrict__ xy, const int32_t *__restrict__ order, int n_order,
                                                        const uint8_t *__restrict__ trash, int w, int h, double radius, uint8_t *__restrict__ keep,
                                                        int32_t *__restrict__ flag) {
    __shared__ int grid[RDVIO_SEL_PGRID_MAX];
    __shared__ __attribute__((aligned(16))) double pts[2 * RDVIO_SEL_PTS_MAX];
    const int t = threadIdx.x;
    const double cellp = radius / sqrt(2.0), r2 = radius * radius;
    constexpr int M = 3;
    const int gx = (int)floor((double)(w - 1) / cellp) + 1 + 2 * M, gy = (int)floor((double)(h - 1) / cellp) + 1 + 2 * M;
    if ((long long)gx * gy > RDVIO_SEL_PGRID_MAX || n_order > RDVIO_SEL_PTS_MAX) {
        if (t == 0) *flag = 1;   // beyond the LDS capacities: the caller takes the host road
        return;
    }
    for (int i = t; i < gx * gy; i += 64) grid[i] = -1;
    __syncthreads();
    int npts = 0;
    for (int k = 0; k < n_order; ++k) {
        const int idx = order[k];
        const double x = xy[2 * idx], y = xy[2 * idx + 1];
        const int cx = (int)floor(x / cellp), cy = (int)floor(y / cellp);
        bool conflict = false;
        if (t < 25) {
            // the reference's walk over [cx-2, cx+2] x [cy-2, cy+2]: it steps BEFORE it looks, so it never visits (cx-2, cy-2) and ends on (cx-2, cy+3)
            const int pos = t + 1;
            const int ax = cx - 2 + pos % 5 + M, ay = cy - 2 + pos / 5 + M;
            if (ax >= 0 && ay >= 0 && ax < gx && ay < gy) {
                const int p = grid[ay * gx + ax];
                if (p >= 0) {
                    const double dx = x - pts[2 * p], dy = y - pts[2 * p + 1];
                    conflict = dx * dx + dy * dy < r2;
                }
            }
        }
        const bool kept = __ballot(conflict) == 0ull && !trash[idx];
        if (kept && t == 0) {
            const int ax = cx + M, ay = cy + M;
            if (ax >= 0 && ay >= 0 && ax < gx && ay < gy) grid[ay * gx + ax] = npts;
            pts[2 * npts] = x;
            pts[2 * npts + 1] = y;
        }
        if (kept) npts++;
        if (t == 0) keep[k] = kept ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (t == 0) *flag = 0;
}

}  // namespace

extern "C" int rdvio_hip_thin_tracks(rdvio_hip_ctx *ctx, int width, int height, double radius, int n_points, const double *xy, int n_order,
                                     const int32_t *order, const uint8_t *trash, uint8_t *keep) {
    if (!ctx || width <= 0 || height <= 0 || !(radius > 0.0) || n_points < 0 || n_order < 0 || (n_order > 0 && (!xy || !order || !trash || !keep)))
        return RDVIO_ERR_INVALID;
    if (n_order == 0) return RDVIO_OK;
    for (int k = 0; k < n_order; ++k)   // shapes are checked on the host: the kernel never indexes out of bounds
        if (order[k] < 0 || order[k] >= n_points) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "thinning order entry %d outside the %d points", order[k], n_points);
    const size_t o_xy = 0, o_ord = (size_t)n_points * 16, o_tr = o_ord + (size_t)n_order * 4, o_keep = (o_tr + (size_t)n_points + 15) & ~(size_t)15,
                 o_flag = (o_keep + (size_t)n_order + 15) & ~(size_t)15, total = o_flag + 16;
    if (total > ctx->thin_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "thinning of %d points exceeds the context's capacity", n_points);
    hipStream_t st = ctx->stream;
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));   // the pinned blob may still be in flight
    uint8_t *hb = (uint8_t *)ctx->thin_host, *db = (uint8_t *)ctx->thin_dev;
    memcpy(hb + o_xy, xy, (size_t)n_points * 16);
    memcpy(hb + o_ord, order, (size_t)n_order * 4);
    memcpy(hb + o_tr, trash, (size_t)n_points);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(db, hb, o_keep, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(thin_tracks_kernel, dim3(1), dim3(64), 0, st, (const double *)(db + o_xy), (const int32_t *)(db + o_ord), n_order,
                       (const uint8_t *)(db + o_tr), width, height, radius, db + o_keep, (int32_t *)(db + o_flag));
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(hb + o_keep, db + o_keep, total - o_keep, hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    if (*(const int32_t *)(hb + o_flag) != 0) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "thinning grid beyond the kernel's LDS capacity");
    memcpy(keep, hb + o_keep, (size_t)n_order);
    return RDVIO_OK;
}

int rdvio_launch_select(rdvio_hip_ctx *ctx, int slot, int max_corners, double gftt_min_dist, double poisson_radius, int n_existing) {
    ImageSlot &S = ctx->slots[slot];
    const int cell = (int)lrint(gftt_min_dist);
    const float md2 = (float)(gftt_min_dist * gftt_min_dist);
    hipLaunchKernelGGL(gftt_select_kernel, dim3(1), dim3(ST), 0, ctx->stream, ctx->harris_cand, ctx->harris_scalars, ctx->harris_cand_cap, S.w,
                       S.h, max_corners, cell, md2, ctx->sel_corners, ctx->sel_hdr);
    hipLaunchKernelGGL(poisson_filter_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->sel_existing, n_existing, ctx->sel_corners, S.w, S.h,
                       poisson_radius, gftt_min_dist >= 1 ? gftt_min_dist * gftt_min_dist : 0.0, ctx->sel_new, ctx->sel_hdr);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}
