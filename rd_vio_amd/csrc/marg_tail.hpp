// Tail of CeresMarginalizationFactor::marginalize(0) on gfx950 (FP64), run by the marginalisation kernel after
// the shared linearisation + normal-equation assembly (solver_kernels.hip).
//
// Reference: /root/reference/src/rdvio_estimation/include/rdvio/estimation/ceres/marginalization_factor.h:74-475
// (called from Map::marginalize_frame, /root/reference/src/rdvio_map/src/map.cpp:50-62):
//   (i)-(iii) J^T J, J^T r of the prior, the victim's preintegration factor and the victim-observed tracks' reprojection
//             factors (no robust loss)                           -> evaluate<true> + build_normal_equations
//   (iv)  landmark Schur:  Lambda -= h_i^T h_j / m,  eta -= h_i^T v / m            (:382-398)
//   (v)   frame Schur with a plain 15x15 inverse                                   (:400-438)
//   (vi)  new sqrt prior from the symmetric eigendecomposition, eigenvalues <= 1e-8 clamped to 0 (:440-474)
//
// The reference permutes the victim frame to the end before (v); with the victim being frame 0 the retained block is
// simply rows/columns 15.. of the natural ordering, so no permutation is materialised here.
//
// (vi) MI355X-first: the eigendecomposition's only observable effect is Lambda+ = S^T S (eigenvalues <= 1e-8
// removed) and eta+ = S^T f.  Structurally-zero rows (frames that carry no information in this prior, e.g. the
// velocity/bias rows of the newest frame) are exact zero eigenvalues and are dropped; on the rest a diagonally-PIVOTED
// Cholesky peels off rank-1 terms l l^T until the largest remaining diagonal entry is <= 1e-8: S = [l_1 .. l_r]^T
// reproduces Lambda up to a remainder below the reference's own clamp threshold, f follows from the same
// elimination applied to eta.  For windows up to RDVIO_LDS_CHOL_MAX_FRAMES frames the packed lower triangle lives in
// LDS (two barriers per pivot, no global round trips in the dependent chain).  Larger windows first try a plain
// blocked Cholesky (exact sqrt factor when the information is positive definite beyond the threshold) and fall back
// to the pivoted factorisation in global memory.  force_eigen runs the reference's literal recipe (parallel-ordered
// two-sided Jacobi eigensolver + clamp).  info[0] reports the path (1 plain Cholesky, 2 pivoted Cholesky, 0 eigen).
#pragma once
#include "block_linalg.hpp"
#include "solver_ws.hpp"

// Parallel-ordered (round-robin) two-sided Jacobi on the symmetric R x R matrix A (row-major, global).
// V accumulates the rotations (columns = eigenvectors).  cs: scratch for 2 * (Rp/2) rotation parameters.
template <int T>
__device__ __attribute__((noinline)) void jacobi_eigen(LdsShared<T> &sh, double *A, double *V, double *cs, int R) {
    const int t = threadIdx.x;
    int phase = 0;
    const int Rp = (R + 1) & ~1, half = Rp / 2;
    for (int i = t; i < R * R; i += T) V[i] = ((i / R) == (i % R)) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = t; i < R * R; i += T) {
            const int r = i / R, c = i - r * R;
            const double v = A[i];
            if (r == c) dg += v * v;
            else off += v * v;
        }
        {
            double v2[2] = {off, dg};
            block_sum_n<T, 2>(sh, v2, phase);
            off = v2[0];
            dg = v2[1];
        }
        if (off <= 1e-60 || off <= 1e-30 * dg) break;
        for (int step = 0; step < Rp - 1; ++step) {
            // circle method: player Rp-1 fixed, the others rotate
            for (int k = t; k < half; k += T) {
                int p = (k == 0) ? Rp - 1 : (step + k) % (Rp - 1);
                int q = (step + Rp - 1 - k) % (Rp - 1);
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                double c = 1.0, s = 0.0;
                if (q < R) {
                    const double apq = A[(size_t)p * R + q];
                    if (apq != 0.0) {
                        const double app = A[(size_t)p * R + p], aqq = A[(size_t)q * R + q];
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(tt * tt + 1.0);
                        s = tt * c;
                    }
                }
                cs[4 * k] = c; cs[4 * k + 1] = s; cs[4 * k + 2] = (double)p; cs[4 * k + 3] = (double)q;
            }
            __syncthreads();
            // column pass: A <- A J, V <- V J
            for (int o = t; o < half * R; o += T) {
                const int k = o / R, i = o - k * R;
                const double c = cs[4 * k], s = cs[4 * k + 1];
                const int p = (int)cs[4 * k + 2], q = (int)cs[4 * k + 3];
                if (q >= R || s == 0.0) continue;
                const double aip = A[(size_t)i * R + p], aiq = A[(size_t)i * R + q];
                A[(size_t)i * R + p] = c * aip - s * aiq;
                A[(size_t)i * R + q] = s * aip + c * aiq;
                const double vip = V[(size_t)i * R + p], viq = V[(size_t)i * R + q];
                V[(size_t)i * R + p] = c * vip - s * viq;
                V[(size_t)i * R + q] = s * vip + c * viq;
            }
            __syncthreads();
            // row pass: A <- J^T A
            for (int o = t; o < half * R; o += T) {
                const int k = o / R, j = o - k * R;
                const double c = cs[4 * k], s = cs[4 * k + 1];
                const int p = (int)cs[4 * k + 2], q = (int)cs[4 * k + 3];
                if (q >= R || s == 0.0) continue;
                const double apj = A[(size_t)p * R + j], aqj = A[(size_t)q * R + j];
                A[(size_t)p * R + j] = c * apj - s * aqj;
                A[(size_t)q * R + j] = s * apj + c * aqj;
            }
            __syncthreads();
        }
    }
}

// Diagonally pivoted Cholesky of the symmetric PSD matrix A (n x n, row-major, GLOBAL memory, destroyed): writes
// factor column j (length n, zero on rows pivoted earlier) to Lc[j * n ..] and f[j]; eta (length n) is eliminated
// alongside.  Stops when the largest remaining diagonal entry is <= tol.  Returns the numerical rank.
template <int T>
__device__ __attribute__((noinline)) int pivoted_cholesky(LdsShared<T> &sh, int &phase, double *A, double *eta, double *Lc, double *fv, double *lcol,
                        int *done, int n, double tol) {
    const int t = threadIdx.x;
    for (int i = t; i < n; i += T) done[i] = 0;
    __syncthreads();
    int r = 0;
    for (int j = 0; j < n; ++j) {
        // arg max of the remaining diagonal (ties: lowest index) -- value reduction, then index reduction
        double best = -1.0;
        for (int i = t; i < n; i += T)
            if (!done[i]) best = fmax(best, A[(size_t)i * n + i]);
        best = block_max(sh, best, phase);
        if (!(best > tol)) break;
        double cand = 1e300;
        for (int i = t; i < n; i += T)
            if (!done[i] && A[(size_t)i * n + i] == best) cand = fmin(cand, (double)i);
        const int p = (int)(-block_max(sh, -cand, phase));
        const double d = sqrt(best);
        for (int i = t; i < n; i += T) {
            const double l = done[i] ? 0.0 : ((i == p) ? d : A[(size_t)i * n + p] / d);
            lcol[i] = l;
            Lc[(size_t)j * n + i] = l;
        }
        const double fj = eta[p] / d;
        __syncthreads();
        if (t == 0) {
            fv[j] = fj;
            done[p] = 1;
        }
        for (int i = t; i < n; i += T)
            if (i != p && lcol[i] != 0.0) eta[i] -= lcol[i] * fj;
        for (int o = t; o < n * n; o += T) {
            const int i = o / n, k = o - i * n;
            const double li = lcol[i], lk = lcol[k];
            if (li != 0.0 && lk != 0.0) A[o] -= li * lk;
        }
        __syncthreads();
        ++r;
    }
    return r;
}

DM int wave_min_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

// The same factorisation with the matrix LDS-resident: packed lower triangle Ap (entry (i, k), k <= i, at tri(i) + k),
// remaining diagonal dg (-1 marks an eliminated row), current column lcol, right-hand side eta -- all in LDS.
// Every wave finds the pivot redundantly (no block reduction); a step costs two barriers.  Each thread owns a fixed
// set of packed entries whose (i, k) coordinates are decoded once.  Factor column j goes straight to row j of S_out
// (scattered through nz to the retained rows' original positions), f[j] to f_out.
template <int T>
__device__ __attribute__((noinline)) int pivoted_cholesky_lds(lds_double *Ap, lds_double *dg, lds_double *lcol, lds_double *eta,
                                                              const double *__restrict__ src, const double *__restrict__ er,
                                                              const int32_t *__restrict__ nz, int n, int R, double tol,
                                                              double *__restrict__ S_out, double *__restrict__ f_out) {
    constexpr int NMAXE = 15 * RDVIO_LDS_CHOL_MAX_FRAMES;
    constexpr int EMAX = (NMAXE * (NMAXE + 1) / 2 + T - 1) / T;
    const int t = threadIdx.x, lane = t & 63;
    const int ne = tri(n);
    unsigned ik[EMAX];
#pragma unroll
    for (int u = 0; u < EMAX; ++u) {
        const int e = t + T * u;
        ik[u] = 0xffffffffu;
        if (e < ne) {
            int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while (tri(i + 1) <= e) ++i;
            while (tri(i) > e) --i;
            const int k = e - tri(i);
            ik[u] = ((unsigned)i << 16) | (unsigned)k;
            Ap[e] = src[(size_t)nz[i] * R + nz[k]];
        }
    }
    for (int i = t; i < n; i += T) {
        dg[i] = src[(size_t)nz[i] * R + nz[i]];
        eta[i] = er[nz[i]];
    }
    __syncthreads();
    int r = 0;
    for (int j = 0; j < n; ++j) {
        // pivot = arg max of the remaining diagonal, lowest index on ties
        double best = -1.0;
        int bi = 0x7fffffff;
        for (int i = lane; i < n; i += 64) {
            const double v = dg[i];
            if (v > best) { best = v; bi = i; }
        }
        const double m = wave_max(best);
        const int p = wave_min_i(best == m ? bi : 0x7fffffff);
        if (!(m > tol)) break;
        const double d = sqrt(m);
        for (int i = t; i < n; i += T) {
            double l = 0.0;
            if (i == p) l = d;
            else if (dg[i] >= 0.0) l = ((i > p) ? Ap[tri(i) + p] : Ap[tri(p) + i]) / d;
            lcol[i] = l;
            S_out[(size_t)j * R + nz[i]] = l;
            const double fj = eta[p] / d;
            if (i == p) f_out[j] = fj;
            else if (l != 0.0) eta[i] -= l * fj;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < EMAX; ++u) {
            if (ik[u] != 0xffffffffu) {
                const int i = (int)(ik[u] >> 16), k = (int)(ik[u] & 0xffffu);
                const double li = lcol[i], lk = lcol[k];
                if (li != 0.0 && lk != 0.0) {
                    const int e = t + T * u;
                    const double v = Ap[e] - li * lk;
                    Ap[e] = v;
                    if (i == k) dg[i] = (i == p) ? -1.0 : v;
                }
            }
        }
        __syncthreads();
        ++r;
    }
    return r;
}

// lds: LDS scratch of at least tri(R) + 3 R + 450 doubles when w.lds_chol, otherwise only the first 450 are used.
template <int T>
__device__ __attribute__((noinline)) void marginalize_tail(LdsWs &w, LdsShared<T> &sh, int &phase, lds_double *lds) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, nw = T / 64;
    const int N = w.N, R = N - 15, nl = w.nl, NA = 6 * w.nfree, NAs = NA + 2;
    lds_double *sM = lds;  // 15 x 30 augmented block for the victim's inverse
    // ---- (iv) landmark Schur: [C | Cg] = A^T diag(1/m) [A | g]; a landmark is skipped if 1/m is not finite (:384-386)
    for (int l = t; l < nl; l += T) {
        const double inv = 1.0 / w.lm_m[l];
        w.lm_w[l] = (w.lfree[l] && isfinite(inv)) ? inv : 0.0;
    }
    __syncthreads();
    if (NA > 0 && nl > 0) {
        // operand staged in the (still idle) LDS buffer with one batch of coalesced loads, as in the solver's schur_reduce
        constexpr size_t LDS_DOUBLES = (size_t)(15 * RDVIO_LDS_CHOL_MAX_FRAMES) * (15 * RDVIO_LDS_CHOL_MAX_FRAMES + 1) / 2 + 225 * RDVIO_LDS_CHOL_MAX_FRAMES;
        if (w.lds_chol && (size_t)nl * NAs + nl <= LDS_DOUBLES) {
            lds_double *As = lds, *ws = lds + nl * NAs;
            stage_to_lds<T, 16>(As, w.A, nl * NAs);
            for (int l = t; l < nl; l += T) ws[l] = w.lm_w[l];
            __syncthreads();
            block_gemm_tn_lds<T>(w.Cm, NAs, As, NAs, As, NAs, ws, true, NA, NA + 1, nl, true);
        } else {
            // (operand too large for LDS: staged a chunk of landmarks at a time; up to 8 x 5 = 40 tiles -- 32 free frames -- else the unstaged walk)
            if (!block_gemm_tn_chunked<T>(w.Cm, NAs, w.A, NAs, w.lm_w, NA, NA + 1, nl, lds, LDS_DOUBLES))
                block_gemm_tn<T>(w.Cm, NAs, w.A, NAs, w.A, NAs, w.lm_w, NA, NA + 1, nl, true);
        }
    }
    __syncthreads();
    // Hs = H - C (full, natural frame order) into Sm; gs = g - Cg into yp
    // (eight entries per trip with all their loads issued first: the stores to Sm may alias H / Cm as far as the
    // compiler knows, so a one-entry loop is a chain of N^2 / T dependent round trips)
    for (int o0 = t; o0 < N * N; o0 += 8 * T) {
        double hv[8], cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + u * T;
            hv[u] = 0.0;
            cv[u] = 0.0;
            if (o < N * N) {
                const int i = o / N, j = o - i * N, fi = i / 15, a = i - 15 * fi, fj = j / 15, b = j - 15 * fj;
                hv[u] = w.H[o];
                if (a < 6 && b < 6 && nl > 0) {
                    const int ri = 6 * fi + a, cj = 6 * fj + b;
                    cv[u] = (ri >= cj) ? w.Cm[(size_t)ri * NAs + cj] : w.Cm[(size_t)cj * NAs + ri];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int o = o0 + u * T;
            if (o < N * N) w.Sm[o] = hv[u] - cv[u];
        }
    }
    for (int i = t; i < N; i += T) {
        const int fi = i / 15, a = i - 15 * fi;
        double v = w.g[i];
        if (a < 6 && nl > 0) v -= w.Cm[(size_t)(6 * fi + a) * NAs + NA];
        w.yp[i] = v;
    }
    __syncthreads();
    // ---- (v) frame Schur of the victim (frame 0): 15x15 inverse (Gauss-Jordan, partial pivoting) in LDS by the first wave
    for (int i = t; i < 15 * 30; i += T) {
        const int r = i / 30, c = i - r * 30;
        sM[i] = (c < 15) ? w.Sm[(size_t)r * N + c] : ((c - 15 == r) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (t < 64) {
        // in registers: lane r holds row r of [M | I]; pivot rows and multipliers travel through v_readlane, so the 15
        // dependent elimination steps never wait for LDS (the LDS version of this loop was ~2 us per column).  Rows are
        // not swapped physically: the lane that supplied the pivot of column c ends up holding row c of the inverse.
        double a[30];
#pragma unroll
        for (int c = 0; c < 30; ++c) a[c] = (lane < 15) ? sM[lane * 30 + c] : 0.0;
        bool used = lane >= 15;
        int mycol = -1;
#pragma unroll
        for (int c = 0; c < 15; ++c) {
            const double v = used ? -1.0 : fabs(a[c]);
            const double m = wave_max(v);
            const unsigned long long tie = __ballot(v == m && !used);
            const int piv = __builtin_amdgcn_readfirstlane(tie ? (int)__builtin_ctzll(tie) : 0);
            const double inv_d = 1.0 / readlane_d(a[c], piv);  // one divide per column; the row is scaled by the reciprocal
            const double f = a[c];
#pragma unroll
            for (int k = 0; k < 30; ++k) {
                const double pr = readlane_d(a[k], piv) * inv_d;
                a[k] = (lane == piv) ? pr : a[k] - f * pr;
            }
            if (lane == piv) { used = true; mycol = c; }
        }
        if (mycol >= 0) {
#pragma unroll
            for (int k = 0; k < 15; ++k) sM[mycol * 30 + 15 + k] = a[15 + k];
        }
    }
    __syncthreads();
    // Tm = H_rm * Minv (R x 15)
    for (int o = t; o < R * 15; o += T) {
        const int i = o / 15, b = o - 15 * i;
        double hv[15];
#pragma unroll
        for (int a = 0; a < 15; ++a) hv[a] = w.Sm[(size_t)(15 + i) * N + a];
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 15; ++a) acc += hv[a] * sM[a * 30 + 15 + b];
        w.m_Tm[o] = acc;
    }
    __syncthreads();
    // Lr = H_rr - Tm H_mr on the matrix cores (K = 15), er = eta_r - Tm eta_m
    {
        const int tn = (R + 15) / 16;
        for (int tile = wave; tile < tn * tn; tile += nw) {
            const int bi = tile / tn, bj = tile - bi * tn;
            const double4_t acc = mfma_tile_f64(w.m_Tm, 1, 15, w.Sm + 15, N, 1, nullptr, 15, 16 * bi, 16 * bj, R, R);
            const int col = 16 * bj + (lane & 15);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int row = 16 * bi + (lane >> 4) + 4 * r4;
                if (row < R && col < R) {
                    const double v = w.Sm[(size_t)(15 + row) * N + 15 + col] - acc[r4];
                    w.m_Lr[(size_t)row * R + col] = v;
                    w.Lambda_out[(size_t)row * R + col] = v;
                    w.S_out[(size_t)row * R + col] = 0.0;
                }
            }
        }
    }
    for (int i = t; i < R; i += T) {
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 15; ++a) acc += w.m_Tm[i * 15 + a] * w.yp[a];
        const double v = w.yp[15 + i] - acc;
        w.m_er[i] = v;
        w.eta_out[i] = v;
        w.f_out[i] = 0.0;
    }
    for (int i = t; i < (w.nfr - 1) * 16; i += T) w.lin_out[i] = w.x[16 + i];
    __syncthreads();
    // ---- (vi) sqrt factor.  Structurally-zero rows are dropped (ballot compaction by the first wave).
    if (wave == 0) {
        int n = 0;
        for (int base = 0; base < R; base += 64) {
            const int i = base + lane;
            const bool nzf = i < R && w.m_Lr[(size_t)i * R + i] != 0.0;
            const unsigned long long mask = __ballot(nzf);
            if (nzf) w.m_nz[n + __popcll(mask & ((1ull << lane) - 1ull))] = i;
            n += __popcll(mask);
        }
        if (lane == 0) sh.flag = n;
    }
    __syncthreads();
    const int Rn = sh.flag, Rb = (Rn + 14) / 15 * 15;
    __syncthreads();
    int path = 0;
    if (!w.marg_force_eigen && w.lds_chol) {
        // The information on the retained rows is normally positive definite: a plain blocked Cholesky (matrix cores, two
        // barriers per 15 pivots instead of two per pivot) whose pivots all stay above the reference's clamp threshold
        // gives the exact factor S = L^T, and f = L^-1 eta falls out of the factorisation as the extra row.  A pivot at
        // or below the threshold sends the (then semi-definite) matrix to the pivoted factorisation.
        // (57 retained rows in steady state -- 8 frame poses + the oldest frame's velocity and biases: 0.25 ms per
        // marginalisation with the blocked factorisation, 0.32 ms with the pivoted one; a first-of-session problem, whose
        // information is rank-deficient, pays 0.03 ms for the failed attempt.)
        int fast = Rn == 0;
        if (Rn > 0) {
            lds_double *Lp = lds + 450, *Dinv = Lp + tri(Rb + 1);
            __attribute__((address_space(3))) int *nzl = (__attribute__((address_space(3))) int *)lds;  // (the 15 x 30 block at the head of the buffer is free again: 900 ints >= Rn)
            for (int i = t; i < Rn; i += T) nzl[i] = w.m_nz[i];
            __syncthreads();
            const int W1 = Rb + 1;
#pragma unroll 4
            for (int o = t; o < W1 * W1; o += T) {
                const int i = o / W1, k = o - i * W1;
                if (k > i) continue;
                double v;
                if (i == Rb) v = (k < Rn) ? w.m_er[nzl[k]] : 0.0;  // right-hand-side row
                else if (i < Rn) v = w.m_Lr[(size_t)nzl[i] * R + nzl[k]];
                else v = (i == k) ? 1.0 : 0.0;  // identity padding up to a multiple of 15
                Lp[tri(i) + k] = v;
            }
            __syncthreads();
            fast = cholesky_lds<T>(sh, Lp, Dinv, Rb, 1.0e-8, false);
            if (fast) {
#pragma unroll 4
                for (int o = t; o < Rn * Rn; o += T) {  // S[i][j] = L[j][i], j >= i
                    const int j = o / Rn, i = o - j * Rn;
                    if (i <= j) w.S_out[(size_t)nzl[i] * R + nzl[j]] = Lp[tri(j) + i];
                }
                for (int i = t; i < Rn; i += T) w.f_out[nzl[i]] = Lp[tri(Rb) + i];
            }
            __syncthreads();
        }
        path = 1;
        if (!fast) {
            lds_double *Ap = lds + 450, *dg = Ap + tri(Rn), *lcol = dg + Rn, *eta = lcol + Rn;
            (void)pivoted_cholesky_lds<T>(Ap, dg, lcol, eta, w.m_Lr, w.m_er, w.m_nz, Rn, R, 1.0e-8, w.S_out, w.f_out);
            path = 2;
        }
    } else if (!w.marg_force_eigen) {
        // large window: plain blocked Cholesky when the information is positive definite beyond the threshold
        // (shifted factorisation as the test), else the pivoted factorisation in global memory
        for (int o = t; o < Rb * Rb; o += T) {
            const int i = o / Rb, j = o - i * Rb;
            double v = (i == j) ? 1.0 : 0.0;
            if (i < Rn && j < Rn) v = w.m_Lr[(size_t)w.m_nz[i] * R + w.m_nz[j]] - ((i == j) ? 1.0e-8 : 0.0);
            w.m_Wk[o] = v;
        }
        __syncthreads();
        int fast = (Rn == 0) ? 1 : cholesky_blocked(sh, w.m_Wk, Rb);
        if (fast) {
            for (int o = t; o < Rb * Rb; o += T) {
                const int i = o / Rb, j = o - i * Rb;
                double v = (i == j) ? 1.0 : 0.0;
                if (i < Rn && j < Rn) v = w.m_Lr[(size_t)w.m_nz[i] * R + w.m_nz[j]];
                w.m_Wk[o] = v;
            }
            __syncthreads();
            fast = (Rn == 0) ? 1 : cholesky_blocked(sh, w.m_Wk, Rb);
        }
        if (fast) {
            // S = L^T on the retained rows/cols;  f = L^-1 eta (forward substitution)
            for (int o = t; o < Rn * Rn; o += T) {
                const int i = o / Rn, j = o - i * Rn;  // S[i][j] = L[j][i], j >= i
                if (j >= i) w.S_out[(size_t)w.m_nz[i] * R + w.m_nz[j]] = w.m_Wk[(size_t)j * Rb + i];
            }
            for (int i = t; i < Rb; i += T) w.m_yv[i] = (i < Rn) ? w.m_er[w.m_nz[i]] : 0.0;
            __syncthreads();
            cholesky_solve(sh, w.m_Wk, Rb, w.m_yv, true, false);
            for (int i = t; i < Rn; i += T) w.f_out[w.m_nz[i]] = w.m_yv[i];
            path = 1;
        } else {
            for (int o = t; o < Rn * Rn; o += T) w.m_Wk[o] = w.m_Lr[(size_t)w.m_nz[o / Rn] * R + w.m_nz[o % Rn]];
            for (int i = t; i < Rn; i += T) w.m_yv[i] = w.m_er[w.m_nz[i]];
            __syncthreads();
            const int rank = pivoted_cholesky(sh, phase, w.m_Wk, w.m_yv, w.m_V, w.m_cs, w.m_Tm, w.m_nz + R + 1, Rn, 1.0e-8);
            __syncthreads();
            for (int o = t; o < rank * Rn; o += T) {
                const int j = o / Rn, i = o - j * Rn;
                w.S_out[(size_t)j * R + w.m_nz[i]] = w.m_V[(size_t)j * Rn + i];
            }
            for (int j = t; j < rank; j += T) w.f_out[j] = w.m_cs[j];
            path = 2;
        }
    } else {
        // literal restatement: eigendecomposition, lambda+ = lambda > 1e-8 ? lambda : 0 (:441-458)
        for (int i = t; i < R * R; i += T) w.m_Wk[i] = w.m_Lr[i];
        __syncthreads();
        jacobi_eigen(sh, w.m_Wk, w.m_V, w.m_cs, R);
        for (int o = t; o < R * R; o += T) {
            const int k = o / R, j = o - k * R;  // S[k][j] = sqrt(lambda_k+) V[j][k]
            const double lam = w.m_Wk[(size_t)k * R + k];
            w.S_out[o] = (lam > 1.0e-8) ? sqrt(lam) * w.m_V[(size_t)j * R + k] : 0.0;
        }
        for (int k = t; k < R; k += T) {
            const double lam = w.m_Wk[(size_t)k * R + k];
            double acc = 0.0;
            for (int j = 0; j < R; ++j) acc += w.m_V[(size_t)j * R + k] * w.m_er[j];
            w.f_out[k] = (lam > 1.0e-8) ? sqrt(1.0 / lam) * acc : 0.0;
        }
    }
    if (t == 0) {
        w.m_info[0] = (double)path;
        w.m_info[1] = (double)Rn;
    }
}
