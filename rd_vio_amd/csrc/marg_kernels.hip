// CeresMarginalizationFactor::marginalize(0) on gfx950 (FP64), one persistent workgroup.
//
// Reference: /root/reference/src/rdvio_estimation/include/rdvio/estimation/ceres/marginalization_factor.h:74-475
// (called from Map::marginalize_frame, /root/reference/src/rdvio_map/src/map.cpp:50-62):
//   (i)   J^T J, J^T r of the current prior, victim frame permuted last          (:95-161)
//   (ii)  the preintegration factor between frames 0 and 1                          (:163-231)
//   (iii) every reprojection factor of the victim-observed tracks, no robust loss  (:233-380)
//   (iv)  landmark Schur:  Lambda -= h_i^T h_j / m,  eta -= h_i^T v / m            (:382-398)
//   (v)   frame Schur with a plain 15x15 inverse                                   (:400-438)
//   (vi)  new sqrt prior from the symmetric eigendecomposition, eigenvalues <= 1e-8 clamped to 0 (:440-474)
//
// (i)-(v) use the same output-stationary, fixed-order assembly as the solver (no atomics).
// (vi) MI355X-first: the eigendecomposition's only observable effect is Lambda+ = S^T S (eigenvalues <= 1e-8
// removed) and eta+ = S^T f.  Structurally-zero rows (frames that carry no information in this prior, e.g.
// the velocity/bias rows of frames >= 2) are exact zero eigenvalues and are dropped; if the remaining block is
// positive definite beyond the 1e-8 threshold (shifted Cholesky succeeds) NO eigenvalue is clamped and
// S = L^T, f = L^-1 eta is an exact sqrt factor of the same (Lambda, eta) -- an O(R^3/3) blocked Cholesky
// instead of ~1000 dependent Jacobi steps.  Otherwise (the usual case: a few directions such as parts of the
// accelerometer bias are numerically unobservable) a diagonally-PIVOTED Cholesky peels off rank-1 terms
// l l^T until the largest remaining diagonal entry is <= 1e-8: S = [l_1 .. l_r]^T reproduces Lambda up to a
// remainder below the reference's own clamp threshold, f follows from the same elimination applied to eta.
// force_eigen runs the reference's literal recipe (parallel-ordered two-sided Jacobi eigensolver + clamp);
// `info[0]` reports which path ran (1 plain Cholesky, 2 pivoted Cholesky, 0 eigendecomposition).
#include "ctx.hpp"
#include "factors.hpp"
#include "marg_ws.hpp"
#include "block_linalg.hpp"

namespace {

constexpr int T = RDVIO_MARG_THREADS;
using Shared = BlockShared<T>;

DM double prior_E(const MargWs &w, int i, int a, int b) {
    if (a < 3 && b < 3) return w.Jri[9 * i + 3 * a + b];
    return a == b ? 1.0 : 0.0;
}

// Parallel-ordered (round-robin) two-sided Jacobi on the symmetric R x R matrix A (row-major, global).
// V accumulates the rotations (columns = eigenvectors).  cs: scratch for 2 * (Rp/2) rotation parameters.
DM void jacobi_eigen(Shared &sh, double *A, double *V, double *cs, int R) {
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

// Diagonally pivoted Cholesky of the symmetric PSD matrix A (n x n, row-major, destroyed): writes factor column j
// (length n, zero on rows pivoted earlier) to Lc[j * n ..] and f[j]; eta (length n) is eliminated alongside.
// Stops when the largest remaining diagonal entry is <= tol.  Returns the numerical rank.
DM int pivoted_cholesky(Shared &sh, int &phase, double *A, double *eta, double *Lc, double *fv, double *lcol, int *done,
                        int n, double tol) {
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

__global__ __launch_bounds__(T) void marginalize_kernel(MargWs w) {
    __shared__ Shared sh;
    __shared__ double sM[15 * 30];
    __shared__ int s_cnt;
    const int t = threadIdx.x;
    const int nfm = w.nfm, N = 15 * nfm, R = N - 15, NA = 6 * nfm, D = w.D;
    const double *W = w.extr + 14;
    int phase = 0;

    // ---- (i) current prior at the current states of its frames: e, Jr^-1, Lambda = S^T S, le = S^T (S e + f)
    if (w.np > 0) {
        for (int i = t; i < w.np; i += T) {
            M3 Jri;
            marginalization_frame_error(w.states + 16 * w.prior_frames[i], w.lin + 16 * i, w.e_m + 15 * i, &Jri);
            for (int q = 0; q < 9; ++q) w.Jri[9 * i + q] = Jri.m[q];
        }
        for (int o = t; o < D * D; o += T) {
            const int a = o / D, b = o - a * D;
            double acc = 0.0;
            for (int q = 0; q < D; ++q) acc += w.S[(size_t)q * D + a] * w.S[(size_t)q * D + b];
            w.Lam[o] = acc;
        }
        __syncthreads();
        for (int row = t; row < D; row += T) {  // r = S e + f
            double acc = 0.0;
            for (int c = 0; c < D; ++c) acc += w.S[(size_t)row * D + c] * w.e_m[c];
            w.r_m[row] = acc + w.f[row];
        }
        __syncthreads();
        for (int a = t; a < D; a += T) {  // le = S^T r
            double acc = 0.0;
            for (int q = 0; q < D; ++q) acc += w.S[(size_t)q * D + a] * w.r_m[q];
            w.le[a] = acc;
        }
    }
    // ---- (ii) preintegration factor (frames 0, 1); bias linearisation = live members => dbg = dba = 0
    if (w.has_pre) {
        for (int i = t; i < 450; i += T) w.G[i] = 0.0;
        __syncthreads();
        if (t == 0)
            preintegration_unwhitened<true>(w.states, w.states + 16, w.preint, w.states + ST_BG, w.extr, w.e_p, w.G, w.G + 225);
        __syncthreads();
        const double *Sic = w.preint + PRE_SIC;
        for (int o = t; o < 15; o += T) {
            double acc = 0.0;
            for (int q = 0; q < 15; ++q) acc += Sic[o * 15 + q] * w.e_p[q];
            w.r_p[o] = acc;
        }
        for (int o = t; o < 450; o += T) {
            const int which = o / 225, rc = o - 225 * which, row = rc / 15, col = rc - 15 * row;
            double acc = 0.0;
            for (int q = 0; q < 15; ++q) acc += Sic[row * 15 + q] * w.G[225 * which + q * 15 + col];
            w.Jp[o] = acc;
        }
    }
    // ---- (iii) reprojection factors, NO robust loss (:233-380)
    for (int k = t; k < w.nf; k += T) {
        double r[2], Jt[12], Jr[12], Jd[2];
        const int l = w.lm[k];
        reprojection_factor<true>(w.states + 16 * w.tgt[k], w.states + 16 * w.ref[k], w.tangent + 9 * (size_t)k,
                                  w.z_ref + 3 * (size_t)l, w.inv_depth[l], w.extr, W, r, Jt, Jr, Jd);
        w.r_f[2 * (size_t)k] = r[0];
        w.r_f[2 * (size_t)k + 1] = r[1];
        for (int i = 0; i < 12; ++i) { w.Jt[12 * (size_t)k + i] = Jt[i]; w.Jr[12 * (size_t)k + i] = Jr[i]; }
        w.Jd[2 * (size_t)k] = Jd[0];
        w.Jd[2 * (size_t)k + 1] = Jd[1];
    }
    for (int i = t; i < N * N; i += T) w.H[i] = 0.0;
    for (int i = t; i < N; i += T) w.eta[i] = 0.0;
    __syncthreads();
    // landmark scalars and coupling rows (frame slots are the PERMUTED indices: victim last)
    for (int l = t; l < w.nl; l += T) {
        double *Arow = w.A + (size_t)l * NA;
        for (int i = 0; i < NA; ++i) Arow[i] = 0.0;
        double m = 0.0, gl = 0.0;
        for (int k = w.lm_first[l]; k < w.lm_first[l] + w.lm_count[l]; ++k) {
            const double d0 = w.Jd[2 * (size_t)k], d1 = w.Jd[2 * (size_t)k + 1];
            m += d0 * d0 + d1 * d1;
            gl += d0 * w.r_f[2 * (size_t)k] + d1 * w.r_f[2 * (size_t)k + 1];
            const int ct = w.fidx[w.tgt[k]], cr = w.fidx[w.ref[k]];
            for (int a = 0; a < 6; ++a) {
                Arow[6 * ct + a] += d0 * w.Jt[12 * (size_t)k + a] + d1 * w.Jt[12 * (size_t)k + 6 + a];
                Arow[6 * cr + a] += d0 * w.Jr[12 * (size_t)k + a] + d1 * w.Jr[12 * (size_t)k + 6 + a];
            }
        }
        w.lm_m[l] = m;
        w.lm_g[l] = gl;
        const double inv = 1.0 / m;
        w.lm_w[l] = (w.lm_count[l] > 0 && isfinite(inv)) ? inv : 0.0;  // skipped if 1/m is not finite (:384-386)
    }
    // reprojection J^T J blocks through the frame-pair factor lists
    for (int o = t; o < w.npairs * 36; o += T) {
        const int p = o / 36, ab = o - 36 * p, a = ab / 6, b = ab - 6 * a;
        const int fi = w.pair_fi[p], fj = w.pair_fj[p];
        double acc = 0.0;
        for (int it = w.pair_off[p]; it < w.pair_off[p + 1]; ++it) {
            const int item = w.pair_item[it], k = item >> 2, code = item & 3;
            const double *Jx = ((code & 1) ? w.Jr : w.Jt) + 12 * (size_t)k;
            const double *Jy = ((code & 2) ? w.Jr : w.Jt) + 12 * (size_t)k;
            acc += Jx[a] * Jy[b] + Jx[6 + a] * Jy[6 + b];
        }
        w.H[(size_t)(15 * fi + a) * N + 15 * fj + b] = acc;
        if (fi != fj) w.H[(size_t)(15 * fj + b) * N + 15 * fi + a] = acc;
    }
    for (int o = t; o < nfm * 6; o += T) {
        const int c = o / 6, a = o - 6 * c, p = w.diag_pair[c];
        double acc = 0.0;
        for (int it = w.pair_off[p]; it < w.pair_off[p + 1]; ++it) {
            const int item = w.pair_item[it], k = item >> 2, code = item & 3;
            const double *Jx = ((code & 1) ? w.Jr : w.Jt) + 12 * (size_t)k;
            acc += Jx[a] * w.r_f[2 * (size_t)k] + Jx[6 + a] * w.r_f[2 * (size_t)k + 1];
        }
        w.eta[15 * c + a] = acc;
    }
    __syncthreads();
    // preintegration blocks
    if (w.has_pre) {
        const int cs[2] = {w.fidx[0], w.fidx[1]};
        for (int o = t; o < 900; o += T) {
            const int xy = o / 225, ab = o - 225 * xy, x = xy >> 1, y = xy & 1, a = ab / 15, b = ab - 15 * a;
            const double *Jx = w.Jp + 225 * x, *Jy = w.Jp + 225 * y;
            double acc = 0.0;
            for (int q = 0; q < 15; ++q) acc += Jx[q * 15 + a] * Jy[q * 15 + b];
            w.H[(size_t)(15 * cs[x] + a) * N + 15 * cs[y] + b] += acc;
        }
        for (int o = t; o < 30; o += T) {
            const int x = o / 15, a = o - 15 * x;
            double acc = 0.0;
            for (int q = 0; q < 15; ++q) acc += w.Jp[225 * x + q * 15 + a] * w.r_p[q];
            w.eta[15 * cs[x] + a] += acc;
        }
        __syncthreads();
    }
    // prior: E^T Lambda E and E^T le
    if (w.np > 0) {
        for (int o = t; o < D * D; o += T) {
            const int ra = o / D, cb = o - ra * D, i = ra / 15, a = ra - 15 * i, j = cb / 15, b = cb - 15 * j;
            const int ci = w.fidx[w.prior_frames[i]], cj = w.fidx[w.prior_frames[j]];
            double acc;
            if (a >= 3 && b >= 3) {
                acc = w.Lam[(size_t)ra * D + cb];
            } else {
                acc = 0.0;
                const int a0 = a < 3 ? 0 : a, a1 = a < 3 ? 3 : a + 1, b0 = b < 3 ? 0 : b, b1 = b < 3 ? 3 : b + 1;
                for (int aa = a0; aa < a1; ++aa)
                    for (int bb = b0; bb < b1; ++bb)
                        acc += prior_E(w, i, aa, a) * w.Lam[(size_t)(15 * i + aa) * D + 15 * j + bb] * prior_E(w, j, bb, b);
            }
            w.H[(size_t)(15 * ci + a) * N + 15 * cj + b] += acc;
        }
        for (int o = t; o < D; o += T) {
            const int i = o / 15, a = o - 15 * i, ci = w.fidx[w.prior_frames[i]];
            double acc = 0.0;
            if (a < 3) {
                for (int aa = 0; aa < 3; ++aa) acc += prior_E(w, i, aa, a) * w.le[15 * i + aa];
            } else {
                acc = w.le[o];
            }
            w.eta[15 * ci + a] += acc;
        }
        __syncthreads();
    }
    // ---- (iv) landmark Schur
    for (int o = t; o < NA * NA; o += T) {
        const int ia = o / NA, jb = o - ia * NA;
        double acc = 0.0;
        for (int l = 0; l < w.nl; ++l) acc += w.A[(size_t)l * NA + ia] * w.lm_w[l] * w.A[(size_t)l * NA + jb];
        w.H[(size_t)(15 * (ia / 6) + ia % 6) * N + 15 * (jb / 6) + jb % 6] -= acc;
    }
    for (int ia = t; ia < NA; ia += T) {
        double acc = 0.0;
        for (int l = 0; l < w.nl; ++l) acc += w.A[(size_t)l * NA + ia] * w.lm_w[l] * w.lm_g[l];
        w.eta[15 * (ia / 6) + ia % 6] -= acc;
    }
    __syncthreads();
    // ---- (v) frame Schur: 15x15 inverse (Gauss-Jordan, partial pivoting) in LDS by the first wave
    for (int i = t; i < 15 * 30; i += T) {
        const int r = i / 30, c = i - r * 30;
        sM[i] = (c < 15) ? w.H[(size_t)(R + r) * N + R + c] : ((c - 15 == r) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (t < 64) {
        for (int c = 0; c < 15; ++c) {
            int piv = c;
            double best = fabs(sM[c * 30 + c]);
            for (int r = c + 1; r < 15; ++r) {
                const double v = fabs(sM[r * 30 + c]);
                if (v > best) { best = v; piv = r; }
            }
            __builtin_amdgcn_wave_barrier();
            if (piv != c && t < 30) {
                const double tmp = sM[c * 30 + t];
                sM[c * 30 + t] = sM[piv * 30 + t];
                sM[piv * 30 + t] = tmp;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const double d = sM[c * 30 + c];
            __builtin_amdgcn_wave_barrier();
            if (t < 30) sM[c * 30 + t] /= d;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const double pc = (t < 30) ? sM[c * 30 + t] : 0.0;
            for (int r = 0; r < 15; ++r) {
                if (r == c) continue;
                const double f = sM[r * 30 + c];
                __builtin_amdgcn_wave_barrier();
                if (t < 30) sM[r * 30 + t] -= f * pc;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    // Tm = H_rm * Minv (R x 15)
    for (int o = t; o < R * 15; o += T) {
        const int i = o / 15, b = o - 15 * i;
        double acc = 0.0;
        for (int a = 0; a < 15; ++a) acc += w.H[(size_t)i * N + R + a] * sM[a * 30 + 15 + b];
        w.Tm[o] = acc;
    }
    __syncthreads();
    for (int o = t; o < R * R; o += T) {
        const int i = o / R, j = o - i * R;
        double acc = 0.0;
        for (int a = 0; a < 15; ++a) acc += w.Tm[i * 15 + a] * w.H[(size_t)(R + a) * N + j];
        w.Lr[o] = w.H[(size_t)i * N + j] - acc;
    }
    for (int i = t; i < R; i += T) {
        double acc = 0.0;
        for (int a = 0; a < 15; ++a) acc += w.Tm[i * 15 + a] * w.eta[R + a];
        w.er[i] = w.eta[i] - acc;
    }
    for (int i = t; i < (nfm - 1) * 16; i += T) w.lin_out[i] = w.states[16 + i];
    __syncthreads();
    for (int i = t; i < R * R; i += T) {
        w.Lambda_out[i] = w.Lr[i];
        w.S_out[i] = 0.0;
    }
    for (int i = t; i < R; i += T) {
        w.eta_out[i] = w.er[i];
        w.f_out[i] = 0.0;
    }
    // ---- (vi) sqrt factor.  Fast path: drop structurally-zero rows, shifted Cholesky as the PD test.
    if (t == 0) {
        int n = 0;
        for (int i = 0; i < R; ++i)
            if (w.Lr[(size_t)i * R + i] != 0.0) w.nz[n++] = i;
        s_cnt = n;
    }
    __syncthreads();
    const int Rn = s_cnt, Rb = (Rn + 14) / 15 * 15;
    int fast = 0;
    if (!w.force_eigen) {
        // work = Lr_nz - 1e-8 I, padded with an identity tail to a multiple of 15
        for (int o = t; o < Rb * Rb; o += T) {
            const int i = o / Rb, j = o - i * Rb;
            double v = (i == j) ? 1.0 : 0.0;
            if (i < Rn && j < Rn) v = w.Lr[(size_t)w.nz[i] * R + w.nz[j]] - ((i == j) ? 1.0e-8 : 0.0);
            w.Wk[o] = v;
        }
        __syncthreads();
        fast = (Rn == 0) ? 1 : cholesky_blocked(sh, w.Wk, Rb);
    }
    if (fast) {
        for (int o = t; o < Rb * Rb; o += T) {
            const int i = o / Rb, j = o - i * Rb;
            double v = (i == j) ? 1.0 : 0.0;
            if (i < Rn && j < Rn) v = w.Lr[(size_t)w.nz[i] * R + w.nz[j]];
            w.Wk[o] = v;
        }
        __syncthreads();
        int ok = (Rn == 0) ? 1 : cholesky_blocked(sh, w.Wk, Rb);
        if (!ok) fast = 0;
    }
    if (fast) {
        // S = L^T on the retained rows/cols;  f = L^-1 eta (forward substitution, blocked through LDS)
        for (int o = t; o < Rn * Rn; o += T) {
            const int i = o / Rn, j = o - i * Rn;  // S[i][j] = L[j][i], j >= i
            if (j >= i) w.S_out[(size_t)w.nz[i] * R + w.nz[j]] = w.Wk[(size_t)j * Rb + i];
        }
        for (int i = t; i < Rb; i += T) w.yv[i] = (i < Rn) ? w.er[w.nz[i]] : 0.0;
        __syncthreads();
        const int nb = Rb / 15;
        for (int kb = 0; kb < nb; ++kb) {
            const int k0 = 15 * kb;
            for (int i = t; i < 225; i += T) sh.blk[(i / 15) * 16 + (i % 15)] = w.Wk[(size_t)(k0 + i / 15) * Rb + k0 + (i % 15)];
            if (t < 15) sh.vec[t] = w.yv[k0 + t];
            __syncthreads();
            if (t == 0)
                for (int c = 0; c < 15; ++c) {
                    double s = sh.vec[c];
                    for (int q = 0; q < c; ++q) s -= sh.blk[c * 16 + q] * sh.vec[q];
                    sh.vec[c] = s / sh.blk[c * 16 + c];
                }
            __syncthreads();
            if (t < 15) w.yv[k0 + t] = sh.vec[t];
            for (int i = k0 + 15 + t; i < Rb; i += T) {
                double s = w.yv[i];
                for (int q = 0; q < 15; ++q) s -= w.Wk[(size_t)i * Rb + k0 + q] * sh.vec[q];
                w.yv[i] = s;
            }
            __syncthreads();
        }
        for (int i = t; i < Rn; i += T) w.f_out[w.nz[i]] = w.yv[i];
    } else if (!w.force_eigen) {
        // rank-deficient information: pivoted Cholesky on the structurally non-zero block
        for (int o = t; o < Rn * Rn; o += T) w.Wk[o] = w.Lr[(size_t)w.nz[o / Rn] * R + w.nz[o % Rn]];
        for (int i = t; i < Rn; i += T) w.yv[i] = w.er[w.nz[i]];
        __syncthreads();
        const int rank = pivoted_cholesky(sh, phase, w.Wk, w.yv, w.V, w.cs, w.Tm, w.nz + R + 1, Rn, 1.0e-8);
        __syncthreads();
        for (int o = t; o < rank * Rn; o += T) {
            const int j = o / Rn, i = o - j * Rn;
            w.S_out[(size_t)j * R + w.nz[i]] = w.V[(size_t)j * Rn + i];
        }
        for (int j = t; j < rank; j += T) w.f_out[j] = w.cs[j];
        fast = 2;
    } else {
        // literal restatement: eigendecomposition, lambda+ = lambda > 1e-8 ? lambda : 0 (:441-458)
        for (int i = t; i < R * R; i += T) w.Wk[i] = w.Lr[i];
        __syncthreads();
        jacobi_eigen(sh, w.Wk, w.V, w.cs, R);
        for (int o = t; o < R * R; o += T) {
            const int k = o / R, j = o - k * R;  // S[k][j] = sqrt(lambda_k+) V[j][k]
            const double lam = w.Wk[(size_t)k * R + k];
            w.S_out[o] = (lam > 1.0e-8) ? sqrt(lam) * w.V[(size_t)j * R + k] : 0.0;
        }
        for (int k = t; k < R; k += T) {
            const double lam = w.Wk[(size_t)k * R + k];
            double acc = 0.0;
            for (int j = 0; j < R; ++j) acc += w.V[(size_t)j * R + k] * w.er[j];
            w.f_out[k] = (lam > 1.0e-8) ? sqrt(1.0 / lam) * acc : 0.0;
        }
    }
    if (t == 0) {
        w.info[0] = (double)fast;
        w.info[1] = (double)Rn;
    }
}

}  // namespace

void rdvio_launch_marginalize(hipStream_t stream, const MargWs &w) {
    hipLaunchKernelGGL(marginalize_kernel, dim3(1), dim3(T), 0, stream, w);
}
