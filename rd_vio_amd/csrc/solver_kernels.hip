// The non-linear solve behind rdvio::Solver::solve on gfx950 (FP64), as ONE persistent single-workgroup
// kernel: the whole trust-region loop (Ceres TrustRegionMinimizer + DoglegStrategy + landmark-Schur normal
// equations, as restated in DESIGN.md "Solver") runs on the device without host round trips.
//
// Reference: /root/reference/src/rdvio_estimation/src/solver.cpp:180-194 (ceres::Solve, DOGLEG,
// SPARSE_SCHUR, CauchyLoss(1.0) on visual factors, update_state_every_iteration) and the factor classes
// in .../estimation/ceres/*.h; graph assembly in /root/reference/src/rdvio/src/sliding_window_tracker.cpp:101-125,
// 226-300, 349-444.
//
// Why one workgroup: at the reference's sizes (15(W+1) <= 270 pose columns, <= ~1000 landmarks, <= 15000
// factors) an iteration is a chain of ~60 dependent phases of a few microseconds each; a kernel boundary
// (~1.5 us) or a grid barrier (~4-10 us) per phase would cost more than the phase.  Inside one workgroup a
// phase boundary is an s_barrier.  All reductions are fixed-order (strided partials + LDS tree), so the
// result is bitwise reproducible run to run.
//
// Normal equations.  With J robustified (sqrt(rho') scaling) and Jacobi-scaled by Sigma:
//   pose block   H = J_p^T J_p  (N x N, N = 15 * free frames), assembled OUTPUT-STATIONARY from per
//                frame-pair factor lists (no atomics);
//   landmarks    scalar m_l = |J_l|^2, coupling row A[l, :] = J_l^T J_p  (dense L x 6 nfree);
//   Schur        S = Sigma (H - A^T W A) Sigma + mu D^2,  w_l = sigma_l^2 / (sigma_l^2 m_l + mu d_l^2)
//   blocked (15-wide) Cholesky of S, landmark back-substitution.
#include "ctx.hpp"
#include "factors.hpp"
#include "solver_ws.hpp"
#include "block_linalg.hpp"

namespace {

constexpr int T = RDVIO_SOLVER_THREADS;

using Shared = BlockShared<T>;

DM double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------------
// cost (and optionally the stored, robustified linearisation) at (states, invd)
// ---------------------------------------------------------------------------------------------
template <bool LIN>
DM double evaluate(const SolverWs &w, Shared &sh, const double *states, const double *invd) {
    const int t = threadIdx.x;
    const double *W = w.extr + 14;
    double cost = 0.0;
    // reprojection factors, CauchyLoss(1): cost 0.5 log(1+s); Corrector with rho'' < 0 => scale r and J by sqrt(rho')
    for (int k = t; k < w.nf; k += T) {
        double r[2], Jt[12], Jr[12], Jd[2];
        const int l = w.lm[k];
        reprojection_factor<LIN>(states + 16 * w.tgt[k], states + 16 * w.ref[k], w.tangent + 9 * (size_t)k,
                                 w.z_ref + 3 * (size_t)l, invd[l], w.extr, W, r, Jt, Jr, Jd);
        const double s = r[0] * r[0] + r[1] * r[1];
        const double sum = 1.0 + s;
        cost += 0.5 * log(sum);
        if (LIN) {
            const double sc = sqrt(fmax(1.0 / sum, 2.2250738585072014e-308));
            w.r_f[2 * (size_t)k] = r[0] * sc;
            w.r_f[2 * (size_t)k + 1] = r[1] * sc;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                w.Jt[12 * (size_t)k + i] = Jt[i] * sc;
                w.Jr[12 * (size_t)k + i] = Jr[i] * sc;
            }
            w.Jd[2 * (size_t)k] = Jd[0] * sc;
            w.Jd[2 * (size_t)k + 1] = Jd[1] * sc;
        }
    }
    for (int k = t; k < w.nrot; k += T) {
        double r[2], J[6];
        rotation_prior_factor<LIN>(states + 16 * w.rot_tgt[k], states + 16 * w.rot_ref[k], w.rot_zref + 3 * k,
                                   w.rot_tangent + 9 * k, w.extr, W, r, J);
        const double s = r[0] * r[0] + r[1] * r[1];
        const double sum = 1.0 + s;
        cost += 0.5 * log(sum);
        if (LIN) {
            const double sc = sqrt(fmax(1.0 / sum, 2.2250738585072014e-308));
            w.r_r[2 * k] = r[0] * sc;
            w.r_r[2 * k + 1] = r[1] * sc;
#pragma unroll
            for (int i = 0; i < 6; ++i) w.Jro[6 * k + i] = J[i] * sc;
        }
    }
    // preintegration factors: unwhitened part by one thread per factor, whitening spread over the block
    if (w.npre > 0) {
        if (LIN) {
            for (int i = t; i < w.npre * 450; i += T) w.G[i] = 0.0;
            __syncthreads();
        }
        for (int k = t; k < w.npre; k += T)
            preintegration_unwhitened<LIN>(states + 16 * w.pre_i[k], states + 16 * w.pre_j[k],
                                           w.preint + (size_t)RDVIO_PREINT_SIZE * k, w.user + 16 * w.pre_i[k] + ST_BG,
                                           w.extr, w.e_p + 15 * k, w.G + 450 * k, w.G + 450 * k + 225);
        __syncthreads();
        double *r_p = LIN ? w.r_p : w.c_p;
        for (int o = t; o < w.npre * 15; o += T) {
            const int k = o / 15, row = o - 15 * k;
            const double *Sic = w.preint + (size_t)RDVIO_PREINT_SIZE * k + PRE_SIC;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 15; ++q) acc += Sic[row * 15 + q] * w.e_p[15 * k + q];
            r_p[o] = acc;
            cost += 0.5 * acc * acc;
        }
        if (LIN) {
            for (int o = t; o < w.npre * 450; o += T) {
                const int k = o / 450, rem = o - 450 * k, which = rem / 225, rc = rem - 225 * which;
                const int row = rc / 15, col = rc - 15 * row;
                const double *Sic = w.preint + (size_t)RDVIO_PREINT_SIZE * k + PRE_SIC;
                const double *G = w.G + 450 * k + 225 * which;
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < 15; ++q) acc += Sic[row * 15 + q] * G[q * 15 + col];
                w.Jp[o] = acc;
            }
        }
    }
    // marginalisation prior: e, r = S e + f; Jacobian handled through Lambda = S^T S (constant) and E
    if (w.np > 0) {
        for (int i = t; i < w.np; i += T) {
            M3 Jri;
            marginalization_frame_error(states + 16 * w.prior_frames[i], w.lin + 16 * i, w.e_m + 15 * i, LIN ? &Jri : nullptr);
            if (LIN)
                for (int q = 0; q < 9; ++q) w.Jri[9 * i + q] = Jri.m[q];
        }
        __syncthreads();
        double *r_m = LIN ? w.r_m : w.c_m;
        for (int row = t; row < w.D; row += T) {
            double acc = 0.0;
            for (int c = 0; c < w.D; ++c) acc += w.S[(size_t)row * w.D + c] * w.e_m[c];
            acc += w.f[row];
            r_m[row] = acc;
            cost += 0.5 * acc * acc;
            if (LIN) {
                double a2 = 0.0;  // (Lambda e + eta0)[row] = (S^T r)[row]
                for (int c = 0; c < w.D; ++c) a2 += w.Lam[(size_t)row * w.D + c] * w.e_m[c];
                w.le[row] = a2 + w.eta0[row];
            }
        }
    }
    return block_sum(sh, cost);
}

// E_i[a, b] of the prior Jacobian J = S E, E = blockdiag(Jr^-1(e_theta), I12) per frame
DM double prior_E(const SolverWs &w, int i, int a, int b) {
    if (a < 3 && b < 3) return w.Jri[9 * i + 3 * a + b];
    return a == b ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------
// normal equations from the stored linearisation: H, g, landmark scalars, coupling rows A
// ---------------------------------------------------------------------------------------------
DM void build_normal_equations(const SolverWs &w, Shared &sh) {
    const int t = threadIdx.x;
    const int N = w.N, nfree = w.nfree, NA = 6 * nfree;
    for (int i = t; i < N * N; i += T) w.H[i] = 0.0;
    // landmarks: factors of one landmark are contiguous (lm sorted); one thread per landmark, fixed order
    for (int l = t; l < w.nl; l += T) {
        double *Arow = w.A + (size_t)l * NA;
        for (int i = 0; i < NA; ++i) Arow[i] = 0.0;
        double m = 0.0, gl = 0.0;
        if (w.lfree[l]) {
            for (int k = w.lm_first[l]; k < w.lm_first[l] + w.lm_count[l]; ++k) {
                const double d0 = w.Jd[2 * (size_t)k], d1 = w.Jd[2 * (size_t)k + 1];
                m += d0 * d0 + d1 * d1;
                gl += d0 * w.r_f[2 * (size_t)k] + d1 * w.r_f[2 * (size_t)k + 1];
                const int ct = w.fcol[w.tgt[k]], cr = w.fcol[w.ref[k]];
                if (ct >= 0)
                    for (int a = 0; a < 6; ++a) Arow[6 * ct + a] += d0 * w.Jt[12 * (size_t)k + a] + d1 * w.Jt[12 * (size_t)k + 6 + a];
                if (cr >= 0)
                    for (int a = 0; a < 6; ++a) Arow[6 * cr + a] += d0 * w.Jr[12 * (size_t)k + a] + d1 * w.Jr[12 * (size_t)k + 6 + a];
            }
        }
        w.lm_m[l] = m;
        w.lm_g[l] = gl;
    }
    __syncthreads();
    // reprojection J_p^T J_p: one thread per (frame pair, a, b), looping the pair's factor list in order
    for (int o = t; o < w.npairs * 36; o += T) {
        const int p = o / 36, ab = o - 36 * p, a = ab / 6, b = ab - 6 * a;
        const int fi = w.pair_fi[p], fj = w.pair_fj[p];
        double acc = 0.0;
        for (int it = w.pair_off[p]; it < w.pair_off[p + 1]; ++it) {
            const int item = w.pair_item[it], k = item >> 2, code = item & 3;
            // code bit0: row block uses Jr (else Jt); bit1: column block uses Jr (else Jt)
            const double *Jx = ((code & 1) ? w.Jr : w.Jt) + 12 * (size_t)k;
            const double *Jy = ((code & 2) ? w.Jr : w.Jt) + 12 * (size_t)k;
            acc += Jx[a] * Jy[b] + Jx[6 + a] * Jy[6 + b];
        }
        w.H[(size_t)(15 * fi + a) * N + 15 * fj + b] = acc;
        if (fi != fj) w.H[(size_t)(15 * fj + b) * N + 15 * fi + a] = acc;
    }
    __syncthreads();
    // rotation priors: theta-theta block of the target frame
    if (w.nrot > 0) {
        for (int o = t; o < nfree * 9; o += T) {
            const int c = o / 9, ab = o - 9 * c, a = ab / 3, b = ab - 3 * a;
            double acc = 0.0;
            for (int k = 0; k < w.nrot; ++k)
                if (w.fcol[w.rot_tgt[k]] == c) acc += w.Jro[6 * k + a] * w.Jro[6 * k + b] + w.Jro[6 * k + 3 + a] * w.Jro[6 * k + 3 + b];
            w.H[(size_t)(15 * c + a) * N + 15 * c + b] += acc;
        }
        __syncthreads();
    }
    // preintegration factors: sequential over factors (neighbouring factors share a diagonal block)
    for (int k = 0; k < w.npre; ++k) {
        const int cs[2] = {w.fcol[w.pre_i[k]], w.fcol[w.pre_j[k]]};
        for (int o = t; o < 900; o += T) {
            const int xy = o / 225, ab = o - 225 * xy, x = xy >> 1, y = xy & 1, a = ab / 15, b = ab - 15 * a;
            if (cs[x] < 0 || cs[y] < 0) continue;
            const double *Jx = w.Jp + 450 * k + 225 * x, *Jy = w.Jp + 450 * k + 225 * y;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 15; ++q) acc += Jx[q * 15 + a] * Jy[q * 15 + b];
            w.H[(size_t)(15 * cs[x] + a) * N + 15 * cs[y] + b] += acc;
        }
        __syncthreads();
    }
    // prior: E^T Lambda E
    if (w.np > 0) {
        for (int o = t; o < w.D * w.D; o += T) {
            const int ra = o / w.D, cb = o - ra * w.D, i = ra / 15, a = ra - 15 * i, j = cb / 15, b = cb - 15 * j;
            const int ci = w.fcol[w.prior_frames[i]], cj = w.fcol[w.prior_frames[j]];
            if (ci < 0 || cj < 0) continue;
            double acc;
            if (a >= 3 && b >= 3) {
                acc = w.Lam[(size_t)ra * w.D + cb];
            } else {
                acc = 0.0;
                const int a0 = a < 3 ? 0 : a, a1 = a < 3 ? 3 : a + 1, b0 = b < 3 ? 0 : b, b1 = b < 3 ? 3 : b + 1;
                for (int aa = a0; aa < a1; ++aa)
                    for (int bb = b0; bb < b1; ++bb)
                        acc += prior_E(w, i, aa, a) * w.Lam[(size_t)(15 * i + aa) * w.D + 15 * j + bb] * prior_E(w, j, bb, b);
            }
            w.H[(size_t)(15 * ci + a) * N + 15 * cj + b] += acc;
        }
        __syncthreads();
    }
    // gradient g = J_p^T r, one thread per entry
    for (int o = t; o < N; o += T) {
        const int c = o / 15, a = o - 15 * c;
        double acc = 0.0;
        if (a < 6) {
            const int p = w.diag_pair[c];
            for (int it = w.pair_off[p]; it < w.pair_off[p + 1]; ++it) {
                const int item = w.pair_item[it], k = item >> 2, code = item & 3;
                const double *Jx = ((code & 1) ? w.Jr : w.Jt) + 12 * (size_t)k;
                acc += Jx[a] * w.r_f[2 * (size_t)k] + Jx[6 + a] * w.r_f[2 * (size_t)k + 1];
            }
            if (a < 3)
                for (int k = 0; k < w.nrot; ++k)
                    if (w.fcol[w.rot_tgt[k]] == c) acc += w.Jro[6 * k + a] * w.r_r[2 * k] + w.Jro[6 * k + 3 + a] * w.r_r[2 * k + 1];
        }
        for (int k = 0; k < w.npre; ++k) {
            for (int x = 0; x < 2; ++x) {
                if (w.fcol[x ? w.pre_j[k] : w.pre_i[k]] != c) continue;
                const double *Jx = w.Jp + 450 * k + 225 * x;
#pragma unroll
                for (int q = 0; q < 15; ++q) acc += Jx[q * 15 + a] * w.r_p[15 * k + q];
            }
        }
        for (int i = 0; i < w.np; ++i) {
            if (w.fcol[w.prior_frames[i]] != c) continue;
            if (a < 3) {
                for (int aa = 0; aa < 3; ++aa) acc += prior_E(w, i, aa, a) * w.le[15 * i + aa];
            } else {
                acc += w.le[15 * i + a];
            }
        }
        w.g[o] = acc;
    }
    __syncthreads();
}

// ||J x||^2 and (J x).r over all residual blocks (x: unscaled-J coordinates: tp pose entries, tl landmarks)
DM void jx_products(const SolverWs &w, Shared &sh, const double *tp, const double *tl, double *sq_out, double *dr_out) {
    const int t = threadIdx.x;
    double sq = 0.0, dr = 0.0;
    for (int k = t; k < w.nf; k += T) {
        const int ct = w.fcol[w.tgt[k]], cr = w.fcol[w.ref[k]], l = w.lm[k];
        const double xl = w.lfree[l] ? tl[l] : 0.0;
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            double v = w.Jd[2 * (size_t)k + row] * xl;
            if (ct >= 0)
                for (int a = 0; a < 6; ++a) v += w.Jt[12 * (size_t)k + 6 * row + a] * tp[15 * ct + a];
            if (cr >= 0)
                for (int a = 0; a < 6; ++a) v += w.Jr[12 * (size_t)k + 6 * row + a] * tp[15 * cr + a];
            sq += v * v;
            dr += v * w.r_f[2 * (size_t)k + row];
        }
    }
    for (int k = t; k < w.nrot; k += T) {
        const int c = w.fcol[w.rot_tgt[k]];
        if (c < 0) continue;
        for (int row = 0; row < 2; ++row) {
            double v = 0.0;
            for (int a = 0; a < 3; ++a) v += w.Jro[6 * k + 3 * row + a] * tp[15 * c + a];
            sq += v * v;
            dr += v * w.r_r[2 * k + row];
        }
    }
    for (int o = t; o < w.npre * 15; o += T) {
        const int k = o / 15, row = o - 15 * k;
        const int ci = w.fcol[w.pre_i[k]], cj = w.fcol[w.pre_j[k]];
        double v = 0.0;
        if (ci >= 0)
            for (int a = 0; a < 15; ++a) v += w.Jp[450 * k + 15 * row + a] * tp[15 * ci + a];
        if (cj >= 0)
            for (int a = 0; a < 15; ++a) v += w.Jp[450 * k + 225 + 15 * row + a] * tp[15 * cj + a];
        sq += v * v;
        dr += v * w.r_p[o];
    }
    if (w.np > 0) {
        // J x = S (E x): ||J x||^2 = (Ex)^T Lambda (Ex), (J x).r = (Ex)^T (Lambda e + eta0)
        for (int o = t; o < w.D; o += T) {
            const int i = o / 15, a = o - 15 * i, c = w.fcol[w.prior_frames[i]];
            double v = 0.0;
            if (c >= 0) {
                if (a < 3) {
                    for (int b = 0; b < 3; ++b) v += w.Jri[9 * i + 3 * a + b] * tp[15 * c + b];
                } else {
                    v = tp[15 * c + a];
                }
            }
            w.Ex[o] = v;
        }
        __syncthreads();
        for (int row = t; row < w.D; row += T) {
            double acc = 0.0;
            for (int c = 0; c < w.D; ++c) acc += w.Lam[(size_t)row * w.D + c] * w.Ex[c];
            sq += w.Ex[row] * acc;
            dr += w.Ex[row] * w.le[row];
        }
    }
    *sq_out = block_sum(sh, sq);
    *dr_out = block_sum(sh, dr);
}

DM double x_norm_of(const SolverWs &w, Shared &sh, const double *st, const double *dep) {
    const int t = threadIdx.x;
    double s = 0.0;
    for (int o = t; o < w.nfr * 16; o += T)
        if (w.fcol[o / 16] >= 0) s += st[o] * st[o];
    for (int l = t; l < w.nl; l += T)
        if (w.lfree[l]) s += dep[l] * dep[l];
    return sqrt(block_sum(sh, s));
}

// gradient_max_norm = || x - Plus(x, -g) ||_inf  (TrustRegionMinimizer::EvaluateGradientAndJacobian)
DM double grad_max_norm(const SolverWs &w, Shared &sh) {
    const int t = threadIdx.x;
    double m = 0.0;
    for (int i = t; i < w.nfr; i += T) {
        const int c = w.fcol[i];
        if (c < 0) continue;
        double d15[15], o[16];
        for (int a = 0; a < 15; ++a) d15[a] = -w.g[15 * c + a];
        state_plus(w.x + 16 * i, d15, o);
        for (int a = 0; a < 16; ++a) m = fmax(m, fabs(w.x[16 * i + a] - o[a]));
    }
    for (int l = t; l < w.nl; l += T)
        if (w.lfree[l]) m = fmax(m, fabs(w.lm_g[l]));
    return block_max(sh, m);
}

__global__ __launch_bounds__(T) void ba_solve_kernel(SolverWs w) {
    __shared__ Shared sh;
    const int t = threadIdx.x;
    const int N = w.N, nl = w.nl, nfree = w.nfree, NA = 6 * nfree;

    // ------------------------------------------------------------------ setup
    for (int i = t; i < w.nfr * 16; i += T) w.user[i] = w.x[i];
    for (int l = t; l < nl; l += T) w.lfree[l] = (w.lm_count[l] > 0 && !w.lm_fixed[l]) ? 1 : 0;
    if (w.np > 0) {
        // Lambda = S^T S, eta0 = S^T f: constant during the solve
        for (int o = t; o < w.D * w.D; o += T) {
            const int a = o / w.D, b = o - a * w.D;
            double acc = 0.0;
            for (int q = 0; q < w.D; ++q) acc += w.S[(size_t)q * w.D + a] * w.S[(size_t)q * w.D + b];
            w.Lam[o] = acc;
        }
        for (int a = t; a < w.D; a += T) {
            double acc = 0.0;
            for (int q = 0; q < w.D; ++q) acc += w.S[(size_t)q * w.D + a] * w.f[q];
            w.eta0[a] = acc;
        }
    }
    __syncthreads();

    double radius = 1e4, mu = 1e-8, alpha = 0.0, dogleg_step_norm = 0.0;
    int reuse = 0, iteration = 0, invalid_steps = 0, last_successful = 0, n_success = 0;
    int term = 1;  // NO_CONVERGENCE
    double x_norm = x_norm_of(w, sh, w.x, w.xd);
    double x_cost = evaluate<true>(w, sh, w.x, w.xd);
    const double initial_cost = x_cost;
    build_normal_equations(w, sh);
    // Jacobi scaling from the iteration-0 Jacobian
    for (int i = t; i < N; i += T) w.sig_p[i] = 1.0 / (1.0 + sqrt(w.H[(size_t)i * N + i]));
    for (int l = t; l < nl; l += T) w.sig_l[l] = 1.0 / (1.0 + sqrt(w.lm_m[l]));
    __syncthreads();
    double grad_max = grad_max_norm(w, sh);

    if (N == 0 && w.n_lfree_hint == 0) term = 0;
    else
        for (;;) {
            if (last_successful) {  // state-updating callback
                for (int i = t; i < w.nfr * 16; i += T) w.user[i] = w.x[i];
                __syncthreads();
            }
            if (iteration >= w.max_iter) { term = 1; break; }
            if (grad_max <= 1e-10) { term = 0; break; }
            if (radius <= 1e-32) { term = 0; break; }
            iteration++;
            last_successful = 0;

            int solve_ok = 1;
            if (!reuse) {
                reuse = 1;
                // dogleg diagonal, scaled gradient
                double gsq = 0.0;
                for (int i = t; i < N; i += T) {
                    const double s = w.sig_p[i];
                    const double d = sqrt(clampd(s * s * w.H[(size_t)i * N + i], 1e-6, 1e32));
                    w.diag_p[i] = d;
                    const double gv = s * w.g[i] / d;
                    w.grad_p[i] = gv;
                    gsq += gv * gv;
                    w.tp[i] = s * gv / d;
                }
                for (int l = t; l < nl; l += T) {
                    if (!w.lfree[l]) { w.diag_l[l] = 1.0; w.grad_l[l] = 0.0; w.tl[l] = 0.0; continue; }
                    const double s = w.sig_l[l];
                    const double d = sqrt(clampd(s * s * w.lm_m[l], 1e-6, 1e32));
                    w.diag_l[l] = d;
                    const double gv = s * w.lm_g[l] / d;
                    w.grad_l[l] = gv;
                    gsq += gv * gv;
                    w.tl[l] = s * gv / d;
                }
                gsq = block_sum(sh, gsq);
                double jsq, jdr;
                jx_products(w, sh, w.tp, w.tl, &jsq, &jdr);
                alpha = gsq / jsq;
                // Gauss-Newton step: (H_s + mu D^2) y = g_s with the landmarks eliminated
                solve_ok = 0;
                while (mu < 1.0) {
                    for (int l = t; l < nl; l += T) {
                        double wl = 0.0;
                        if (w.lfree[l]) {
                            const double s2 = w.sig_l[l] * w.sig_l[l];
                            wl = s2 / (s2 * w.lm_m[l] + mu * w.diag_l[l] * w.diag_l[l]);
                        }
                        w.lm_w[l] = wl;
                    }
                    __syncthreads();
                    // S = Sigma (H - A^T W A) Sigma + mu D^2   (lower triangle is what the factorisation reads)
                    for (int o = t; o < N * N; o += T) {
                        const int i = o / N, j = o - i * N;
                        if (j > i) continue;
                        const int fi = i / 15, a = i - 15 * fi, fj = j / 15, b = j - 15 * fj;
                        double v = w.H[o];
                        if (a < 6 && b < 6) {
                            const int ia = 6 * fi + a, jb = 6 * fj + b;
                            double acc = 0.0;
                            for (int l = 0; l < nl; ++l) acc += w.A[(size_t)l * NA + ia] * w.lm_w[l] * w.A[(size_t)l * NA + jb];
                            v -= acc;
                        }
                        v *= w.sig_p[i] * w.sig_p[j];
                        if (i == j) v += mu * w.diag_p[i] * w.diag_p[i];
                        w.Sm[o] = v;
                    }
                    for (int i = t; i < N; i += T) {
                        const int fi = i / 15, a = i - 15 * fi;
                        double v = w.g[i];
                        if (a < 6) {
                            double acc = 0.0;
                            for (int l = 0; l < nl; ++l) acc += w.A[(size_t)l * NA + 6 * fi + a] * w.lm_w[l] * w.lm_g[l];
                            v -= acc;
                        }
                        w.yp[i] = v * w.sig_p[i];
                    }
                    __syncthreads();
                    int ok = (N == 0) ? 1 : cholesky_blocked(sh, w.Sm, N);
                    if (ok && N > 0) cholesky_solve(sh, w.Sm, N, w.yp);
                    double bad = 0.0;
                    if (ok) {
                        for (int l = t; l < nl; l += T) {
                            double y = 0.0;
                            if (w.lfree[l]) {
                                double s = w.lm_g[l];
                                for (int i = 0; i < NA; ++i) {
                                    const int col = 15 * (i / 6) + (i % 6);
                                    s -= w.A[(size_t)l * NA + i] * w.sig_p[col] * w.yp[col];
                                }
                                const double s2 = w.sig_l[l] * w.sig_l[l];
                                y = w.sig_l[l] * s / (s2 * w.lm_m[l] + mu * w.diag_l[l] * w.diag_l[l]);
                                if (!isfinite(y)) bad = 1.0;
                            }
                            w.yl[l] = y;
                        }
                        for (int i = t; i < N; i += T)
                            if (!isfinite(w.yp[i])) bad = 1.0;
                        bad = block_max(sh, bad);
                    }
                    if (!ok || bad > 0.0) {
                        mu *= 10.0;
                        continue;
                    }
                    solve_ok = 1;
                    break;
                }
                if (solve_ok) {
                    for (int i = t; i < N; i += T) w.gn_p[i] = -w.yp[i] * w.diag_p[i];
                    for (int l = t; l < nl; l += T) w.gn_l[l] = -w.yl[l] * w.diag_l[l];
                    __syncthreads();
                }
            }
            int step_valid = 0;
            double model_cost_change = 0.0;
            if (solve_ok) {
                double a0 = 0.0, a1 = 0.0, a2 = 0.0;
                for (int i = t; i < N; i += T) { a0 += w.grad_p[i] * w.grad_p[i]; a1 += w.gn_p[i] * w.gn_p[i]; a2 += w.grad_p[i] * w.gn_p[i]; }
                for (int l = t; l < nl; l += T) { a0 += w.grad_l[l] * w.grad_l[l]; a1 += w.gn_l[l] * w.gn_l[l]; a2 += w.grad_l[l] * w.gn_l[l]; }
                const double gnorm = sqrt(block_sum(sh, a0)), gn_norm = sqrt(block_sum(sh, a1)), gdotgn = block_sum(sh, a2);
                double ca, cb;
                bool need_norm = false;
                if (gn_norm <= radius) { ca = 0.0; cb = 1.0; dogleg_step_norm = gn_norm; }
                else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0.0; dogleg_step_norm = radius; }
                else {
                    const double b_dot_a = -alpha * gdotgn;
                    const double a_sq = (alpha * gnorm) * (alpha * gnorm);
                    const double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
                    const double c = b_dot_a - a_sq;
                    const double d = sqrt(c * c + bma_sq * (radius * radius - a_sq));
                    const double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
                    ca = -alpha * (1.0 - beta);
                    cb = beta;
                    need_norm = true;
                }
                double sn = 0.0;
                for (int i = t; i < N; i += T) {
                    const double v = ca * w.grad_p[i] + cb * w.gn_p[i];
                    sn += v * v;
                    w.tp[i] = w.sig_p[i] * (v / w.diag_p[i]);  // delta = step * jacobi scaling
                }
                for (int l = t; l < nl; l += T) {
                    const double v = ca * w.grad_l[l] + cb * w.gn_l[l];
                    sn += v * v;
                    w.tl[l] = w.lfree[l] ? w.sig_l[l] * (v / w.diag_l[l]) : 0.0;
                }
                sn = block_sum(sh, sn);
                if (need_norm) dogleg_step_norm = sqrt(sn);
                double jsq, jdr;
                jx_products(w, sh, w.tp, w.tl, &jsq, &jdr);
                model_cost_change = -(jdr + 0.5 * jsq);
                step_valid = model_cost_change > 0.0;
            }
            if (!step_valid) {
                if (++invalid_steps >= 5) { term = 2; break; }
                mu *= 10.0;
                reuse = 0;
                continue;
            }
            invalid_steps = 0;
            // candidate = Plus(x, delta)
            for (int i = t; i < w.nfr; i += T) {
                const int c = w.fcol[i];
                if (c < 0) {
                    for (int a = 0; a < 16; ++a) w.xc[16 * i + a] = w.x[16 * i + a];
                } else {
                    state_plus(w.x + 16 * i, w.tp + 15 * c, w.xc + 16 * i);
                }
            }
            for (int l = t; l < nl; l += T) w.xdc[l] = w.xd[l] + (w.lfree[l] ? w.tl[l] : 0.0);
            __syncthreads();
            double cand_cost = evaluate<false>(w, sh, w.xc, w.xdc);
            if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
            double sn2 = 0.0;
            for (int o = t; o < w.nfr * 16; o += T)
                if (w.fcol[o / 16] >= 0) { const double e = w.x[o] - w.xc[o]; sn2 += e * e; }
            for (int l = t; l < nl; l += T)
                if (w.lfree[l]) { const double e = w.xd[l] - w.xdc[l]; sn2 += e * e; }
            const double step_norm = sqrt(block_sum(sh, sn2));
            if (step_norm <= 1e-8 * (x_norm + 1e-8)) { term = 0; break; }
            const double cost_change = x_cost - cand_cost;
            if (fabs(cost_change) <= 1e-6 * x_cost) { term = 0; break; }
            const double rel = (cand_cost >= 1.7976931348623157e308) ? -1.7976931348623157e308 : (x_cost - cand_cost) / model_cost_change;
            if (rel > 1e-3) {
                for (int o = t; o < w.nfr * 16; o += T) w.x[o] = w.xc[o];
                for (int l = t; l < nl; l += T) w.xd[l] = w.xdc[l];
                __syncthreads();
                x_norm = x_norm_of(w, sh, w.x, w.xd);
                x_cost = evaluate<true>(w, sh, w.x, w.xd);  // `user` still holds the previous point here
                build_normal_equations(w, sh);
                grad_max = grad_max_norm(w, sh);
                last_successful = 1;
                n_success++;
                if (rel < 0.25) radius *= 0.5;
                if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
                mu = fmax(1e-8, 2.0 * mu / 10.0);
                reuse = 0;
            } else {
                radius *= 0.5;
                reuse = 1;
            }
        }
    if (t == 0) {
        w.summary[0] = (double)iteration;
        w.summary[1] = (double)n_success;
        w.summary[2] = initial_cost;
        w.summary[3] = x_cost;
        w.summary[4] = (double)term;
    }
}

}  // namespace

void rdvio_launch_ba_solve(hipStream_t stream, const SolverWs &w) {
    hipLaunchKernelGGL(ba_solve_kernel, dim3(1), dim3(T), 0, stream, w);
}
