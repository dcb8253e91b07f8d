// Estimation-side kernels on gfx950 (FP64): batched reprojection residual/Jacobian evaluation and
// IMU preintegration.
//
//  * reprojection_eval_kernel  <- CeresReprojectionErrorFactor::Evaluate
//      /root/reference/src/rdvio_estimation/include/rdvio/estimation/ceres/reprojection_factor.h:24-89
//  * preintegrate_kernel       <- PreIntegrator::{reset,increment,integrate,compute_sqrt_inv_cov}
//      /root/reference/src/rdvio_estimation/src/preintegrator.cpp:7-100
#include "ctx.hpp"
#include "factors.hpp"

namespace {

// One thread per factor.  Per factor: 2 int32 frame indices + landmark index, 72 B tangent, 24 B z_ref,
// 8 B inverse depth read (poses are shared, L2/L1-resident); 16 B + 208 B written when r/J are materialised.
__global__ __launch_bounds__(256) void reprojection_eval_kernel(int nf, const int32_t *__restrict__ tgt,
                                                               const int32_t *__restrict__ ref,
                                                               const int32_t *__restrict__ lm,
                                                               const double *__restrict__ tangent,
                                                               const double *__restrict__ z_ref_all,
                                                               const double *__restrict__ inv_depth_all,
                                                               const double *__restrict__ states,
                                                               const double *__restrict__ extr,
                                                               const double *__restrict__ W, double *__restrict__ r_out,
                                                               double *__restrict__ Jt_out, double *__restrict__ Jr_out,
                                                               double *__restrict__ Jd_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nf) return;
    const int l = lm[k];
    double r[2], Jt[12], Jr[12], Jd[2];
    if (Jt_out) {
        reprojection_factor<true>(states + 16 * tgt[k], states + 16 * ref[k], tangent + 9 * (size_t)k,
                                  z_ref_all + 3 * (size_t)l, inv_depth_all[l], extr, W, r, Jt, Jr, Jd);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            Jt_out[12 * (size_t)k + i] = Jt[i];
            Jr_out[12 * (size_t)k + i] = Jr[i];
        }
        Jd_out[2 * (size_t)k] = Jd[0];
        Jd_out[2 * (size_t)k + 1] = Jd[1];
    } else {
        reprojection_factor<false>(states + 16 * tgt[k], states + 16 * ref[k], tangent + 9 * (size_t)k,
                                   z_ref_all + 3 * (size_t)l, inv_depth_all[l], extr, W, r, Jt, Jr, Jd);
    }
    r_out[2 * (size_t)k] = r[0];
    r_out[2 * (size_t)k + 1] = r[1];
}

// CeresRotationPriorFactor::Evaluate for a batch (unit-parity entry; inside a solve the same device routine runs in
// ba_solve_kernel).  /root/reference/src/rdvio_estimation/include/rdvio/estimation/ceres/rotation_factor.h:22-58
__global__ __launch_bounds__(256) void rotation_prior_eval_kernel(int n, const int32_t *__restrict__ tgt, const int32_t *__restrict__ ref,
                                                                 const double *__restrict__ zref, const double *__restrict__ tangent,
                                                                 const double *__restrict__ states, const double *__restrict__ extr,
                                                                 const double *__restrict__ W, double *__restrict__ r_out,
                                                                 double *__restrict__ J_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double r[2], J[6];
    if (J_out) {
        rotation_prior_factor<true>(states + 16 * tgt[k], states + 16 * ref[k], zref + 3 * (size_t)k, tangent + 9 * (size_t)k, extr, W, r, J);
#pragma unroll
        for (int i = 0; i < 6; ++i) J_out[6 * (size_t)k + i] = J[i];
    } else {
        rotation_prior_factor<false>(states + 16 * tgt[k], states + 16 * ref[k], zref + 3 * (size_t)k, tangent + 9 * (size_t)k, extr, W, r, J);
    }
    r_out[2 * (size_t)k] = r[0];
    r_out[2 * (size_t)k + 1] = r[1];
}

// ---------------------------------------------------------------------------------------------
// IMU preintegration, one wavefront per segment.
//  phase 1 (data-parallel over samples): lane k computes its sample's increment
//      e_k = exp((w_k - bg) dt_k), Jr_k = Jr((w_k - bg) dt_k), a_k, dt_k           -- the transcendental part
//  phase 2 (wavefront scan): inclusive prefix product of the e_k over lanes (Hillis-Steele, 6 steps of
//      quaternion multiplies through cross-lane shuffles) gives every lane the PRE-update dq_k the
//      reference's recurrence would see (preintegrator.cpp:72-75).  The reference renormalises after every
//      step; a product of unit quaternions is unit up to rounding, so the scan differs from the sequential
//      result by a few ulp (documented tolerance 1e-12).  dp/dv are then plain prefix sums of
//      dt*(dq_k a_k) terms.
//  phase 3 (sequential over samples, parallel over matrix entries): the 9x9 covariance recurrence
//      cov <- A cov A^T + B Q B^T and the five 3x3 bias Jacobians, one matrix entry per lane through LDS.
// Segments longer than 64 samples are processed in chunks of 64 with the running state carried over.
// ---------------------------------------------------------------------------------------------
DM double shfl_d(double v, int src) { return __shfl(v, src); }
DM double readlane_dd(double v, int src_lane) {  // src_lane wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
DM double wave_max_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
DM Q4 shfl_q(const Q4 &q, int src) { return Q4{shfl_d(q.x, src), shfl_d(q.y, src), shfl_d(q.z, src), shfl_d(q.w, src)}; }
DM V3 shfl_v(const V3 &v, int src) { return V3{shfl_d(v.x, src), shfl_d(v.y, src), shfl_d(v.z, src)}; }

__global__ __launch_bounds__(128) void preintegrate_kernel(int nseg, const int32_t *__restrict__ seg_off,
                                                          const double *__restrict__ imu,
                                                          const double *__restrict__ par /* nseg x 7: t_end,bg,ba */,
                                                          const double *__restrict__ noise, int cj, int cc,
                                                          double *__restrict__ out) {
    __shared__ double s_cov[15 * 15];
    __shared__ double s_T[9 * 9];
    __shared__ double s_jac[5 * 9];
    // per-sample operands of the chunk, built by the sample's own lane (row stride odd: conflict-free):
    // A_k (9x9), N_k = B Q B^T (9x9), and the 3x3 blocks R(dq), R hat(a), E^T, Jr for the bias Jacobians
    __shared__ double s_Ak[64][81];
    __shared__ double s_Nk[64][81];
    __shared__ double s_Mk[64][37];
    const int seg = blockIdx.x;
    if (seg >= nseg) return;
    // Two wavefronts per segment.  Wavefront 0 (tid 0..63) does everything a one-wavefront block did -- per-sample increments,
    // the scans, the sample matrices, the inverse / LLT at the end.  In the covariance recurrence, which walks the samples one by
    // one and is bound by the instruction issue of a lone wavefront, the 81 matrix entries are spread over 81 THREADS (one
    // dot-product chain each instead of two on the first 17 lanes) and the Jacobian / bias-block updates move to wavefront 1.
    // Every entry is still the same 9-term dot product in the same order: results are bit-identical to the one-wavefront kernel.
    const int tid = threadIdx.x, lane = tid & 63;
    const bool w0 = tid < 64;
    const int s0 = seg_off[seg], n = seg_off[seg + 1] - s0;
    double *o = out + (size_t)seg * RDVIO_PREINT_SIZE;
    if (n <= 0) {  // PreIntegrator::integrate returns false on empty data (:80-81): leave a reset() state
        for (int i = lane; i < RDVIO_PREINT_SIZE; i += 64) o[i] = (i == PRE_Q + 3) ? 1.0 : 0.0;
        return;
    }
    const double t_end = par[7 * seg];
    const V3 bg = v3_load(par + 7 * seg + 1), ba = v3_load(par + 7 * seg + 4);

    for (int i = lane; i < 225; i += 64) s_cov[i] = 0.0;
    if (lane < 45) s_jac[lane] = 0.0;
    Q4 run_q = q_identity();
    V3 run_p{0, 0, 0}, run_v{0, 0, 0};
    double run_t = 0.0;
    __syncthreads();

    for (int base = 0; base < n; base += 64) {
        const int cnt = min(64, n - base);
        const int k = base + lane;
        // ---- phase 1: per-sample increments
        double dt = 0.0;
        V3 a{0, 0, 0}, wdt{0, 0, 0};
        Q4 e = q_identity();
        if (lane < cnt) {
            const double *d = imu + 7 * (size_t)(s0 + k);
            const double t_next = (k + 1 < n) ? imu[7 * (size_t)(s0 + k + 1)] : t_end;
            dt = t_next - d[0];
            V3 w = v3_load(d + 1) - bg;
            a = v3_load(d + 4) - ba;
            wdt = w * dt;
            e = expmap(wdt);
        }
        // ---- phase 2a: exclusive prefix product of e (pre-update dq per sample)
        Q4 inc = e;  // inclusive scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            Q4 other = shfl_q(inc, max(lane - off, 0));
            if (lane >= off) inc = other * inc;
        }
        Q4 excl = shfl_q(inc, max(lane - 1, 0));
        if (lane == 0) excl = q_identity();
        const Q4 dq_pre = normalized(run_q * excl);   // dq the sequential loop holds before sample k
        // ---- phase 2b: dv, dp prefix sums.  dv_{k+1} = dv_k + dt (dq_k a);  dp_{k+1} = dp_k + dt dv_k + 0.5 dt^2 (dq_k a)
        const V3 qa = rot(dq_pre, a);
        V3 dv_inc = dt * qa;  // inclusive sum -> dv after sample k
        double t_inc = dt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            V3 ov = shfl_v(dv_inc, max(lane - off, 0));
            double ot = shfl_d(t_inc, max(lane - off, 0));
            if (lane >= off) { dv_inc = ov + dv_inc; t_inc = ot + t_inc; }
        }
        V3 dv_excl = shfl_v(dv_inc, max(lane - 1, 0));
        if (lane == 0) dv_excl = V3{0, 0, 0};
        const V3 dv_pre = run_v + dv_excl;
        V3 dp_inc = dt * dv_pre + (0.5 * dt * dt) * qa;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            V3 ov = shfl_v(dp_inc, max(lane - off, 0));
            if (lane >= off) dp_inc = ov + dp_inc;
        }
        // ---- phase 3: covariance / Jacobian recurrences.  First every lane builds its own sample's matrices
        // (parallel over samples), then the recurrence walks the samples with two barriers per step and nothing but
        // 9-term dot products in the dependent chain.
        if (cj || cc) {
            if (w0 && lane < cnt) {   // (wavefront 1 computed the same increments and scans redundantly; only wavefront 0 publishes)
                const M3 R = to_mat(dq_pre);
                const M3 RHa = R * hat(a);
                const M3 Et = to_mat(conj(e));
                const M3 Jr = right_jacobian(wdt);
                double *mk = s_Mk[lane];
#pragma unroll
                for (int i = 0; i < 9; ++i) { mk[i] = R.m[i]; mk[9 + i] = RHa.m[i]; mk[18 + i] = Et.m[i]; mk[27 + i] = Jr.m[i]; }
                mk[36] = dt;
                if (cc) {
                    // G_w = Jr Q_w Jr^T, G_a = R Q_a R^T: the two distinct 3x3 blocks of B Q B^T
                    double Gw[9], Ga[9];
#pragma unroll
                    for (int ri = 0; ri < 3; ++ri)
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) {
                            double gw = 0.0, ga = 0.0;
#pragma unroll
                            for (int x = 0; x < 3; ++x)
#pragma unroll
                                for (int y = 0; y < 3; ++y) {
                                    gw += Jr.m[ri * 3 + x] * noise[x * 3 + y] * Jr.m[ci * 3 + y];
                                    ga += R.m[ri * 3 + x] * noise[9 + x * 3 + y] * R.m[ci * 3 + y];
                                }
                            Gw[ri * 3 + ci] = gw;
                            Ga[ri * 3 + ci] = ga;
                        }
                    const double inv_dt = 1.0 / fmax(dt, 1.0e-7);
                    double *Ak = s_Ak[lane], *Nk = s_Nk[lane];
#pragma unroll
                    for (int i = 0; i < 81; ++i) {
                        const int r = i / 9, c = i - r * 9, rb = r / 3, cb = c / 3, ri = r % 3, ci = c % 3;
                        double av = (r == c) ? 1.0 : 0.0;
                        if (rb == 0 && cb == 0) av = Et.m[ri * 3 + ci];                         // A[q,q] = E^T
                        else if (rb == 2 && cb == 0) av = -dt * RHa.m[ri * 3 + ci];             // A[v,q]
                        else if (rb == 1 && cb == 0) av = -0.5 * dt * dt * RHa.m[ri * 3 + ci];  // A[p,q]
                        else if (rb == 1 && cb == 2) av = (ri == ci) ? dt : 0.0;                // A[p,v]
                        Ak[i] = av;
                        // B rows: q: [dt Jr, 0], p: [0, 0.5 dt^2 R], v: [0, dt R];  Q = diag(cov_w, cov_a)/max(dt,1e-7)
                        const double sr = (rb == 0) ? dt : (rb == 1 ? 0.5 * dt * dt : dt);
                        const double sc = (cb == 0) ? dt : (cb == 1 ? 0.5 * dt * dt : dt);
                        double nv = 0.0;
                        if ((rb == 0) == (cb == 0)) nv = ((rb == 0) ? Gw[ri * 3 + ci] : Ga[ri * 3 + ci]) * (sr * sc * inv_dt);
                        Nk[i] = nv;
                    }
                }
            }
            __syncthreads();
            // one matrix entry per thread (81 of the 128), Jacobian entries on threads 96..104, bias blocks on 105..122
            const int e_a = tid, r_a = e_a / 9, c_a = e_a - r_a * 9;
            const bool has = e_a < 81;
            const int jl = tid - 96, bq = tid - 105;
            for (int j = 0; j < cnt; ++j) {
                const double *Aj = s_Ak[j], *Nj = s_Nk[j], *mk = s_Mk[j];
                const double dt_j = mk[36];
                double T0 = 0.0;
                if (cc && has) {  // T = A * cov9 (fused multiply-adds: the recurrence is a serial chain of such dot products)
#pragma unroll
                    for (int x = 0; x < 9; ++x) T0 = __builtin_fma(Aj[r_a * 9 + x], s_cov[x * 15 + c_a], T0);
                }
                // one entry of each 3x3 Jacobian per thread (preintegrator.cpp:59-70; old values feed p and v)
                double n_dq_dbg = 0, n_dp_dbg = 0, n_dp_dba = 0, n_dv_dbg = 0, n_dv_dba = 0;
                if (cj && jl >= 0 && jl < 9) {
                    const int r = jl / 3, c = jl % 3;
                    const double *dq_dbg = s_jac, *dv_dbg = s_jac + 27, *dv_dba = s_jac + 36;
                    double t_rha = 0.0, t_et = 0.0;
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        t_rha += mk[9 + r * 3 + x] * dq_dbg[x * 3 + c];
                        t_et += mk[18 + r * 3 + x] * dq_dbg[x * 3 + c];
                    }
                    const double Rrc = mk[jl];
                    n_dp_dbg = s_jac[9 + jl] + dt_j * dv_dbg[jl] - 0.5 * dt_j * dt_j * t_rha;
                    n_dp_dba = s_jac[18 + jl] + dt_j * dv_dba[jl] - 0.5 * dt_j * dt_j * Rrc;
                    n_dv_dbg = dv_dbg[jl] - dt_j * t_rha;
                    n_dv_dba = dv_dba[jl] - dt_j * Rrc;
                    n_dq_dbg = t_et - dt_j * mk[27 + jl];
                }
                if (cc && has) s_T[e_a] = T0;
                __syncthreads();
                if (cj && jl >= 0 && jl < 9) {
                    s_jac[jl] = n_dq_dbg;
                    s_jac[9 + jl] = n_dp_dbg;
                    s_jac[18 + jl] = n_dp_dba;
                    s_jac[27 + jl] = n_dv_dbg;
                    s_jac[36 + jl] = n_dv_dba;
                }
                if (cc) {  // cov9 = T A^T + N; bias random walks
                    if (has) {
                        double acc_a = 0.0;
#pragma unroll
                        for (int x = 0; x < 9; ++x) acc_a = __builtin_fma(s_T[r_a * 9 + x], Aj[c_a * 9 + x], acc_a);
                        s_cov[r_a * 15 + c_a] = acc_a + Nj[e_a];
                    } else if (bq >= 0 && bq < 18) {
                        const int which = bq / 9, i9 = bq % 9, r = i9 / 3, c = i9 % 3;
                        const int o0 = which ? ES_BA : ES_BG;
                        s_cov[(o0 + r) * 15 + o0 + c] += noise[18 + 9 * which + i9] * dt_j;
                    }
                }
                __syncthreads();
            }
        }
        // carry the running state to the next chunk (values after the chunk's last sample)
        const int last = cnt - 1;
        const Q4 q_last = shfl_q(normalized(dq_pre * e), last);
        const V3 v_last = shfl_v(run_v + dv_inc, last);
        const V3 p_last = shfl_v(dp_inc, last);
        const double t_last = shfl_d(t_inc, last);
        run_q = q_last;
        run_p = run_p + p_last;
        run_v = v_last;
        run_t += t_last;
    }
    __syncthreads();
    // ---- outputs
    if (tid == 0) {
        o[PRE_T] = run_t;
        q_store(o + PRE_Q, run_q);
        v3_store(o + PRE_P, run_p);
        v3_store(o + PRE_V, run_v);
    }
    if (w0) {
        for (int i = lane; i < 225; i += 64) o[PRE_COV + i] = s_cov[i];
        if (lane < 45) o[PRE_JAC + lane] = s_jac[lane];
    }
    if (!cc) {
        if (w0)
            for (int i = lane; i < 225; i += 64) o[PRE_SIC + i] = 0.0;
        return;
    }
    if (!w0) return;   // (the inverse and its LLT below run in the registers of wavefront 0; a terminated wavefront does not take part in barriers)
    // ---- compute_sqrt_inv_cov (:97-100): inverse by Gauss-Jordan with partial pivoting on [cov | I], then LLT of the
    //      inverse, transposed.  Both run in registers: lane r holds row r, pivot rows / multipliers travel through
    //      v_readlane, so the 15-step dependent chains never touch LDS.  Rows are not swapped physically; the lane that
    //      supplied the pivot of column c ends up holding row c of the inverse.
    __shared__ double s_inv[15 * 15];
    {
        double a[30];
#pragma unroll
        for (int c = 0; c < 30; ++c) a[c] = (lane < 15) ? ((c < 15) ? s_cov[lane * 15 + c] : ((c - 15 == lane) ? 1.0 : 0.0)) : 0.0;
        bool used = lane >= 15;
        int mycol = -1;
#pragma unroll
        for (int c = 0; c < 15; ++c) {
            const double v = used ? -1.0 : fabs(a[c]);
            const double m = wave_max_d(v);
            const unsigned long long tie = __ballot(v == m && !used);
            const int piv = __builtin_amdgcn_readfirstlane(tie ? (int)__builtin_ctzll(tie) : 0);
            const double inv_d = 1.0 / readlane_dd(a[c], piv);  // one divide per column; the row is scaled by the reciprocal
            const double f = a[c];
#pragma unroll
            for (int k = 0; k < 30; ++k) {
                const double pr = readlane_dd(a[k], piv) * inv_d;
                a[k] = (lane == piv) ? pr : a[k] - f * pr;
            }
            if (lane == piv) { used = true; mycol = c; }
        }
        if (mycol >= 0) {
#pragma unroll
            for (int k = 0; k < 15; ++k) s_inv[mycol * 15 + k] = a[15 + k];
        }
    }
    __syncthreads();
    {
        double a[15];
#pragma unroll
        for (int c = 0; c < 15; ++c) a[c] = (lane < 15 && c <= lane) ? s_inv[lane * 15 + c] : 0.0;
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            const double dj = sqrt(readlane_dd(a[j], j));
            const double lj = (lane == j) ? dj : a[j] * (1.0 / dj);
            a[j] = lj;
#pragma unroll
            for (int c = j + 1; c < 15; ++c) {
                const double lc = readlane_dd(lj, c);
                a[c] -= lj * lc;
            }
        }
        if (lane < 15) {
#pragma unroll
            for (int r = 0; r < 15; ++r) o[PRE_SIC + r * 15 + lane] = (r <= lane) ? a[r] : 0.0;  // matrixL().transpose()
        }
    }
}

}  // namespace

int rdvio_launch_reprojection(rdvio_hip_ctx *ctx, int nf, int with_jac) {
    if (nf <= 0) return RDVIO_OK;
    const int32_t *tgt = ctx->ba_idx, *ref = ctx->ba_idx + ctx->max_factors, *lm = ctx->ba_idx + 2 * ctx->max_factors;
    hipLaunchKernelGGL(reprojection_eval_kernel, dim3((nf + 255) / 256), dim3(256), 0, ctx->stream, nf, tgt, ref, lm,
                       ctx->ba_tangent, ctx->ba_zref, ctx->ba_invd, ctx->ba_states, ctx->ba_extr, ctx->ba_extr + 14,
                       ctx->ba_r, with_jac ? ctx->ba_Jt : nullptr, ctx->ba_Jr, ctx->ba_Jd);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}

int rdvio_launch_rotation_prior(rdvio_hip_ctx *ctx, int n, int with_jac) {
    if (n <= 0) return RDVIO_OK;
    const int32_t *tgt = ctx->ba_idx, *ref = ctx->ba_idx + ctx->max_factors;
    hipLaunchKernelGGL(rotation_prior_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, tgt, ref, ctx->ba_zref,
                       ctx->ba_tangent, ctx->ba_states, ctx->ba_extr, ctx->ba_extr + 14, ctx->ba_r, with_jac ? ctx->ba_Jt : nullptr);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}

int rdvio_launch_preintegrate(rdvio_hip_ctx *ctx, hipStream_t st, int nseg, const int32_t *off, const double *imu, const double *par,
                              const double *noise, int cj, int cc, double *out) {
    if (nseg <= 0) return RDVIO_OK;
    hipLaunchKernelGGL(preintegrate_kernel, dim3(nseg), dim3(128), 0, st, nseg, off, imu, par, noise, cj, cc, out);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}
