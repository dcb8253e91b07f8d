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
DM Q4 shfl_q(const Q4 &q, int src) { return Q4{shfl_d(q.x, src), shfl_d(q.y, src), shfl_d(q.z, src), shfl_d(q.w, src)}; }
DM V3 shfl_v(const V3 &v, int src) { return V3{shfl_d(v.x, src), shfl_d(v.y, src), shfl_d(v.z, src)}; }

__global__ __launch_bounds__(64) void preintegrate_kernel(int nseg, const int32_t *__restrict__ seg_off,
                                                          const double *__restrict__ imu,
                                                          const double *__restrict__ par /* nseg x 7: t_end,bg,ba */,
                                                          const double *__restrict__ noise, int cj, int cc,
                                                          double *__restrict__ out) {
    __shared__ double s_cov[15 * 15];
    __shared__ double s_A[9 * 9];
    __shared__ double s_T[9 * 9];
    __shared__ double s_N[9 * 9];   // B Q B^T
    __shared__ double s_jac[5 * 9];
    __shared__ double s_m[3][9];    // per-step 3x3 operands: R(dq), R hat(a), E^T ; Jr kept in s_jr
    __shared__ double s_jr[9];
    const int seg = blockIdx.x;
    if (seg >= nseg) return;
    const int lane = threadIdx.x;
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
        // ---- phase 3: covariance / Jacobian recurrences, sequential over the chunk's samples
        if (cj || cc) {
            for (int j = 0; j < cnt; ++j) {
                // broadcast sample j's quantities to the whole wave
                const Q4 dq_j = shfl_q(dq_pre, j);
                const V3 a_j = shfl_v(a, j), wdt_j = shfl_v(wdt, j);
                const Q4 e_j = shfl_q(e, j);
                const double dt_j = shfl_d(dt, j);
                if (lane == 0) {
                    M3 R = to_mat(dq_j);
                    M3 RHa = R * hat(a_j);
                    M3 Et = to_mat(conj(e_j));
                    M3 Jr = right_jacobian(wdt_j);
#pragma unroll
                    for (int i = 0; i < 9; ++i) { s_m[0][i] = R.m[i]; s_m[1][i] = RHa.m[i]; s_m[2][i] = Et.m[i]; s_jr[i] = Jr.m[i]; }
                }
                __syncthreads();
                if (cc) {
                    // A (9x9) and N = B Q B^T (9x9), built entry-wise
                    for (int i = lane; i < 81; i += 64) {
                        int r = i / 9, c = i - r * 9, rb = r / 3, cb = c / 3, ri = r % 3, ci = c % 3;
                        double av = (r == c) ? 1.0 : 0.0;
                        if (rb == 0 && cb == 0) av = s_m[2][ri * 3 + ci];                       // A[q,q] = E^T
                        else if (rb == 2 && cb == 0) av = -dt_j * s_m[1][ri * 3 + ci];          // A[v,q]
                        else if (rb == 1 && cb == 0) av = -0.5 * dt_j * dt_j * s_m[1][ri * 3 + ci];  // A[p,q]
                        else if (rb == 1 && cb == 2) av = (ri == ci) ? dt_j : 0.0;              // A[p,v]
                        s_A[i] = av;
                        // B rows: q: [dt Jr, 0], p: [0, 0.5 dt^2 R], v: [0, dt R];  Q = diag(cov_w, cov_a)/max(dt,1e-7)
                        const double inv_dt = 1.0 / fmax(dt_j, 1.0e-7);
                        double nv = 0.0;
                        // N[r,c] = sum_{x,y} B[r,x] Q[x,y] B[c,y]; B has one non-zero 3x3 block per block-row
                        const double *Br = (rb == 0) ? s_jr : s_m[0];
                        const double *Bc = (cb == 0) ? s_jr : s_m[0];
                        const double sr = (rb == 0) ? dt_j : (rb == 1 ? 0.5 * dt_j * dt_j : dt_j);
                        const double sc = (cb == 0) ? dt_j : (cb == 1 ? 0.5 * dt_j * dt_j : dt_j);
                        const bool same = ((rb == 0) == (cb == 0));  // both gyro-driven or both acc-driven
                        if (same) {
                            const double *Qm = noise + ((rb == 0) ? 0 : 9);
#pragma unroll
                            for (int x = 0; x < 3; ++x)
#pragma unroll
                                for (int y = 0; y < 3; ++y) nv += Br[ri * 3 + x] * Qm[x * 3 + y] * Bc[ci * 3 + y];
                            nv *= sr * sc * inv_dt;
                        }
                        s_N[i] = nv;
                    }
                    __syncthreads();
                    for (int i = lane; i < 81; i += 64) {  // T = A * cov9
                        int r = i / 9, c = i - r * 9;
                        double acc = 0.0;
#pragma unroll
                        for (int x = 0; x < 9; ++x) acc += s_A[r * 9 + x] * s_cov[x * 15 + c];
                        s_T[i] = acc;
                    }
                    __syncthreads();
                    for (int i = lane; i < 81; i += 64) {  // cov9 = T A^T + N
                        int r = i / 9, c = i - r * 9;
                        double acc = 0.0;
#pragma unroll
                        for (int x = 0; x < 9; ++x) acc += s_T[r * 9 + x] * s_A[c * 9 + x];
                        s_cov[r * 15 + c] = acc + s_N[i];
                    }
                    if (lane < 18) {
                        int which = lane / 9, i9 = lane % 9, r = i9 / 3, c = i9 % 3;
                        int o0 = which ? ES_BA : ES_BG;
                        s_cov[(o0 + r) * 15 + o0 + c] += noise[18 + 9 * which + i9] * dt_j;
                    }
                }
                // one entry of each 3x3 Jacobian per lane (preintegrator.cpp:59-70; old values feed p and v)
                double n_dq_dbg = 0, n_dp_dbg = 0, n_dp_dba = 0, n_dv_dbg = 0, n_dv_dba = 0;
                if (cj && lane < 9) {
                    const int r = lane / 3, c = lane % 3;
                    const double *dq_dbg = s_jac, *dv_dbg = s_jac + 27, *dv_dba = s_jac + 36;
                    double t_rha = 0.0, t_et = 0.0;
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        t_rha += s_m[1][r * 3 + x] * dq_dbg[x * 3 + c];
                        t_et += s_m[2][r * 3 + x] * dq_dbg[x * 3 + c];
                    }
                    const double Rrc = s_m[0][lane];
                    n_dp_dbg = s_jac[9 + lane] + dt_j * dv_dbg[lane] - 0.5 * dt_j * dt_j * t_rha;
                    n_dp_dba = s_jac[18 + lane] + dt_j * dv_dba[lane] - 0.5 * dt_j * dt_j * Rrc;
                    n_dv_dbg = dv_dbg[lane] - dt_j * t_rha;
                    n_dv_dba = dv_dba[lane] - dt_j * Rrc;
                    n_dq_dbg = t_et - dt_j * s_jr[lane];
                }
                __syncthreads();
                if (cj && lane < 9) {
                    s_jac[lane] = n_dq_dbg;
                    s_jac[9 + lane] = n_dp_dbg;
                    s_jac[18 + lane] = n_dp_dba;
                    s_jac[27 + lane] = n_dv_dbg;
                    s_jac[36 + lane] = n_dv_dba;
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
    if (lane == 0) {
        o[PRE_T] = run_t;
        q_store(o + PRE_Q, run_q);
        v3_store(o + PRE_P, run_p);
        v3_store(o + PRE_V, run_v);
    }
    for (int i = lane; i < 225; i += 64) o[PRE_COV + i] = s_cov[i];
    if (lane < 45) o[PRE_JAC + lane] = s_jac[lane];
    if (!cc) {
        for (int i = lane; i < 225; i += 64) o[PRE_SIC + i] = 0.0;
        return;
    }
    // ---- compute_sqrt_inv_cov (:97-100): inverse by Gauss-Jordan with partial pivoting on [cov | I]
    //      (15 x 30 in LDS, one column per lane), then LLT of the inverse, transposed.
    __shared__ double s_M[15 * 30];
    __shared__ double s_L[15 * 15];
    for (int i = lane; i < 15 * 30; i += 64) {
        int r = i / 30, c = i - r * 30;
        s_M[i] = (c < 15) ? s_cov[r * 15 + c] : ((c - 15 == r) ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int c = 0; c < 15; ++c) {
        int piv = c;
        double best = fabs(s_M[c * 30 + c]);
        for (int r = c + 1; r < 15; ++r) {
            double v = fabs(s_M[r * 30 + c]);
            if (v > best) { best = v; piv = r; }
        }
        __syncthreads();
        if (piv != c && lane < 30) {
            double t = s_M[c * 30 + lane];
            s_M[c * 30 + lane] = s_M[piv * 30 + lane];
            s_M[piv * 30 + lane] = t;
        }
        __syncthreads();
        const double d = s_M[c * 30 + c];
        __syncthreads();
        if (lane < 30) s_M[c * 30 + lane] /= d;
        __syncthreads();
        if (lane < 30) {
            const double pc = s_M[c * 30 + lane];
            for (int r = 0; r < 15; ++r) {
                if (r == c) continue;
                const double f = s_M[r * 30 + c];
                // every lane reads column c of row r before any lane overwrites it: lane c writes last value itself
                __builtin_amdgcn_wave_barrier();
                s_M[r * 30 + lane] -= f * pc;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }
    // LLT of inv (lower), sequential over columns, rows in parallel
    for (int i = lane; i < 225; i += 64) s_L[i] = 0.0;
    __syncthreads();
    for (int j = 0; j < 15; ++j) {
        double sdiag = s_M[j * 30 + 15 + j];
        for (int k2 = 0; k2 < j; ++k2) sdiag -= s_L[j * 15 + k2] * s_L[j * 15 + k2];
        const double dj = sqrt(sdiag);
        __syncthreads();
        if (lane == j) s_L[j * 15 + j] = dj;
        if (lane > j && lane < 15) {
            double t = s_M[lane * 30 + 15 + j];
            for (int k2 = 0; k2 < j; ++k2) t -= s_L[lane * 15 + k2] * s_L[j * 15 + k2];
            s_L[lane * 15 + j] = t / dj;
        }
        __syncthreads();
    }
    for (int i = lane; i < 225; i += 64) {
        int r = i / 15, c = i - r * 15;
        o[PRE_SIC + i] = s_L[c * 15 + r];  // matrixL().transpose()
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

int rdvio_launch_preintegrate(rdvio_hip_ctx *ctx, int nseg, const int32_t *off, const double *imu, const double *par,
                              const double *noise, int cj, int cc, double *out) {
    if (nseg <= 0) return RDVIO_OK;
    hipLaunchKernelGGL(preintegrate_kernel, dim3(nseg), dim3(64), 0, ctx->stream, nseg, off, imu, par, noise, cj, cc, out);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}
