#include <cstdio>
#include <cstdlib>
// Host side of rdvio_hip_marginalize.  The marginalisation graph
// (/root/reference/src/rdvio_estimation/include/rdvio/estimation/ceres/marginalization_factor.h:95-380: the current
// prior, the preintegration factor between frames 0 and 1, every reprojection factor of the victim-observed tracks) is
// a BA problem with every frame and landmark free and no robust loss, so it is indexed and packed by the solver's
// own host code (solver_host.hip) into a slot of its own; the kernel then runs the shared linearisation and
// normal-equation assembly followed by the marginalisation tail (solver_kernels.hip, marg_tail.hpp).
#include <vector>

#include "ctx.hpp"

static int marg_prepare(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb) {
    if (!pb) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null marginalisation problem");
    const int nfm = pb->n_frames, np = pb->n_prior, nf = pb->n_factors, nl = pb->n_landmarks;
    if (nfm < 2 || np < 0 || nf < 0 || nl < 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad marginalisation sizes");
    if (nfm > ctx->max_window + 2 || np > nfm) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d frames exceed window capacity", nfm);
    std::vector<uint8_t> frame_free(nfm, 0), lm_free(nl > 0 ? nl : 1, 0);
    const int32_t pre_i = 0, pre_j = 1;  // the victim's preintegration factor (:163-231)
    rdvio_ba_problem q;
    memset(&q, 0, sizeof q);
    q.n_frames = nfm;
    q.states = pb->states;
    q.frame_fixed = frame_free.data();
    q.extr = pb->extr;
    q.sqrt_inv_cov = pb->sqrt_inv_cov;
    q.n_landmarks = nl;
    q.z_ref = pb->z_ref;
    q.inv_depth = pb->inv_depth;
    q.lm_fixed = lm_free.data();
    q.n_factors = nf;
    q.tgt = pb->tgt;
    q.ref = pb->ref;
    q.lm = pb->lm;
    q.tangent = pb->tangent;
    q.n_preint = pb->preint01 ? 1 : 0;
    q.pre_i = &pre_i;
    q.pre_j = &pre_j;
    q.preint = pb->preint01;
    q.n_prior = np;
    q.prior_frames = pb->prior_frames;
    q.prior_lin = pb->prior_lin;
    q.prior_S = pb->prior_S;
    q.prior_f = pb->prior_f;
    return rdvio_ba_prepare(ctx, ctx->marg, &q, ctx->marg_bytes, true);
}

extern "C" {

int rdvio_hip_marginalize_upload(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb) {
    if (!ctx) return RDVIO_ERR_INVALID;
    ctx->marg.ready = false;
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[RDVIO_LANE_MARG]));
    if (int rc = marg_prepare(ctx, pb)) return rc;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->marg.arena, ctx->marg.host, ctx->marg.in_bytes, hipMemcpyHostToDevice, ctx->lane[RDVIO_LANE_MARG]));
    return RDVIO_OK;
}

int rdvio_hip_marginalize_resident(rdvio_hip_ctx *ctx, int force_eigen) {
    if (!ctx) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::BaSlot &S = ctx->marg;
    if (!S.ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no marginalisation problem uploaded");
    SolverWs &w = S.ws;
    w.marg_force_eigen = force_eigen ? 1 : 0;
    rdvio_launch_marginalize(ctx->lane[RDVIO_LANE_MARG], w);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}

int rdvio_hip_marginalize_fetch(rdvio_hip_ctx *ctx, double *S_out, double *f_out, double *lin_out, double *Lambda_out,
                                double *eta_out, int *used_fast_path) {
    if (!ctx) return RDVIO_ERR_INVALID;
    if (!ctx->marg.ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no marginalisation problem uploaded");
    const SolverWs &w = ctx->marg.ws;
    const size_t R = (size_t)15 * (w.nfr - 1);
    hipStream_t st = ctx->lane[RDVIO_LANE_MARG];
    const rdvio_hip_ctx::BaSlot &slot = ctx->marg;
    // S | f | lin | info in one device-to-host copy into the slot's pinned blob (behind the uploaded inputs)
    const size_t n_out = (size_t)(w.m_info + 4 - w.S_out);
    const size_t host_off = (slot.in_bytes + 63) & ~(size_t)63;
    if (host_off + n_out * sizeof(double) > slot.host_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "result does not fit the pinned blob");
    double *down = (double *)((uint8_t *)slot.host + host_off);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down, w.S_out, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Lambda_out) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(Lambda_out, w.Lambda_out, R * R * sizeof(double), hipMemcpyDeviceToHost, st));
    if (eta_out) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(eta_out, w.eta_out, R * sizeof(double), hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    if (const unsigned ug = rdvio_ug_violations()) return rdvio_fail(ctx, RDVIO_ERR_HIP, "RDVIO_UG was applied to an LDS address %u times (checking build)", ug);
    if (S_out) memcpy(S_out, down, R * R * sizeof(double));
    if (f_out) memcpy(f_out, down + (w.f_out - w.S_out), R * sizeof(double));
    if (lin_out) memcpy(lin_out, down + (w.lin_out - w.S_out), (size_t)(w.nfr - 1) * 16 * sizeof(double));
    const double *info = down + (w.m_info - w.S_out);
    if (used_fast_path) *used_fast_path = (int)info[0];  // 1 plain Cholesky, 2 pivoted Cholesky, 0 eigen
    if (getenv("RDVIO_DEBUG_MARG")) fprintf(stderr, "marg: nfr %d N %d nl %d nf %d lds %d path %d Rn %d\n", w.nfr, w.N, w.nl, w.nf, w.lds_chol, (int)info[0], (int)info[1]);
    return RDVIO_OK;
}

int rdvio_hip_marginalize(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb, int force_eigen, double *S_out, double *f_out,
                          double *lin_out, double *Lambda_out, double *eta_out, int *used_fast_path) {
    if (int rc = rdvio_hip_marginalize_upload(ctx, pb)) return rc;
    if (int rc = rdvio_hip_marginalize_resident(ctx, force_eigen)) return rc;
    return rdvio_hip_marginalize_fetch(ctx, S_out, f_out, lin_out, Lambda_out, eta_out, used_fast_path);
}

}  // extern "C"
