// Host side of rdvio_hip_marginalize: validation, graph index lists (victim frame permuted last,
// /root/reference/src/rdvio_estimation/include/rdvio/estimation/ceres/marginalization_factor.h:95-105), one pinned
// blob, one upload, one kernel.
#include <algorithm>
#include <vector>

#include "ctx.hpp"
#include "marg_ws.hpp"

namespace {
struct Pk {
    uint8_t *base;
    size_t cap, off = 0;
    bool ok = true;
    template <class Tp>
    size_t put(const Tp *src, size_t n) {
        off = (off + 15) & ~(size_t)15;
        const size_t at = off, bytes = n * sizeof(Tp);
        if (at + bytes > cap) { ok = false; return at; }
        if (n && src) memcpy(base + at, src, bytes);
        off += bytes;
        return at;
    }
    size_t reserve(size_t bytes) {
        off = (off + 15) & ~(size_t)15;
        const size_t at = off;
        if (at + bytes > cap) ok = false;
        off += bytes;
        return at;
    }
};
}  // namespace

static int marg_prepare(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb) {
    if (!pb) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null marginalisation problem");
    const int nfm = pb->n_frames, np = pb->n_prior, nf = pb->n_factors, nl = pb->n_landmarks;
    if (nfm < 2 || np < 0 || nf < 0 || nl < 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "bad marginalisation sizes");
    if (nfm > ctx->max_window + 2 || np > nfm) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d frames exceed window capacity", nfm);
    if (nf > ctx->max_factors || nl > ctx->max_factors) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "too many factors/landmarks");
    if (!pb->states || !pb->extr || !pb->sqrt_inv_cov) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null frame arrays");
    if (np > 0 && (!pb->prior_frames || !pb->prior_lin || !pb->prior_S || !pb->prior_f)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null prior arrays");
    if (nf > 0 && (!pb->tgt || !pb->ref || !pb->lm || !pb->tangent || !pb->z_ref || !pb->inv_depth)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null factor arrays");
    for (int i = 0; i < np; ++i)
        if (pb->prior_frames[i] < 0 || pb->prior_frames[i] >= nfm) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "prior frame out of range");
    for (int k = 0; k < nf; ++k) {
        if (pb->tgt[k] < 0 || pb->tgt[k] >= nfm || pb->ref[k] < 0 || pb->ref[k] >= nfm || pb->lm[k] < 0 || pb->lm[k] >= nl || pb->tgt[k] == pb->ref[k])
            return rdvio_fail(ctx, RDVIO_ERR_INVALID, "factor %d indexes out of range", k);
        if (k > 0 && pb->lm[k] < pb->lm[k - 1]) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "factors must be ordered by landmark");
    }
    // frame_indices: victim (0) last, the others shift down (:95-105)
    std::vector<int32_t> fidx(nfm);
    for (int i = 0; i < nfm; ++i) fidx[i] = (i == 0) ? nfm - 1 : i - 1;
    const int npairs = nfm * (nfm + 1) / 2;
    std::vector<int32_t> pair_fi(npairs), pair_fj(npairs), diag_pair(nfm), pair_index((size_t)nfm * nfm, -1);
    {
        int p = 0;
        for (int i = 0; i < nfm; ++i)
            for (int j = i; j < nfm; ++j) {
                pair_fi[p] = i; pair_fj[p] = j; pair_index[(size_t)i * nfm + j] = p;
                if (i == j) diag_pair[i] = p;
                ++p;
            }
    }
    std::vector<int32_t> pair_off(npairs + 1, 0);
    auto visit = [&](auto &&emit) {
        for (int k = 0; k < nf; ++k) {
            const int ct = fidx[pb->tgt[k]], cr = fidx[pb->ref[k]];
            emit(pair_index[(size_t)ct * nfm + ct], k * 4 + 0);
            emit(pair_index[(size_t)cr * nfm + cr], k * 4 + 3);
            if (ct < cr) emit(pair_index[(size_t)ct * nfm + cr], k * 4 + 2);
            else emit(pair_index[(size_t)cr * nfm + ct], k * 4 + 1);
        }
    };
    visit([&](int p, int) { pair_off[p + 1]++; });
    for (int p = 0; p < npairs; ++p) pair_off[p + 1] += pair_off[p];
    std::vector<int32_t> pair_item((size_t)std::max(pair_off[npairs], 1));
    {
        std::vector<int32_t> cur(pair_off.begin(), pair_off.end() - 1);
        visit([&](int p, int item) { pair_item[cur[p]++] = item; });
    }
    std::vector<int32_t> lm_first(std::max(nl, 1), 0), lm_count(std::max(nl, 1), 0);
    for (int k = 0; k < nf; ++k) {
        if (lm_count[pb->lm[k]] == 0) lm_first[pb->lm[k]] = k;
        lm_count[pb->lm[k]]++;
    }
    const int N = 15 * nfm, R = N - 15, D = 15 * np, NA = 6 * nfm;
    Pk P{(uint8_t *)ctx->marg_host, ctx->marg_bytes};
    double extr18[18];
    memcpy(extr18, pb->extr, 14 * sizeof(double));
    memcpy(extr18 + 14, pb->sqrt_inv_cov, 4 * sizeof(double));
    const size_t o_st = P.put(pb->states, (size_t)nfm * 16), o_ex = P.put(extr18, 18);
    const size_t o_pf = P.put(pb->prior_frames, (size_t)np), o_lin = P.put(pb->prior_lin, (size_t)np * 16);
    const size_t o_S = P.put(pb->prior_S, (size_t)D * D), o_f = P.put(pb->prior_f, (size_t)D);
    const size_t o_pre = P.put(pb->preint01, pb->preint01 ? (size_t)RDVIO_PREINT_SIZE : 0);
    const size_t o_z = P.put(pb->z_ref, (size_t)nl * 3), o_d = P.put(pb->inv_depth, (size_t)nl);
    const size_t o_t = P.put(pb->tgt, (size_t)nf), o_r = P.put(pb->ref, (size_t)nf), o_l = P.put(pb->lm, (size_t)nf);
    const size_t o_tan = P.put(pb->tangent, (size_t)nf * 9);
    const size_t o_fx = P.put(fidx.data(), (size_t)nfm), o_lf = P.put(lm_first.data(), (size_t)nl), o_lc = P.put(lm_count.data(), (size_t)nl);
    const size_t o_pi = P.put(pair_fi.data(), (size_t)npairs), o_pj = P.put(pair_fj.data(), (size_t)npairs);
    const size_t o_po = P.put(pair_off.data(), (size_t)npairs + 1), o_pit = P.put(pair_item.data(), pair_item.size());
    const size_t o_dp = P.put(diag_pair.data(), (size_t)nfm);
    if (!P.ok) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "marginalisation problem does not fit the staging buffer");
    const size_t in_bytes = P.off;
    Pk Sx{nullptr, ctx->marg_bytes};
    Sx.off = in_bytes;
    auto dd = [&](size_t n) { return Sx.reserve(std::max<size_t>(n, 1) * sizeof(double)); };
    const int Rb = (R + 14) / 15 * 15, Wn = std::max(R, Rb);
    const size_t s_em = dd(D), s_rm = dd(D), s_Jri = dd((size_t)np * 9), s_Lam = dd((size_t)D * D), s_le = dd(D);
    const size_t s_ep = dd(15), s_G = dd(450), s_rp = dd(15), s_Jp = dd(450);
    const size_t s_rf = dd((size_t)nf * 2), s_Jt = dd((size_t)nf * 12), s_Jr = dd((size_t)nf * 12), s_Jd = dd((size_t)nf * 2);
    const size_t s_A = dd((size_t)nl * NA), s_lmm = dd(nl), s_lmg = dd(nl), s_lmw = dd(nl);
    const size_t s_H = dd((size_t)N * N), s_eta = dd(N), s_Tm = dd((size_t)R * 15), s_Lr = dd((size_t)R * R), s_er = dd(R);
    const size_t s_Wk = dd((size_t)Wn * Wn), s_V = dd((size_t)R * R), s_cs = dd((size_t)4 * (R / 2 + 2) + R), s_yv = dd(Wn);
    const size_t s_nz = Sx.reserve((size_t)(2 * R + 2) * sizeof(int32_t));  // nz list + pivot `done` flags
    const size_t s_So = dd((size_t)R * R), s_fo = dd(R), s_lo = dd((size_t)(nfm - 1) * 16), s_Lo = dd((size_t)R * R), s_eo = dd(R), s_info = dd(4);
    if (!Sx.ok) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "marginalisation problem does not fit the device arena");
    uint8_t *B = (uint8_t *)ctx->marg_arena;
    MargWs &w = ctx->marg_ws;
    memset(&w, 0, sizeof w);
    w.nfm = nfm; w.np = np; w.D = D; w.nf = nf; w.nl = nl; w.npairs = npairs; w.has_pre = pb->preint01 ? 1 : 0;
#define DP(o) ((double *)(B + (o)))
#define IP(o) ((const int32_t *)(B + (o)))
    w.states = DP(o_st); w.extr = DP(o_ex); w.prior_frames = IP(o_pf); w.lin = DP(o_lin); w.S = DP(o_S); w.f = DP(o_f);
    w.preint = DP(o_pre); w.z_ref = DP(o_z); w.inv_depth = DP(o_d); w.tgt = IP(o_t); w.ref = IP(o_r); w.lm = IP(o_l);
    w.tangent = DP(o_tan); w.fidx = IP(o_fx); w.lm_first = IP(o_lf); w.lm_count = IP(o_lc); w.pair_fi = IP(o_pi);
    w.pair_fj = IP(o_pj); w.pair_off = IP(o_po); w.pair_item = IP(o_pit); w.diag_pair = IP(o_dp);
    w.e_m = DP(s_em); w.r_m = DP(s_rm); w.Jri = DP(s_Jri); w.Lam = DP(s_Lam); w.le = DP(s_le);
    w.e_p = DP(s_ep); w.G = DP(s_G); w.r_p = DP(s_rp); w.Jp = DP(s_Jp);
    w.r_f = DP(s_rf); w.Jt = DP(s_Jt); w.Jr = DP(s_Jr); w.Jd = DP(s_Jd);
    w.A = DP(s_A); w.lm_m = DP(s_lmm); w.lm_g = DP(s_lmg); w.lm_w = DP(s_lmw);
    w.H = DP(s_H); w.eta = DP(s_eta); w.Tm = DP(s_Tm); w.Lr = DP(s_Lr); w.er = DP(s_er); w.Wk = DP(s_Wk); w.V = DP(s_V);
    w.cs = DP(s_cs); w.yv = DP(s_yv); w.nz = (int32_t *)(B + s_nz);
    w.S_out = DP(s_So); w.f_out = DP(s_fo); w.lin_out = DP(s_lo); w.Lambda_out = DP(s_Lo); w.eta_out = DP(s_eo); w.info = DP(s_info);
#undef DP
#undef IP
    ctx->marg_in_bytes = in_bytes;
    ctx->marg_ready = true;
    return RDVIO_OK;
}

extern "C" {

int rdvio_hip_marginalize_upload(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb) {
    if (!ctx) return RDVIO_ERR_INVALID;
    ctx->marg_ready = false;
    RDVIO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (int rc = marg_prepare(ctx, pb)) return rc;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(ctx->marg_arena, ctx->marg_host, ctx->marg_in_bytes, hipMemcpyHostToDevice, ctx->stream));
    return RDVIO_OK;
}

int rdvio_hip_marginalize_resident(rdvio_hip_ctx *ctx, int force_eigen) {
    if (!ctx) return RDVIO_ERR_INVALID;
    if (!ctx->marg_ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no marginalisation problem uploaded");
    ctx->marg_ws.force_eigen = force_eigen ? 1 : 0;
    rdvio_launch_marginalize(ctx->stream, ctx->marg_ws);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}

int rdvio_hip_marginalize_fetch(rdvio_hip_ctx *ctx, double *S_out, double *f_out, double *lin_out, double *Lambda_out,
                                double *eta_out, int *used_fast_path) {
    if (!ctx) return RDVIO_ERR_INVALID;
    if (!ctx->marg_ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no marginalisation problem uploaded");
    const MargWs &w = ctx->marg_ws;
    const size_t R = (size_t)15 * (w.nfm - 1);
    hipStream_t st = ctx->stream;
    double info[4] = {0};
    if (S_out) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(S_out, w.S_out, R * R * sizeof(double), hipMemcpyDeviceToHost, st));
    if (f_out) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(f_out, w.f_out, R * sizeof(double), hipMemcpyDeviceToHost, st));
    if (lin_out) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(lin_out, w.lin_out, (size_t)(w.nfm - 1) * 16 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Lambda_out) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(Lambda_out, w.Lambda_out, R * R * sizeof(double), hipMemcpyDeviceToHost, st));
    if (eta_out) RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(eta_out, w.eta_out, R * sizeof(double), hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(info, w.info, sizeof info, hipMemcpyDeviceToHost, st));
    RDVIO_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (used_fast_path) *used_fast_path = (int)info[0];  // 1 plain Cholesky, 2 pivoted Cholesky, 0 eigen
    return RDVIO_OK;
}

int rdvio_hip_marginalize(rdvio_hip_ctx *ctx, const rdvio_marg_problem *pb, int force_eigen, double *S_out, double *f_out,
                          double *lin_out, double *Lambda_out, double *eta_out, int *used_fast_path) {
    if (int rc = rdvio_hip_marginalize_upload(ctx, pb)) return rc;
    if (int rc = rdvio_hip_marginalize_resident(ctx, force_eigen)) return rc;
    return rdvio_hip_marginalize_fetch(ctx, S_out, f_out, lin_out, Lambda_out, eta_out, used_fast_path);
}

}  // extern "C"
