// Host side of rdvio_hip_ba_solve: validates the SoA problem, builds the graph index structures
// (frame-pair factor lists, per-landmark factor ranges -- the SoA export of what refine_window assembles
// through pointers, /root/reference/src/rdvio/src/sliding_window_tracker.cpp:226-300), packs everything into
// one pinned blob, uploads it with a single copy and launches the persistent solver kernel.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ctx.hpp"
#include "solver_ws.hpp"

namespace {

struct Packer {
    uint8_t *base;
    size_t cap, off = 0;
    bool ok = true;
    template <class Tp>
    size_t put(const Tp *src, size_t n) {
        off = (off + 15) & ~(size_t)15;
        const size_t at = off, bytes = n * sizeof(Tp);
        if (at + bytes > cap) {
            ok = false;
            return at;
        }
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

int rdvio_ba_prepare(rdvio_hip_ctx *ctx, rdvio_hip_ctx::BaSlot &slot, const rdvio_ba_problem *pb, size_t cap, bool with_marg_tail) {
    if (!pb) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null BA problem");
    const int nfr = pb->n_frames, nl = pb->n_landmarks, nf = pb->n_factors, nrot = pb->n_rot, npre = pb->n_preint,
              np = pb->n_prior;
    if (nfr <= 0 || nl < 0 || nf < 0 || nrot < 0 || npre < 0 || np < 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "negative BA sizes");
    if (nfr > 64) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "at most 64 frames per solve (32 of them free)");
    if (nf > ctx->max_factors || nl > ctx->max_factors || nrot > ctx->max_factors)
        return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d factors / %d landmarks exceed capacity %d", nf, nl, ctx->max_factors);
    if (npre > nfr + 8 || np > nfr) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "too many preintegration factors / prior frames");
    // the prior error vector e (15 per prior frame) is staged in the kernel's LDS operand xv[RDVIO_SOLVER_XV]
    if (15 * np > RDVIO_SOLVER_XV) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "at most %d frames in the marginalisation prior", RDVIO_SOLVER_XV / 15);
    if (!pb->states || !pb->frame_fixed || !pb->extr || !pb->sqrt_inv_cov) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null frame arrays");
    if (nl > 0 && (!pb->z_ref || !pb->inv_depth || !pb->lm_fixed)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null landmark arrays");
    if (nf > 0 && (!pb->tgt || !pb->ref || !pb->lm || !pb->tangent)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null factor arrays");
    if (nrot > 0 && (!pb->rot_tgt || !pb->rot_ref || !pb->rot_zref || !pb->rot_tangent)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null rotation-prior arrays");
    const int njobs = pb->n_pre_jobs;
    if (njobs != 0 && (njobs != npre || with_marg_tail)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "n_pre_jobs must be 0 or n_preint");
    if (npre > 0 && (!pb->pre_i || !pb->pre_j || (!pb->preint && njobs == 0))) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null preintegration arrays");
    int job_samples = 0;
    if (njobs > 0) {
        if (!pb->job_seg_off || !pb->job_imu || !pb->job_par || !pb->job_noise || !pb->job_preint_out) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null preintegration job arrays");
        if (pb->job_seg_off[0] != 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "job_seg_off[0] must be 0");
        for (int k = 0; k < njobs; ++k)
            if (pb->job_seg_off[k + 1] < pb->job_seg_off[k]) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "job_seg_off must be non-decreasing");
        job_samples = pb->job_seg_off[njobs];
        if (job_samples > ctx->pre_max_samples) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d samples exceed capacity %d", job_samples, ctx->pre_max_samples);
    }
    if (np > 0 && (!pb->prior_frames || !pb->prior_lin || !pb->prior_S || !pb->prior_f)) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "null prior arrays");
    // every index is range-checked here so that the kernel can never read out of bounds
    for (int k = 0; k < nf; ++k) {
        if (pb->tgt[k] < 0 || pb->tgt[k] >= nfr || pb->ref[k] < 0 || pb->ref[k] >= nfr || pb->lm[k] < 0 || pb->lm[k] >= nl)
            return rdvio_fail(ctx, RDVIO_ERR_INVALID, "factor %d indexes out of range", k);
        if (k > 0 && pb->lm[k] < pb->lm[k - 1]) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "factors must be ordered by landmark (factor %d)", k);
    }
    // a track is anchored in ONE frame (Track::first_frame) and observed at most once per frame (track.h keypoint_refs)
    for (int k = 0; k < nf;) {
        int e = k;
        uint64_t seen = 0;  // one bit per frame (nfr <= 64)
        while (e < nf && pb->lm[e] == pb->lm[k]) {
            if (pb->ref[e] != pb->ref[k]) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "landmark %d has two anchor frames", pb->lm[k]);
            if (pb->tgt[e] == pb->ref[e] || (seen >> pb->tgt[e]) & 1ull) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "landmark %d observed twice in one frame", pb->lm[k]);
            seen |= 1ull << pb->tgt[e];
            ++e;
        }
        k = e;
    }
    for (int k = 0; k < nrot; ++k)
        if (pb->rot_tgt[k] < 0 || pb->rot_tgt[k] >= nfr || pb->rot_ref[k] < 0 || pb->rot_ref[k] >= nfr)
            return rdvio_fail(ctx, RDVIO_ERR_INVALID, "rotation prior %d indexes out of range", k);
    for (int k = 0; k < npre; ++k)
        if (pb->pre_i[k] < 0 || pb->pre_i[k] >= nfr || pb->pre_j[k] < 0 || pb->pre_j[k] >= nfr)
            return rdvio_fail(ctx, RDVIO_ERR_INVALID, "preintegration factor %d indexes out of range", k);
    for (int i = 0; i < np; ++i)
        if (pb->prior_frames[i] < 0 || pb->prior_frames[i] >= nfr) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "prior frame %d out of range", i);

    // ---- graph index structures
    std::vector<int32_t> fcol(nfr);
    int nfree = 0;
    for (int i = 0; i < nfr; ++i) fcol[i] = (pb->frame_fixed[i] == 1) ? -1 : nfree++;
    if (nfree > 32) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "at most 32 free frames per solve");
    if (nfree > ctx->max_window + 2) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "%d free frames exceed window capacity %d", nfree, ctx->max_window + 2);
    const int N = 15 * nfree, D = 15 * np;
    const int npairs = nfree * (nfree + 1) / 2;
    std::vector<int32_t> pair_fi(npairs), pair_fj(npairs), diag_pair(std::max(nfree, 1));
    std::vector<int> pair_index((size_t)std::max(nfree * nfree, 1), -1);
    {
        int p = 0;
        for (int i = 0; i < nfree; ++i)
            for (int j = i; j < nfree; ++j) {
                pair_fi[p] = i;
                pair_fj[p] = j;
                pair_index[(size_t)i * nfree + j] = p;
                if (i == j) diag_pair[i] = p;
                ++p;
            }
    }
    // Factor groups: every factor belongs to exactly ONE group = the unordered pair (lo, hi) of its free frames
    // (lo == hi when only one of its two frames is free).  Its record [J_lo | J_hi | r] is written in group order at
    // linearisation time; one MFMA accumulation per group then yields the four blocks (lo,lo) (lo,hi) (hi,lo) (hi,hi)
    // and the two gradient segments at once.
    std::vector<int32_t> grp_off(npairs + 1, 0), gslot(std::max(nf, 1), -1), gflip(std::max(nf, 1), 0);
    auto group_of = [&](int k, int &flip) {
        const int ct = fcol[pb->tgt[k]], cr = fcol[pb->ref[k]];
        flip = 0;
        if (ct < 0 && cr < 0) return -1;
        if (ct >= 0 && cr >= 0) {
            flip = cr < ct;  // first slot = J of the lower-numbered frame: Jr when the anchor comes first
            return pair_index[(size_t)std::min(ct, cr) * nfree + std::max(ct, cr)];
        }
        flip = ct < 0;       // only the anchor is free: first slot = Jr, second slot = 0
        const int c = ct >= 0 ? ct : cr;
        return pair_index[(size_t)c * nfree + c];
    };
    for (int k = 0; k < nf; ++k) {
        int flip;
        const int g = group_of(k, flip);
        if (g >= 0) grp_off[g + 1]++;
    }
    for (int p = 0; p < npairs; ++p) grp_off[p + 1] += grp_off[p];
    {
        std::vector<int32_t> cur(grp_off.begin(), grp_off.end() - 1);
        for (int k = 0; k < nf; ++k) {
            int flip;
            const int g = group_of(k, flip);
            if (g < 0) continue;
            gslot[k] = cur[g]++;
            gflip[k] = flip;
        }
    }
    const int nrec = grp_off[npairs];
    // preintegration sources per block of the (block-)band and per gradient block; at most two factors touch a block
    std::vector<int32_t> band_src((size_t)std::max(nfree, 1) * 6, -1), g_src((size_t)std::max(nfree, 1) * 2, -1), pcol(std::max(nfree, 1), -1);
    for (int k = 0; k < npre; ++k) {
        const int cs[2] = {fcol[pb->pre_i[k]], fcol[pb->pre_j[k]]};
        for (int x = 0; x < 2; ++x) {
            if (cs[x] < 0) continue;
            for (int y = 0; y < 2; ++y) {
                if (cs[y] < 0) continue;
                const int which = cs[y] - cs[x] + 1;
                if (which < 0 || which > 2) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "preintegration factor %d links non-adjacent free frames", k);
                int32_t *dst = &band_src[((size_t)cs[x] * 3 + which) * 2];
                if (dst[0] < 0) dst[0] = k * 4 + x * 2 + y;
                else if (dst[1] < 0) dst[1] = k * 4 + x * 2 + y;
                else return rdvio_fail(ctx, RDVIO_ERR_INVALID, "more than two preintegration factors on one block");
            }
            int32_t *gd = &g_src[(size_t)cs[x] * 2];
            if (gd[0] < 0) gd[0] = k * 2 + x;
            else if (gd[1] < 0) gd[1] = k * 2 + x;
            else return rdvio_fail(ctx, RDVIO_ERR_INVALID, "more than two preintegration factors on one frame");
        }
    }
    for (int i = 0; i < np; ++i) {
        const int c = fcol[pb->prior_frames[i]];
        if (c < 0) continue;
        if (pcol[c] >= 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "frame listed twice in the prior");
        pcol[c] = i;
    }
    std::vector<int32_t> lm_first(std::max(nl, 1), 0), lm_count(std::max(nl, 1), 0);
    for (int k = 0; k < nf; ++k) {
        if (lm_count[pb->lm[k]] == 0) lm_first[pb->lm[k]] = k;
        lm_count[pb->lm[k]]++;
    }
    int n_lfree = 0;
    for (int l = 0; l < nl; ++l)
        if (lm_count[l] > 0 && !pb->lm_fixed[l]) n_lfree++;

    // ---- pack inputs into the pinned blob (one H2D copy), then carve device scratch behind it
    Packer P{(uint8_t *)slot.host, cap};
    double extr18[18];
    memcpy(extr18, pb->extr, 14 * sizeof(double));
    memcpy(extr18 + 14, pb->sqrt_inv_cov, 4 * sizeof(double));
    const size_t o_states = P.put(pb->states, (size_t)nfr * 16);
    const size_t o_invd = P.put(pb->inv_depth, (size_t)nl);
    const size_t o_lmfixed = P.put(pb->lm_fixed, (size_t)nl);
    const size_t o_extr = P.put(extr18, 18);
    const size_t o_zref = P.put(pb->z_ref, (size_t)nl * 3);
    const size_t o_tgt = P.put(pb->tgt, (size_t)nf), o_ref = P.put(pb->ref, (size_t)nf), o_lm = P.put(pb->lm, (size_t)nf);
    const size_t o_tan = P.put(pb->tangent, (size_t)nf * 9);
    const size_t o_rt = P.put(pb->rot_tgt, (size_t)nrot), o_rr = P.put(pb->rot_ref, (size_t)nrot);
    const size_t o_rz = P.put(pb->rot_zref, (size_t)nrot * 3), o_rtan = P.put(pb->rot_tangent, (size_t)nrot * 9);
    const size_t o_pi = P.put(pb->pre_i, (size_t)npre), o_pj = P.put(pb->pre_j, (size_t)npre);
    const size_t o_pre = P.put(njobs > 0 ? (const double *)nullptr : pb->preint, (size_t)npre * RDVIO_PREINT_SIZE);
    const size_t o_joff = P.put(pb->job_seg_off, njobs > 0 ? (size_t)njobs + 1 : 0), o_jimu = P.put(pb->job_imu, (size_t)job_samples * 7);
    const size_t o_jpar = P.put(pb->job_par, (size_t)njobs * 7), o_jnoise = P.put(pb->job_noise, njobs > 0 ? 36 : 0);
    const size_t o_pf = P.put(pb->prior_frames, (size_t)np), o_lin = P.put(pb->prior_lin, (size_t)np * 16);
    const size_t o_S = P.put(pb->prior_S, (size_t)D * D), o_f = P.put(pb->prior_f, (size_t)D);
    const size_t o_user0 = P.put((const double *)nullptr, (size_t)nfr * 16);   // (rdvio_hip_ba_linearize fills it)
    const size_t o_fcol = P.put(fcol.data(), (size_t)nfr);
    const size_t o_ffix = P.put(pb->frame_fixed, (size_t)nfr);
    const size_t o_lmf = P.put(lm_first.data(), (size_t)nl), o_lmc = P.put(lm_count.data(), (size_t)nl);
    const size_t o_pfi = P.put(pair_fi.data(), (size_t)npairs), o_pfj = P.put(pair_fj.data(), (size_t)npairs);
    const size_t o_goff = P.put(grp_off.data(), (size_t)npairs + 1);
    const size_t o_dp = P.put(diag_pair.data(), diag_pair.size());
    const size_t o_gslot = P.put(gslot.data(), (size_t)nf), o_gflip = P.put(gflip.data(), (size_t)nf);
    const size_t o_band = P.put(band_src.data(), band_src.size()), o_gsrc = P.put(g_src.data(), g_src.size()), o_pcol = P.put(pcol.data(), pcol.size());
    if (!P.ok) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "BA problem does not fit the context's staging buffer");
    const size_t in_bytes = P.off;

    // scratch (device only): bump-allocate behind the inputs
    Packer Sx{nullptr, cap};
    Sx.off = in_bytes;
    auto dd = [&](size_t n) { return Sx.reserve(std::max<size_t>(n, 1) * sizeof(double)); };
    // x | xd | summary are adjacent: the result travels back with one copy
    const size_t s_x = dd((size_t)nfr * 16), s_xd = dd(nl), s_sum = dd(80);  // summary[0..7] + diagnostic phase stamps
    const size_t s_sync = dd(8), s_partial = dd(RDVIO_MAX_SOLVER_WGS);
    const size_t s_xc = dd((size_t)nfr * 16), s_xdc = dd(nl), s_user = dd((size_t)nfr * 16);
    const size_t s_lfree = Sx.reserve(std::max(nl, 1));
    const size_t s_fac = dd((size_t)nf * RDVIO_FAC_STRIDE), s_prec = dd((size_t)nrec * RDVIO_REC_STRIDE), s_GP = dd((size_t)npairs * 256);
    const size_t s_PP = dd((size_t)npre * 900), s_Pg = dd((size_t)npre * 30), s_ST = dd((size_t)D * D);
    const size_t s_rr = dd((size_t)nrot * 2), s_Jro = dd((size_t)nrot * 6);
    const size_t s_ep = dd((size_t)npre * 15), s_G = dd((size_t)npre * 450), s_rp = dd((size_t)npre * 15), s_cp = dd((size_t)npre * 15), s_Jp = dd((size_t)npre * 450);
    const size_t s_em = dd(D), s_rm = dd(D), s_cm = dd(D), s_Jri = dd((size_t)np * 9), s_Lam = dd((size_t)D * D), s_eta0 = dd(D), s_le = dd(D), s_Ex = dd(D);
    const size_t s_H = dd((size_t)N * N), s_Sm = dd((size_t)(N + 1) * N), s_g = dd(N), s_yp = dd(N);  // Sm: + the right-hand-side row
    const size_t s_Cm = dd((size_t)(6 * nfree + 2) * (6 * nfree + 2));
    // the helper team: from helper_min_factors (4096) factors on; windows whose reduced system is factored in global memory
    // (more than RDVIO_LDS_CHOL_MAX_FRAMES free frames: no speculative trial steps to lose) already from 1500 factors
    // (3183 factors, 13 free frames: 3.85 -> 3.16 ms)
    const bool global_chol = nfree > RDVIO_LDS_CHOL_MAX_FRAMES;
    const int team_from = global_chol ? std::min(ctx->helper_min_factors, 1500) : ctx->helper_min_factors;
    const int n_wg = (!with_marg_tail && nf >= team_from && ctx->solver_wgs > 1) ? ctx->solver_wgs : 1;
    const size_t s_Cmp = dd(n_wg > 1 ? (size_t)n_wg * (6 * nfree + 2) * (6 * nfree + 2) : 1);
    const size_t s_lmm = dd(nl), s_lmg = dd(nl), s_lmw = dd(nl), s_A = dd((size_t)nl * (6 * nfree + 2)), s_yl = dd(nl);
    const size_t s_sigp = dd(N), s_sigl = dd(nl), s_dgp = dd(N), s_dgl = dd(nl), s_grp = dd(N), s_grl = dd(nl), s_gnp = dd(N), s_gnl = dd(nl), s_tp = dd(N), s_tl = dd(nl);
    // marginalisation tail (victim = frame 0): R = N - 15 retained rows
    const int R = N >= 15 ? N - 15 : 0, Rb = (R + 14) / 15 * 15, Wn = std::max(R, Rb);
    size_t s_mTm = 0, s_mLr = 0, s_mer = 0, s_mWk = 0, s_mV = 0, s_mcs = 0, s_myv = 0, s_mnz = 0, s_So = 0, s_fo = 0, s_lo = 0, s_Lo = 0, s_eo = 0, s_info = 0;
    if (with_marg_tail) {
        s_mTm = dd((size_t)R * 15); s_mLr = dd((size_t)R * R); s_mer = dd(R); s_mWk = dd((size_t)Wn * Wn); s_mV = dd((size_t)R * R);
        s_mcs = dd((size_t)4 * (R / 2 + 2) + R); s_myv = dd(Wn);
        s_mnz = Sx.reserve((size_t)(2 * R + 2) * sizeof(int32_t));  // nz list + pivot `done` flags
        // S | f | lin | info adjacent: the new prior travels back with one copy
        s_So = dd((size_t)R * R); s_fo = dd(R); s_lo = dd((size_t)(nfr - 1) * 16); s_info = dd(4); s_Lo = dd((size_t)R * R); s_eo = dd(R);
    }
    if (!Sx.ok) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "BA problem does not fit the context's device arena");

    uint8_t *B = (uint8_t *)slot.arena;
    SolverWs &w = slot.ws;
    memset(&w, 0, sizeof w);
    w.nfr = nfr; w.nl = nl; w.nf = nf; w.nrot = nrot; w.npre = npre; w.np = np; w.D = D; w.nfree = nfree; w.N = N;
    w.npairs = npairs; w.n_lfree_hint = n_lfree; w.nrec = nrec;
    {
        w.lds_chol = (nfree > 0 && nfree <= RDVIO_LDS_CHOL_MAX_FRAMES) ? 1 : 0;
        w.lds_bytes = 0;
    }
#define DP(off) ((double *)(B + (off)))
#define IP(off) ((const int32_t *)(B + (off)))
    w.lm_fixed = B + o_lmfixed;
    w.extr = DP(o_extr); w.z_ref = DP(o_zref);
    w.tgt = IP(o_tgt); w.ref = IP(o_ref); w.lm = IP(o_lm); w.tangent = DP(o_tan);
    w.rot_tgt = IP(o_rt); w.rot_ref = IP(o_rr); w.rot_zref = DP(o_rz); w.rot_tangent = DP(o_rtan);
    w.pre_i = IP(o_pi); w.pre_j = IP(o_pj); w.preint = DP(o_pre);
    w.prior_frames = IP(o_pf); w.lin = DP(o_lin); w.S = DP(o_S); w.f = DP(o_f);
    w.fcol = IP(o_fcol); w.frame_fixed = B + o_ffix; w.lm_first = IP(o_lmf); w.lm_count = IP(o_lmc);
    w.pair_fi = IP(o_pfi); w.pair_fj = IP(o_pfj); w.grp_off = IP(o_goff); w.diag_pair = IP(o_dp);
    w.gslot = IP(o_gslot); w.gflip = IP(o_gflip); w.band_src = IP(o_band); w.g_src = IP(o_gsrc); w.pcol = IP(o_pcol);
    w.x0 = DP(o_states); w.xd0 = DP(o_invd);
    w.x = DP(s_x); w.xd = DP(s_xd); w.xc = DP(s_xc); w.xdc = DP(s_xdc); w.user = DP(s_user); w.lfree = B + s_lfree;
    w.fac = DP(s_fac); w.prec = DP(s_prec); w.GP = DP(s_GP); w.PP = DP(s_PP); w.Pg = DP(s_Pg); w.ST = DP(s_ST); w.r_r = DP(s_rr); w.Jro = DP(s_Jro);
    w.e_p = DP(s_ep); w.G = DP(s_G); w.r_p = DP(s_rp); w.c_p = DP(s_cp); w.Jp = DP(s_Jp);
    w.e_m = DP(s_em); w.r_m = DP(s_rm); w.c_m = DP(s_cm); w.Jri = DP(s_Jri); w.Lam = DP(s_Lam); w.eta0 = DP(s_eta0); w.le = DP(s_le); w.Ex = DP(s_Ex);
    w.H = DP(s_H); w.Sm = DP(s_Sm); w.g = DP(s_g); w.yp = DP(s_yp); w.Cm = DP(s_Cm); w.Cmp = DP(s_Cmp);
    w.lm_m = DP(s_lmm); w.lm_g = DP(s_lmg); w.lm_w = DP(s_lmw); w.A = DP(s_A); w.yl = DP(s_yl);
    w.sig_p = DP(s_sigp); w.sig_l = DP(s_sigl); w.diag_p = DP(s_dgp); w.diag_l = DP(s_dgl); w.grad_p = DP(s_grp); w.grad_l = DP(s_grl);
    w.gn_p = DP(s_gnp); w.gn_l = DP(s_gnl); w.tp = DP(s_tp); w.tl = DP(s_tl);
    w.summary = DP(s_sum);
    w.sync = (unsigned *)(B + s_sync); w.partial = DP(s_partial);
    // helper workgroups: a command round trip costs ~8 us (L2 atomics, barriers, re-staging the states), one evaluation
    // of F factors on the leader alone ~F / 90 us -- measured break-even near 2000 factors; the marginalisation kernel
    // (a single linearisation) never uses them
    w.n_wg = n_wg;
    if (with_marg_tail) {
        w.no_loss = 1;
        w.m_Tm = DP(s_mTm); w.m_Lr = DP(s_mLr); w.m_er = DP(s_mer); w.m_Wk = DP(s_mWk); w.m_V = DP(s_mV); w.m_cs = DP(s_mcs);
        w.m_yv = DP(s_myv); w.m_nz = (int32_t *)(B + s_mnz);
        w.S_out = DP(s_So); w.f_out = DP(s_fo); w.lin_out = DP(s_lo); w.Lambda_out = DP(s_Lo); w.eta_out = DP(s_eo); w.m_info = DP(s_info);
    }
#undef DP
#undef IP
    {
        // algorithmic FP64 flops from SURVEY.md 8(d)'s per-unit figures: per linearisation 0.5 kflop per reprojection factor
        // + 15 kflop per preintegration factor + 2 D^2 16 W for the prior's Jacobian product + 182 MAC per factor and 36 m^2 MAC
        // per free landmark with m observations (normal equations + landmark Schur) + n^3 / 3 for the reduced Cholesky (n = 15 x
        // free frames); per trial-step cost evaluation 0.2 kflop per reprojection factor + 2 kflop per preintegration factor +
        // 2 D^2 for the prior
        double m2 = 0.0;
        for (int l = 0; l < nl; ++l)
            if (lm_count[l] > 0 && !pb->lm_fixed[l]) m2 += (double)lm_count[l] * lm_count[l];
        const double F = nf, P = npre, Dd = D, Wp = np, n3 = (double)N * N * N;
        slot.flops_lin = 500.0 * F + 15000.0 * P + 2.0 * Dd * Dd * 16.0 * Wp + 2.0 * 182.0 * F + 2.0 * 36.0 * m2 + n3 / 3.0;
        slot.flops_eval = 200.0 * F + 2000.0 * P + 2.0 * Dd * Dd;
    }
    slot.user0_off = o_user0;
    slot.n_jobs = njobs;
    slot.job_off = (const int32_t *)(B + o_joff);
    slot.job_imu = (const double *)(B + o_jimu);
    slot.job_par = (const double *)(B + o_jpar);
    slot.job_noise = (const double *)(B + o_jnoise);
    slot.job_out_host = pb->job_preint_out;
    slot.in_states_off = o_states;
    slot.in_invd_off = o_invd;
    slot.in_bytes = in_bytes;
    slot.host_bytes = cap;
    slot.ready = true;
    return RDVIO_OK;
}

static bool bad_slot(int slot) { return slot < 0 || slot >= RDVIO_BA_SLOTS; }

extern "C" {

int rdvio_hip_ba_upload(rdvio_hip_ctx *ctx, int slot, const rdvio_ba_problem *pb) {
    if (!ctx || bad_slot(slot)) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::BaSlot &S = ctx->ba[slot];
    S.ready = false;
    // the pinned blob may still be in flight from a previous upload on this stream
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[RDVIO_LANE_SOLVER]));
    if (int rc = rdvio_ba_prepare(ctx, S, pb, ctx->ba_arena_bytes, false)) return rc;
    S.ws.chain_src = nullptr;
    S.ws.chain_frame = 0;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(S.arena, S.host, S.in_bytes, hipMemcpyHostToDevice, ctx->lane[RDVIO_LANE_SOLVER]));
    // fused PreIntegrator::integrate: the records of this solve's preintegration factors, straight into their slots of the arena
    if (S.n_jobs > 0)
        if (int rc = rdvio_launch_preintegrate(ctx, ctx->lane[RDVIO_LANE_SOLVER], S.n_jobs, S.job_off, S.job_imu, S.job_par, S.job_noise, 1, 1,
                                               const_cast<double *>(S.ws.preint)))
            return rc;
    return RDVIO_OK;
}

int rdvio_hip_ba_upload_chained(rdvio_hip_ctx *ctx, int slot, const rdvio_ba_problem *pb, int from_slot, int from_frame, int to_frame) {
    if (!ctx || bad_slot(slot) || bad_slot(from_slot) || from_slot == slot) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::BaSlot &S = ctx->ba[slot], &F = ctx->ba[from_slot];
    if (!F.ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no BA problem uploaded in slot %d to chain from", from_slot);
    if (!pb || from_frame < 0 || from_frame >= F.ws.nfr || to_frame < 0 || to_frame >= pb->n_frames)
        return rdvio_fail(ctx, RDVIO_ERR_INVALID, "chained frame index out of range");
    S.ready = false;
    if (!ctx->chain_stream) {
        RDVIO_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->chain_stream, hipStreamNonBlocking));
        RDVIO_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->chain_ev, hipEventDisableTiming));
    }
    // only THIS slot's pinned blob must have left for the device (the solve of the other slot may still be running: that is the point)
    if (S.up_ev) RDVIO_HIP_CHECK(ctx, hipEventSynchronize(S.up_ev));
    else RDVIO_HIP_CHECK(ctx, hipEventCreateWithFlags(&S.up_ev, hipEventDisableTiming));
    if (int rc = rdvio_ba_prepare(ctx, S, pb, ctx->ba_arena_bytes, false)) return rc;
    // Inputs and preintegration jobs on the side stream, BESIDE the solve this one continues (copies and kernels that alternate on
    // one stream cost an engine hand-over each, and the jobs would sit between the two solves); the solver lane only waits for them.
    hipStream_t side = ctx->chain_stream;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(S.arena, S.host, S.in_bytes, hipMemcpyHostToDevice, side));
    RDVIO_HIP_CHECK(ctx, hipEventRecord(S.up_ev, side));
    if (S.n_jobs > 0)
        if (int rc = rdvio_launch_preintegrate(ctx, side, S.n_jobs, S.job_off, S.job_imu, S.job_par, S.job_noise, 1, 1, const_cast<double *>(S.ws.preint)))
            return rc;
    RDVIO_HIP_CHECK(ctx, hipEventRecord(ctx->chain_ev, side));
    RDVIO_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->lane[RDVIO_LANE_SOLVER], ctx->chain_ev, 0));
    // the initial value of frame to_frame = the other solve's result for its frame from_frame: read by this solve's own setup (it
    // runs behind that solve on the solver lane)
    S.ws.chain_src = F.ws.x + 16 * (size_t)from_frame;
    S.ws.chain_frame = to_frame;
    return RDVIO_OK;
}

int rdvio_hip_ba_solve_resident(rdvio_hip_ctx *ctx, int slot, int max_iterations) {
    if (!ctx || bad_slot(slot)) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::BaSlot &S = ctx->ba[slot];
    if (!S.ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no BA problem uploaded in slot %d", slot);
    if (max_iterations < 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "negative iteration limit");
    SolverWs &w = S.ws;
    w.max_iter = max_iterations;
    S.retried = false;
    S.down_enqueued = false;
    // several contexts live on this device (sequences sharing a GPU): a team's helper workgroups are not guaranteed a compute unit
    // each, so large solves stay on one workgroup there (slower per solve, never a time-out)
    if (w.n_wg > 1 && rdvio_live_contexts(ctx->device) > 1) {
        const char *force = getenv("RDVIO_TEST_FORCE_TEAM");   // test switch: the team launch although other contexts are alive
        if (!(force && force[0] == '1')) w.n_wg = 1;
    }
    // the kernel (re)starts from the uploaded initial values (SolverWs::x0 / xd0): no host traffic, no extra copies
    if (w.n_wg > 1) RDVIO_HIP_CHECK(ctx, hipMemsetAsync(w.sync, 0, 8 * sizeof(double), ctx->lane[RDVIO_LANE_SOLVER]));
    // (k > 1 times PAIRS of consecutive launches -- the two solves of a frame -- of every k-th pair: a stride over single launches
    // would sample the alternating localisation / window launches unevenly)
    // and the pairs are picked by a hash of their index, not by a stride: keyframes come every sixth frame or so, and any fixed
    // stride would sample the window solves unevenly (every fourth frame never met one)
    {
        const unsigned long pair = (unsigned long)(ctx->kt_seen++ / 2);
        const unsigned h = (unsigned)((pair * 2654435761ul) >> 13);
        S.timed_launch = ctx->kernel_timing == 1 || (ctx->kernel_timing > 1 && h % (unsigned)ctx->kernel_timing == 0);
    }
    if (S.timed_launch) {
        if (!S.ev0) {
            RDVIO_HIP_CHECK(ctx, hipEventCreate(&S.ev0));
            RDVIO_HIP_CHECK(ctx, hipEventCreate(&S.ev1));
        }
    }
    // (timed: the two events ride on the launch itself -- the dispatch's own start / stop timestamps; two hipEventRecord calls around
    // it put marker packets on the lane and cost the pipeline up to a tenth of its rate)
    rdvio_launch_ba_solve(ctx->lane[RDVIO_LANE_SOLVER], w, S.timed_launch ? S.ev0 : nullptr, S.timed_launch ? S.ev1 : nullptr);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    return RDVIO_OK;
}

// the device-to-host copies of a slot's result, without the wait: a caller with several solves in flight (chained solves) enqueues
// all of them and waits once
int rdvio_hip_ba_fetch_enqueue(rdvio_hip_ctx *ctx, int slot) {
    if (!ctx || bad_slot(slot)) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::BaSlot &S = ctx->ba[slot];
    if (!S.ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no BA problem uploaded in slot %d", slot);
    if (S.down_enqueued) return RDVIO_OK;
    SolverWs &w = S.ws;
    // x | xd | summary[0..7] in one device-to-host copy into the slot's pinned blob (behind the uploaded inputs)
    const size_t n_out = (size_t)(w.summary + 8 - w.x);
    const size_t host_off = (S.in_bytes + 63) & ~(size_t)63;
    if (host_off + n_out * sizeof(double) > S.host_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "result does not fit the pinned blob");
    double *down = (double *)((uint8_t *)S.host + host_off);
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down, w.x, n_out * sizeof(double), hipMemcpyDeviceToHost, ctx->lane[RDVIO_LANE_SOLVER]));
    double *down_pre = down + ((n_out + 7) & ~(size_t)7);
    const size_t pre_doubles = (size_t)S.n_jobs * RDVIO_PREINT_SIZE;
    if (S.n_jobs > 0) {
        if ((size_t)((uint8_t *)(down_pre + pre_doubles) - (uint8_t *)S.host) > S.host_bytes) return rdvio_fail(ctx, RDVIO_ERR_CAPACITY, "records do not fit the pinned blob");
        RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(down_pre, w.preint, pre_doubles * sizeof(double), hipMemcpyDeviceToHost, ctx->lane[RDVIO_LANE_SOLVER]));
    }
    S.down_enqueued = true;
    return RDVIO_OK;
}

int rdvio_hip_ba_fetch(rdvio_hip_ctx *ctx, int slot, double *states_out, double *inv_depth_out, rdvio_ba_summary *summary) {
    if (!ctx || bad_slot(slot)) return RDVIO_ERR_INVALID;
    rdvio_hip_ctx::BaSlot &S = ctx->ba[slot];
    if (!S.ready) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "no BA problem uploaded in slot %d", slot);
    SolverWs &w = S.ws;
    if (int rc = rdvio_hip_ba_fetch_enqueue(ctx, slot)) return rc;
    S.down_enqueued = false;
    const size_t n_out = (size_t)(w.summary + 8 - w.x);
    const size_t host_off = (S.in_bytes + 63) & ~(size_t)63;
    double *down = (double *)((uint8_t *)S.host + host_off);
    double *down_pre = down + ((n_out + 7) & ~(size_t)7);
    const size_t pre_doubles = (size_t)S.n_jobs * RDVIO_PREINT_SIZE;
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[RDVIO_LANE_SOLVER]));
    if (S.n_jobs > 0 && S.job_out_host) memcpy(S.job_out_host, down_pre, pre_doubles * sizeof(double));
    if (states_out) memcpy(states_out, down, (size_t)w.nfr * 16 * sizeof(double));
    if (inv_depth_out && w.nl > 0) memcpy(inv_depth_out, down + (w.xd - w.x), (size_t)w.nl * sizeof(double));
    const double *sum = down + (w.summary - w.x);
    if (S.timed_launch) {   // the lane has just been waited for: both events are complete
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, S.ev0, S.ev1) == hipSuccess) {
            ctx->kt_launches += 1.0;
            ctx->kt_ms += (double)ms;
            ctx->kt_flops += (sum[1] + 1.0) * S.flops_lin + sum[0] * S.flops_eval;
            ctx->kt_iterations += sum[0];
        }
        S.timed_launch = false;
    }
    if (summary) {
        summary->iterations = (int32_t)sum[0];
        summary->successful_steps = (int32_t)sum[1];
        summary->initial_cost = sum[2];
        summary->final_cost = sum[3];
        summary->termination = (int32_t)sum[4];
    }
    // a helper workgroup that never answered (solver_kernels.hip, collect_partials): the outputs hold the last accepted
    // point and termination FAILURE, and the caller is told
    if (const unsigned ug = rdvio_ug_violations()) return rdvio_fail(ctx, RDVIO_ERR_HIP, "RDVIO_UG was applied to an LDS address %u times (checking build)", ug);
    if (sum[6] != 0.0) return rdvio_fail(ctx, RDVIO_ERR_HIP, "solver wavefronts disagreed on the trust-region loop's scalars (internal error)");
    if (sum[5] != 0.0) {
        // A helper workgroup did not answer within the spin limit -- it found no compute unit (a device shared with other work) or
        // was muted by the test switch.  The solve is not lost: it is repeated once from the uploaded initial values on the leader
        // alone (the same kernel, n_wg = 1); only a second failure is reported.
        if (w.n_wg > 1 && !S.retried) {
            const int iters = w.max_iter;
            w.n_wg = 1;
            if (int rc = rdvio_hip_ba_solve_resident(ctx, slot, iters)) return rc;
            S.retried = true;
            ctx->team_retries++;
            return rdvio_hip_ba_fetch(ctx, slot, states_out, inv_depth_out, summary);
        }
        return rdvio_fail(ctx, RDVIO_ERR_TIMEOUT, "a solver helper workgroup did not answer within the spin limit");
    }
    return RDVIO_OK;
}

int rdvio_hip_ba_linearize(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, const double *lin_states, int robust_loss, rdvio_ba_linearization *out) {
    if (!ctx || !pb || !out) return RDVIO_ERR_INVALID;
    if (pb->n_pre_jobs != 0) return rdvio_fail(ctx, RDVIO_ERR_INVALID, "rdvio_hip_ba_linearize takes integrated records");
    rdvio_hip_ctx::BaSlot &S = ctx->ba[0];
    S.ready = false;
    hipStream_t st = ctx->lane[RDVIO_LANE_SOLVER];
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    // helper workgroups off: this entry is one workgroup whatever the size
    const int wgs = ctx->solver_wgs;
    ctx->solver_wgs = 1;
    const int rc = rdvio_ba_prepare(ctx, S, pb, ctx->ba_arena_bytes, false);
    ctx->solver_wgs = wgs;
    if (rc) return rc;
    SolverWs &w = S.ws;
    if (lin_states) {
        memcpy((uint8_t *)S.host + S.user0_off, lin_states, (size_t)w.nfr * 16 * sizeof(double));
        w.user0 = (const double *)((uint8_t *)S.arena + S.user0_off);
    }
    w.no_loss = robust_loss ? 0 : 1;
    w.max_iter = 0;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(S.arena, S.host, S.in_bytes, hipMemcpyHostToDevice, st));
    rdvio_launch_ba_linearize(st, w);
    RDVIO_HIP_CHECK(ctx, hipGetLastError());
    const int N = w.N, D = w.D, nl = w.nl, npre = w.npre, np = w.np;
    std::vector<double> Jri((size_t)std::max(np, 1) * 9), Sm((size_t)(N + 1) * std::max(N, 1));
    auto down = [&](double *dst, const double *src, size_t n) -> hipError_t {
        return (dst && n) ? hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, st) : hipSuccess;
    };
    RDVIO_HIP_CHECK(ctx, down(out->r_preint, w.r_p, (size_t)npre * 15));
    RDVIO_HIP_CHECK(ctx, down(out->J_preint, w.Jp, (size_t)npre * 450));
    RDVIO_HIP_CHECK(ctx, down(out->r_prior, w.r_m, (size_t)D));
    RDVIO_HIP_CHECK(ctx, down(out->J_prior ? Jri.data() : nullptr, w.Jri, (size_t)np * 9));
    RDVIO_HIP_CHECK(ctx, down(out->H, w.H, (size_t)N * N));
    RDVIO_HIP_CHECK(ctx, down(out->g, w.g, (size_t)N));
    RDVIO_HIP_CHECK(ctx, down(out->lm_info, w.lm_m, (size_t)nl));
    RDVIO_HIP_CHECK(ctx, down(out->lm_grad, w.lm_g, (size_t)nl));
    RDVIO_HIP_CHECK(ctx, down((out->S_reduced || out->c_reduced) ? Sm.data() : nullptr, w.Sm, (size_t)(N + 1) * N));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, st));
    if (out->S_reduced)   // lower triangle -> full symmetric
        for (int i = 0; i < N; ++i)
            for (int j = 0; j <= i; ++j) out->S_reduced[(size_t)i * N + j] = out->S_reduced[(size_t)j * N + i] = Sm[(size_t)i * N + j];
    if (out->c_reduced) memcpy(out->c_reduced, &Sm[(size_t)N * N], (size_t)N * sizeof(double));
    if (out->J_prior) {
        // ceres/marginalization_factor.h:46-67: d r / d theta_i = S[:, 15 i .. +3] Jr^-1(log(q0^-1 q)), every other column block is S's
        for (int r = 0; r < D; ++r)
            for (int c = 0; c < D; ++c) {
                const int fi = c / 15, a = c % 15;
                double v = pb->prior_S[(size_t)r * D + c];
                if (a < 3) {
                    v = 0.0;
                    for (int k = 0; k < 3; ++k) v += pb->prior_S[(size_t)r * D + 15 * fi + k] * Jri[9 * (size_t)fi + 3 * k + a];
                }
                out->J_prior[(size_t)r * D + c] = v;
            }
    }
    out->N = N;
    return RDVIO_OK;
}

long rdvio_hip_ctx_team_retries(const rdvio_hip_ctx *ctx) { return ctx ? ctx->team_retries : -1; }

int rdvio_hip_ctx_set_kernel_timing(rdvio_hip_ctx *ctx, int on) {
    if (!ctx) return RDVIO_ERR_INVALID;
    ctx->kernel_timing = on < 0 ? 0 : on;
    ctx->kt_seen = 0;
    ctx->kt_launches = ctx->kt_ms = ctx->kt_flops = ctx->kt_iterations = 0.0;
    return RDVIO_OK;
}

int rdvio_hip_ctx_get_kernel_timing(rdvio_hip_ctx *ctx, double *out4) {
    if (!ctx || !out4) return RDVIO_ERR_INVALID;
    out4[0] = ctx->kt_launches;
    out4[1] = ctx->kt_ms;
    out4[2] = ctx->kt_flops;
    out4[3] = ctx->kt_iterations;
    return RDVIO_OK;
}

// diagnostic (RDVIO_PROF builds): per-phase ticks / counts of the last solve in a slot; not part of the public header
int rdvio_hip_debug_ba_prof(rdvio_hip_ctx *ctx, int slot, double *out64) {
    if (!ctx || bad_slot(slot) || !ctx->ba[slot].ready) return RDVIO_ERR_INVALID;
    RDVIO_HIP_CHECK(ctx, hipMemcpyAsync(out64, ctx->ba[slot].ws.summary + 8, 72 * sizeof(double), hipMemcpyDeviceToHost, ctx->lane[RDVIO_LANE_SOLVER]));
    RDVIO_HIP_CHECK(ctx, rdvio_wait(ctx, ctx->lane[RDVIO_LANE_SOLVER]));
    return RDVIO_OK;
}

int rdvio_hip_ba_solve(rdvio_hip_ctx *ctx, const rdvio_ba_problem *pb, int max_iterations, double *states_out,
                       double *inv_depth_out, rdvio_ba_summary *summary) {
    if (int rc = rdvio_hip_ba_upload(ctx, 0, pb)) return rc;
    if (int rc = rdvio_hip_ba_solve_resident(ctx, 0, max_iterations)) return rc;
    return rdvio_hip_ba_fetch(ctx, 0, states_out, inv_depth_out, summary);
}

}  // extern "C"
