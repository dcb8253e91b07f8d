#include <cstdlib>
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
// factors) an iteration is a chain of dependent phases of a few microseconds each; a kernel boundary
// (~1.5 us) or a grid barrier (~4-10 us) per phase would cost more than the phase.  Inside one workgroup a
// phase boundary is an s_barrier.  All reductions are fixed-order, so the result is bitwise reproducible.
//
// Normal equations.  With J robustified (sqrt(rho') scaling) and Jacobi-scaled by Sigma:
//   pose block   H = J_p^T J_p  (N x N, N = 15 * free frames): reprojection part OUTPUT-STATIONARY, one wavefront
//                per frame pair walking that pair's factor list (36 + 6 lanes = block entries + gradient), no atomics;
//   landmarks    scalar m_l = |J_l|^2, coupling row A[l, :] = J_l^T J_p (dense L x 6 nfree), built from per-factor
//                products stored at linearisation time;
//   Schur        S = Sigma (H - A^T W A) Sigma + mu D^2,  w_l = sigma_l^2 / (sigma_l^2 m_l + mu d_l^2);
//                A^T W A and the prior's S^T S are v_mfma_f64_16x16x4 tile products (block_linalg.hpp);
//   blocked (15-wide) Cholesky of S with MFMA trailing updates, landmark back-substitution.
// Model quantities (Cauchy point, model cost change) are evaluated from the assembled H, A, m, g instead of
// re-walking the factors:  |J x|^2 = xp^T H xp + 2 sum_l xl (A_l . xp) + sum_l m_l xl^2.
#include "ctx.hpp"
#include "factors.hpp"
#include <hip/hip_ext.h>
#include "solver_ws.hpp"
#include "block_linalg.hpp"
#include "marg_tail.hpp"

namespace {

constexpr int T = RDVIO_SOLVER_THREADS;
constexpr int NW = T / 64;
using Shared = LdsShared<T>;

#ifdef RDVIO_PROF
// diagnostic build only: accumulate wall-clock ticks (100 MHz) per phase into summary[8 + id], counts into [40 + id]
#define STAMP(id)                                                   \
    do {                                                            \
        if (threadIdx.x == 0) {                                     \
            const unsigned long long n__ = wall_clock64();          \
            w.summary[8 + (id)] += (double)(n__ - prof_last);       \
            w.summary[40 + (id)] += 1.0;                            \
            prof_last = n__;                                        \
        }                                                           \
    } while (0)
#else
#define STAMP(id) do {} while (0)
#endif

DM double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// E_i[a, b] of the prior Jacobian J = S E, E = blockdiag(Jr^-1(e_theta), I12) per frame
DM double prior_E(const Shared &sh, int i, int a, int b) {
    if (a < 3 && b < 3) return sh.Jri[9 * i + 3 * a + b];
    return a == b ? 1.0 : 0.0;
}

// one reprojection factor of the stored linearisation (robustified residual + Jacobians, per-factor landmark products,
// the target slot of the coupling row, the group-ordered record); returns the factor's cost
template <class WS>
DM double linearize_factor(const WS &w, Shared &sh, int k, const double *states, const double *invd, const double *extr, const double *W) {
    double cost = 0.0;
    double r[2], Jt[12], Jr[12], Jd[2];
    const int l = w.lm[k];
    reprojection_factor<true>(states + 16 * w.tgt[k], states + 16 * w.ref[k], w.tangent + 9 * (size_t)k,
                             w.z_ref + 3 * (size_t)l, invd[l], extr, W, r, Jt, Jr, Jd);
    const double s = r[0] * r[0] + r[1] * r[1];
    const double sum = 1.0 + s;
    cost += w.no_loss ? 0.5 * s : 0.5 * log(sum);
    if (true) {
        const double sc = w.no_loss ? 1.0 : sqrt(fmax(1.0 / sum, 2.2250738585072014e-308));
        r[0] *= sc;
        r[1] *= sc;
        Jd[0] *= sc;
        Jd[1] *= sc;
        double *o = w.fac + RDVIO_FAC_STRIDE * (size_t)k;
        // pose-constant frames (frame_fixed == 2) keep their columns, with zero pose Jacobians
        const double zt = sh.pfix[w.tgt[k]] ? 0.0 : sc, zr = sh.pfix[w.ref[k]] ? 0.0 : sc;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            Jt[i] *= zt;
            Jr[i] *= zr;
        }
        // per-factor landmark products: ht = Jd^T Jt, hr = Jd^T Jr, m = Jd^T Jd, g = Jd^T r.  Only hr, m, g are read back
        // (landmark pass, record entries 34..41); ht goes straight into the coupling row below, and the Jacobians and the
        // residual travel in the group-ordered record -- storing them here as well was a third of the phase's stores.
        double ht[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            ht[a] = Jd[0] * Jt[a] + Jd[1] * Jt[6 + a];
            o[34 + a] = Jd[0] * Jr[a] + Jd[1] * Jr[6 + a];
        }
        o[40] = Jd[0] * Jd[0] + Jd[1] * Jd[1];
        o[41] = Jd[0] * r[0] + Jd[1] * r[1];
        // coupling row of the landmark: the target frame's slot belongs to this factor alone (one observation per
        // (track, frame)), so it is a plain store; the anchor slot is summed in the landmark pass
        {
            const int ctc = sh.fcol[w.tgt[k]];
            if (ctc >= 0 && w.lfree[l]) {
                double *Arow = w.A + (size_t)l * (6 * w.nfree + 2) + 6 * ctc;
#pragma unroll
                for (int a = 0; a < 6; ++a) Arow[a] = ht[a];
            }
        }
        // group-ordered record [J_lo | J_hi | r]: the assembly streams these with no indirection
        const int gs = w.gslot[k];
        if (gs >= 0) {
            double *q = w.prec + RDVIO_REC_STRIDE * (size_t)gs;
            const bool flip = w.gflip[k] != 0;          // first slot holds Jr
            const bool both = sh.fcol[w.tgt[k]] >= 0 && sh.fcol[w.ref[k]] >= 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                q[i] = flip ? Jr[i] : Jt[i];
                q[12 + i] = both ? (flip ? Jt[i] : Jr[i]) : 0.0;
            }
            q[24] = r[0];
            q[25] = r[1];
        }
    }
    return cost;
}

// ---------------------------------------------------------------------------------------------
// Helper workgroups.  The factor evaluation is issue-bound on one CU, so a solve may be launched
// with n_wg > 1 workgroups: workgroup 0 runs the trust-region loop, the others wait for "evaluate your share of the
// factors" commands.  Protocol (all words in w.sync, agent-scope atomics):
//   sync[0] (sequence number << 12) | command   (release-stored by the leader, acquire-polled by the helpers);
//           command: bit 0 = with linearisation, bit 1 = candidate states (xc / xdc); 4 / 8 / 16 = this workgroup's share of
//           the group products / the H blocks / the Schur product (normal equations of large windows); 0x100 = exit
//   sync[1] completion counter                  (release-incremented by each helper, acquire-polled by the leader)
// partial[g] is written with a plain store before the helper's release-increment and read with an atomic load (never
// through the scalar cache).
// Every wait is bounded (RDVIO_SPIN_LIMIT polls of ~0.2 us): a helper that never hears from the leader exits; a leader
// that never hears from a helper sets sh.lost, leaves the loop with termination FAILURE (2) and flags summary[5], which
// rdvio_hip_ba_fetch turns into RDVIO_ERR_TIMEOUT -- the grid always drains and the caller always hears about it.  The launch assumes the workgroups are
// co-resident (one per CU on an otherwise idle stream), which is what a <= 16-workgroup grid on 256 CUs gets.
// ---------------------------------------------------------------------------------------------
#define RDVIO_SPIN_LIMIT 1000000
enum { CMD_LIN = 1, CMD_CAND = 2, CMD_PAIRS = 4, CMD_HBLK = 8, CMD_GEMM = 16, CMD_EXIT = 0x100 };

// Hand-off forms (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   producer: plain stores -> every storing wavefront's s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane's agent-scope
//             release fence (L2 write-back) -> s_waitcnt vmcnt(0) -> relaxed agent-scope flag store;
//   consumer: relaxed (L1-bypassing) polls by one lane -> ONE agent-scope acquire (invalidates this CU's L1) ->
//             s_waitcnt vmcnt(0) -> workgroup barrier -> plain loads.
// A cost-only answer has no payload but the 8-byte partial sum, which travels as an agent-scope store / load itself:
// no fence at all on that path.  (Round 1 fenced from every thread on both sides and polled with acquire loads: ~8 us
// per command round trip.)
DM unsigned sync_poll(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DM void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// leader: publish a command (all earlier global writes of the workgroup become visible to the helpers)
template <class WS>
DM void post_command(const WS &w, Shared &sh, unsigned cmd) {
    vm_drain();
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        vm_drain();
        __hip_atomic_store(w.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vm_drain();   // the counter is reset before any helper can see the command
        sh.seq += 1;
        __hip_atomic_store(w.sync, ((unsigned)sh.seq << 12) | cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// leader: wait for all helpers; returns the sum of their partial costs in workgroup order (NaN on timeout).
// with_payload: the helpers also wrote linearisation records that this workgroup reads with plain loads afterwards.
template <class WS>
DM double collect_partials(const WS &w, Shared &sh, bool with_payload) {
    const int G = w.n_wg - 1;
    if (threadIdx.x == 0) {
        int spins = sh.lost ? RDVIO_SPIN_LIMIT : 0;   // (a helper already went silent: the solve is ending, do not wait again)
        while (spins < RDVIO_SPIN_LIMIT && sync_poll(w.sync + 1) != (unsigned)G) { ++spins; __builtin_amdgcn_s_sleep(4); }
        sh.flag = spins < RDVIO_SPIN_LIMIT ? 1 : 0;
        if (spins >= RDVIO_SPIN_LIMIT) sh.lost = 1;  // the trust-region loop stops with FAILURE at its next check
        if (with_payload) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            vm_drain();
        }
    }
    __syncthreads();
    double extra = 0.0;
    if (threadIdx.x == 0) {
        if (!sh.flag) extra = __builtin_nan("");
        else
            for (int g = 1; g <= G; ++g)
                extra += __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)(w.partial + g), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    return extra;
}

// cost-only pass over the factors gid, gid + P, ...: two independent factors per trip so that their FP64 dependency
// chains interleave (a workgroup has only two wavefronts per SIMD to hide latency with); fixed summation order
template <class WS>
DM double cost_factors(const WS &w, const Shared &sh, const double *invd, const double *W, int gid, int P) {
    double cost = 0.0;
    for (int k = gid; k < w.nf; k += 2 * P) {
        const int k2 = k + P;
        const bool has2 = k2 < w.nf;
        const int kb = has2 ? k2 : k;
        double ra[2], rb[2];
        const int la = w.lm[k], lb = w.lm[kb];
        reprojection_residual(RDVIO_GEN(sh.cam) + 12 * w.tgt[k], RDVIO_GEN(sh.cam) + 12 * w.ref[k], w.tangent + 9 * (size_t)k, w.z_ref + 3 * (size_t)la, invd[la], W, ra);
        reprojection_residual(RDVIO_GEN(sh.cam) + 12 * w.tgt[kb], RDVIO_GEN(sh.cam) + 12 * w.ref[kb], w.tangent + 9 * (size_t)kb, w.z_ref + 3 * (size_t)lb, invd[lb], W, rb);
        const double sa = ra[0] * ra[0] + ra[1] * ra[1], sb = rb[0] * rb[0] + rb[1] * rb[1];
        cost += w.no_loss ? 0.5 * sa : 0.5 * log(1.0 + sa);
        if (has2) cost += w.no_loss ? 0.5 * sb : 0.5 * log(1.0 + sb);
    }
    return cost;
}

// rotation-prior factors gid, gid + P, ...
template <bool LIN, class WS>
DM double rotation_factors(const WS &w, Shared &sh, const double *states, const double *extr, const double *W, int gid, int P) {
    double cost = 0.0;
    for (int k = gid; k < w.nrot; k += P) {
        double r[2], J[6];
        rotation_prior_factor<LIN>(states + 16 * w.rot_tgt[k], states + 16 * w.rot_ref[k], w.rot_zref + 3 * k, w.rot_tangent + 9 * k, extr, W, r, J);
        const double s = r[0] * r[0] + r[1] * r[1];
        const double sum = 1.0 + s;
        cost += 0.5 * log(sum);
        if (LIN) {
            const double sc = sqrt(fmax(1.0 / sum, 2.2250738585072014e-308));
            w.r_r[2 * k] = r[0] * sc;
            w.r_r[2 * k + 1] = r[1] * sc;
            const double zs = sh.pfix[w.rot_tgt[k]] ? 0.0 : sc;
#pragma unroll
            for (int i = 0; i < 6; ++i) w.Jro[6 * k + i] = J[i] * zs;
        }
    }
    return cost;
}

template <class WS>
DM double prior_part(const WS &w, const Shared &sh, int pi, int a, int pj, int b) {
    const int D = w.D;
    if (a >= 3 && b >= 3) return w.Lam[(size_t)(15 * pi + a) * D + 15 * pj + b];
    // rows {0,1,2} (a < 3) or {a}, columns {0,1,2} (b < 3) or {b}: fixed 3 x 3 trip counts with the unused terms masked, so
    // that the (up to nine) loads are issued together instead of one per trip of a run-time loop; same summation order
    double lam[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool valid = (a < 3 || i == 0) && (b < 3 || j == 0);
            const int aa = a < 3 ? i : a, bb = b < 3 ? j : b;
            lam[3 * i + j] = valid ? w.Lam[(size_t)(15 * pi + aa) * D + 15 * pj + bb] : 0.0;
        }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool valid = (a < 3 || i == 0) && (b < 3 || j == 0);
            const int aa = a < 3 ? i : a, bb = b < 3 ? j : b;
            if (valid) acc += prior_E(sh, pi, aa, a) * lam[3 * i + j] * prior_E(sh, pj, bb, b);
        }
    return acc;
}

// index of the frame pair (lo <= hi) in the host's enumeration
DM int pair_id(int lo, int hi, int nfree) { return lo * nfree - lo * (lo - 1) / 2 + (hi - lo); }
// p-th lower block of the nfree x nfree block grid in row-major order -> fi * nfree + fj
DM int lower_block_of(int p, int nfree) {
    int fi = 0;
    while ((fi + 1) * (fi + 2) / 2 <= p) ++fi;
    return fi * nfree + (p - fi * (fi + 1) / 2);
}

// per factor group, X^T X with X = [J_lo | J_hi | r] (2 n_g x 13) on the matrix cores, groups g0, g0 + gstride, ...
// (one wavefront per group; lane l feeds A[i = l & 15][k] and B[k][j = l & 15] -- the same record element for i = j < 12 --,
// k = (item, row))
__device__ __attribute__((noinline)) void ne_pair_products(LdsWs &w, int g0_, int gstride_) {
    const int lane = threadIdx.x & 63;
    // (group quantities in scalar registers, the lane's record element and item offset computed once: see ne_h_blocks)
    const int g0 = __builtin_amdgcn_readfirstlane(g0_), gstride = __builtin_amdgcn_readfirstlane(gstride_);
    const int npairs = __builtin_amdgcn_readfirstlane(w.npairs);
    cgdouble *prec = RDVIO_UG(w.prec);
    gdouble *GPw = RDVIO_UGW(w.GP);
    const int *grp_off = w.grp_off;
    const int i = lane & 15, kk = lane >> 4;
    const int item_off = kk >> 1, row = kk & 1;
    // element of the record this lane supplies: i < 6: first[row][i]; 6 <= i < 12: second[row][i - 6]; i == 12: r[row]
    const int eo = (i < 6) ? row * 6 + i : (i < 12 ? 12 + row * 6 + (i - 6) : 24 + row);
    const bool has = i < 13;
    const int lane_off = RDVIO_REC_STRIDE * item_off + eo;
    const int out_off = 16 * (lane >> 4) + (lane & 15);
    // the group offsets (npairs + 1 ints) first, one per lane: reading them group by group put a dependent round trip in front
    // of every group's record loads
    const bool tab = npairs + 1 <= 64;
    const int off_tab = (tab && lane <= npairs) ? grp_off[lane] : 0;
    auto group_range = [&](int g, int &base, int &n) {
        int go, g1;
        if (tab) {
            go = __builtin_amdgcn_readlane(off_tab, g);
            g1 = __builtin_amdgcn_readlane(off_tab, g + 1);
        } else {
            go = __builtin_amdgcn_readfirstlane(grp_off[g]);
            g1 = __builtin_amdgcn_readfirstlane(grp_off[g + 1]);
        }
        base = RDVIO_REC_STRIDE * go;
        n = g1 - go;
    };
    // one group: X^T X accumulated over its items (k ascending); groups of up to 32 items -- the usual size -- are one trip
    auto group_loads = [&](int base, int n, int it, double (&v)[16]) {
#pragma unroll
        for (int u = 0; u < 16; ++u)   // (branch-free: masked lanes read the group's first record)
            v[u] = rdvio_ldm(prec, base + RDVIO_REC_STRIDE * (it + 2 * u) + lane_off, has && it + 2 * u + item_off < n, base);
    };
    auto group_mfma = [&](int n, int it, const double (&v)[16], double4_t &acc) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (it + 2 * u < n) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], v[u], acc, 0, 0, 0);   // (A = B: rows 12.. of the tile are never read)
    };
    auto group_store = [&](int g, const double4_t &acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) GPw[256 * g + 64 * r + out_off] = acc[r];
    };
    // two groups per trip: the first 32 items of both are loaded before either is used
#ifdef RDVIO_PROF_HBLK
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    const unsigned long long tq_in = wall_clock64();
    const unsigned long long cq_in = clock64();
#endif
    for (int g = g0; g < npairs; g += 2 * gstride) {
        const bool two = g + gstride < npairs;
        const int gB = two ? g + gstride : g;
        int baseA, nA, baseB, nB;
#ifdef RDVIO_PROF_HBLK
        const unsigned long long t0 = wall_clock64();
#endif
        group_range(g, baseA, nA);
        group_range(gB, baseB, nB);
        double vA[16], vB[16];
        group_loads(baseA, nA, 0, vA);
        group_loads(baseB, two ? nB : 0, 0, vB);
#ifdef RDVIO_PROF_HBLK
        const unsigned long long t1 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = wall_clock64();
        q0 += t1 - t0; q1 += t2 - t1;
#endif
        double4_t accA = {0.0, 0.0, 0.0, 0.0}, accB = {0.0, 0.0, 0.0, 0.0};
        group_mfma(nA, 0, vA, accA);
        for (int it = 32; it < nA; it += 32) {
            group_loads(baseA, nA, it, vA);
            group_mfma(nA, it, vA, accA);
        }
        group_store(g, accA);
        if (two) {
            group_mfma(nB, 0, vB, accB);
            for (int it = 32; it < nB; it += 32) {
                group_loads(baseB, nB, it, vB);
                group_mfma(nB, it, vB, accB);
            }
            group_store(gB, accB);
        }
#ifdef RDVIO_PROF_HBLK
        const unsigned long long t3 = wall_clock64();
        q2 += t3 - t2;
#endif
    }
#ifdef RDVIO_PROF_HBLK
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    q3 = wall_clock64() - tq_in;
    const unsigned long long cq = clock64() - cq_in;   // shader clocks over the same interval: the clock the CU actually runs at
    if (threadIdx.x == 0) { w.summary[76] += (double)q0; w.summary[77] += (double)q1; w.summary[78] += (double)q2; w.summary[79] += (double)q3; w.summary[75] += (double)cq; }
#endif
}

// every entry of H, output-stationary: prior + preintegration band + reprojection groups.  One wavefront per 15 x 15 lower
// block, in the matrix cores' result layout -- lane l holds the entries (row a = (l >> 4) + 4 r, column b = l & 15), r = 0..3:
//   prior part  E_i^T Lam_ij E_j  (E = blockdiag(Jr^-1, I12)) as two 15 x 15 x 15 products: T = Lam_ij E_j, then E_i^T T -- the
//               result registers of the first are, as they stand, the B operand of the second; the identity part of E costs
//               exact multiplications by one / zero, so entries outside the rotation rows / columns are Lam_ij itself;
//   band, group tiles: plain loads in the same layout.
// EVERY load of a block (4 Lam + 8 band + 2 or 2 nfree group entries per lane) is issued before the first use, for two blocks
// at a time: one L2 round trip per pair of blocks (the per-entry conditionals this replaces were each a round trip of its
// own: 62 waits per trip).  Block-level conditions are wave-uniform.  Wavefront p0 of pstride takes the lower blocks
// p0, p0 + 2 pstride, ...
__device__ __attribute__((noinline)) void ne_h_blocks(LdsWs &w, const Shared &sh, int p0_, int pstride_) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    // Addresses = wave-uniform part (which block: scalar registers, scalar arithmetic) + per-lane part (where in the block:
    // computed ONCE per call).  Measured with stamps inside this routine (RDVIO_PROF_HBLK): the wait for the loads of a trip
    // was 0.1 us, ISSUING them 2.7 us -- ~25 VALU instructions of 64-bit per-lane address arithmetic per load, on a
    // workgroup with two wavefronts per SIMD; the routine is instruction-issue bound.
    const int p0 = __builtin_amdgcn_readfirstlane(p0_), pstride = __builtin_amdgcn_readfirstlane(pstride_);
    const int N = __builtin_amdgcn_readfirstlane(w.N), nfree = __builtin_amdgcn_readfirstlane(w.nfree), D = __builtin_amdgcn_readfirstlane(w.D);
    cgdouble *Lam = RDVIO_UG(w.Lam), *PP = RDVIO_UG(w.PP), *GP = RDVIO_UG(w.GP);
    gdouble *H = RDVIO_UGW(w.H);
    // per-lane parts: entry (a = lk + 4 u, b = li) of a block in the matrix cores' result layout
    int lam_off[4], pp_off[4], h_off[4], ht_off[4];
    bool lam_ok[4], ok15[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = 4 * u + lk, a = lk + 4 * u;
        lam_off[u] = k * D + li;          // mirror entry Lam[15 pj + k][15 pi + li] of A[i = li][k] (Lam is exactly symmetric)
        lam_ok[u] = li < 15 && k < 15;
        pp_off[u] = 30 * a + li;
        ok15[u] = a < 15 && li < 15;
        h_off[u] = a * N + li;
        ht_off[u] = li * N + a;
    }
    int gpt_off[2], gd_off[2];
    bool ok6[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {         // pose rows a < 6 live in r = 0 (a = lk) and r = 1 (a = lk + 4 < 6)
        const int a = lk + 4 * u;
        gpt_off[u] = 16 * li + 6 + a;     // transposed read of the cross quadrant
        gd_off[u] = 16 * a + li;
        ok6[u] = a < 6 && li < 6;
    }
    constexpr int GD = 12;   // group tiles of a diagonal block loaded with the block (windows of up to 12 free frames: all)
    struct Loads {
        double lam[4], pp0[4], pp1[4], gp[2], gd[2][GD];
    };
    // block-level (scalar) quantities
    struct Blk {
        int fi, fj, pi, pj, src0, src1;
    };
    auto block_of = [&](int p) {  // p-th lower block in row-major order
        int fi = 0;
        while (tri(fi + 1) <= p) ++fi;
        Blk B;
        B.fi = fi;
        B.fj = p - tri(fi);
        B.pi = __builtin_amdgcn_readfirstlane(sh.pcol[B.fi]);
        B.pj = __builtin_amdgcn_readfirstlane(sh.pcol[B.fj]);
        const int which = B.fj - B.fi + 1;   // (lower blocks: 0 or 1)
        B.src0 = (which >= 0 && which <= 2) ? __builtin_amdgcn_readfirstlane(sh.band_src[(B.fi * 3 + which) * 2]) : -1;
        B.src1 = (which >= 0 && which <= 2) ? __builtin_amdgcn_readfirstlane(sh.band_src[(B.fi * 3 + which) * 2 + 1]) : -1;
        return B;
    };
    auto block_load = [&](const Blk &B, Loads &L) {
        const bool has_prior = B.pi >= 0 && B.pj >= 0;
        const int lam_base = (15 * B.pj) * D + 15 * B.pi;
        const int pp0_base = 900 * (B.src0 >> 2) + 450 * ((B.src0 >> 1) & 1) + 15 * (B.src0 & 1);
        const int pp1_base = 900 * (B.src1 >> 2) + 450 * ((B.src1 >> 1) & 1) + 15 * (B.src1 & 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (lane masks stay exec masks here: a vector load costs the memory pipeline in proportion to its ACTIVE lanes -- with all
            // eight wavefronts issuing their ~60 loads of a trip together the pass is bound by that rate; the clamped-load form
            // (rdvio_ldm) doubled the issue time of this routine, 14 -> 30 us over a solve)
            L.lam[u] = (has_prior && lam_ok[u]) ? Lam[lam_base + lam_off[u]] : 0.0;
            L.pp0[u] = (B.src0 >= 0 && ok15[u]) ? PP[pp0_base + pp_off[u]] : 0.0;
            L.pp1[u] = (B.src1 >= 0 && ok15[u]) ? PP[pp1_base + pp_off[u]] : 0.0;
        }
        if (B.fi != B.fj) {   // off-diagonal lower block (fi > fj): the (lo = fj, hi = fi) group's cross quadrant, transposed
            const int gp_base = 256 * pair_id(B.fj, B.fi, nfree);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                L.gp[u] = ok6[u] ? GP[gp_base + gpt_off[u]] : 0.0;
#pragma unroll
                for (int q = 0; q < GD; ++q) L.gd[u][q] = 0.0;
            }
        } else {              // diagonal block: the nfree group tiles that touch this frame (summed in f2 order)
#pragma unroll
            for (int u = 0; u < 2; ++u) L.gp[u] = 0.0;
#pragma unroll
            for (int q = 0; q < GD; ++q) {
                const int f2 = q, fi = B.fi;
                const int lo = f2 < fi ? f2 : fi, hi = f2 < fi ? fi : f2;
                const int off = (fi == lo) ? 0 : 6;  // quadrant (lo,lo) or (hi,hi); the single group (f,f) uses (lo,lo)
                const int gd_base = 256 * pair_id(lo, hi, nfree) + 17 * off;
#pragma unroll
                for (int u = 0; u < 2; ++u) L.gd[u][q] = (f2 < nfree && ok6[u]) ? GP[gd_base + gd_off[u]] : 0.0;
            }
        }
    };
    auto block_finish = [&](const Blk &B, const Loads &L) {
        const int fi = B.fi, fj = B.fj, pi = B.pi, pj = B.pj;
        const bool has_prior = pi >= 0 && pj >= 0;
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        if (has_prior) {
            double4_t T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // B[k = 4 u + lk][j = li] = E_j[k][j]
                const int k = 4 * u + lk;
                const double e = (k < 3 && li < 3) ? sh.Jri[9 * pj + 3 * k + li] : ((k == li && k < 15) ? 1.0 : 0.0);
                T = __builtin_amdgcn_mfma_f64_16x16x4f64(L.lam[u], e, T, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // A[i = li][k = 4 u + lk] = E_i[k][i];  B[k][j] = T[k][j] = result register u of T
                const int k = 4 * u + lk;
                const double e = (k < 3 && li < 3) ? sh.Jri[9 * pi + 3 * k + li] : ((k == li && k < 15) ? 1.0 : 0.0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(e, T[u], acc, 0, 0, 0);
            }
        }
        const bool fxi = __builtin_amdgcn_readfirstlane(sh.pfixc[fi]) != 0, fxj = __builtin_amdgcn_readfirstlane(sh.pfixc[fj]) != 0;
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = lk + 4 * u, b = li;
            double x = ((a < 6 && fxi) || (b < 6 && fxj)) ? 0.0 : acc[u];   // pose-constant frames: no prior Jacobian in those columns
            x += L.pp0[u];
            x += L.pp1[u];
            if (u < 2) {
                if (fi == fj) {
#pragma unroll
                    for (int q = 0; q < GD; ++q)
                        if (q < nfree) x += L.gd[u][q];
                    if (a < 6 && b < 6) {
                        for (int f0 = GD; f0 < nfree; f0 += GD) {   // (windows of more than GD free frames)
                            double gv[GD];
#pragma unroll
                            for (int q = 0; q < GD; ++q) {
                                const int f2 = f0 + q;
                                const int lo = f2 < fi ? f2 : fi, hi = f2 < fi ? fi : f2;
                                const int off = (fi == lo) ? 0 : 6;
                                gv[q] = f2 < nfree ? GP[256 * pair_id(lo, hi, nfree) + 17 * off + gd_off[u]] : 0.0;
                            }
#pragma unroll
                            for (int q = 0; q < GD; ++q)
                                if (f0 + q < nfree) x += gv[q];
                        }
                        if (a < 3 && b < 3)
                            for (int k = 0; k < w.nrot; ++k)
                                if (sh.fcol[w.rot_tgt[k]] == fi) x += w.Jro[6 * k + a] * w.Jro[6 * k + b] + w.Jro[6 * k + 3 + a] * w.Jro[6 * k + 3 + b];
                    }
                } else {
                    x += L.gp[u];
                }
            }
            v[u] = x;
        }
        // H is symmetric: only the lower blocks are formed, an off-diagonal block is stored a second time transposed
        const int h_base = (15 * fi) * N + 15 * fj, ht_base = (15 * fj) * N + 15 * fi;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok15[u]) {
                H[h_base + h_off[u]] = v[u];
                if (fi != fj) H[ht_base + ht_off[u]] = v[u];
            }
    };
    const int n_lower = nfree * (nfree + 1) / 2;
#ifdef RDVIO_PROF_HBLK
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#endif
    for (int p = p0; p < n_lower; p += 2 * pstride) {
        Loads L0, L1;
        const bool two = p + pstride < n_lower;
        const Blk B0 = block_of(p), B1 = block_of(two ? p + pstride : p);
#ifdef RDVIO_PROF_HBLK
        const unsigned long long t0 = wall_clock64();
#endif
        block_load(B0, L0);
        block_load(B1, L1);
#ifdef RDVIO_PROF_HBLK
        const unsigned long long t1 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t2 = wall_clock64();
#endif
        block_finish(B0, L0);
        if (two) block_finish(B1, L1);
#ifdef RDVIO_PROF_HBLK
        const unsigned long long t3 = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t4 = wall_clock64();
        c0 += t1 - t0; c1 += t2 - t1; c2 += t3 - t2; c3 += t4 - t3;
#endif
    }
#ifdef RDVIO_PROF_HBLK
    if (threadIdx.x == 0) { w.summary[72] += (double)c0; w.summary[73] += (double)c1; w.summary[74] += (double)(c2 + c3); }   // ([75]: shader clocks of the group products)
#endif
}

// workgroup g's share of the Schur product [C | Cg] = A^T W [A | g] of a multi-workgroup launch: the landmarks are split
// into n_wg runs (multiples of four rows: whole MFMA steps), every workgroup forms the product of ITS run -- one LDS-staged
// chunk at config-5 size -- into its own partial w.Cmp[g]; the consumers (schur_cm) add the partials in workgroup order.
__device__ __attribute__((noinline)) void schur_gemm_share(LdsWs &w, lds_double *lds, size_t lds_cap, int g) {
    const int NA = 6 * w.nfree, NAs = NA + 2;
    const int per = (((w.nl + w.n_wg - 1) / w.n_wg) + 3) & ~3;
    const int k0 = g * per < w.nl ? g * per : w.nl, k1 = k0 + per < w.nl ? k0 + per : w.nl, K = k1 - k0;
    double *C = w.Cmp + (size_t)g * NAs * NAs;
    const double *A = w.A + (size_t)k0 * NAs;
    if ((size_t)K * NAs + K <= lds_cap) {   // the whole run in one staged operand (config 5: 125 landmarks x 98 doubles)
        lds_double *As = lds, *ws = lds + (size_t)K * NAs;
        stage_to_lds<T, 16>(As, A, K * NAs);
        stage_to_lds<T, 1>(ws, w.lm_w + k0, K);
        __syncthreads();
        block_gemm_tn_lds<T>(C, NAs, As, NAs, As, NAs, ws, true, NA, NA + 1, K, true);
    } else if (!block_gemm_tn_chunked<T>(C, NAs, A, NAs, w.lm_w + k0, NA, NA + 1, K, lds, lds_cap)) {
        block_gemm_tn<T>(C, NAs, A, NAs, A, NAs, w.lm_w + k0, NA, NA + 1, K, true);
    }
}
// entry idx of the Schur product: one matrix, or the sum of the workgroups' partials (fixed order)
template <class WS>
DM double schur_cm(const WS &w, size_t idx, bool split, size_t stride) {
    if (!split) return w.Cm[idx];
    double acc = 0.0;
    for (int g0 = 0; g0 < w.n_wg; g0 += 8) {   // eight partials in flight (the default team has eight workgroups)
        double v[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = g0 + g < w.n_wg ? w.Cmp[(g0 + g) * stride + idx] : 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g0 + g < w.n_wg) acc += v[g];
    }
    return acc;
}

// the helpers' side: serve evaluation commands until told to exit (or until the leader goes silent)
__device__ __attribute__((noinline)) void helper_loop(LdsWs &w, Shared &sh, lds_double *lds, size_t lds_cap, int rank) {
    const int t = threadIdx.x;
    constexpr int TFm = T - 64;
    const int P = TFm + (w.n_wg - 1) * T, gid = TFm + (rank - 1) * T + t;
    int phase = 0;
    unsigned seen = 0;
    for (;;) {
        if (t == 0) {
            int spins = 0;
            unsigned s;
            while ((s = sync_poll(w.sync)) == seen && ++spins < RDVIO_SPIN_LIMIT) __builtin_amdgcn_s_sleep(4);
            sh.flag = (s == seen) ? -1 : (int)s;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // one acquire per CU: the command's data is read with plain loads below
            vm_drain();
        }
        __syncthreads();
        const int s = sh.flag;
        if (s < 0) return;  // the leader went silent
        seen = (unsigned)s;
        const unsigned cmd = seen & 0xfffu;
        if (cmd & CMD_EXIT) return;
        if (cmd & (CMD_PAIRS | CMD_HBLK | CMD_GEMM)) {
            // a share of the normal equations (the small tables were mirrored on the first command, an evaluation)
            if (cmd & CMD_PAIRS) {
                ne_pair_products(w, rank * NW + (t >> 6), w.n_wg * NW);
            } else if (cmd & CMD_HBLK) {
                for (int i = t; i < 9 * w.np; i += T) sh.Jri[i] = w.Jri[i];   // (the leader's linearisation left them in memory)
                __syncthreads();
                ne_h_blocks(w, sh, rank * NW + (t >> 6), w.n_wg * NW);
            } else {
                schur_gemm_share(w, lds, lds_cap, rank);
            }
            vm_drain();
            __syncthreads();
            if (t == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                vm_drain();
                __hip_atomic_fetch_add(w.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        const double *states = (cmd & CMD_CAND) ? w.xc : w.x, *invd = (cmd & CMD_CAND) ? w.xdc : w.xd;
        if ((seen >> 12) == 1) {  // first command: the leader's setup is complete, mirror the small tables
            for (int i = t; i < 64; i += T) {
                sh.fcol[i] = (i < w.nfr) ? w.fcol[i] : -1;
                sh.pfix[i] = (i < w.nfr && w.frame_fixed[i] == 2) ? 1 : 0;
                if (i < 32) {
                    sh.pcol[i] = (i < w.nfree) ? w.pcol[i] : -1;
                    sh.pfixc[i] = 0;
                }
            }
            for (int i = t; i < 18; i += T) sh.ext[i] = w.extr[i];
            for (int i = t; i < w.nfree * 6; i += T) sh.band_src[i] = w.band_src[i];
            __syncthreads();
            for (int i = t; i < w.nfr; i += T)
                if (w.fcol[i] >= 0) sh.pfixc[w.fcol[i]] = (w.frame_fixed[i] == 2) ? 1 : 0;
        }
        for (int i = t; i < w.nfr * 16; i += T) sh.st[i] = states[i];
        if (!(cmd & CMD_LIN))
            for (int i = t; i < w.nfr; i += T) camera_pose_of(states + 16 * i, w.extr, RDVIO_GEN(sh.cam) + 12 * i);
        __syncthreads();
        const double *W = RDVIO_GEN(sh.ext) + 14, *extr = RDVIO_GEN(sh.ext);
        double cost = 0.0;
        if (cmd & CMD_LIN) {
            for (int k = gid; k < w.nf; k += P) cost += linearize_factor(w, sh, k, RDVIO_GEN(sh.st), invd, extr, W);
            cost += rotation_factors<true>(w, sh, RDVIO_GEN(sh.st), extr, W, gid, P);
        } else {
            cost += cost_factors(w, sh, invd, W, gid, P);
            cost += rotation_factors<false>(w, sh, RDVIO_GEN(sh.st), extr, W, gid, P);
        }
        cost = block_sum(sh, cost, phase);
        if (cmd & CMD_LIN) {   // the linearisation records this workgroup wrote must reach memory before the answer does
            vm_drain();
            __syncthreads();
            if (t == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                vm_drain();
            }
        }
        if (t == 0) {
            __hip_atomic_store((unsigned long long *)(w.partial + rank), (unsigned long long)__double_as_longlong(cost), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            vm_drain();   // the partial sum is at memory scope before the arrival is counted
            __hip_atomic_fetch_add(w.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cost (and optionally the stored, robustified linearisation) at (states, invd).
// Waves 0..NW-2 evaluate reprojection factors while the last wave evaluates the (long, serial) preintegration
// factors and the prior's per-frame errors, so the two overlap.
// ---------------------------------------------------------------------------------------------
// CAND (cost-only): the candidate Plus(x, delta) is formed HERE -- straight into LDS, with the camera poses and the
// squared ambient step norm -- instead of being written to global memory, fenced and read back (a rejected trust-region
// iteration is a chain of such 1-5 us phases); *sn2_out receives the step norm squared.
template <bool LIN, bool CAND = false>
__device__ __attribute__((noinline)) double evaluate(LdsWs &w, Shared &sh, int phase, const double *states, const double *invd, unsigned long long &prof_last,
                                                     double *sn2_out = nullptr, double ca = 0.0, double cb = 0.0) {
    const int t = threadIdx.x;
    double sn2 = 0.0;
    // small hot data in LDS: every factor reads two frame states; L2 round trips would dominate the evaluation
    if (CAND) {
        for (int i = T - 1 - t; i < w.nfr; i += T) {  // (from the top: the last wave has no factor work)
            const int c = sh.fcol[i];
            double o[16];
            if (c < 0) {
#pragma unroll
                for (int a = 0; a < 16; ++a) o[a] = w.x[16 * i + a];
            } else {
                // delta = (ca * scaled gradient + cb * Gauss-Newton step) / dogleg diagonal * Jacobi scaling, 15 entries
                double d15[15], sg[15], gr[15], gn[15], dg[15];
#pragma unroll
                for (int a = 0; a < 15; ++a) {
                    sg[a] = w.sig_p[15 * c + a]; gr[a] = w.grad_p[15 * c + a]; gn[a] = w.gn_p[15 * c + a]; dg[a] = w.diag_p[15 * c + a];
                }
#pragma unroll
                for (int a = 0; a < 15; ++a) d15[a] = sg[a] * ((ca * gr[a] + cb * gn[a]) / dg[a]);
                state_plus(w.x + 16 * i, d15, o);
                if (sh.pfix[i])
#pragma unroll
                    for (int a = 0; a < 7; ++a) o[a] = w.x[16 * i + a];  // constant pose block
#pragma unroll
                for (int a = 0; a < 16; ++a)
                    if (!(sh.pfix[i] && a < 7)) { const double e = w.x[16 * i + a] - o[a]; sn2 += e * e; }
            }
#pragma unroll
            for (int a = 0; a < 16; ++a) { w.xc[16 * i + a] = o[a]; sh.st[16 * i + a] = o[a]; }
            camera_pose_of(o, w.extr, RDVIO_GEN(sh.cam) + 12 * i);
        }
        for (int l = t; l < w.nl; l += T) {
            const double v = w.xd[l] + (w.lfree[l] ? w.sig_l[l] * ((ca * w.grad_l[l] + cb * w.gn_l[l]) / w.diag_l[l]) : 0.0);
            w.xdc[l] = v;
            if (w.lfree[l]) { const double e = w.xd[l] - v; sn2 += e * e; }
        }
    } else {
        for (int i = t; i < w.nfr * 16; i += T) sh.st[i] = states[i];
        if (!LIN)  // camera poses for the cost-only residuals (the last wave is idle here: its work starts after the barrier)
            for (int i = T - 1 - t; i < w.nfr; i += T) camera_pose_of(states + 16 * i, w.extr, RDVIO_GEN(sh.cam) + 12 * i);
    }
    for (int i = t; i < w.nfr * 6; i += T) sh.ub[i] = w.user[16 * (i / 6) + ST_BG + (i % 6)];
    if (LIN)  // the preintegration Jacobian blocks are sparse: cleared here, ahead of the two wavefronts that fill them
        for (int i = t; i < w.npre * 450; i += T) w.G[i] = 0.0;
    __syncthreads();
    STAMP(LIN ? 30 : 31);
    states = RDVIO_GEN(sh.st);
    const double *W = RDVIO_GEN(sh.ext) + 14, *extr = RDVIO_GEN(sh.ext);
    double cost = 0.0;
    constexpr int TF = T - 64;
    // threads that evaluate reprojection factors: TF of this workgroup + every thread of the helper workgroups
    const int P = TF + (w.n_wg - 1) * T, gid = t;
    if (w.n_wg > 1) post_command(w, sh, (LIN ? CMD_LIN : 0u) | ((invd == w.xdc) ? CMD_CAND : 0u));
    if (t < TF) {
        // reprojection factors, CauchyLoss(1): cost 0.5 log(1+s); Corrector with rho'' < 0 => scale r, J by sqrt(rho')
        if (!LIN) cost += cost_factors(w, sh, invd, W, gid, P);
        for (int k = gid; LIN && k < w.nf; k += P) cost += linearize_factor(w, sh, k, states, invd, extr, W);
        cost += rotation_factors<LIN>(w, sh, states, extr, W, gid, P);
        // the prior's per-frame errors ride on the tail of the last factor wave (they used to share the preintegration
        // wave, where the two divergent branches ran one after the other); e goes straight into the LDS operand of S e
        for (int i = t - (TF - 64); i >= 0 && i < w.np; i += 64) {
            M3 Jri;
            double e15[15];
            marginalization_frame_error(states + 16 * w.prior_frames[i], w.lin + 16 * i, e15, LIN ? &Jri : nullptr);
#pragma unroll
            for (int a = 0; a < 15; ++a) {
                w.e_m[15 * i + a] = e15[a];
                sh.xv[15 * i + a] = e15[a];
            }
            if (LIN)
                for (int q = 0; q < 9; ++q) {
                    sh.Jri[9 * i + q] = Jri.m[q];
                    if (w.n_wg > 1) w.Jri[9 * i + q] = Jri.m[q];   // (the helpers' share of the H blocks reads them)
                }
        }
        // the residual-independent block groups of the preintegration Jacobians on lanes 16..63 of the same (last factor)
        // wavefront -- its threads carry the fewest reprojection factors -- while the last wavefront evaluates the residuals
        // and the rotation rows: the one-lane-per-factor chain there was the longest pole of the linearisation
        if (LIN)
            for (int k = t - (TF - 64) - 16; k >= 0 && k < w.npre; k += 48) {
                double *G = w.G + 450 * k;
                preintegration_translation_bias_blocks(states + 16 * w.pre_i[k], states + 16 * w.pre_j[k],
                                                       w.preint + (size_t)RDVIO_PREINT_SIZE * k, extr, G, G + 225);
            }
    } else {
        const int j = t - TF;  // last wave: the (long, serial) preintegration factors, one per lane
        for (int k = j; k < w.npre; k += 64) {
            double *G = w.G + 450 * k;
            preintegration_unwhitened<LIN, true>(states + 16 * w.pre_i[k], states + 16 * w.pre_j[k], w.preint + (size_t)RDVIO_PREINT_SIZE * k,
                                                 RDVIO_GEN(sh.ub) + 6 * w.pre_i[k], extr, w.e_p + 15 * k, G, G + 225);
        }
    }
    STAMP(LIN ? 17 : 20);
    __syncthreads();
    STAMP(LIN ? 18 : 21);
    // whitening of the preintegration residuals (and Jacobians)
    if (w.npre > 0) {
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
            // Jp = Sic G per factor and side: one 15 x 15 x 15 product per wavefront trip on the matrix cores, two per trip
            const int ntile = w.npre * 2, wave = t >> 6, lane = t & 63;
            auto store = [&](const double4_t &acc, int tile) {
                const int k = tile >> 1, which = tile & 1;
                const bool fix = sh.pfix[which ? w.pre_j[k] : w.pre_i[k]];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int row = (lane >> 4) + 4 * r4, col = lane & 15;
                    if (row < 15 && col < 15) w.Jp[450 * k + 225 * which + 15 * row + col] = (col < 6 && fix) ? 0.0 : acc[r4];
                }
            };
            for (int tile = wave; tile < ntile; tile += 2 * NW) {
                const bool two = tile + NW < ntile;
                const int t1 = two ? tile + NW : tile;
                double a0[4], b0[4], a1[4], b1[4];
                mfma_load15(w.preint + (size_t)RDVIO_PREINT_SIZE * (tile >> 1) + PRE_SIC, 1, 15, w.G + 450 * (tile >> 1) + 225 * (tile & 1), 15, 1, a0, b0);
                mfma_load15(w.preint + (size_t)RDVIO_PREINT_SIZE * (t1 >> 1) + PRE_SIC, 1, 15, w.G + 450 * (t1 >> 1) + 225 * (t1 & 1), 15, 1, a1, b1);
                store(mfma_run15(a0, b0), tile);
                if (two) store(mfma_run15(a1, b1), t1);
            }
        }
    }
    STAMP(LIN ? 19 : 22);
    // marginalisation prior: r = S e + f (S^T stored, so consecutive threads read consecutive addresses);
    // Jacobian handled through Lambda = S^T S (constant, symmetric) and E
    if (w.np > 0) {
        double *r_m = LIN ? w.r_m : w.c_m;
        const int D = w.D;
        // (sh.xv = e was filled by the factor wave's tail before the barrier above)
        cgdouble *STg = RDVIO_UG(w.ST), *Lamg = RDVIO_UG(w.Lam);   // (typed: scalar base + 32-bit lane offset; the operand vector is LDS)
        const int Ds = __builtin_amdgcn_readfirstlane(D);
        for (int base = 0; base < D; base += T / 4) {
            const int row = base + (t >> 2), part = t & 3;
            if (row < D) {
                const double r = quad_col_dot_u(STg, Ds, sh.xv, Ds, row, part) + w.f[row];
                if (part == 0) {
                    r_m[row] = r;
                    cost += 0.5 * r * r;
                }
                if (LIN) {
                    const double le = quad_col_dot_u(Lamg, Ds, sh.xv, Ds, row, part) + w.eta0[row];
                    if (part == 0) w.le[row] = le;
                }
            }
        }
    }
    if (w.n_wg > 1) cost += collect_partials(w, sh, LIN);  // (thread 0 carries the helpers' partial sums into the reduction)
    if (CAND) {
        double v2[2] = {cost, sn2};
        block_sum_n<T, 2>(sh, v2, phase);
        *sn2_out = v2[1];
        return v2[0];
    }
    return block_sum(sh, cost, phase);
}

// ---------------------------------------------------------------------------------------------
// normal equations from the stored linearisation: H, g, landmark scalars, coupling rows A.
// Phase 1 (independent loops, no barrier between them): per-pair reprojection products HP (one wavefront per
// frame pair streaming that pair's contiguous records), landmark rows, per-factor preintegration products.
// Phase 2: one output-stationary pass writes every H / g entry as prior + preintegration + reprojection part.
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) void build_normal_equations(LdsWs &w, Shared &sh, unsigned long long &prof_last) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int N = w.N, nfree = w.nfree, NA = 6 * nfree, NAs = NA + 2;
    // ---- phase 1a: the group products (ne_pair_products); large windows share them with the helper workgroups
    const bool multi = w.n_wg > 1;
    const int GW = multi ? w.n_wg * NW : NW;
    if (multi) post_command(w, sh, CMD_PAIRS);
    ne_pair_products(w, wave, GW);
    STAMP(12);
    // ---- phase 1b: landmarks (factors of one landmark are contiguous): scalars m, g and the anchor slot of the
    // coupling row (target slots were stored at linearisation time; untouched slots stay zero from the setup).
    // Column NA of the row holds g_l so that A^T W [A | g] yields the Schur gradient term with the same GEMM.
    // (loads batched: the landmark's flags / range in one round trip, then four factors per trip -- a window of eight keyframes
    // is two trips -- with the anchor frame's index riding on the first; summation order k ascending, as ever)
    for (int l = t; l < w.nl; l += T) {
        const bool lf = w.lfree[l] != 0;
        const int k0 = w.lm_first[l], cnt = w.lm_count[l];
        if (!lf) continue;
        const int k1 = k0 + cnt;
        double m = 0.0, gl = 0.0, ha[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int refk0 = 0;
        for (int k = k0; k < k1; k += 4) {
            double h[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double *o = w.fac + RDVIO_FAC_STRIDE * (size_t)(k + q < k1 ? k + q : k) + 34;
#pragma unroll
                for (int a = 0; a < 8; ++a) h[q][a] = o[a];
            }
            if (k == k0) refk0 = w.ref[k0];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (k + q < k1) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) ha[a] += h[q][a];
                    m += h[q][6];
                    gl += h[q][7];
                }
        }
        double *Arow = w.A + (size_t)l * NAs;
        const int ca = sh.fcol[refk0];
        if (ca >= 0)
#pragma unroll
            for (int a = 0; a < 6; ++a) Arow[6 * ca + a] = ha[a];
        Arow[NA] = gl;
        w.lm_m[l] = m;
        w.lm_g[l] = gl;
    }
    STAMP(13);
    // ---- phase 1c: per preintegration factor [Ji Jj]^T [Ji Jj] (30 x 30) and [Ji Jj]^T r (30)
    // three 15 x 15 quadrants per factor on the matrix cores -- (Ji,Ji), (Jj,Ji), (Jj,Jj); the fourth is the transpose of
    // the second -- two tiles per trip so that both tiles' operand loads share one memory round trip
    {
        const int ntile = w.npre * 3;
        auto quadrant = [&](int tile, const double *&Xa, const double *&Xb, int &k, int &qa, int &qb) {
            k = tile / 3;
            const int qd = tile - 3 * k;
            qa = qd >= 1;
            qb = qd == 2;
            Xa = w.Jp + 450 * k + 225 * qa;
            Xb = w.Jp + 450 * k + 225 * qb;
        };
        auto store = [&](const double4_t &acc, int k, int qa, int qb) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int row = (lane >> 4) + 4 * r4, col = lane & 15;
                if (row < 15 && col < 15) {
                    w.PP[900 * k + (15 * qa + row) * 30 + 15 * qb + col] = acc[r4];
                    if (qa != qb) w.PP[900 * k + (15 * qb + col) * 30 + 15 * qa + row] = acc[r4];
                }
            }
        };
        for (int tile = wave; tile < ntile; tile += 2 * NW) {
            const bool two = tile + NW < ntile;
            const double *Xa0, *Xb0, *Xa1, *Xb1;
            int k0, qa0, qb0, k1, qa1, qb1;
            quadrant(tile, Xa0, Xb0, k0, qa0, qb0);
            quadrant(two ? tile + NW : tile, Xa1, Xb1, k1, qa1, qb1);
            double a0[4], b0[4], a1[4], b1[4];
            mfma_load15(Xa0, 15, 1, Xb0, 15, 1, a0, b0);
            mfma_load15(Xa1, 15, 1, Xb1, 15, 1, a1, b1);
            store(mfma_run15(a0, b0), k0, qa0, qb0);
            if (two) store(mfma_run15(a1, b1), k1, qa1, qb1);
        }
    }
    for (int o = t; o < w.npre * 30; o += T) {
        const int k = o / 30, ra = o - 30 * k;
        const double *Jx = w.Jp + 450 * k + 225 * (ra / 15) + (ra % 15);
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 15; ++q) acc += Jx[q * 15] * w.r_p[15 * k + q];
        w.Pg[o] = acc;
    }
    STAMP(14);
    if (multi) (void)collect_partials(w, sh, true);   // (ends with the workgroup barrier; the helpers' tiles are acquired)
    else __syncthreads();
    STAMP(15);
    // ---- phase 2: the H blocks (ne_h_blocks), shared with the helper workgroups like phase 1a
    if (multi) post_command(w, sh, CMD_HBLK);
    ne_h_blocks(w, sh, wave, GW);
    STAMP(27);
    // gradient: every load of an entry (up to 12 group tiles, two preintegration terms, three prior terms) is issued before the
    // first use -- the conditionals used to be a round trip each; same summation order
    for (int o = t; o < N; o += T) {
        const int c = o / 15, a = o - 15 * c;
        const bool pose = a < 6;
        double gv[12];
#pragma unroll
        for (int u4 = 0; u4 < 12; ++u4) {
            const int f2 = u4;
            const int lo = f2 < c ? f2 : c, hi = f2 < c ? c : f2;
            const int off = (c == lo) ? 0 : 6;
            gv[u4] = (pose && f2 < nfree) ? w.GP[256 * (size_t)pair_id(lo, hi, nfree) + 16 * (off + a) + 12] : 0.0;
        }
        const int src0 = sh.g_src[2 * c], src1 = sh.g_src[2 * c + 1];
        const double pg0 = src0 >= 0 ? w.Pg[30 * (size_t)(src0 >> 1) + 15 * (src0 & 1) + a] : 0.0;
        const double pg1 = src1 >= 0 ? w.Pg[30 * (size_t)(src1 >> 1) + 15 * (src1 & 1) + a] : 0.0;
        const int pi = sh.pcol[c];
        const bool prior = pi >= 0 && !(pose && sh.pfixc[c]);
        double le3[3];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) le3[aa] = prior ? w.le[15 * pi + (a < 3 ? aa : a)] : 0.0;
        double acc = 0.0;
        if (pose) {
#pragma unroll
            for (int u4 = 0; u4 < 12; ++u4)
                if (u4 < nfree) acc += gv[u4];
            for (int f0 = 12; f0 < nfree; f0 += 12) {
                double gw[12];
#pragma unroll
                for (int u4 = 0; u4 < 12; ++u4) {
                    const int f2 = f0 + u4;
                    const int lo = f2 < c ? f2 : c, hi = f2 < c ? c : f2;
                    const int off = (c == lo) ? 0 : 6;
                    gw[u4] = f2 < nfree ? w.GP[256 * (size_t)pair_id(lo, hi, nfree) + 16 * (off + a) + 12] : 0.0;
                }
#pragma unroll
                for (int u4 = 0; u4 < 12; ++u4)
                    if (f0 + u4 < nfree) acc += gw[u4];
            }
            if (a < 3)
                for (int k = 0; k < w.nrot; ++k)
                    if (sh.fcol[w.rot_tgt[k]] == c) acc += w.Jro[6 * k + a] * w.r_r[2 * k] + w.Jro[6 * k + 3 + a] * w.r_r[2 * k + 1];
        }
        if (src0 >= 0) acc += pg0;
        if (src1 >= 0) acc += pg1;
        if (prior) {
            if (a < 3) {
#pragma unroll
                for (int aa = 0; aa < 3; ++aa) acc += prior_E(sh, pi, aa, a) * le3[aa];
            } else {
                acc += le3[0];
            }
        }
        w.g[o] = acc;
    }
    if (multi) (void)collect_partials(w, sh, true);   // (the helpers' H blocks are acquired; ends with the workgroup barrier)
    else __syncthreads();
    STAMP(16);
}

// q = x^T (J^T J) x and l = (J x) . r from the assembled normal equations; x = (xp: N pose entries, xl: landmarks)
template <class WS>
DM void model_products(const WS &w, Shared &sh, int &phase, const double *xp, const double *xl, double *q_out,
                       double *l_out) {
    const int t = threadIdx.x;
    const int N = w.N, NA = 6 * w.nfree;
    double v[2] = {0.0, 0.0};
    for (int c = t; c < N; c += T) sh.xv[c] = xp[c];
    __syncthreads();
    for (int r = t; r < N; r += T) {
        const double acc = dot_strided(w.H + r, N, sh.xv, 1, N);
        v[0] += sh.xv[r] * acc;
        v[1] += w.g[r] * sh.xv[r];
    }
    for (int l = t; l < w.nl; l += T) {
        if (!w.lfree[l]) continue;
        const double *Arow = w.A + (size_t)l * NA;
        double dotp = 0.0;
        for (int f = 0; f < w.nfree; ++f) {
            double av[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) av[a] = Arow[6 * f + a];
#pragma unroll
            for (int a = 0; a < 6; ++a) dotp += av[a] * sh.xv[15 * f + a];
        }
        v[0] += 2.0 * xl[l] * dotp + w.lm_m[l] * xl[l] * xl[l];
        v[1] += w.lm_g[l] * xl[l];
    }
    block_sum_n<T, 2>(sh, v, phase);
    *q_out = v[0];
    *l_out = v[1];
}

// Quadratic-model scalars of the two dogleg directions u = Sigma gradient_/D (steepest descent) and v = Sigma gn/D
// (Gauss-Newton), both in unscaled-J coordinates:  q_xy = (J x)^T (J y),  l_x = (J x)^T r.  Every dogleg step is
// delta = ca u + cb v, so |J delta|^2 and (J delta).r follow from these five numbers for ANY trust-region radius:
// a rejected step re-interpolates without touching H again.
// Only windows with 3 N > RDVIO_SOLVER_XV come here (the others take the fused gauss_newton_step_and_model): those are
// exactly the windows with more than RDVIO_LDS_CHOL_MAX_FRAMES free frames, whose reduced system is factored in global
// memory -- so u and v (2 N <= 960 doubles) live in the idle LDS Cholesky buffer, not in sh.xv (512 doubles).
static_assert(3 * 15 * RDVIO_LDS_CHOL_MAX_FRAMES <= RDVIO_SOLVER_XV && 3 * 15 * (RDVIO_LDS_CHOL_MAX_FRAMES + 1) > RDVIO_SOLVER_XV,
              "fused post-solve pass <=> LDS-resident Cholesky");
__device__ __attribute__((noinline)) void model_scalars(LdsWs &w, Shared &sh, lds_double *lds, int phase, double (&out)[5]) {
    const int t = threadIdx.x;
    const int N = w.N, NA = 6 * w.nfree;
    double *u = RDVIO_GEN(lds), *v = RDVIO_GEN(lds) + N;
    for (int i = t; i < N; i += T) {
        const double sd = w.sig_p[i] / w.diag_p[i];
        u[i] = sd * w.grad_p[i];
        v[i] = sd * w.gn_p[i];
    }
    __syncthreads();
    double a5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base < N; base += T / 4) {
        const int r = base + (t >> 2), part = t & 3;
        if (r < N) {
            double hu, hv;
            quad_col_dot2(w.H, N, u, v, N, r, part, hu, hv);
            if (part == 0) {
                a5[0] += u[r] * hu;
                a5[1] += u[r] * hv;
                a5[2] += v[r] * hv;
                a5[3] += w.g[r] * u[r];
                a5[4] += w.g[r] * v[r];
            }
        }
    }
    for (int l = t; l < w.nl; l += T) {
        if (!w.lfree[l]) continue;
        const double sd = w.sig_l[l] / w.diag_l[l];
        const double ul = sd * w.grad_l[l], vl = sd * w.gn_l[l];
        const double *Arow = w.A + (size_t)l * (NA + 2);
        double au = 0.0, av = 0.0;
        for (int c0 = 0; c0 < NA; c0 += 30) {  // five frames of the coupling row per trip: their loads share one round trip
            double x30[30];
#pragma unroll
            for (int q = 0; q < 30; ++q) x30[q] = (c0 + q < NA) ? Arow[c0 + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 30; ++q)
                if (c0 + q < NA) {
                    const int o = 15 * (c0 / 6 + q / 6) + q % 6;
                    au += x30[q] * u[o];
                    av += x30[q] * v[o];
                }
        }
        const double m = w.lm_m[l];
        a5[0] += 2.0 * ul * au + m * ul * ul;
        a5[1] += ul * av + vl * au + m * ul * vl;
        a5[2] += 2.0 * vl * av + m * vl * vl;
        a5[3] += w.lm_g[l] * ul;
        a5[4] += w.lm_g[l] * vl;
    }
    block_sum_n<T, 5>(sh, a5, phase);
#pragma unroll
    for (int i = 0; i < 5; ++i) out[i] = a5[i];
}

__device__ __attribute__((noinline)) double x_norm_of(LdsWs &w, Shared &sh, int phase, const double *st, const double *dep) {
    const int t = threadIdx.x;
    double s = 0.0;
    for (int o = t; o < w.nfr * 16; o += T)
        if (w.fcol[o / 16] >= 0 && !(sh.pfix[o / 16] && (o & 15) < 7)) s += st[o] * st[o];
    for (int l = t; l < w.nl; l += T)
        if (w.lfree[l]) s += dep[l] * dep[l];
    return sqrt(block_sum(sh, s, phase));
}

// gradient_max_norm = || x - Plus(x, -g) ||_inf  (TrustRegionMinimizer::EvaluateGradientAndJacobian)
__device__ __attribute__((noinline)) double grad_max_norm(LdsWs &w, Shared &sh, int phase) {
    const int t = threadIdx.x;
    double m = 0.0;
    for (int i = t; i < w.nfr; i += T) {
        const int c = sh.fcol[i];
        if (c < 0) continue;
        double d15[15], x16[16], o[16];
#pragma unroll
        for (int a = 0; a < 15; ++a) d15[a] = -w.g[15 * c + a];
#pragma unroll
        for (int a = 0; a < 16; ++a) x16[a] = w.x[16 * i + a];
        state_plus(x16, d15, o);
        const bool pf = sh.pfix[i];
#pragma unroll
        for (int a = 0; a < 16; ++a)  // (static indices: a run-time start index would put o[] in scratch)
            if (!(pf && a < 7)) m = fmax(m, fabs(x16[a] - o[a]));
    }
    if (w.n_lfree_hint > 0)
        for (int l = t; l < w.nl; l += T)
            if (w.lfree[l]) m = fmax(m, fabs(w.lm_g[l]));
    return block_max(sh, m, phase);
}

// per-launch setup shared by the solver and the marginalisation kernel: user state, free-landmark flags, LDS index
// tables, zeroed coupling rows, and the constant parts of the prior (S^T, Lambda = S^T S, eta0 = S^T f)
__device__ __attribute__((noinline)) void solver_setup(LdsWs &w, Shared &sh, lds_double *lds, size_t lds_cap, unsigned long long &prof_last) {
    const int t = threadIdx.x;
    const int nl = w.nl, nfree = w.nfree, NAs = 6 * nfree + 2;
    if (t == 0) { sh.seq = 0; sh.lost = 0; }
    for (int i = t; i < w.nfr * 16; i += T) {
        double v = w.x0[i];
        if (w.chain_src && i / 16 == w.chain_frame) v = w.chain_src[i % 16];   // (stream order: that solve has finished)
        w.x[i] = v;
        w.user[i] = w.user0 ? w.user0[i] : v;
    }
    for (int l = t; l < nl; l += T) w.xd[l] = w.xd0[l];
    for (int l = t; l < nl; l += T) w.lfree[l] = (w.lm_count[l] > 0 && !w.lm_fixed[l]) ? 1 : 0;
    for (int i = t; i < 64; i += T) {
        sh.fcol[i] = (i < w.nfr) ? w.fcol[i] : -1;
        sh.pfix[i] = (i < w.nfr && w.frame_fixed[i] == 2) ? 1 : 0;
        if (i < 32) {
            sh.pcol[i] = (i < nfree) ? w.pcol[i] : -1;
            sh.pfixc[i] = 0;
        }
    }
    __syncthreads();
    for (int i = t; i < w.nfr; i += T)
        if (w.fcol[i] >= 0) sh.pfixc[w.fcol[i]] = (w.frame_fixed[i] == 2) ? 1 : 0;
    for (int i = t; i < 18; i += T) sh.ext[i] = w.extr[i];
    for (size_t i = t; i < (size_t)nl * NAs; i += T) w.A[i] = 0.0;  // slots of frames that do not observe a landmark stay zero
    for (int i = t; i < nfree * 6; i += T) sh.band_src[i] = w.band_src[i];
    for (int i = t; i < nfree * 2; i += T) sh.g_src[i] = w.g_src[i];
    STAMP(23);
    if (w.np > 0) {
        const int D = w.D;
        // constant parts of the prior: S^T (coalesced S e), Lambda = S^T S (MFMA tiles), eta0 = S^T f.  S is staged in LDS
        // first (one batch of coalesced loads): the K-loops of the GEMM and of S^T f then run at LDS latency.
        if ((size_t)D * D + D <= lds_cap) {
            lds_double *Sl = lds, *fl = Sl + D * D;
            stage_to_lds<T, 16>(Sl, w.S, D * D);
            for (int q = t; q < D; q += T) fl[q] = w.f[q];
            __syncthreads();
            for (int o = t; o < D * D; o += T) w.ST[o] = Sl[(o % D) * D + o / D];
            STAMP(24);
            block_gemm_tn_g<T, const lds_double *, false, true>(w.Lam, D, Sl, D, Sl, D, Sl, D, D, D, true);
            STAMP(25);
            for (int a = t; a < D; a += T) {
                double acc = 0.0;
                for (int q = 0; q < D; ++q) acc += Sl[q * D + a] * fl[q];
                w.eta0[a] = acc;
            }
        } else {
            const double *Sp = w.S;
            for (int o = t; o < D * D; o += T) w.ST[o] = Sp[(size_t)(o % D) * D + o / D];
            STAMP(24);
            block_gemm_tn<T>(w.Lam, D, Sp, D, Sp, D, nullptr, D, D, D, true);
            STAMP(25);
            for (int a = t; a < D; a += T) {
                double acc = 0.0;
                for (int q = 0; q < D; ++q) acc += Sp[(size_t)q * D + a] * w.f[q];
                w.eta0[a] = acc;
            }
            __syncthreads();
            // mirror the lower tiles (the assembly reads Lambda as a full symmetric matrix)
            for (int o = t; o < D * D; o += T) {
                const int r = o / D, c = o - r * D;
                if ((c >> 4) > (r >> 4)) w.Lam[o] = w.Lam[(size_t)c * D + r];
            }
        }
        STAMP(26);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Phases of one trust-region iteration.  Each is its own (noinline) function on the LDS-resident workspace: the kernel
// body keeps only the scalar accept / reject logic, so that its registers are not shared with -- and spilled around --
// every phase's working set.
// ---------------------------------------------------------------------------------------------
#define PHASE_FN __device__ __attribute__((noinline))

// Jacobi scaling from the iteration-0 Jacobian
PHASE_FN void jacobi_scaling(LdsWs &w) {
    const int t = threadIdx.x, N = w.N, nl = w.nl;
    for (int i = t; i < N; i += T) w.sig_p[i] = 1.0 / (1.0 + sqrt(w.H[(size_t)i * N + i]));
    for (int l = t; l < nl; l += T) w.sig_l[l] = 1.0 / (1.0 + sqrt(w.lm_m[l]));
    __syncthreads();
}

// state-updating callback: `user` (the bias linearisation point of the preintegration factors) follows an accepted step
PHASE_FN void publish_user_state(LdsWs &w) {
    for (int i = threadIdx.x; i < w.nfr * 16; i += T) w.user[i] = w.x[i];
    __syncthreads();
}

// dogleg diagonal and scaled gradient; returns |scaled gradient|^2
PHASE_FN double dogleg_prepare(LdsWs &w, Shared &sh, int phase) {
    const int t = threadIdx.x, N = w.N, nl = w.nl;
    double gsq = 0.0;
    for (int i = t; i < N; i += T) {
        const double s = w.sig_p[i];
        const double d = sqrt(clampd(s * s * w.H[(size_t)i * N + i], 1e-6, 1e32));
        w.diag_p[i] = d;
        const double gv = s * w.g[i] / d;
        w.grad_p[i] = gv;
        gsq += gv * gv;
    }
    for (int l = t; l < nl && w.n_lfree_hint > 0; l += T) {  // (no free landmark: diag_l / grad_l are never read)
        double d = 1.0, gv = 0.0;
        if (w.lfree[l]) {
            const double s = w.sig_l[l];
            d = sqrt(clampd(s * s * w.lm_m[l], 1e-6, 1e32));
            gv = s * w.lm_g[l] / d;
        }
        w.diag_l[l] = d;
        w.grad_l[l] = gv;
        gsq += gv * gv;
    }
    return block_sum(sh, gsq, phase);
}

// landmark elimination for the damping mu:  [C | Cg] = A^T W [A | g] on the matrix cores, then
// S = Sigma (H - C) Sigma + mu D^2 (lower triangle, LDS-packed when the window fits) and the reduced right-hand side
PHASE_FN void schur_reduce(LdsWs &w, Shared &sh, lds_double *lds, size_t lds_cap, double mu, unsigned long long &prof_last) {
    const int t = threadIdx.x;
    const int N = w.N, nl = w.nl, nfree = w.nfree, NA = 6 * nfree, NAs = NA + 2;
    const bool has_lm = nl > 0 && w.n_lfree_hint > 0;
    const bool split = w.n_wg > 1 && !(w.lds_chol && (size_t)nl * NAs + nl <= lds_cap);   // Schur product shared with the helpers
    const size_t cstride = (size_t)NAs * NAs;
    lds_double *Sl = lds;
    // the landmark weights go straight into the staged operand's weight vector when the operand fits LDS (no pass and barrier
    // of their own); the other roads read them from memory
    const bool staged = NA > 0 && has_lm && w.lds_chol && (size_t)nl * NAs + nl <= lds_cap;
    if (has_lm) {
        lds_double *ws = lds + nl * NAs;
        for (int l = t; l < nl; l += T) {
            const bool lf = w.lfree[l] != 0;
            const double sl = w.sig_l[l], lm = w.lm_m[l], dl = w.diag_l[l];
            double lw = 0.0;
            if (lf) {
                const double s2 = sl * sl;
                lw = s2 / (s2 * lm + mu * dl * dl);
            }
            w.lm_w[l] = lw;
            if (staged) ws[l] = lw;
        }
        if (!staged) __syncthreads();
    }
    // The operand is first staged into the (currently idle) LDS Cholesky buffer with one batch of coalesced loads: a
    // K-loop over global memory is a chain of ~nl/16 dependent L2 round trips per tile.
    if (NA > 0 && has_lm) {
        if (staged) {
            lds_double *As = lds, *ws = lds + nl * NAs;
            stage_to_lds<T, 16>(As, w.A, nl * NAs);
            __syncthreads();
            block_gemm_tn_lds<T>(w.Cm, NAs, As, NAs, As, NAs, ws, true, NA, NA + 1, nl, true);
        } else {
            // (operand too large for LDS: staged a chunk of landmarks at a time; up to 8 x 5 = 40 tiles -- 32 free frames -- else the unstaged walk)
            if (split) {   // every workgroup forms the product of its run of landmarks (schur_gemm_share)
                post_command(w, sh, CMD_GEMM);
                schur_gemm_share(w, lds, lds_cap, 0);
                (void)collect_partials(w, sh, true);
            } else if (!block_gemm_tn_chunked<T>(w.Cm, NAs, w.A, NAs, w.lm_w, NA, NA + 1, nl, lds, lds_cap))
                block_gemm_tn<T>(w.Cm, NAs, w.A, NAs, w.A, NAs, w.lm_w, NA, NA + 1, nl, true);
        }
    }
    __syncthreads();
    STAMP(29);
    // one wavefront per LOWER 15 x 15 block, two blocks per trip; the loads of both blocks are issued before the first use.
    // Block quantities are scalars (readfirstlane), the lane's four entries (a, b) and their offsets are computed once: this
    // pass, like the H blocks, is bound by the issue of address arithmetic, not by memory.
    for (int i = t; i < N; i += T) sh.xv[i] = w.sig_p[i];
    __syncthreads();
    {
        const int lane = t & 63, n_lower = nfree * (nfree + 1) / 2;
        const int Ns = __builtin_amdgcn_readfirstlane(N), NAss = __builtin_amdgcn_readfirstlane(NAs);
        const bool lds_chol = __builtin_amdgcn_readfirstlane(w.lds_chol) != 0;
        cgdouble *Hg = RDVIO_UG(w.H), *Cg = RDVIO_UG(w.Cm);   // (never w.diag_p & co.: the small vectors may live in LDS -- generic pointers only)
        int ea[4], eb[4], h_off[4], c_off[4];
        bool in_blk[4], in_pose[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = lane + 64 * u;
            ea[u] = e / 15;
            eb[u] = e - 15 * ea[u];
            in_blk[u] = e < 225;
            in_pose[u] = in_blk[u] && ea[u] < 6 && eb[u] < 6 && has_lm;
            h_off[u] = ea[u] * Ns + eb[u];
            c_off[u] = ea[u] * NAss + eb[u];
        }
        for (int p = __builtin_amdgcn_readfirstlane(t >> 6); p < n_lower; p += 2 * NW) {
            const bool two = p + NW < n_lower;
            int fi[2], fj[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int blk = lower_block_of((q == 0 || two) ? p + q * NW : p, nfree);
                fi[q] = blk / nfree;
                fj[q] = blk - fi[q] * nfree;
            }
            double hv[2][4], cv[2][4], dv[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int h_base = 15 * fi[q] * Ns + 15 * fj[q], c_base = 6 * fi[q] * NAss + 6 * fj[q];
                const bool diag = fi[q] == fj[q];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = in_blk[u] && (!diag || eb[u] <= ea[u]);
                    hv[q][u] = ok ? Hg[h_base + h_off[u]] : 0.0;
                    dv[q][u] = (ok && diag && ea[u] == eb[u]) ? w.diag_p[15 * fi[q] + ea[u]] : 0.0;
                    const bool pose = ok && in_pose[u];
                    if (!split) cv[q][u] = pose ? Cg[c_base + c_off[u]] : 0.0;   // (workgroup-uniform branch)
                    else cv[q][u] = pose ? schur_cm(w, (size_t)(c_base + c_off[u]), true, cstride) : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool diag = fi[q] == fj[q];
                if (q == 0 || two) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool ok = in_blk[u] && (!diag || eb[u] <= ea[u]);
                        if (ok) {
                            const int i = 15 * fi[q] + ea[u], j = 15 * fj[q] + eb[u];
                            double v = hv[q][u];
                            if (in_pose[u]) v -= cv[q][u];
                            v *= sh.xv[i] * sh.xv[j];
                            if (i == j) v += mu * dv[q][u] * dv[q][u];
                            if (lds_chol) Sl[tri(i) + j] = v;
                            else w.Sm[(size_t)i * N + j] = v;
                        }
                    }
                }
            }
        }
    }
    for (int i = t; i < N; i += T) {
        const int fi = i / 15, a = i - 15 * fi;
        double v = w.g[i];
        if (a < 6 && has_lm) v -= schur_cm(w, (size_t)(6 * fi + a) * NAs + NA, split, cstride);
        v *= w.sig_p[i];
        w.yp[i] = v;
        if (w.lds_chol) Sl[tri(N) + i] = v;  // right-hand-side row of the packed triangle
        else w.Sm[(size_t)N * N + i] = v;   // ... or row N of the global-memory matrix (ld N)
    }
    __syncthreads();
}

// landmark part of the Gauss-Newton solve from the frame part in yp; returns > 0 when a component is not finite
PHASE_FN double back_substitute(LdsWs &w, Shared &sh, int phase, double mu) {
    const int t = threadIdx.x;
    const int N = w.N, nl = w.nl, nfree = w.nfree, NAs = 6 * nfree + 2;
    double bad = 0.0;
    for (int i = t; i < N; i += T) sh.xv[i] = w.sig_p[i] * w.yp[i];
    __syncthreads();
    for (int l = t; l < nl; l += T) {
        double y = 0.0;
        if (w.lfree[l]) {
            double s = w.lm_g[l];
            const double *Arow = w.A + (size_t)l * NAs;
            for (int c0 = 0; c0 < 6 * nfree; c0 += 30) {  // five frames of the coupling row per trip
                double av[30];
#pragma unroll
                for (int q = 0; q < 30; ++q) av[q] = (c0 + q < 6 * nfree) ? Arow[c0 + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 30; ++q)
                    if (c0 + q < 6 * nfree) s -= av[q] * sh.xv[15 * (c0 / 6 + q / 6) + q % 6];
            }
            const double s2 = w.sig_l[l] * w.sig_l[l];
            y = w.sig_l[l] * s / (s2 * w.lm_m[l] + mu * w.diag_l[l] * w.diag_l[l]);
            if (!isfinite(y)) bad = 1.0;
        }
        w.yl[l] = y;
    }
    for (int i = t; i < N; i += T)
        if (!isfinite(w.yp[i])) bad = 1.0;
    return block_max(sh, bad, phase);
}

// Gauss-Newton step in dogleg coordinates and the three norms the step selection needs: |g|^2, |gn|^2, g . gn
PHASE_FN void gauss_newton_norms(LdsWs &w, Shared &sh, int phase, double (&a3)[3]) {
    const int t = threadIdx.x, N = w.N, nl = w.nl;
    a3[0] = a3[1] = a3[2] = 0.0;
    for (int i = t; i < N; i += T) {
        const double gn = -w.yp[i] * w.diag_p[i];
        w.gn_p[i] = gn;
        a3[0] += w.grad_p[i] * w.grad_p[i];
        a3[1] += gn * gn;
        a3[2] += w.grad_p[i] * gn;
    }
    for (int l = t; l < nl; l += T) {
        const double gn = -w.yl[l] * w.diag_l[l];
        w.gn_l[l] = gn;
        a3[0] += w.grad_l[l] * w.grad_l[l];
        a3[1] += gn * gn;
        a3[2] += w.grad_l[l] * gn;
    }
    block_sum_n<T, 3>(sh, a3, phase);
}

// back_substitute + gauss_newton_norms + model_scalars in one pass (windows whose three staged pose vectors fit the LDS
// operand: 3 N <= 512): every landmark's coupling row is read once instead of twice, the pose vectors are staged once,
// and the nine scalars share one reduction.  out = {|g|^2, |gn|^2, g.gn, q_uu, q_uv, q_vv, l_u, l_v}; returns > 0 when a
// component of the solve is not finite.  Per-thread accumulation order is that of the three separate routines.
PHASE_FN double gauss_newton_step_and_model(LdsWs &w, Shared &sh, lds_double *big, int phase, double mu, double (&out)[8]) {
    const int t = threadIdx.x;
    const int N = w.N, nl = w.nl, nfree = w.nfree, NA = 6 * nfree, NAs = NA + 2;
    // the three staged pose vectors: sh.xv for windows whose 3 N doubles fit it, the idle LDS Cholesky buffer otherwise
    // (those windows factor in global memory)
    // (LDS-typed: ds_read / ds_write with immediate offsets instead of FLAT accesses through 64-bit addresses)
    lds_double *y = (3 * N <= RDVIO_SOLVER_XV) ? (lds_double *)sh.xv : big, *u = y + N, *v = y + 2 * N;
    cgdouble *Hg = RDVIO_UG(w.H);
    const int Ns = __builtin_amdgcn_readfirstlane(N);
    double r[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // r[8]: count of non-finite components
    for (int i = t; i < N; i += T) {
        const double yp = w.yp[i], sg = w.sig_p[i], dg = w.diag_p[i], gr = w.grad_p[i];
        if (!isfinite(yp)) r[8] += 1.0;
        y[i] = sg * yp;
        const double gn = -yp * dg;
        w.gn_p[i] = gn;
        r[0] += gr * gr;
        r[1] += gn * gn;
        r[2] += gr * gn;
        const double sd = sg / dg;
        u[i] = sd * gr;
        v[i] = sd * gn;
    }
    __syncthreads();
    for (int base = 0; base < N; base += T / 4) {
        const int row = base + (t >> 2), part = t & 3;
        if (row < N) {
            double hu, hv;
            quad_col_dot2_u(Hg, Ns, u, v, Ns, row, part, hu, hv);
            if (part == 0) {
                r[3] += u[row] * hu;
                r[4] += u[row] * hv;
                r[5] += v[row] * hv;
                r[6] += w.g[row] * u[row];
                r[7] += w.g[row] * v[row];
            }
        }
    }
    for (int l = t; l < nl && w.n_lfree_hint > 0; l += T) {  // (no free landmark: nothing below contributes)
        double yl = 0.0, gnl = 0.0;
        const bool lf = w.lfree[l];
        const double gl = w.grad_l[l], dl = w.diag_l[l];
        if (lf) {
            const double sl = w.sig_l[l], m = w.lm_m[l], lg = w.lm_g[l];
            double s = lg, au = 0.0, av = 0.0;
            const double *Arow = w.A + (size_t)l * NAs;
            for (int c0 = 0; c0 < NA; c0 += 30) {  // five frames of the coupling row per trip
                double a30[30];
#pragma unroll
                for (int q = 0; q < 30; ++q) a30[q] = (c0 + q < NA) ? Arow[c0 + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 30; ++q)
                    if (c0 + q < NA) {
                        const int o = 15 * (c0 / 6 + q / 6) + q % 6;
                        s -= a30[q] * y[o];
                        au += a30[q] * u[o];
                        av += a30[q] * v[o];
                    }
            }
            const double s2 = sl * sl;
            yl = sl * s / (s2 * m + mu * dl * dl);
            if (!isfinite(yl)) r[8] += 1.0;
            gnl = -yl * dl;
            const double sd = sl / dl;
            const double ul = sd * gl, vl = sd * gnl;
            r[3] += 2.0 * ul * au + m * ul * ul;
            r[4] += ul * av + vl * au + m * ul * vl;
            r[5] += 2.0 * vl * av + m * vl * vl;
            r[6] += lg * ul;
            r[7] += lg * vl;
        }
        w.yl[l] = yl;
        w.gn_l[l] = gnl;
        r[0] += gl * gl;
        r[1] += gnl * gnl;
        r[2] += gl * gnl;
    }
    block_sum_n<T, 9>(sh, r, phase);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = r[i];
    return r[8];
}

// One free frame and no free landmark (localize_newframe, sliding_window_tracker.cpp:101-125: N = 15): the damped system, its
// factorisation, both substitutions and the post-solve scalars on ONE wavefront, rows in lanes -- no workgroup barrier, no
// round trip through memory between the steps (the general road is five phases of a few microseconds each, whatever the
// size).  out as gauss_newton_step_and_model; returns 1 when the factorisation succeeded and every component is finite.
PHASE_FN int small_system_solve(LdsWs &w, Shared &sh, lds_double *Sl, double mu, double (&out)[8]) {
    const int t = threadIdx.x;
    if (t < 64) {
        const int r = t < 15 ? t : 14;
        const bool on = t < 15;
        double h[15];
#pragma unroll
        for (int c = 0; c < 15; ++c) h[c] = w.H[15 * r + c];
        const double sg = w.sig_p[r], dg = w.diag_p[r], gr = w.grad_p[r], g = w.g[r];
        // S = Sigma H Sigma + mu D^2 (lower triangle, packed), right-hand side Sigma g
#pragma unroll
        for (int c = 0; c < 15; ++c) {
            double v = h[c] * (sg * readlane_d(sg, c));
            if (c == r) v += mu * dg * dg;
            if (on && c <= r) Sl[tri(r) + c] = v;
        }
        if (t == 0) sh.flag = 1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cholesky_diag_block<T>(sh, Sl, 0, 0.0);   // L in place, sh.vec[j] = 1 / L_jj, sh.flag = 0 on a bad pivot
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double l[15];
#pragma unroll
        for (int c = 0; c < 15; ++c) l[c] = (on && c <= r) ? Sl[tri(r) + c] : 0.0;
        // forward substitution, column by column: z_j is final once the columns before it have been applied
        double z = on ? sg * g : 0.0;
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            const double zj = readlane_d(z, j) * sh.vec[j];
            if (t == j) z = zj;
            else if (on && t > j) z = __builtin_fma(-l[j], zj, z);
        }
        // backward substitution with L^T: lane i < j needs L[j][i] (row j lives in lane j): read from LDS
        double y = z;
#pragma unroll
        for (int j = 14; j >= 0; --j) {
            const double yj = readlane_d(y, j) * sh.vec[j];
            if (t == j) y = yj;
            else if (t < j) y = __builtin_fma(-Sl[tri(j) + r], yj, y);
        }
        const bool fin = !on || isfinite(y);
        const int ok = (__ballot(!fin) == 0ull && sh.flag) ? 1 : 0;
        // Gauss-Newton step in dogleg coordinates, the two model directions and the scalars
        const double gn = on ? -y * dg : 0.0;
        const double sd = sg / dg;
        const double u = on ? sd * gr : 0.0, v = on ? sd * gn : 0.0;
        double hu = 0.0, hv = 0.0;
#pragma unroll
        for (int c = 0; c < 15; ++c) {
            hu = __builtin_fma(h[c], readlane_d(u, c), hu);
            hv = __builtin_fma(h[c], readlane_d(v, c), hv);
        }
        const double grz = on ? gr : 0.0;
        double s8[8] = {grz * grz, gn * gn, grz * gn, u * hu, u * hv, v * hv, on ? g * u : 0.0, on ? g * v : 0.0};
#pragma unroll
        for (int i = 0; i < 8; ++i) s8[i] = wave_sum(s8[i]);
        if (on) {
            w.yp[r] = y;
            w.gn_p[r] = gn;
        }
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sh.blk[32 + i] = s8[i];
            sh.blk[40] = (double)ok;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = sh.blk[32 + i];
    const int ok = sh.blk[40] != 0.0;
    __syncthreads();   // (sh.blk is free again)
    return ok;
}

// the Gauss-Newton solve for one damping mu and everything the step selection needs from it:
//   out = {|g|^2, |gn|^2, g.gn, q_uu, q_uv, q_vv, l_u, l_v};
// returns bit 0 = solved (0: the damping must grow), bit 1 = an odd number of block reductions ran (the caller's bank flips)
PHASE_FN int solve_step(LdsWs &w, Shared &sh, lds_double *lds, size_t lds_cap, int phase, double mu, double (&out)[8], unsigned long long &prof_last) {
    const int N = w.N;
    if (w.small_system) {
        const int ok = small_system_solve(w, sh, lds, mu, out);
        STAMP(7);
        return ok;
    }
    lds_double *Sl = lds, *Dinv = lds + (size_t)(N + 1) * (N + 2) / 2;
    schur_reduce(w, sh, lds, lds_cap, mu, prof_last);
    STAMP(4);
    int ok = 1;
    if (N > 0) ok = w.lds_chol ? cholesky_lds(sh, Sl, Dinv, N) : cholesky_blocked(sh, w.Sm, N, N + 1);
    STAMP(5);
    if (!ok) return 0;
    if (N > 0) {
        if (w.lds_chol) cholesky_solve_lds(sh, Sl, Dinv, N, w.yp);
        else cholesky_solve(sh, w.Sm, N, w.yp, false, true, true);   // (row N holds L^-1 b)
    }
    STAMP(6);
    // (3 N doubles fit sh.xv for every LDS-factored window, the idle LDS buffer otherwise)
    const double bad = gauss_newton_step_and_model(w, sh, lds, phase, mu, out);
    return (bad > 0.0 ? 0 : 1) | 2;
}

// ---------------------------------------------------------------------------------------------
// Speculative evaluation of up to four trial steps in ONE pass.  After a rejected step the next trial is known in
// advance: the radius halves, the dogleg step is re-interpolated from the same model scalars, nothing else changes --
// so a run of rejections (the usual end of a window solve: the radius shrinks until a termination test fires) is a
// chain of identical phases, each a few barriers and memory round trips long.  Here the K candidates of the radii
// r, r/2, r/4, r/8 are formed and costed together; the kernel then replays the accept / reject decisions in order on the
// K results, so the iteration sequence is exactly the sequential one (every thread accumulates every candidate's terms
// in the order the one-candidate evaluation does).  Candidates live in the idle LDS Cholesky buffer:
//   states [4][nfr*16] | camera poses [4][nfr*12] | inverse depths [4][nl] | preintegration errors [4][npre*15].
// ---------------------------------------------------------------------------------------------
constexpr int KC = 4;   // trial steps evaluated per speculative pass (radii r, r/2, r/4, r/8).  Eight per pass measured slower: sixteen
                        // accumulators per thread spill in the factor loop, and what the callee clobbers costs the kernel body registers too
DM size_t candidates_lds_doubles(int nfr, int nl, int npre, int D) { return KC * ((size_t)28 * nfr + nl + 15 * npre + D); }

// one candidate state: Plus(x_i, delta_i) for the trial step (ca, cb) of frame i, its camera pose, and the squared ambient
// step; written to the candidate's LDS slots.  Out of line: its ~100 live doubles (state, step, four scaling vectors,
// quaternion algebra) would otherwise be spilled inside evaluate_candidates.
PHASE_FN double form_candidate_state(LdsWs &w, Shared &sh, lds_double *st_out, lds_double *cam_out, int i, double cak, double cbk) {
    const int c = sh.fcol[i];
    double x16[16], o[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) x16[a] = w.x[16 * i + a];
    double e2 = 0.0;
    if (c < 0) {
#pragma unroll
        for (int a = 0; a < 16; ++a) o[a] = x16[a];
    } else {
        double d15[15];
#pragma unroll
        for (int a = 0; a < 15; ++a) d15[a] = w.sig_p[15 * c + a] * ((cak * w.grad_p[15 * c + a] + cbk * w.gn_p[15 * c + a]) / w.diag_p[15 * c + a]);
        state_plus(x16, d15, o);
        if (sh.pfix[i])
#pragma unroll
            for (int a = 0; a < 7; ++a) o[a] = x16[a];  // constant pose block
#pragma unroll
        for (int a = 0; a < 16; ++a)
            if (!(sh.pfix[i] && a < 7)) { const double e = x16[a] - o[a]; e2 += e * e; }
    }
#pragma unroll
    for (int a = 0; a < 16; ++a) st_out[a] = o[a];
    double cam[12];
    camera_pose_of(o, w.extr, cam);
#pragma unroll
    for (int a = 0; a < 12; ++a) cam_out[a] = cam[a];
    return e2;
}

// Trial-step coefficients arrive through sh.blk[16 .. 16 + 2 KC) (ca, then cb; written by the caller in front of a barrier)
// and the 2 KC results leave through sh.blk[0 .. 2 KC) (cost, then squared step norm, per candidate): arrays handed over by
// reference would live in scratch memory on both sides of the call.
PHASE_FN void evaluate_candidates(LdsWs &w, Shared &sh, lds_double *lds, int phase, int K, unsigned long long &prof_last) {
    const int t = threadIdx.x, nfr = w.nfr, nl = w.nl, npre = w.npre, D = w.D;
    lds_double *stK = lds, *camK = stK + KC * nfr * 16, *xdK = camK + KC * nfr * 12, *epK = xdK + KC * nl, *eK = epK + KC * npre * 15;
    double ca[KC], cb[KC], cost[KC], sn2[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) { ca[k] = sh.blk[16 + k]; cb[k] = sh.blk[16 + KC + k]; cost[k] = 0.0; sn2[k] = 0.0; }
    // the candidate loops below are real loops (the bodies are large: four unrolled copies overflow the instruction
    // cache); the per-candidate accumulators are selected with compares so that they stay in registers
    auto add_to = [](double (&acc)[KC], int k, double v) {
#pragma unroll
        for (int q = 0; q < KC; ++q)
            if (q == k) acc[q] += v;
    };
    auto pick = [](const double (&arr)[KC], int k) {
        double v = arr[0];
#pragma unroll
        for (int q = 1; q < KC; ++q)
            if (q == k) v = arr[q];
        return v;
    };
    // candidate states: one (candidate, frame) pair per lane of the last wavefront (16 frames per candidate); the squared
    // step of frame i is then handed to the lane that owns frame i in the one-candidate evaluation, so that the step
    // norm is reduced from the same lanes, in the same order
    auto candidate_state = [&](int k, int i, double cak, double cbk, double &e2) {
        e2 = form_candidate_state(w, sh, stK + (k * nfr + i) * 16, camK + (k * nfr + i) * 12, i, cak, cbk);
    };
    if (nfr <= 16) {
        if (t >= T - 64) {
            const int L = T - 1 - t, i = L & 15;  // (L = 0 is the workgroup's last thread)
#pragma unroll
            for (int p = 0; p < KC / 4; ++p) {   // four candidates per pass of the wavefront
                const int k = 4 * p + (L >> 4);
                double e2 = 0.0;
                if (4 * p < K) {   // (wave-uniform)
                    if (k < K && i < nfr) candidate_state(k, i, pick(ca, k), pick(cb, k), e2);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double v = __shfl(e2, 63 - ((L & 15) + 16 * q));  // from the lane that formed candidate 4 p + q of this frame
                        if (L < 16 && L < nfr && 4 * p + q < K && sh.fcol[L] >= 0) sn2[4 * p + q] += v;
                    }
                }
            }
        }
    } else {
        for (int i = T - 1 - t; i < nfr; i += T) {
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                double e2;
                candidate_state(k, i, pick(ca, k), pick(cb, k), e2);
                if (sh.fcol[i] >= 0) add_to(sn2, k, e2);
            }
        }
    }
    for (int l = t; l < nl; l += T) {
        const double xd = w.xd[l];
        const bool lf = w.lfree[l];
        double sl = 0.0, gl = 0.0, gnl = 0.0, dl = 1.0;
        if (lf) { sl = w.sig_l[l]; gl = w.grad_l[l]; gnl = w.gn_l[l]; dl = w.diag_l[l]; }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (k >= K) break;
            const double v = xd + (lf ? sl * ((ca[k] * gl + cb[k] * gnl) / dl) : 0.0);
            xdK[k * nl + l] = v;
            if (lf) { const double e = xd - v; sn2[k] += e * e; }
        }
    }
    for (int i = t; i < nfr * 6; i += T) sh.ub[i] = w.user[16 * (i / 6) + ST_BG + (i % 6)];
    __syncthreads();
    STAMP(31);
    const double *W = RDVIO_GEN(sh.ext) + 14, *extr = RDVIO_GEN(sh.ext);
    constexpr int TF = T - 64;
    if (t < TF) {
        // reprojection factors, two per trip as in cost_factors; the factor's own data is loaded once for all candidates
        for (int f = t; f < w.nf; f += 2 * TF) {
            const int f2 = f + TF;
            const bool has2 = f2 < w.nf;
            const int fb = has2 ? f2 : f;
            const int la = w.lm[f], lb = w.lm[fb];
            const int ta = w.tgt[f], ra_ = w.ref[f], tb = w.tgt[fb], rb_ = w.ref[fb];
            double Ta[9], Tb[9], za[3], zb[3];
#pragma unroll
            for (int q = 0; q < 9; ++q) { Ta[q] = w.tangent[9 * (size_t)f + q]; Tb[q] = w.tangent[9 * (size_t)fb + q]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) { za[q] = w.z_ref[3 * (size_t)la + q]; zb[q] = w.z_ref[3 * (size_t)lb + q]; }
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const double *cam = RDVIO_GEN(camK) + (size_t)k * nfr * 12;
                double ra[2], rb[2];
                reprojection_residual(cam + 12 * ta, cam + 12 * ra_, Ta, za, xdK[k * nl + la], W, ra);
                reprojection_residual(cam + 12 * tb, cam + 12 * rb_, Tb, zb, xdK[k * nl + lb], W, rb);
                const double sa = ra[0] * ra[0] + ra[1] * ra[1], sb = rb[0] * rb[0] + rb[1] * rb[1];
                add_to(cost, k, w.no_loss ? 0.5 * sa : 0.5 * log(1.0 + sa));
                if (has2) add_to(cost, k, w.no_loss ? 0.5 * sb : 0.5 * log(1.0 + sb));
            }
        }
#pragma unroll 1
        for (int k = 0; k < K; ++k) add_to(cost, k, rotation_factors<false>(w, sh, RDVIO_GEN(stK) + (size_t)k * nfr * 16, extr, W, t, TF));
        // the prior's per-frame errors on the tail of the last factor wave, one operand vector per candidate
        if (w.np <= 16) {  // one (candidate, prior frame) pair per lane, four candidates per pass
            const int j = t - (TF - 64), i = j & 15;
            for (int k = j >> 4; j >= 0 && k < K && i < w.np; k += 4) {
                double e15[15];
                marginalization_frame_error(RDVIO_GEN(stK) + ((size_t)k * nfr + w.prior_frames[i]) * 16, w.lin + 16 * i, e15, nullptr);
#pragma unroll
                for (int a = 0; a < 15; ++a) eK[k * D + 15 * i + a] = e15[a];
            }
        } else {
            for (int i = t - (TF - 64); i >= 0 && i < w.np; i += 64) {
#pragma unroll 1
                for (int k = 0; k < K; ++k) {
                    double e15[15];
                    marginalization_frame_error(RDVIO_GEN(stK) + ((size_t)k * nfr + w.prior_frames[i]) * 16, w.lin + 16 * i, e15, nullptr);
#pragma unroll
                    for (int a = 0; a < 15; ++a) eK[k * D + 15 * i + a] = e15[a];
                }
            }
        }
    } else {
        // last wave: one (candidate, preintegration factor) pair per lane
        for (int j = t - TF; j < K * npre; j += 64) {
            const int k = j / npre, f = j - k * npre;
            const double *st = RDVIO_GEN(stK) + (size_t)k * nfr * 16;
            double e15[15];
            preintegration_unwhitened<false>(st + 16 * w.pre_i[f], st + 16 * w.pre_j[f], w.preint + (size_t)RDVIO_PREINT_SIZE * f,
                                             RDVIO_GEN(sh.ub) + 6 * w.pre_i[f], extr, e15, nullptr, nullptr);
#pragma unroll
            for (int a = 0; a < 15; ++a) epK[(k * npre + f) * 15 + a] = e15[a];
        }
    }
    STAMP(20);
    __syncthreads();
    STAMP(21);
    for (int o = t; o < npre * 15; o += T) {
        const int f = o / 15, row = o - 15 * f;
        const double *Sic = w.preint + (size_t)RDVIO_PREINT_SIZE * f + PRE_SIC;
        double sr[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) sr[q] = Sic[row * 15 + q];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (k >= K) break;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 15; ++q) acc += sr[q] * epK[(k * npre + f) * 15 + q];
            cost[k] += 0.5 * acc * acc;
        }
    }
    STAMP(22);
    if (w.np > 0) {
        cgdouble *STg = RDVIO_UG(w.ST);
        const int Ds = __builtin_amdgcn_readfirstlane(D);
        for (int base = 0; base < D; base += T / 4) {
            const int row = base + (t >> 2), part = t & 3;
            if (row < D) {
                double rk[KC];
                quad_col_dotk_u<KC>(STg, Ds, eK, Ds, K, Ds, row, part, rk);
                if (part == 0) {
                    const double fr = w.f[row];
#pragma unroll
                    for (int k = 0; k < KC; ++k)
                        if (k < K) { const double r = rk[k] + fr; cost[k] += 0.5 * r * r; }
                }
            }
        }
    }
    double v2k[2 * KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) { v2k[k] = cost[k]; v2k[KC + k] = sn2[k]; }
    block_sum_n<T, 2 * KC>(sh, v2k, phase);
    if (t == 0)
#pragma unroll
        for (int k = 0; k < 2 * KC; ++k) sh.blk[k] = v2k[k];
    __syncthreads();
}

// dogleg step for a trust-region radius from the scalars of the current linearisation (no memory traffic): coefficients of
// delta = ca u + cb v, its norm and the model cost change; returns whether the step is valid (model change > 0)
DM int dogleg_step_of(double rad, double gn_norm, double gnorm, double alpha, double gdotgn, double m0, double m1, double m2, double m3, double m4,
                      double &ca, double &cb, double &step_norm, double &model_change) {
    bool need_norm = false;
    if (gn_norm <= rad) { ca = 0.0; cb = 1.0; step_norm = gn_norm; }
    else if (gnorm * alpha >= rad) { ca = -(rad / gnorm); cb = 0.0; step_norm = rad; }
    else {
        const double b_dot_a = -alpha * gdotgn;
        const double a_sq = (alpha * gnorm) * (alpha * gnorm);
        const double bma_sq = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
        const double c = b_dot_a - a_sq;
        const double d = sqrt(c * c + bma_sq * (rad * rad - a_sq));
        const double beta = (c <= 0) ? (d - c) / bma_sq : (rad * rad - a_sq) / (d + c);
        ca = -alpha * (1.0 - beta);
        cb = beta;
        need_norm = true;
    }
    // delta = (ca grad + cb gn) / D * Jacobi scaling is formed inside the candidate evaluation
    if (need_norm) step_norm = sqrt(ca * ca * gnorm * gnorm + 2.0 * ca * cb * gdotgn + cb * cb * gn_norm * gn_norm);
    const double jsq = ca * ca * m0 + 2.0 * ca * cb * m1 + cb * cb * m2;
    const double jdr = ca * m3 + cb * m4;
    model_change = -(jdr + 0.5 * jsq);
    return model_change > 0.0;
}

// A run of rejections, up to KC trial steps in one pass: the coefficients of the radii r, r/2, ... (as long as each is a valid
// step within the iteration limit), their evaluation (evaluate_candidates) and the replay of exactly the decisions the
// sequential loop would take.  Out of line and fed through LDS (sh.spec, sh.blk), so that none of it lives in the registers of
// the kernel body.  Returns -1 when fewer than two trials qualify (nothing evaluated, no reduction run); otherwise
//   bits 0..3 trials consumed, bit 4 a termination test fired (convergence), bits 5..8 accepted trial + 1 (0: none);
// step norm / model change of trial k stay in sh.blk[16 + 2 KC + k] / [16 + 3 KC + k], its cost in sh.blk[k].
PHASE_FN int speculative_trials(LdsWs &w, Shared &sh, lds_double *lds, int phase, double radius, int iteration, double x_cost, double x_norm,
                                unsigned long long &prof_last) {
    const int t = threadIdx.x;
    __syncthreads();   // (thread 0's stores below stay behind the last replay's reads)
    const double gn_norm = sh.spec[0], gnorm = sh.spec[1], alpha = sh.spec[2], gdotgn = sh.spec[3];
    const double m0 = sh.spec[4], m1 = sh.spec[5], m2 = sh.spec[6], m3 = sh.spec[7], m4 = sh.spec[8];
    int Kc = 0;
    {
        double rk = radius;
        bool open = true;
#pragma unroll 1
        for (int k = 0; k < KC; ++k) {
            double ca_k = 0.0, cb_k = 0.0, dsn_k = 0.0, mcc_k = 1.0;
            if (open && iteration + k < w.max_iter && rk > 1e-32 && dogleg_step_of(rk, gn_norm, gnorm, alpha, gdotgn, m0, m1, m2, m3, m4, ca_k, cb_k, dsn_k, mcc_k)) {
                Kc = k + 1;
                rk *= 0.5;
            } else {
                open = false;
            }
            if (t == 0) { sh.blk[16 + k] = ca_k; sh.blk[16 + KC + k] = cb_k; sh.blk[16 + 2 * KC + k] = dsn_k; sh.blk[16 + 3 * KC + k] = mcc_k; }
        }
    }
    if (Kc < 2) return -1;
    __syncthreads();
    evaluate_candidates(w, sh, lds, phase, Kc, prof_last);
    STAMP(9);
    int n = 0, finished = 0, accepted = -1;
#pragma unroll 1
    for (int k = 0; k < Kc; ++k) {  // replay: exactly the decisions of Kc sequential iterations
        ++n;
        const double ck = sh.blk[k], sk = sh.blk[KC + k];
        const double cand_cost = isfinite(ck) ? ck : 1.7976931348623157e308;
        const double rel = (cand_cost >= 1.7976931348623157e308) ? -1.7976931348623157e308 : (x_cost - cand_cost) / sh.blk[16 + 3 * KC + k];
        if (sqrt(sk) <= 1e-8 * (x_norm + 1e-8)) { finished = 1; break; }
        if (fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { finished = 1; break; }
        if (rel > 1e-3) { accepted = k; break; }
    }
    return n | (finished << 4) | ((accepted + 1) << 5);
}

// x <- speculative candidate k (still in the LDS buffer evaluate_candidates filled)
PHASE_FN double accept_speculative(LdsWs &w, Shared &sh, int phase, lds_double *lds, int k) {
    const int t = threadIdx.x, nfr = w.nfr, nl = w.nl;
    const lds_double *stK = lds + (size_t)k * nfr * 16, *xdK = lds + KC * nfr * 28 + (size_t)k * nl;
    double s = 0.0;  // |x|^2 of the accepted point, accumulated exactly as x_norm_of accumulates it
    for (int o = t; o < nfr * 16; o += T) {
        const double v = stK[o];
        w.x[o] = v;
        if (sh.fcol[o / 16] >= 0 && !(sh.pfix[o / 16] && (o & 15) < 7)) s += v * v;
    }
    for (int l = t; l < nl; l += T) {
        const double v = xdK[l];
        w.xd[l] = v;
        if (w.lfree[l]) s += v * v;
    }
    return sqrt(block_sum(sh, s, phase));  // (the reduction's barrier also publishes the copies)
}

// x <- candidate
PHASE_FN double accept_candidate(LdsWs &w, Shared &sh, int phase) {
    const int t = threadIdx.x;
    double s = 0.0;
    for (int o = t; o < w.nfr * 16; o += T) {
        const double v = w.xc[o];
        w.x[o] = v;
        if (sh.fcol[o / 16] >= 0 && !(sh.pfix[o / 16] && (o & 15) < 7)) s += v * v;
    }
    for (int l = t; l < w.nl; l += T) {
        const double v = w.xdc[l];
        w.xd[l] = v;
        if (w.lfree[l]) s += v * v;
    }
    return sqrt(block_sum(sh, s, phase));
}

// Test switch RDVIO_TEST_POISON_LDS: LDS is not cleared between workgroups, so a field that some path reads before
// anybody wrote it holds whatever the previous kernel left there -- the mechanism behind the round-1 hang of the
// helper-workgroup launch (helper workgroups never run solver_setup).  With the switch on, every
// workgroup first fills its shared block and scratch buffer with 0xFF bytes (NaN doubles, -1 integers): any dependence on
// uninitialised LDS then shows up in the parity tests instead of depending on what ran before.
DM void poison_lds(void *p, size_t bytes) {
    __attribute__((address_space(3))) unsigned *q = (__attribute__((address_space(3))) unsigned *)p;
    for (size_t i = threadIdx.x; i < bytes / 4; i += T) q[i] = 0xffffffffu;
}

__global__ __launch_bounds__(T) void ba_solve_kernel(SolverWs w) {
    __shared__ BlockShared<T> sh_store;
    Shared &sh = *(Shared *)&sh_store;
    // packed 15x15 blocks of S and the inverses of its diagonal factors, LDS-resident when the window has at most
    // RDVIO_LDS_CHOL_MAX_FRAMES free frames (138.6 KB of the CU's 160 KB); larger windows factor in global memory
    constexpr int NMAX = 15 * RDVIO_LDS_CHOL_MAX_FRAMES;
    constexpr size_t LDS_CAP = (NMAX + 1) * (NMAX + 2) / 2 + 225 * RDVIO_LDS_CHOL_MAX_FRAMES;  // (+ the right-hand-side row)
    __shared__ __attribute__((aligned(16))) double lds_chol_buf[LDS_CAP];
    const int t = threadIdx.x;
    // Workgroups of a multi-workgroup launch: the grid is wg_stride x n_wg blocks and only every wg_stride-th block takes part
    // (the others end here).  Blocks are dealt round-robin over the 8 XCDs, so with wg_stride = 8 the team shares one XCD's L2:
    // what one workgroup hands to another (records, tiles, coupling rows -- megabytes at config-5 size) is then read from that
    // L2 instead of from memory.  Placement is a speed matter only: every hand-off keeps its agent-scope release / acquire.
    const int rank = (int)blockIdx.x / w.wg_stride;
    if ((int)blockIdx.x % w.wg_stride) return;
    __shared__ SolverWs w_lds;
    if (w.poison_lds) {
        poison_lds(&sh_store, sizeof(sh_store));
        poison_lds(lds_chol_buf, sizeof(lds_chol_buf));
        __syncthreads();
    }
    for (int i = t; i < (int)(sizeof(SolverWs) / 8); i += T)
        ((__attribute__((address_space(3))) unsigned long long *)&w_lds)[i] = ((const unsigned long long *)&w)[i];
    __syncthreads();
    LdsWs &wl = *(LdsWs *)&w_lds;
    if (rank > 0) {  // helper workgroup: factor evaluation and shares of the normal equations on request
        if (w.mute_helpers) return;  // test switch: the leader's bounded wait must turn this into FAILURE
        helper_loop(wl, sh, RDVIO_LDS(lds_chol_buf), LDS_CAP, rank);
        return;
    }
    const int N = w.N;
    int phase = 0;
    unsigned long long prof_last = 0;
#ifdef RDVIO_PROF
    prof_last = wall_clock64();
    if (t == 0) for (int i = 8; i < 80; ++i) w.summary[i] = 0.0;
#endif

    solver_setup(wl, sh, RDVIO_LDS(lds_chol_buf), LDS_CAP, prof_last);

    // ---- small vectors resident in LDS.  The phases between two linearisations are chains of short passes over N- and
    // nl-sized vectors (scalings, gradient, Gauss-Newton step, landmark scalars, frame states): in global memory every pass
    // is an L2 round trip per read plus the store drain in front of its barrier.  When the window leaves room behind the
    // packed triangle (config 2: 5 000 of the buffer's 16 336 doubles; a one-frame localisation: nearly all of it), the
    // descriptor's pointers to those vectors are redirected into that room: the phases read the descriptor, so they follow
    // (generic pointers into LDS: FLAT accesses at LDS latency).  lds_cap shrinks accordingly for every scratch user of
    // the buffer (operand staging, candidate states); x / xd are copied back before the summary.
    size_t lds_cap = LDS_CAP;
    int resident = 0;
    if (w.n_wg == 1 && w.lds_chol && !w.no_lds_vectors) {
        const int NAs = 6 * w.nfree + 2;
        size_t need = (size_t)(N + 1) * (N + 2) / 2 + 225 * (size_t)(N / 15);                          // packed triangle + diagonal inverses
        const size_t stage = (w.nl > 0 && w.n_lfree_hint > 0) ? (size_t)w.nl * NAs + w.nl : 0;          // Schur operand staging
        const size_t cand = candidates_lds_doubles(w.nfr, w.nl, w.npre, w.D);                           // speculative candidates
        if (stage > need) need = stage;
        if (cand > need) need = cand;
        need = (need + 1) & ~(size_t)1;
        const size_t lfree_d = ((size_t)w.nl + 7) / 8;
        const size_t want = 6 * (size_t)N + 10 * (size_t)w.nl + lfree_d + 48 * (size_t)w.nfr;
        if (need + want <= LDS_CAP) {
            resident = 1;
            lds_cap = need;
            if (t == 0) {
                double *p = lds_chol_buf + need;
                auto take = [&](size_t n) { double *q = p; p += n; return q; };
                w_lds.g = take(N); w_lds.yp = take(N); w_lds.sig_p = take(N); w_lds.diag_p = take(N); w_lds.grad_p = take(N); w_lds.gn_p = take(N);
                w_lds.lm_m = take(w.nl); w_lds.lm_g = take(w.nl); w_lds.lm_w = take(w.nl); w_lds.yl = take(w.nl);
                w_lds.sig_l = take(w.nl); w_lds.diag_l = take(w.nl); w_lds.grad_l = take(w.nl); w_lds.gn_l = take(w.nl);
                w_lds.xd = take(w.nl); w_lds.xdc = take(w.nl);
                w_lds.x = take(16 * (size_t)w.nfr); w_lds.xc = take(16 * (size_t)w.nfr); w_lds.user = take(16 * (size_t)w.nfr);
                w_lds.lfree = (uint8_t *)take(lfree_d);
            }
            __syncthreads();
            // what the setup left in the global copies
            for (int i = t; i < w.nfr * 16; i += T) { wl.x[i] = w.x[i]; wl.user[i] = w.user[i]; }
            for (int l = t; l < w.nl; l += T) { wl.xd[l] = w.xd[l]; wl.lfree[l] = w.lfree[l]; }
            __syncthreads();
        }
    }

    double radius = 1e4, mu = 1e-8, alpha = 0.0, dogleg_step_norm = 0.0;
    double gnorm = 0.0, gn_norm = 0.0, gdotgn = 0.0, gsq_keep = 0.0;
    double msc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};  // q_uu q_uv q_vv l_u l_v of the current linearisation
    int reuse = 0, iteration = 0, invalid_steps = 0, last_successful = 0, n_success = 0;
    int term = 1;  // NO_CONVERGENCE
    STAMP(0);
    auto dogleg_step = [&](double rad, double &ca, double &cb, double &step_norm, double &model_change) -> int {
        return dogleg_step_of(rad, gn_norm, gnorm, alpha, gdotgn, msc[0], msc[1], msc[2], msc[3], msc[4], ca, cb, step_norm, model_change);
    };
    int prev_rejected = 0;
    double x_norm = 0.0, x_cost = 0.0, grad_max = 0.0;
    // bookkeeping of an accepted step (x already holds the candidate): new linearisation, radius and damping updates
    // linearised: the candidate was evaluated with its linearisation (fused trial, below) -- its records, residuals and cost ARE
    // those of the accepted point (`user` still holds the previous point in both cases)
    auto accepted_step = [&](double rel, bool linearised = false, double cost_there = 0.0) {  // (x_norm was set by the accept function)
        STAMP(10);
        if (linearised) {
            x_cost = cost_there;
        } else {
            x_cost = evaluate<true>(wl, sh, phase, wl.x, wl.xd, prof_last);  // `user` still holds the previous point here
            phase ^= 1;  // (one reduction inside)
        }
        STAMP(1);
        build_normal_equations(wl, sh, prof_last);
        STAMP(2);
        grad_max = grad_max_norm(wl, sh, phase);
        phase ^= 1;  // (one reduction inside)
        STAMP(11);
        last_successful = 1;
        n_success++;
        if (rel < 0.25) radius *= 0.5;
        if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
        mu = fmax(1e-8, 2.0 * mu / 10.0);
        reuse = 0;
        prev_rejected = 0;
    };
    // a run of rejections is evaluated four trial radii at a time (evaluate_candidates) when the candidates fit the LDS
    const bool speculate = !w.no_speculation && w.n_wg == 1 && w.lds_chol && candidates_lds_doubles(w.nfr, w.nl, w.npre, w.D) <= lds_cap;

    x_norm = x_norm_of(wl, sh, phase, wl.x, wl.xd);
    phase ^= 1;  // (one reduction inside)
    x_cost = evaluate<true>(wl, sh, phase, wl.x, wl.xd, prof_last);
    phase ^= 1;  // (one reduction inside)
    STAMP(1);
    const double initial_cost = x_cost;
    build_normal_equations(wl, sh, prof_last);
    STAMP(2);
    jacobi_scaling(wl);
    grad_max = grad_max_norm(wl, sh, phase);
    phase ^= 1;  // (one reduction inside)

    if (N == 0 && w.n_lfree_hint == 0) term = 0;
    else
        for (;;) {
            if (w.n_wg > 1 && sh.lost) { term = 2; break; }  // a helper workgroup went silent: FAILURE, x = last accepted point
            if (last_successful) publish_user_state(wl);
            if (iteration >= w.max_iter) { term = 1; break; }
            if (grad_max <= 1e-10) { term = 0; break; }
            if (radius <= 1e-32) { term = 0; break; }
            if (speculate && prev_rejected && reuse) {
                // the trial steps the next iterations would take one by one, evaluated together and replayed (speculative_trials)
                const int oc = speculative_trials(wl, sh, RDVIO_LDS(lds_chol_buf), phase, radius, iteration, x_cost, x_norm, prof_last);
                if (oc >= 0) {
                    phase ^= 1;  // (one reduction inside)
                    const int n = oc & 15, finished = (oc >> 4) & 1, accepted = ((oc >> 5) & 15) - 1;
                    iteration += n;
                    last_successful = 0;
                    invalid_steps = 0;
                    for (int q = (finished || accepted >= 0) ? 1 : 0; q < n; ++q) radius *= 0.5;   // (one halving per rejected trial)
                    if (finished) { term = 0; break; }
                    if (accepted >= 0) {
                        const double ck = sh.blk[accepted];
                        const double rel_acc = (x_cost - (isfinite(ck) ? ck : 1.7976931348623157e308)) / sh.blk[16 + 3 * KC + accepted];
                        const double dsn_acc = sh.blk[16 + 2 * KC + accepted];
                        x_norm = accept_speculative(wl, sh, phase, RDVIO_LDS(lds_chol_buf), accepted);
                        phase ^= 1;  // (one reduction inside)
                        dogleg_step_norm = dsn_acc;
                        accepted_step(rel_acc);
                    }
                    continue;
                }
            }
            iteration++;
            last_successful = 0;
            STAMP(10);

            int solve_ok = 1;
            if (!reuse) {
                reuse = 1;
                gsq_keep = dogleg_prepare(wl, sh, phase);
                phase ^= 1;  // (one reduction inside)
                STAMP(3);
                // Gauss-Newton step: (H_s + mu D^2) y = g_s with the landmarks eliminated
                solve_ok = 0;
                while (mu < 1.0) {
                    double pm[8];
                    const int st = solve_step(wl, sh, RDVIO_LDS(lds_chol_buf), lds_cap, phase, mu, pm, prof_last);
                    phase ^= (st >> 1) & 1;
                    if (!(st & 1)) {
                        mu *= 10.0;
                        continue;
                    }
                    gnorm = sqrt(pm[0]);
                    gn_norm = sqrt(pm[1]);
                    gdotgn = pm[2];
#pragma unroll
                    for (int i = 0; i < 5; ++i) msc[i] = pm[3 + i];
                    solve_ok = 1;
                    break;
                }
                if (solve_ok) {
                    alpha = gsq_keep / msc[0];
                    if (t == 0) {   // for speculative_trials (read behind its entry barrier)
                        sh.spec[0] = gn_norm; sh.spec[1] = gnorm; sh.spec[2] = alpha; sh.spec[3] = gdotgn;
#pragma unroll
                        for (int i = 0; i < 5; ++i) sh.spec[4 + i] = msc[i];
                    }
                }
                STAMP(7);
            }
            int step_valid = 0;
            double model_cost_change = 0.0, step_ca = 0.0, step_cb = 0.0;
            if (solve_ok) step_valid = dogleg_step(radius, step_ca, step_cb, dogleg_step_norm, model_cost_change);
            STAMP(8);
            if (!step_valid) {
                if (++invalid_steps >= 5) { term = 2; break; }
                mu *= 10.0;
                reuse = 0;
                prev_rejected = 0;
                continue;
            }
            invalid_steps = 0;
            // candidate = Plus(x, delta), its cost and the ambient step norm in one pass
            STAMP(28);
            // A trial that follows an accepted step is accepted as a rule (small solves accept nearly every step; a window solve's
            // run of rejections starts with one miss): it is evaluated WITH its linearisation, so that its acceptance needs no second
            // pass over the factors (cost-only 6.7 us, linearisation 11.3 us at 335 factors: 18.0 -> 11.3 us per accepted step, 4.6 us
            // lost on a miss).  The trials behind a rejection stay cost-only.
            const bool fused = w.fuse_accept && !prev_rejected;
            double sn2 = 0.0;
            double cand_cost = fused ? evaluate<true, true>(wl, sh, phase, wl.xc, wl.xdc, prof_last, &sn2, step_ca, step_cb)
                                     : evaluate<false, true>(wl, sh, phase, wl.xc, wl.xdc, prof_last, &sn2, step_ca, step_cb);
            phase ^= 1;  // (one reduction inside)
            if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
            STAMP(9);
            const double step_norm = sqrt(sn2);
            if (step_norm <= 1e-8 * (x_norm + 1e-8)) { term = 0; break; }
            const double cost_change = x_cost - cand_cost;
            if (fabs(cost_change) <= 1e-6 * x_cost) { term = 0; break; }
            const double rel = (cand_cost >= 1.7976931348623157e308) ? -1.7976931348623157e308 : (x_cost - cand_cost) / model_cost_change;
            if (rel > 1e-3) {
                x_norm = accept_candidate(wl, sh, phase);
                phase ^= 1;  // (one reduction inside)
                accepted_step(rel, fused, cand_cost);
            } else {
                radius *= 0.5;
                reuse = 1;
                prev_rejected = 1;
            }
        }
    if (w.n_wg > 1) post_command(w, sh, CMD_EXIT);
    if (resident) {   // the result lives in LDS: back to the arena ([x | xd | summary] travels to the host with one copy)
        for (int i = t; i < w.nfr * 16; i += T) w.x[i] = wl.x[i];
        for (int l = t; l < w.nl; l += T) w.xd[l] = wl.xd[l];
    }
    // Every thread carries its own copy of the loop's scalars and branches on it; they are computed from identical inputs in
    // identical order, so all copies must agree.  Checked once per launch (one LDS word per wavefront): a disagreement means
    // wavefronts took different paths through the loop -- reported as summary[6] -> RDVIO_ERR_HIP from rdvio_hip_ba_fetch.
    __syncthreads();
    if ((t & 63) == 0) sh.red[0][0][t >> 6] = (double)(iteration * 8 + n_success * 2048 + term);
    __syncthreads();
    int disagree = 0;
    for (int q = 1; q < NW; ++q) disagree |= sh.red[0][0][q] != sh.red[0][0][0];
    if (t == 0) {
        w.summary[6] = disagree ? 1.0 : 0.0;
        w.summary[0] = (double)iteration;
        w.summary[1] = (double)n_success;
        w.summary[2] = initial_cost;
        w.summary[3] = x_cost;
        w.summary[4] = (double)term;
        w.summary[5] = (w.n_wg > 1 && sh.lost) ? 1.0 : 0.0;
#ifdef RDVIO_PROF_CHOL
        for (int i = 0; i < 5; ++i) { w.summary[72 + i] = (double)rdvio_chol_prof[i]; rdvio_chol_prof[i] = 0; }
#endif
    }
}

// CeresMarginalizationFactor::marginalize(0): linearise the marginalisation graph at the current states (no robust
// loss), assemble the normal equations with the solver's own routines, then Schur out the landmarks and the victim
// frame and rebuild the sqrt prior (marg_tail.hpp).
__global__ __launch_bounds__(T) void marginalize_kernel(SolverWs w) {
    __shared__ BlockShared<T> sh_store;
    Shared &sh = *(Shared *)&sh_store;
    constexpr int NMAX = 15 * RDVIO_LDS_CHOL_MAX_FRAMES;
    __shared__ __attribute__((aligned(16))) double lds_buf[NMAX * (NMAX + 1) / 2 + 225 * RDVIO_LDS_CHOL_MAX_FRAMES];
    __shared__ SolverWs w_lds;
    if (w.poison_lds) {
        poison_lds(&sh_store, sizeof(sh_store));
        poison_lds(lds_buf, sizeof(lds_buf));
        __syncthreads();
    }
    for (int i = threadIdx.x; i < (int)(sizeof(SolverWs) / 8); i += T)
        ((__attribute__((address_space(3))) unsigned long long *)&w_lds)[i] = ((const unsigned long long *)&w)[i];
    __syncthreads();
    LdsWs &wl = *(LdsWs *)&w_lds;
    int phase = 0;
    unsigned long long prof_last = 0;
#ifdef RDVIO_PROF
    prof_last = wall_clock64();
    if (threadIdx.x == 0) for (int i = 8; i < 80; ++i) w.summary[i] = 0.0;
#endif
    solver_setup(wl, sh, RDVIO_LDS(lds_buf), sizeof(lds_buf) / sizeof(double), prof_last);
    STAMP(0);
    (void)evaluate<true>(wl, sh, phase, w.x, w.xd, prof_last);
    phase ^= 1;  // (one reduction inside)
    STAMP(1);
    build_normal_equations(wl, sh, prof_last);
    STAMP(2);
    marginalize_tail<T>(wl, sh, phase, RDVIO_LDS(lds_buf));
    STAMP(3);
}

// Unit-parity entry (rdvio_hip_ba_linearize): ONE linearisation of the problem with the solver's own device routines -- setup,
// factor evaluation with Jacobians, normal equations, landmark elimination with unit scaling and zero damping (so that the
// reduced system is plainly H - A^T W A, g - A^T W g_l) -- and nothing else; the host copies the workspace arrays out.
__global__ __launch_bounds__(T) void linearize_kernel(SolverWs w) {
    __shared__ BlockShared<T> sh_store;
    Shared &sh = *(Shared *)&sh_store;
    constexpr int NMAX = 15 * RDVIO_LDS_CHOL_MAX_FRAMES;
    constexpr size_t LDS_CAP = (NMAX + 1) * (NMAX + 2) / 2 + 225 * RDVIO_LDS_CHOL_MAX_FRAMES;
    __shared__ __attribute__((aligned(16))) double lds_buf[LDS_CAP];
    __shared__ SolverWs w_lds;
    const int t = threadIdx.x;
    for (int i = t; i < (int)(sizeof(SolverWs) / 8); i += T)
        ((__attribute__((address_space(3))) unsigned long long *)&w_lds)[i] = ((const unsigned long long *)&w)[i];
    __syncthreads();
    LdsWs &wl = *(LdsWs *)&w_lds;
    int phase = 0;
    unsigned long long prof_last = 0;
    solver_setup(wl, sh, RDVIO_LDS(lds_buf), LDS_CAP, prof_last);
    (void)evaluate<true>(wl, sh, phase, w.x, w.xd, prof_last);
    phase ^= 1;
    for (int i = t; i < 9 * w.np; i += T) w.Jri[i] = sh.Jri[i];   // (the prior's Jr^-1 blocks live in LDS; the host assembles S E from them)
    build_normal_equations(wl, sh, prof_last);
    for (int i = t; i < w.N; i += T) { w.sig_p[i] = 1.0; w.diag_p[i] = 1.0; }
    for (int l = t; l < w.nl; l += T) { w.sig_l[l] = 1.0; w.diag_l[l] = 1.0; }
    __syncthreads();
    schur_reduce(wl, sh, RDVIO_LDS(lds_buf), LDS_CAP, 0.0, prof_last);
    if (w.lds_chol) {   // the packed LDS triangle (+ right-hand-side row) -> the global matrix the host reads (ld N, row N = rhs)
        const int N = w.N;
        for (int o = t; o < (N + 1) * N; o += T) {
            const int i = o / N, j = o - i * N;
            if (j <= i || i == N) w.Sm[(size_t)i * N + j] = lds_buf[tri(i) + j];
        }
    }
}

}  // namespace

// checking builds: RDVIO_UG violations since the last call (0 in product builds)
unsigned rdvio_ug_violations() {
#ifdef RDVIO_CHECK_UG
    unsigned v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_rdvio_ug_violations), sizeof v) != hipSuccess) return 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rdvio_ug_violations), &zero, sizeof zero);
    return v;
#else
    return 0;
#endif
}

void rdvio_launch_ba_linearize(hipStream_t stream, const SolverWs &w0) {
    SolverWs w = w0;
    w.n_wg = 1;
    w.wg_stride = 1;
    hipLaunchKernelGGL(linearize_kernel, dim3(1), dim3(T), 0, stream, w);
}

void rdvio_launch_marginalize(hipStream_t stream, const SolverWs &w0) {
    SolverWs w = w0;
    const char *pl = getenv("RDVIO_TEST_POISON_LDS");
    w.poison_lds = (pl && pl[0] == '1') ? 1 : 0;
    hipLaunchKernelGGL(marginalize_kernel, dim3(1), dim3(T), 0, stream, w);
}

void rdvio_launch_ba_solve(hipStream_t stream, const SolverWs &w0, hipEvent_t ev0, hipEvent_t ev1) {
    SolverWs w = w0;
    const char *ns = getenv("RDVIO_NO_SPECULATION");
    w.no_speculation = (ns && ns[0] == '1') ? 1 : 0;
    const char *mh = getenv("RDVIO_TEST_MUTE_HELPERS");
    w.mute_helpers = (mh && mh[0] == '1') ? 1 : 0;
    const char *pl = getenv("RDVIO_TEST_POISON_LDS");
    w.poison_lds = (pl && pl[0] == '1') ? 1 : 0;
    const char *ss = getenv("RDVIO_NO_SMALL_SOLVE");   // diagnostic: one-frame problems take the general road
    w.small_system = (w.nfree == 1 && w.N == 15 && w.n_lfree_hint == 0 && w.lds_chol && w.n_wg == 1 && !(ss && ss[0] == '1')) ? 1 : 0;
    const char *nf = getenv("RDVIO_NO_FUSED_ACCEPT");   // diagnostic: every trial cost-only, a second pass on acceptance (the round-2 loop)
    w.fuse_accept = (w.n_wg == 1 && !(nf && nf[0] == '1')) ? 1 : 0;
    const char *nv = getenv("RDVIO_NO_LDS_VECTORS");   // diagnostic: keep every vector in global memory (the A/B of the LDS-resident vectors)
    w.no_lds_vectors = (nv && nv[0] == '1') ? 1 : 0;
    const char *sp = getenv("RDVIO_SOLVER_SPREAD");   // diagnostic: one team member per XCD (the round-robin placement of a plain grid)
    w.wg_stride = (w.n_wg > 1 && !(sp && sp[0] == '1')) ? 8 : 1;
    if (ev0 && ev1) hipExtLaunchKernelGGL(ba_solve_kernel, dim3((w.n_wg > 1 ? w.n_wg : 1) * w.wg_stride), dim3(T), 0, stream, ev0, ev1, 0, w);
    else hipLaunchKernelGGL(ba_solve_kernel, dim3((w.n_wg > 1 ? w.n_wg : 1) * w.wg_stride), dim3(T), 0, stream, w);
}
