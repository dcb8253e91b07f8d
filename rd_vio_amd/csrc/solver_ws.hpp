// Workspace descriptor of the persistent BA solver kernel (device pointers; passed by value).
#pragma once
#include <cstdint>

#ifndef RDVIO_SOLVER_THREADS
#define RDVIO_SOLVER_THREADS 512
#endif
// per-factor record of the stored linearisation: Jt[12] Jr[12] Jd[2] r[2] ht[6] hr[6] m g
#define RDVIO_FAC_STRIDE 42
#define RDVIO_REC_STRIDE 26
#define RDVIO_LDS_CHOL_MAX_FRAMES 11
#define RDVIO_MAX_SOLVER_WGS 16
// doubles of the LDS vector operand BlockShared::xv: bounds 15 * n_prior (prior error), N = 15 * free frames (<= 480)
#define RDVIO_SOLVER_XV 512
#define RDVIO_HELPER_MIN_FACTORS 4096

struct SolverWs {
    // ---- problem (read-only on the device)
    int nfr, nl, nf, nrot, npre, np, D, nfree, N, npairs, max_iter, n_lfree_hint;
    const uint8_t *lm_fixed;
    const double *extr;  // 14 extrinsics + 4 sqrt_inv_cov
    const double *z_ref;
    const int32_t *tgt, *ref, *lm;
    const double *tangent;
    const int32_t *rot_tgt, *rot_ref;
    const double *rot_zref, *rot_tangent;
    const int32_t *pre_i, *pre_j;
    const double *preint;
    const int32_t *prior_frames;
    const double *lin, *S, *f;
    // graph index structures (built on the host with the problem: A16, sliding_window_tracker.cpp:226-300)
    const int32_t *fcol;                 // frame -> free slot or -1
    const uint8_t *frame_fixed;          // per frame: 0 free, 1 constant, 2 pose constant / motion free
    const int32_t *lm_first, *lm_count;  // factors of a landmark are contiguous
    const int32_t *pair_fi, *pair_fj, *grp_off, *diag_pair;  // frame pairs (lo <= hi); factor groups per pair
    const int32_t *gslot, *gflip;        // per factor: record slot in group order (-1: none), 1 if the first slot holds Jr
    const int32_t *band_src;             // nfree x 3 x 2: preintegration sources of H block (c, c+which-1): k*4+x*2+y or -1
    const int32_t *g_src;                // nfree x 2: preintegration sources of g block c: k*2+x or -1
    const int32_t *pcol;                 // nfree: prior frame index of a free column block or -1
    int nrec;
    int lds_chol, lds_bytes;             // packed LDS Cholesky when 15-blocks of S (+ inverses) fit in LDS
    int n_wg;                            // workgroups of the launch: 1 leader + helpers for the factor evaluation
    unsigned *sync;                      // [0] command sequence, [1] completion counter, [2] command (zeroed before each launch)
    double *partial;                     // per-workgroup partial costs
    // ---- state
    const double *x0, *xd0;              // uploaded initial values (every launch restarts from them)
    const double *user0;                 // unit-parity entry only: the bias linearisation states (NULL: x0, as Solver::solve starts)
    double *x, *xd;                      // in/out: frame states, inverse depths
    double *xc, *xdc, *user;
    uint8_t *lfree;
    // ---- stored linearisation
    double *fac;                         // nf x RDVIO_FAC_STRIDE
    double *prec;                        // nrec x RDVIO_REC_STRIDE: group-ordered [J_lo(12) J_hi(12) r(2)] records
    double *GP;                          // npairs x 256: per group the 16x16 MFMA tile [J_lo J_hi r]^T [J_lo J_hi r]
    double *PP, *Pg;                     // npre x 900 / npre x 30: per preintegration factor [Ji Jj]^T [Ji Jj], [Ji Jj]^T r
    double *ST;                          // D x D: transpose of the prior's S (coalesced S e)
    double *r_r, *Jro;
    double *e_p, *G, *r_p, *c_p, *Jp;    // Jp: per factor [Ji 225 | Jj 225]
    double *e_m, *r_m, *c_m, *Jri, *Lam, *eta0, *le, *Ex;
    // ---- normal equations / step
    double *H, *Sm, *g, *yp, *Cm;        // Cm: (6 nfree + 2)^2 landmark Schur term A^T W [A | g] (gradient part in column 6 nfree)
    double *Cmp;                         // n_wg > 1: n_wg partial Schur terms (one per workgroup's run of landmarks), summed by the consumers
    double *lm_m, *lm_g, *lm_w, *A, *yl;
    double *sig_p, *sig_l, *diag_p, *diag_l, *grad_p, *grad_l, *gn_p, *gn_l, *tp, *tl;
    double *summary;
    // ---- marginalisation mode (rdvio_hip_marginalize): the same linearisation + normal equations without the robust
    // loss, then the victim frame's Schur complement and the new sqrt prior (marg_tail.hpp)
    int no_loss, marg_force_eigen;
    int no_speculation;                  // diagnostic switch (RDVIO_NO_SPECULATION): trial steps one by one, for the equivalence test
    int mute_helpers;                    // test switch (RDVIO_TEST_MUTE_HELPERS): helper workgroups exit at once
    int poison_lds;                      // test switch (RDVIO_TEST_POISON_LDS): every workgroup fills its LDS with 0xFF bytes first
    int no_lds_vectors;                  // diagnostic switch (RDVIO_NO_LDS_VECTORS): no LDS-resident small vectors
    int fuse_accept;                     // trial steps that follow an accepted step are evaluated WITH their linearisation (set at launch)
    // a solve that continues another one (rdvio_hip_ba_upload_chained): frame chain_frame starts from the 16 doubles at chain_src
    // (the other solve's result, in its arena) instead of its row of x0
    const double *chain_src;
    int chain_frame;
    int small_system;                    // one free frame, no free landmark (N = 15): the one-wavefront solve (set at launch)
    int wg_stride;                       // multi-workgroup launches: every wg_stride-th block of the grid is a team member (8: one XCD)
    double *m_Tm, *m_Lr, *m_er, *m_Wk, *m_V, *m_cs, *m_yv;
    int32_t *m_nz;
    double *S_out, *f_out, *lin_out, *Lambda_out, *eta_out, *m_info;
    double *prof;                        // diagnostic phase stamps (RDVIO_PROF builds only)                     // iterations, successful steps, initial cost, final cost, termination
};

#ifdef __HIPCC__
// The by-value kernel argument cannot be handed to a noinline function without a private (scratch) copy, and every
// `w.field` read there would be an L2 round trip ahead of the access it feeds; the kernels keep one copy in LDS instead.
typedef const __attribute__((address_space(3))) SolverWs LdsWs;
#endif

// ev0 / ev1 (both or neither): HIP events attached to the launch itself (hipExtLaunchKernelGGL: the dispatch's own start / stop
// timestamps, no marker packets on the stream)
void rdvio_launch_ba_solve(hipStream_t stream, const SolverWs &w, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
void rdvio_launch_marginalize(hipStream_t stream, const SolverWs &w);
void rdvio_launch_ba_linearize(hipStream_t stream, const SolverWs &w);
unsigned rdvio_ug_violations();   // RDVIO_CHECK_UG builds: global-typed accesses that were handed an LDS address (else 0)
