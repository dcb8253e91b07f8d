// Workspace descriptor of the marginalisation kernel (device pointers; passed by value).
#pragma once
#include <cstdint>

#define RDVIO_MARG_THREADS 512

struct MargWs {
    int nfm, np, D, nf, nl, npairs, has_pre, force_eigen;
    const double *states, *extr;  // extr: 14 + 4
    const int32_t *prior_frames;
    const double *lin, *S, *f, *preint;
    const double *z_ref, *inv_depth;
    const int32_t *tgt, *ref, *lm;
    const double *tangent;
    const int32_t *fidx, *lm_first, *lm_count, *pair_fi, *pair_fj, *pair_off, *pair_item, *diag_pair;
    // scratch
    double *e_m, *r_m, *Jri, *Lam, *le;
    double *e_p, *G, *r_p, *Jp;
    double *r_f, *Jt, *Jr, *Jd;
    double *A, *lm_m, *lm_g, *lm_w;
    double *H, *eta, *Tm, *Lr, *er, *Wk, *V, *cs, *yv;
    int32_t *nz;
    // outputs
    double *S_out, *f_out, *lin_out, *Lambda_out, *eta_out, *info;
    double *prof;  // diagnostic phase stamps (RDVIO_PROF builds only)
};

void rdvio_launch_marginalize(hipStream_t stream, const MargWs &w);
