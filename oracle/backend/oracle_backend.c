/* ORACLE / TEST INFRASTRUCTURE: an rdvio_backend (include/rdvio_pipeline.h) over the CPU oracle, so that the test-suite and
 * bench.py's cpu_baseline leg can run the product's host orchestration over the CPU path and compare trajectories / feature
 * indices with the HIP path (SURVEY.md 8d metrics 2 and 3).  Built by oracle.build_backend() into oracle/_build/; never part
 * of the product (nothing under rd_vio_amd/ refers to it). */
#include <stdlib.h>
#include <string.h>

#include "../../include/rdvio_pipeline.h"
#include "../rdvio_oracle.h"

typedef struct {
    uint8_t *gray;
    int w, h;
    ro_pyr_layout L;
    uint8_t *pyr_img;
    int16_t *pyr_deriv;
} ob_image;

static int ob_image_create(void *user, const uint8_t *gray, int w, int h, int stride, void **out) {
    (void)user;
    ob_image *im = (ob_image *)calloc(1, sizeof *im);
    im->w = w;
    im->h = h;
    im->gray = (uint8_t *)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y) memcpy(im->gray + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
    *out = im;
    return RDVIO_OK;
}

static int ob_image_preprocess(void *user, void *image, double clip, int tx, int ty) {
    (void)user;
    ob_image *im = (ob_image *)image;
    ro_pyr_layout_init(im->w, im->h, 3, &im->L);
    im->pyr_img = (uint8_t *)calloc((size_t)im->L.img_bytes, 1);
    im->pyr_deriv = (int16_t *)calloc((size_t)im->L.deriv_elems, sizeof(int16_t));
    ro_preprocess(im->gray, im->w, im->h, im->w, clip, tx, ty, &im->L, im->pyr_img, im->pyr_deriv);
    free(im->gray);
    im->gray = NULL;
    return RDVIO_OK;
}

static int ob_image_detect(void *user, void *image, double *kps, int n_existing, int capacity, int max_points, double min_distance,
                           int *n_out) {
    (void)user;
    ob_image *im = (ob_image *)image;
    if (!im->pyr_img || capacity < n_existing + max_points) return RDVIO_ERR_CAPACITY;
    const uint8_t *lvl0 = im->pyr_img + im->L.img_off[0] + (int64_t)im->L.border * im->L.stride[0] + im->L.border;
    *n_out = ro_detect_keypoints(lvl0, im->w, im->h, im->L.stride[0], max_points, min_distance, kps, n_existing);
    return RDVIO_OK;
}

static int ob_image_track(void *user, void *curr, void *next, int n, const double *curr_xy, double *next_xy, int has_guess,
                          uint8_t *status) {
    (void)user;
    ob_image *a = (ob_image *)curr, *b = (ob_image *)next;
    if (!a->pyr_img || !b->pyr_img) return RDVIO_ERR_INVALID;
    ro_track_keypoints(&a->L, a->pyr_img, a->pyr_deriv, b->pyr_img, b->pyr_deriv, n, curr_xy, next_xy, has_guess, status);
    return RDVIO_OK;
}

static void ob_image_release(void *user, void *image) {
    (void)user;
    ob_image *im = (ob_image *)image;
    free(im->pyr_img);
    free(im->pyr_deriv);
    im->pyr_img = NULL;
    im->pyr_deriv = NULL;
}

static void ob_image_destroy(void *user, void *image) {
    ob_image *im = (ob_image *)image;
    ob_image_release(user, image);
    free(im->gray);
    free(im);
}

static int ob_preintegrate(void *user, int nseg, const int32_t *seg_off, const double *imu, const double *t_end, const double *bg,
                           const double *ba, const double *noise, int cj, int cc, double *out) {
    (void)user;
    for (int i = 0; i < nseg; ++i)
        ro_preintegrate(seg_off[i + 1] - seg_off[i], imu + 7 * (size_t)seg_off[i], t_end[i], bg + 3 * i, ba + 3 * i, noise, cj, cc,
                        out + (size_t)RO_PREINT_SIZE * i);
    return RDVIO_OK;
}

static int ob_ba_solve(void *user, const rdvio_ba_problem *pb, int max_iter, double *states, double *invd, rdvio_ba_summary *sm) {
    (void)user;
    ro_ba_problem q;
    memset(&q, 0, sizeof q);
    q.n_frames = pb->n_frames; q.frame_fixed = pb->frame_fixed; q.extr = pb->extr; q.sqrt_inv_cov = pb->sqrt_inv_cov;
    q.n_landmarks = pb->n_landmarks; q.lm_fixed = pb->lm_fixed; q.z_ref = pb->z_ref;
    q.n_factors = pb->n_factors; q.tgt = pb->tgt; q.ref = pb->ref; q.lm = pb->lm; q.tangent = pb->tangent;
    q.n_rot = pb->n_rot; q.rot_tgt = pb->rot_tgt; q.rot_ref = pb->rot_ref; q.rot_zref = pb->rot_zref; q.rot_tangent = pb->rot_tangent;
    q.n_preint = pb->n_preint; q.pre_i = pb->pre_i; q.pre_j = pb->pre_j; q.preint = pb->preint;
    if (pb->n_pre_jobs > 0) { /* the fused form: integrate the records first (PreIntegrator::integrate with covariance and Jacobians) */
        for (int k = 0; k < pb->n_pre_jobs; ++k)
            ro_preintegrate(pb->job_seg_off[k + 1] - pb->job_seg_off[k], pb->job_imu + 7 * (size_t)pb->job_seg_off[k], pb->job_par[7 * k],
                            pb->job_par + 7 * k + 1, pb->job_par + 7 * k + 4, pb->job_noise, 1, 1, pb->job_preint_out + (size_t)RO_PREINT_SIZE * k);
        q.preint = pb->job_preint_out;
    }
    q.np = pb->n_prior; q.prior_frames = pb->prior_frames; q.lin = pb->prior_lin; q.S = pb->prior_S; q.f = pb->prior_f;
    memcpy(states, pb->states, sizeof(double) * 16 * (size_t)pb->n_frames);
    if (pb->n_landmarks > 0) memcpy(invd, pb->inv_depth, sizeof(double) * (size_t)pb->n_landmarks);
    ro_ba_summary s;
    memset(&s, 0, sizeof s);
    ro_ba_solve(&q, max_iter, states, invd, &s);
    if (sm) {
        sm->iterations = s.iterations; sm->successful_steps = s.successful_steps;
        sm->initial_cost = s.initial_cost; sm->final_cost = s.final_cost; sm->termination = s.termination;
    }
    return RDVIO_OK;
}

static int ob_marginalize(void *user, const rdvio_marg_problem *pb, double *S, double *f, double *lin) {
    (void)user;
    ro_marg_problem q;
    memset(&q, 0, sizeof q);
    q.nframes = pb->n_frames; q.states = pb->states; q.extr = pb->extr; q.sqrt_inv_cov = pb->sqrt_inv_cov;
    q.np = pb->n_prior; q.prior_frames = pb->prior_frames; q.lin = pb->prior_lin; q.S = pb->prior_S; q.f = pb->prior_f;
    q.preint01 = pb->preint01;
    q.nfac = pb->n_factors; q.tgt = pb->tgt; q.ref = pb->ref; q.lm = pb->lm; q.tangent = pb->tangent;
    q.nlm = pb->n_landmarks; q.z_ref = pb->z_ref; q.inv_depth = pb->inv_depth;
    ro_marginalize(&q, S, f, lin, NULL, NULL);
    return RDVIO_OK;
}

static const char *ob_last_error(void *user) {
    (void)user;
    return "oracle backend";
}

void rdvio_oracle_backend_fill(rdvio_backend *b) {
    b->user = NULL;
    b->image_create = ob_image_create;
    b->image_preprocess = ob_image_preprocess;
    b->image_detect = ob_image_detect;
    b->image_track = ob_image_track;
    b->image_release = ob_image_release;
    b->image_destroy = ob_image_destroy;
    b->preintegrate = ob_preintegrate;
    b->ba_solve = ob_ba_solve;
    b->marginalize = ob_marginalize;
    b->last_error = ob_last_error;
    b->destroy = NULL;
    b->parsac_score = NULL; /* the CPU path scores hypotheses with the orchestration's own host code (parsac.hpp) */
    b->parsac_fetch = NULL;
    b->preintegrate_estimator = NULL; /* ro_preintegrate keeps no state: safe from both threads */
    b->thread_attach = NULL;
    b->marginalize_begin = NULL;
    b->marginalize_end = NULL;
    b->ransac_generate_score = NULL; /* the CPU path runs the gates and the thinning in the orchestration's host code */
    b->ransac_fetch = NULL;
    b->thin_tracks = NULL;
    b->parsac_generate_score = NULL;
    b->preintegrate_estimator_begin = NULL;
    b->preintegrate_estimator_end = NULL;
    b->ba_solve_begin = NULL; /* the CPU path solves one problem after the other */
    b->ba_solve_end = NULL;
}
