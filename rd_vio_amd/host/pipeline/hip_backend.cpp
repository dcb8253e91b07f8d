// The product backend of the orchestration: every seam call goes to librdvio_hip.so (include/rdvio_hip.h).
// Images live in the context's two pyramid slots -- the tracker keeps exactly two consecutive frames alive
// (feature_tracker.cpp:94), so frame k uses slot k % 2.
#include <atomic>
#include <cstring>
#include <memory>
#include <vector>

#include "../../../include/rdvio_pipeline.h"

namespace {

struct HipImage {
    std::vector<uint8_t> gray;  // OpenCvImage::image: a private copy, only while the image could not go to the device at once
    int width, height, slot;
    bool preprocessed = false;
    double clip = 0.0;
    int tx = 0, ty = 0;
};

struct HipBackend {
    rdvio_hip_ctx *ctx;
    // CLAHE parameters of the configuration: what the prefetch at image_create applies (preprocess re-runs if asked otherwise)
    double clip;
    int tx, ty;
    int next_slot = 0;
    // which image holds each pyramid slot (nullptr = free).  image_destroy can run on the estimator's thread (the last clone of
    // a frame dies there), everything else on the tracker's.
    std::atomic<HipImage *> owner[2] = {nullptr, nullptr};
};

// Odometry::addFrame hands the image over before the tracker asks for it (handler.cpp:113-138 only enqueues): when the slot this
// frame will use is free -- the frame two before it was released at the end of the previous tracker step -- the pixels go to
// the device and CLAHE + pyramid are enqueued right away, behind whatever the frontend lane still has to do; the tracker's
// preprocess call then finds the work done (or under way).  Otherwise the image waits in a private host copy, as before.
int image_create(void *user, const uint8_t *gray, int width, int height, int stride, void **out) {
    auto *b = static_cast<HipBackend *>(user);
    auto *img = new HipImage();
    img->width = width;
    img->height = height;
    img->slot = -1;
    *out = img;
    const int slot = b->next_slot;
    HipImage *expected = nullptr;
    if (b->owner[slot].compare_exchange_strong(expected, img)) {
        b->next_slot ^= 1;
        img->slot = slot;
        int rc = rdvio_hip_image_upload(b->ctx, slot, gray, width, height, stride);
        if (rc == RDVIO_OK) rc = rdvio_hip_image_preprocess_uploaded(b->ctx, slot, b->clip, b->tx, b->ty);
        img->preprocessed = rc == RDVIO_OK;
        img->clip = b->clip; img->tx = b->tx; img->ty = b->ty;
        return rc;
    }
    img->gray.resize((size_t)width * height);
    for (int y = 0; y < height; ++y) std::memcpy(&img->gray[(size_t)y * width], gray + (size_t)y * stride, (size_t)width);
    return RDVIO_OK;
}

int image_preprocess(void *user, void *image, double clip, int tx, int ty) {
    auto *b = static_cast<HipBackend *>(user);
    auto *img = static_cast<HipImage *>(image);
    if (img->preprocessed) {
        if (img->clip == clip && img->tx == tx && img->ty == ty) return RDVIO_OK;
        // other parameters than the prefetch assumed: the pixels are still in the slot's device staging buffer
        img->clip = clip; img->tx = tx; img->ty = ty;
        return rdvio_hip_image_preprocess_uploaded(b->ctx, img->slot, clip, tx, ty);
    }
    if (img->gray.empty()) return RDVIO_ERR_INVALID;
    const int slot = b->next_slot;
    b->next_slot ^= 1;
    b->owner[slot].store(img);   // (the tracker keeps two consecutive frames alive: whoever held the slot is done with it)
    img->slot = slot;
    const int rc = rdvio_hip_image_preprocess(b->ctx, slot, img->gray.data(), img->width, img->height, img->width, clip, tx, ty);
    std::vector<uint8_t>().swap(img->gray);  // rdvio_hip_image_preprocess has read the pixels when it returns
    img->preprocessed = rc == RDVIO_OK;
    img->clip = clip; img->tx = tx; img->ty = ty;
    return rc;
}

int image_detect(void *user, void *image, double *kps, int n_existing, int capacity, int max_points, double min_distance, int *n_out) {
    auto *b = static_cast<HipBackend *>(user);
    return rdvio_hip_detect_keypoints(b->ctx, static_cast<HipImage *>(image)->slot, kps, n_existing, capacity, max_points, min_distance, n_out);
}

int image_track(void *user, void *curr, void *next, int n, const double *curr_xy, double *next_xy, int has_guess, uint8_t *status) {
    auto *b = static_cast<HipBackend *>(user);
    return rdvio_hip_track_keypoints(b->ctx, static_cast<HipImage *>(curr)->slot, static_cast<HipImage *>(next)->slot, n, curr_xy, next_xy,
                                     has_guess, status);
}

void image_release(void *user, void *image) {
    auto *b = static_cast<HipBackend *>(user);
    auto *img = static_cast<HipImage *>(image);
    if (img->slot >= 0) {
        HipImage *expected = img;
        if (b->owner[img->slot].compare_exchange_strong(expected, nullptr)) (void)rdvio_hip_image_release(b->ctx, img->slot);
    }
    img->slot = -1;
}

void image_destroy(void *user, void *image) {
    auto *b = static_cast<HipBackend *>(user);
    auto *img = static_cast<HipImage *>(image);
    // (an image that still holds its slot -- never released by the tracker -- gives it back; no device call from this thread)
    for (int s = 0; s < 2; ++s) {
        HipImage *expected = img;
        (void)b->owner[s].compare_exchange_strong(expected, nullptr);
    }
    delete img;
}

int preintegrate(void *user, int nseg, const int32_t *seg_off, const double *imu, const double *t_end, const double *bg, const double *ba,
                 const double *noise, int cj, int cc, double *out) {
    return rdvio_hip_preintegrate(static_cast<HipBackend *>(user)->ctx, nseg, seg_off, imu, t_end, bg, ba, noise, cj, cc, out);
}

int preintegrate_estimator(void *user, int nseg, const int32_t *seg_off, const double *imu, const double *t_end, const double *bg, const double *ba,
                           const double *noise, int cj, int cc, double *out) {
    return rdvio_hip_preintegrate_estimator(static_cast<HipBackend *>(user)->ctx, nseg, seg_off, imu, t_end, bg, ba, noise, cj, cc, out);
}

int thread_attach(void *user) { return rdvio_hip_ctx_attach_thread(static_cast<HipBackend *>(user)->ctx); }

int ba_solve(void *user, const rdvio_ba_problem *pb, int max_iter, double *states, double *invd, rdvio_ba_summary *sm) {
    return rdvio_hip_ba_solve(static_cast<HipBackend *>(user)->ctx, pb, max_iter, states, invd, sm);
}

int ba_solve_begin(void *user, int slot, const rdvio_ba_problem *pb, int max_iter, int from_slot, int from_frame, int to_frame) {
    rdvio_hip_ctx *ctx = static_cast<HipBackend *>(user)->ctx;
    if (int rc = from_slot >= 0 ? rdvio_hip_ba_upload_chained(ctx, slot, pb, from_slot, from_frame, to_frame) : rdvio_hip_ba_upload(ctx, slot, pb)) return rc;
    if (int rc = rdvio_hip_ba_solve_resident(ctx, slot, max_iter)) return rc;
    if (from_slot < 0) return RDVIO_OK;
    // the last solve of a chain: both results' copies go onto the lane now, behind it -- the two ends then share one wait
    if (int rc = rdvio_hip_ba_fetch_enqueue(ctx, from_slot)) return rc;
    return rdvio_hip_ba_fetch_enqueue(ctx, slot);
}
int ba_solve_end(void *user, int slot, double *states, double *invd, rdvio_ba_summary *sm) {
    return rdvio_hip_ba_fetch(static_cast<HipBackend *>(user)->ctx, slot, states, invd, sm);
}

int marginalize(void *user, const rdvio_marg_problem *pb, double *S, double *f, double *lin) {
    return rdvio_hip_marginalize(static_cast<HipBackend *>(user)->ctx, pb, 0, S, f, lin, nullptr, nullptr, nullptr);
}

int marginalize_begin(void *user, const rdvio_marg_problem *pb) {
    rdvio_hip_ctx *ctx = static_cast<HipBackend *>(user)->ctx;
    if (int rc = rdvio_hip_marginalize_upload(ctx, pb)) return rc;
    return rdvio_hip_marginalize_resident(ctx, 0);
}

int marginalize_end(void *user, double *S, double *f, double *lin) {
    return rdvio_hip_marginalize_fetch(static_cast<HipBackend *>(user)->ctx, S, f, lin, nullptr, nullptr, nullptr);
}

int parsac_score(void *user, const rdvio_parsac_batch *batch, rdvio_parsac_result *results) {
    return rdvio_hip_parsac_score(static_cast<HipBackend *>(user)->ctx, batch, results);
}

int parsac_fetch(void *user, int model, uint8_t *mask, int32_t *bin_inliers) {
    return rdvio_hip_parsac_fetch(static_cast<HipBackend *>(user)->ctx, model, mask, bin_inliers);
}

int ransac_generate_score(void *user, int kind, int n, int changed, const double *pa, const double *pb, double thr, int n_iter, const int32_t *samples,
                          int32_t *per_iter, double *models, int32_t *counts) {
    return rdvio_hip_ransac_generate_score(static_cast<HipBackend *>(user)->ctx, kind, n, changed, pa, pb, thr, n_iter, samples, per_iter, models, counts);
}
int ransac_fetch(void *user, int model, uint8_t *mask) { return rdvio_hip_ransac_fetch(static_cast<HipBackend *>(user)->ctx, model, mask); }
int preintegrate_estimator_begin(void *user, int nseg, const int32_t *seg_off, const double *imu, const double *t_end, const double *bg, const double *ba,
                                 const double *noise, int cj, int cc) {
    return rdvio_hip_preintegrate_estimator_begin(static_cast<HipBackend *>(user)->ctx, nseg, seg_off, imu, t_end, bg, ba, noise, cj, cc);
}
int preintegrate_estimator_end(void *user, double *out) { return rdvio_hip_preintegrate_estimator_end(static_cast<HipBackend *>(user)->ctx, out); }

int thin_tracks(void *user, int w, int h, double radius, int n, const double *xy, int n_order, const int32_t *order, const uint8_t *trash, uint8_t *keep) {
    return rdvio_hip_thin_tracks(static_cast<HipBackend *>(user)->ctx, w, h, radius, n, xy, n_order, order, trash, keep);
}

int parsac_generate_score(void *user, const rdvio_parsac_batch *batch, int n_iter, const int32_t *samples, int32_t *per_iter, double *models,
                          rdvio_parsac_result *results) {
    return rdvio_hip_parsac_generate_score(static_cast<HipBackend *>(user)->ctx, batch, n_iter, samples, per_iter, models, results);
}

const char *last_error(void *user) { return rdvio_hip_last_error(static_cast<HipBackend *>(user)->ctx); }

void destroy(void *user) { delete static_cast<HipBackend *>(user); }

}  // namespace

extern "C" int rdvio_pipeline_create_hip(rdvio_pipeline **out, const rdvio_pipeline_config *cfg, rdvio_hip_ctx *ctx) {
    if (!out || !cfg || !ctx) return RDVIO_ERR_INVALID;
    // tracker and estimator on two host threads: each needs a stream of its own (a shared stream would make every wait of
    // one thread wait for the other's kernels)
    if (cfg->threading == 2)
        if (int rc = rdvio_hip_ctx_ensure_lane_streams(ctx)) return rc;
    auto *b = new HipBackend{ctx, cfg->feature_tracker_clahe_clip_limit, cfg->feature_tracker_clahe_width, cfg->feature_tracker_clahe_height};
    rdvio_backend fn;
    fn.user = b;
    fn.image_create = image_create;
    fn.image_preprocess = image_preprocess;
    fn.image_detect = image_detect;
    fn.image_track = image_track;
    fn.image_release = image_release;
    fn.image_destroy = image_destroy;
    fn.preintegrate = preintegrate;
    fn.ba_solve = ba_solve;
    fn.marginalize = marginalize;
    fn.last_error = last_error;
    fn.destroy = destroy;
    fn.parsac_score = parsac_score;
    fn.parsac_fetch = parsac_fetch;
    fn.preintegrate_estimator = preintegrate_estimator;
    fn.thread_attach = thread_attach;
    fn.parsac_generate_score = parsac_generate_score;
    fn.ransac_generate_score = ransac_generate_score;
    fn.ransac_fetch = ransac_fetch;
    fn.thin_tracks = thin_tracks;
    fn.ba_solve_begin = ba_solve_begin;
    fn.ba_solve_end = ba_solve_end;
    fn.preintegrate_estimator_begin = preintegrate_estimator_begin;
    fn.preintegrate_estimator_end = preintegrate_estimator_end;
    fn.marginalize_begin = marginalize_begin;
    fn.marginalize_end = marginalize_end;
    const int rc = rdvio_pipeline_create(out, cfg, &fn);
    if (rc != RDVIO_OK) delete b;
    return rc;
}
