#include "host_select.hpp"

#include <algorithm>
#include <cmath>
#include <unordered_map>
#include <vector>

namespace {

// PoissonDiskFilter<2>, /root/reference/src/rdvio_util/include/rdvio/util/poisson_disk_filter.h.
// Kept literally: a sparse grid with ONE point index per cell (later points overwrite, :20-24) and the
// reference's neighbourhood walk, which skips the first cell and visits one cell past the end (:77-92).
class PoissonDisk2 {
  public:
    explicit PoissonDisk2(double radius)
        : r2_(radius * radius), cell_(radius / std::sqrt(2.0)), span_((int)std::ceil(std::sqrt(2.0))) {}

    void preset(double x, double y) {
        grid_[key(ix(x), ix(y))] = (int)pts_.size() / 2;
        pts_.push_back(x);
        pts_.push_back(y);
    }
    bool insert(double x, double y) {
        const int cx = ix(x), cy = ix(y);
        const int bx = cx - span_, by = cy - span_, ex = cx + span_, ey = cy + span_;
        int x0 = bx, y0 = by;
        while (y0 <= ey) {
            ++x0;
            if (x0 > ex) {
                x0 = bx;
                ++y0;
            }
            auto it = grid_.find(key(x0, y0));
            if (it != grid_.end()) {
                const double dx = x - pts_[2 * it->second], dy = y - pts_[2 * it->second + 1];
                if (dx * dx + dy * dy < r2_) return false;
            }
        }
        grid_[key(cx, cy)] = (int)pts_.size() / 2;
        pts_.push_back(x);
        pts_.push_back(y);
        return true;
    }

  private:
    int ix(double v) const { return (int)std::floor(v / cell_); }
    static int64_t key(int x, int y) { return ((int64_t)x << 32) ^ (uint32_t)y; }
    double r2_, cell_;
    int span_;
    std::vector<double> pts_;
    std::unordered_map<int64_t, int> grid_;
};

}  // namespace

int rdvio_host_select_keypoints(HarrisCand *cand, int nc, int w, int h, int max_corners, double gftt_min_dist,
                                double poisson_radius, double *keypoints, int n_existing, int capacity) {
    // std::sort(tmpCorners, greaterThanPtr()): by response, ties by higher address == higher pixel index -- a strict total
    // order, so the sequence is the same however it is produced.  The greedy selection below usually stops after a few
    // hundred candidates (max_corners accepted), so the list is sorted lazily, one chunk of the best remaining
    // candidates at a time (nth_element + sort of the chunk) instead of all at once.
    const auto before = [](const HarrisCand &a, const HarrisCand &b) { return a.v > b.v || (a.v == b.v && a.idx > b.idx); };
    int sorted = 0;
    const int chunk = max_corners > 0 ? std::max(256, 4 * max_corners) : nc;
    auto ensure_sorted = [&](int upto) {  // candidates [0, upto) in final order
        while (sorted < upto) {
            const int end = std::min(nc, sorted + chunk);
            if (end < nc) std::nth_element(cand + sorted, cand + end, cand + nc, before);
            std::sort(cand + sorted, cand + end, before);
            sorted = end;
        }
    };
    // greedy minDistance selection on a cell grid (goodFeaturesToTrack)
    std::vector<float> corners;  // x,y
    if (gftt_min_dist >= 1) {
        const int cell = (int)std::lrint(gftt_min_dist);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        const double md2 = gftt_min_dist * gftt_min_dist;
        // per-cell singly linked lists in two flat arrays (reused between calls: no per-frame allocation)
        static thread_local std::vector<int> head, next;
        head.assign((size_t)gw * gh, -1);
        next.clear();
        corners.reserve(max_corners > 0 ? 2 * (size_t)max_corners : 512);
        for (int i = 0; i < nc; ++i) {
            ensure_sorted(i + 1);
            const int y = cand[i].idx / w, x = cand[i].idx - y * w;
            const int xc = x / cell, yc = y / cell;
            const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
            const int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (int c = head[(size_t)yy * gw + xx]; c >= 0; c = next[c]) {
                        const float dx = (float)x - corners[2 * c], dy = (float)y - corners[2 * c + 1];
                        if ((double)(dx * dx + dy * dy) < md2) {
                            good = false;
                            break;
                        }
                    }
            if (!good) continue;
            const int id = (int)corners.size() / 2;
            next.push_back(head[(size_t)yc * gw + xc]);
            head[(size_t)yc * gw + xc] = id;
            corners.push_back((float)x);
            corners.push_back((float)y);
            if (max_corners > 0 && id + 1 == max_corners) break;
        }
    } else {
        ensure_sorted(max_corners > 0 ? std::min(nc, max_corners) : nc);
        for (int i = 0; i < nc && (max_corners <= 0 || i < max_corners); ++i) {
            corners.push_back((float)(cand[i].idx % w));
            corners.push_back((float)(cand[i].idx / w));
        }
    }
    // (the reference re-sorts by KeyPoint::response here, opencv_image.cpp:47-50; GFTT output is already in that order)
    int total = n_existing;
    if (!corners.empty()) {
        PoissonDisk2 filter(poisson_radius);
        for (int i = 0; i < n_existing; ++i) filter.preset(keypoints[2 * i], keypoints[2 * i + 1]);
        for (size_t i = 0; i < corners.size() / 2; ++i) {
            const double x = corners[2 * i], y = corners[2 * i + 1];
            if (!filter.insert(x, y)) continue;
            if (x < 20 || y < 20 || x >= w - 20 || y >= h - 20) continue;
            if (total >= capacity) return -1;
            keypoints[2 * total] = x;
            keypoints[2 * total + 1] = y;
            ++total;
        }
    }
    return total;
}
