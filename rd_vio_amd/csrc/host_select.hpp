// Host-side, sequential part of keypoint detection (tiny: a few thousand candidates per frame):
// cv::goodFeaturesToTrack's ordering + greedy minDistance grid, then the reference's
// PoissonDiskFilter<2> and 20-px border test (opencv_image.cpp:46-72).
#pragma once
#include "ctx.hpp"

// cand: unordered Harris local maxima from the GPU.  keypoints: in/out (x,y) doubles.
// Returns the new total number of keypoints, or -1 if `capacity` is too small.
int rdvio_host_select_keypoints(HarrisCand *cand, int nc, int w, int h, int max_corners, double gftt_min_dist,
                                double poisson_radius, double *keypoints, int n_existing, int capacity);
