// LDS capacities of the device-side keypoint selection (select_kernels.hip); inputs beyond them take the host road
// (host_select.cpp), which computes the same result.
#pragma once
#define RDVIO_SEL_NC_MAX 8192        // Harris local maxima (typical frame: 300 - 2500)
#define RDVIO_SEL_GCELLS_MAX 4096    // cells of the minDistance grid (1280x720 at 20 px: 2304)
#define RDVIO_SEL_CORNERS_MAX 4096   // maxCorners
#define RDVIO_SEL_PGRID_MAX 21504    // cells of the Poisson-disk grid (1280x720 at radius 10: 20492)
#define RDVIO_SEL_PTS_MAX 3072       // existing keypoints + corners
