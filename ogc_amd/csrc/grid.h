// grid.h — shared declarations of the cell-list acceleration (grid.hip) used by ball_query.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace ogc_grid {

struct GridHdr { // one per cloud, written by grid_build_kernel
    float minx, miny, minz, inv_h;
    int gx, gy, gz, npts; // npts = points with finite coordinates (the ones inserted)
    int dense;            // 1: the 27-cell neighbourhood holds a large share of the cloud -> all-pairs scan instead
    int heavy;            // 1: 120 or more candidates per centre: lists too long for ball_query_cells_kernel's register sort
    int knn_general;      // k-NN builds: 1 = knn_cells_kernel leaves this cloud to knn_grid_kernel (cells shorter than the radius, or crowded)
    int pending;          // set by knn_cells_kernel when it left rows of this cloud to knn_grid_kernel (marked idx[row][0] = -1)
    int fast;             // which coordinate runs fastest in the cell order: the grid's axes (fast, mid, slow) are (x, y, z) for 0,
                          // (y, x, z) for 1, (z, x, y) for 2; minx / gx belong to the fast axis, miny / gy to the middle one, ...
    int slab;             // 1: gx <= 2 — the cells (any x, y - 1 .. y + 1) of one z are ONE contiguous run: three runs per centre
};

} // namespace ogc_grid

// Queues build + query on `s` (clouds the build flags dense are scanned in index order by the same kernel).
// Returns OGC_OK, or OGC_ERR_UNSUPPORTED when the caller should run the all-pairs scan instead.
int ogc_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                        int *idx, hipStream_t s);

// Exact k-NN over cell lists (mode 0: squared distances, 1: sqrt + radius clamp).  OGC_OK, or OGC_ERR_UNSUPPORTED when
// the caller should run the all-pairs scan (small clouds, k too large for the LDS budget).
int ogc_knn_grid(int mode, int b, int n, int m, int k, float radius, const float *unknown, const float *known,
                 float *dist, int *idx, hipStream_t s);

// three_nn over cell lists (known clouds of 1024 points or more).  OGC_OK, or OGC_ERR_UNSUPPORTED: the caller runs its scan.
int ogc_three_nn_grid(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, hipStream_t s);
