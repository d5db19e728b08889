// grid.hip — uniform-grid (cell-list) acceleration of the fixed-radius query, with results identical to the
// brute-force scan.
//
// The reference's ball query (ball_query_gpu.cu:9-45) tests every centre against every point: 8*N*M flop for
// ~2.3 MB of input/output per 8192-point cloud, i.e. compute-bound by two orders of magnitude.  A radius query only
// needs the points of the 27 cells around the centre when the cell edge is >= the radius.  Per call:
//   grid_build_kernel   one workgroup per cloud: bounding box -> cell edge h >= 1.01 r (enlarged until the grid has
//                       <= GRID_MAX_CELLS cells) -> LDS histogram -> scan -> scatter: cell_start[], and the points
//                       re-ordered by cell (indices + coordinates, so the query reads contiguous runs);
//   ball_query_grid     one lane per centre, centres taken in cell order (neighbouring lanes walk the same runs);
//                       the 3 x-adjacent cells of a (y, z) pair are one contiguous run, so 9 runs per centre; the
//                       squared distance uses the reference's exact fp32 expression; hits go to a per-lane LDS
//                       max-heap keyed by point index that keeps the `nsample` SMALLEST indices; heap-sort ->
//                       ascending -> padded with the first -> the row the reference produces by scanning in index
//                       order and stopping after nsample hits.
// Exactness: a hit satisfies |dx| < r in every axis, the cell coordinate is floor((x - min) / h) with h >= 1.01 r, so
// the cell coordinates of a centre and any of its hits differ by at most one even with fp32 rounding of the
// quotient (relative error 1e-7 * up to 16384 cells << 0.01); points with non-finite coordinates can never be
// hits (their distance is inf/NaN) and are left out of the grid.
#include "ogc_common.h"
#include "grid.h"

namespace ogc_grid {

constexpr int GRID_MAX_CELLS = 16384;
constexpr int BUILD_THREADS = 1024;

__device__ __forceinline__ int cell_coord(float x, float mn, float inv_h, int g) {
    // floor((x - mn) * inv_h) clamped to [-2, g + 1]; NaN -> -2 (outside every neighbourhood)
    const float f = floorf((x - mn) * inv_h);
    if (!(f >= -2.0f)) return -2;
    if (f > (float)(g + 1)) return g + 1;
    return (int)f;
}

__global__ __launch_bounds__(BUILD_THREADS) void grid_build_kernel(int n, float radius, int knn_k, int stride_cells,
                                                                   const float *__restrict__ xyz,
                                                                   GridHdr *__restrict__ hdrs,
                                                                   int *__restrict__ cell_start,
                                                                   int *__restrict__ sorted_idx,
                                                                   float *__restrict__ sorted_xyz) {
    __shared__ int s_cnt[GRID_MAX_CELLS]; // histogram -> exclusive starts -> scatter cursors
    __shared__ float s_red[6][BUILD_THREADS / 64];
    __shared__ int s_part[BUILD_THREADS];
    __shared__ GridHdr s_hdr;
    __shared__ int s_tail; // cursor for points left out of the grid (non-finite coordinates)
    const int t = threadIdx.x, b = blockIdx.x;
    const float *pts = xyz + (size_t)b * n * 3;

    // 1. bounding box of the finite points
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = t; k < n; k += BUILD_THREADS) {
        const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
            mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
        if ((t & 63) == 0) {
            s_red[a][t >> 6] = mn[a];
            s_red[3 + a][t >> 6] = mx[a];
        }
    }
    for (int c = t; c < GRID_MAX_CELLS; c += BUILD_THREADS) s_cnt[c] = 0;
    __syncthreads();
    if (t == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = s_red[a][0];
            hi[a] = s_red[3 + a][0];
            for (int w = 1; w < BUILD_THREADS / 64; ++w) {
                lo[a] = fminf(lo[a], s_red[a][w]);
                hi[a] = fmaxf(hi[a], s_red[3 + a][w]);
            }
        }
        GridHdr h;
        const bool any = lo[0] <= hi[0];
        double ext[3];
        for (int a = 0; a < 3; ++a) ext[a] = any ? (double)hi[a] - (double)lo[a] : 0.0;
        double edge = (double)radius * 1.01;
        const double maxext = fmax(ext[0], fmax(ext[1], ext[2]));
        if (knn_k > 0) {
            // k-NN mode: pick the edge from the mean density so that the 3^d block around a query holds ~2.5 k points
            // (d = number of axes with a non-negligible extent: flat or linear clouds get fewer cells per block)
            int dims = 0;
            double vol = 1.0;
            for (int a = 0; a < 3; ++a)
                if (ext[a] > 1e-3 * maxext && ext[a] > 0.0) { ++dims; vol *= ext[a]; }
            int finite_pts = 0;
            (void)finite_pts;
            const double block = dims == 3 ? 27.0 : (dims == 2 ? 9.0 : 3.0);
            const double per_cell = fmax(2.5 * (double)knn_k / block, 1.0);
            edge = dims > 0 ? pow(vol * per_cell / (double)max(n, 1), 1.0 / (double)dims) : 1.0;
        }
        if (!(edge > 0.0) || !isfinite(edge)) edge = fmax(maxext, 1.0);      // degenerate: one cell per axis
        edge = fmax(edge, maxext * 1e-6);                                    // keep the quotient well inside int range
        double g[3];
        for (int it = 0; it < 64; ++it) {
            for (int a = 0; a < 3; ++a) g[a] = floor(ext[a] / edge) + 1.0;
            const double total = g[0] * g[1] * g[2];
            if (total <= (double)GRID_MAX_CELLS) break;
            edge *= cbrt(total / (double)GRID_MAX_CELLS) * 1.02;
        }
        h.minx = any ? lo[0] : 0.f; h.miny = any ? lo[1] : 0.f; h.minz = any ? lo[2] : 0.f;
        h.inv_h = (float)(1.0 / edge);
        h.gx = (int)g[0]; h.gy = (int)g[1]; h.gz = (int)g[2];
        h.npts = 0;
        h.dense = 0;
        s_hdr = h;
    }
    __syncthreads();
    const GridHdr h = s_hdr;
    const int ncell = h.gx * h.gy * h.gz;

    // 2. histogram (LDS atomics)
    for (int k = t; k < n; k += BUILD_THREADS) {
        const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            const int cx = min(max(cell_coord(x, h.minx, h.inv_h, h.gx), 0), h.gx - 1);
            const int cy = min(max(cell_coord(y, h.miny, h.inv_h, h.gy), 0), h.gy - 1);
            const int cz = min(max(cell_coord(z, h.minz, h.inv_h, h.gz), 0), h.gz - 1);
            atomicAdd(&s_cnt[cx + h.gx * (cy + h.gy * cz)], 1);
        }
    }
    __syncthreads();

    // 3. exclusive scan of s_cnt[0..ncell): each thread owns a contiguous chunk
    const int per = (ncell + BUILD_THREADS - 1) / BUILD_THREADS;
    const int c0 = min(t * per, ncell), c1 = min(c0 + per, ncell);
    int sum = 0;
    for (int c = c0; c < c1; ++c) sum += s_cnt[c];
    s_part[t] = sum;
    __syncthreads();
    for (int off = 1; off < BUILD_THREADS; off <<= 1) { // Hillis-Steele inclusive scan over the 1024 partials
        const int v = t >= off ? s_part[t - off] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = t > 0 ? s_part[t - 1] : 0;
    int *cs = cell_start + (size_t)b * stride_cells;
    for (int c = c0; c < c1; ++c) {
        const int cnt = s_cnt[c];
        s_cnt[c] = run; // becomes the scatter cursor
        cs[c] = run;
        run += cnt;
    }
    if (t == BUILD_THREADS - 1) {
        cs[ncell] = s_part[BUILD_THREADS - 1];
        s_tail = s_part[BUILD_THREADS - 1];
        GridHdr out = h;
        out.npts = s_part[BUILD_THREADS - 1];
        // mean number of candidates a centre would test (27 cells at the mean occupancy).  When that is a large share
        // of the cloud the cell lists buy nothing, and rows saturate early, which the index-ordered all-pairs scan
        // exploits (it stops after nsample hits) while a cell-ordered scan cannot.
        const double per_query = 27.0 * (double)out.npts / (double)ncell;
        out.dense = per_query > 0.25 * (double)n ? 1 : 0;
        hdrs[b] = out;
    }
    __syncthreads();

    // 4. scatter (order inside a cell is arbitrary; the query sorts its hits by index)
    int *sidx = sorted_idx + (size_t)b * n;
    float *sxyz = sorted_xyz + (size_t)b * n * 3;
    for (int k = t; k < n; k += BUILD_THREADS) {
        const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            const int cx = min(max(cell_coord(x, h.minx, h.inv_h, h.gx), 0), h.gx - 1);
            const int cy = min(max(cell_coord(y, h.miny, h.inv_h, h.gy), 0), h.gy - 1);
            const int cz = min(max(cell_coord(z, h.minz, h.inv_h, h.gz), 0), h.gz - 1);
            const int pos = atomicAdd(&s_cnt[cx + h.gx * (cy + h.gy * cz)], 1);
            sidx[pos] = k;
            sxyz[pos * 3] = x; sxyz[pos * 3 + 1] = y; sxyz[pos * 3 + 2] = z;
        } else { // not in any cell; listed after the cells so that a same-set query still emits its (empty) row
            const int pos = atomicAdd(&s_tail, 1);
            sidx[pos] = k;
            sxyz[pos * 3] = NAN; sxyz[pos * 3 + 1] = NAN; sxyz[pos * 3 + 2] = NAN;
        }
    }
}

constexpr int SUB = 8;               // lanes cooperating on one centre
constexpr int QPW = OGC_WAVE / SUB;  // centres per wavefront

// EIGHT lanes per centre.  With one lane per centre a 16 x 8192 batch is only 2048 wavefronts (two per SIMD) of
// long serial pointer-walks; eight lanes per centre give 16384 short wavefronts, so the chip hides the cache
// latency of the candidate reads by switching waves.  The centres are the grid's own points in cell order; the
// three x-adjacent cells of a (y, z) pair are ONE contiguous run of the cell-sorted arrays, so a centre has nine
// runs, and the eight lanes stride through each run together (consecutive candidates -> one cache line).
//   hits       : slot = cnt + (number of hitting lanes below me in my group), from one wave ballot — no atomics;
//   row full   : the group keeps the nsample SMALLEST point indices (replace the current maximum, re-scan it);
//   finish     : rank sort by the eight lanes (indices are distinct), pad with the smallest, 32-byte stores.
__global__ __launch_bounds__(OGC_WAVE) void ball_query_grid_kernel(int n, int m, float radius2, int nsample,
                                                                   int stride_cells,
                                                                   const GridHdr *__restrict__ hdrs,
                                                                   const int *__restrict__ cell_start,
                                                                   const int *__restrict__ sorted_idx,
                                                                   const float *__restrict__ sorted_xyz,
                                                                   int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) int gq_smem[];
    const int lane = threadIdx.x, b = blockIdx.y;
    const GridHdr h = hdrs[b];
    if (h.dense) return; // this cloud is handled by the all-pairs kernel
    const int sub = lane & (SUB - 1), qi = lane >> 3;
    int *kept = gq_smem + qi * nsample;                 // [QPW][nsample] indices kept so far (unordered)
    int *outr = gq_smem + (QPW + qi) * nsample;         // [QPW][nsample] sorted + padded row
    const int p = blockIdx.x * QPW + qi;
    const int *cs = cell_start + (size_t)b * stride_cells;
    const int *sidx = sorted_idx + (size_t)b * n;
    const float *sxyz = sorted_xyz + (size_t)b * n * 3;
    const unsigned below = (1u << sub) - 1u;

    int q = -1;
    float qx = NAN, qy = NAN, qz = NAN;
    if (p < n) { // positions >= h.npts hold the non-finite points (NaN coordinates -> no hit -> zero row)
        q = sidx[p];
        qx = sxyz[p * 3]; qy = sxyz[p * 3 + 1]; qz = sxyz[p * 3 + 2];
    }
    int cnt = 0, maxv = -1, maxpos = 0; // uniform within the group
    // (re)compute the largest kept index of a FULL row: each lane scans nsample/8 entries, then a 3-step butterfly
    auto rescan_max = [&]() {
        int mv = -1, mp = 0;
        for (int e = sub; e < nsample; e += SUB) {
            const int v = kept[e];
            if (v > mv) { mv = v; mp = e; }
        }
#pragma unroll
        for (int off = 1; off < SUB; off <<= 1) {
            const int ov = __shfl_xor(mv, off, 64), op = __shfl_xor(mp, off, 64);
            if (ov > mv) { mv = ov; mp = op; }
        }
        maxv = mv;
        maxpos = mp;
    };
    if (p < h.npts) {
        const int cx = cell_coord(qx, h.minx, h.inv_h, h.gx);
        const int cy = cell_coord(qy, h.miny, h.inv_h, h.gy);
        const int cz = cell_coord(qz, h.minz, h.inv_h, h.gz);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.gx - 1);
        int lo[9], hi[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) { // eighteen independent loads (the same addresses for the 8 lanes of a group)
            const int y = cy + (r % 3) - 1, z = cz + (r / 3) - 1;
            const bool ok = y >= 0 && y < h.gy && z >= 0 && z < h.gz && x0 <= x1;
            const int rowc = h.gx * (y + h.gy * z);
            lo[r] = ok ? cs[rowc + x0] : 0;
            hi[r] = ok ? cs[rowc + x1 + 1] : 0;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            for (int j = lo[r] + sub; __builtin_amdgcn_ballot_w64(j < hi[r]) != 0; j += SUB) {
                bool hit = false;
                int v = 0;
                if (j < hi[r]) {
                    hit = ogc_sqdist(qx, qy, qz, sxyz[j * 3], sxyz[j * 3 + 1], sxyz[j * 3 + 2]) < radius2;
                    if (hit) v = sidx[j];
                }
                const unsigned long long ball = __builtin_amdgcn_ballot_w64(hit);
                if (ball == 0) continue;
                const unsigned slice = (unsigned)(ball >> (qi * SUB)) & 0xFFu;
                if (slice == 0) continue;
                const int nh = __popc(slice);
                if (cnt + nh <= nsample) { // common case: room for all of the group's hits
                    if (hit) kept[cnt + __popc(slice & below)] = v;
                    cnt += nh;
                    if (cnt == nsample) rescan_max();
                } else { // row (nearly) full: take the hits one by one, keep the nsample smallest indices
                    for (int t = 0; t < SUB; ++t) {
                        if (!((slice >> t) & 1u)) continue;
                        const int vt = __shfl(v, qi * SUB + t, 64);
                        if (cnt < nsample) {
                            if (sub == 0) kept[cnt] = vt;
                            if (++cnt == nsample) rescan_max();
                        } else if (vt < maxv) {
                            if (sub == 0) kept[maxpos] = vt;
                            rescan_max();
                        }
                    }
                }
            }
        }
    }
    // rank sort (the kept indices are distinct): element e goes to position #{f : kept[f] < kept[e]}
    for (int e = sub; e < cnt; e += SUB) {
        const int ve = kept[e];
        int rank = 0;
        for (int f = 0; f < cnt; ++f) rank += kept[f] < ve ? 1 : 0;
        outr[rank] = ve;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (q >= 0) {
        const int first = cnt > 0 ? outr[0] : 0;
        int *o = idx_out + ((size_t)b * m + q) * nsample;
        for (int j = sub; j < nsample; j += SUB) o[j] = j < cnt ? outr[j] : first;
    }
}

typedef unsigned long long u64;

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int off) {
    const unsigned lo = __shfl_xor((unsigned)v, off, 64), hi = __shfl_xor((unsigned)(v >> 32), off, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    const unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
    return ((u64)hi << 32) | lo;
}

// Exact k nearest neighbours over the cell lists: "the k smallest (distance, index) keys", which is what the
// reference's stable insertion computes (interpolate_gpu.cu:36-52).  EIGHT lanes per query, as in the ball query.
// The query's block of (2R+1)^3 cells is scanned shell by shell (R = 1, 2, ...): every point closer than R*h lies
// inside the block (the query is projected into the box first; projection is contractive per axis), so the search
// stops as soon as k keys are held and the k-th distance is below (R*h)^2 (with a 0.1 % guard for the fp32 cell
// quotient) — or the block covers the whole grid.  The kept set is an unordered LDS array with its maximum tracked;
// a candidate is admitted iff its key is below that maximum (strict '<' on (distance, index)), exactly the
// reference's rule whatever the order in which candidates are met.
// MODE 0: squared distances (ogc_knn).  MODE 1: sqrt + radius clamp of the indices (ogc_knn_clamped).
template <int MODE>
__global__ __launch_bounds__(OGC_WAVE) void knn_grid_kernel(int n, int m, int k, float radius, int stride_cells,
                                                            const float *__restrict__ unknown,
                                                            const GridHdr *__restrict__ hdrs,
                                                            const int *__restrict__ cell_start,
                                                            const int *__restrict__ sorted_idx,
                                                            const float *__restrict__ sorted_xyz,
                                                            float *__restrict__ dist_out, int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) u64 kq_smem[];
    const int lane = threadIdx.x, b = blockIdx.y;
    const int sub = lane & (SUB - 1), qi = lane >> 3;
    u64 *kept = kq_smem + (size_t)qi * k;           // [QPW][k]
    u64 *outk = kq_smem + (size_t)(QPW + qi) * k;   // [QPW][k]
    const int p = blockIdx.x * QPW + qi;
    const GridHdr h = hdrs[b];
    const int *cs = cell_start + (size_t)b * stride_cells;
    const int *sidx = sorted_idx + (size_t)b * m;
    const float *sxyz = sorted_xyz + (size_t)b * m * 3;
    const unsigned below = (1u << sub) - 1u;

    float qx = NAN, qy = NAN, qz = NAN;
    if (p < n) {
        const float *u = unknown + ((size_t)b * n + p) * 3;
        qx = u[0]; qy = u[1]; qz = u[2];
    }
    int cnt = 0, maxpos = 0;
    u64 maxkey = 0;
    auto rescan_max = [&]() {
        u64 mk = 0;
        int mp = 0;
        for (int e = sub; e < k; e += SUB) {
            const u64 v = kept[e];
            if (v >= mk) { mk = v; mp = e; }
        }
#pragma unroll
        for (int off = 1; off < SUB; off <<= 1) {
            const u64 ov = shfl_xor_u64(mk, off);
            const int op = __shfl_xor(mp, off, 64);
            if (ov > mk) { mk = ov; mp = op; }
        }
        maxkey = mk;
        maxpos = mp;
    };
    // scan the run [j0, j1) of the cell-sorted arrays with the 8 lanes of the group
    auto scan_run = [&](int j0, int j1) {
        for (int j = j0 + sub; __builtin_amdgcn_ballot_w64(j < j1) != 0; j += SUB) {
            bool adm = false;
            u64 key = 0;
            if (j < j1) {
                const float d = ogc_sqdist(qx, qy, qz, sxyz[j * 3], sxyz[j * 3 + 1], sxyz[j * 3 + 2]);
                if (d < INFINITY) { // NaN / inf are never selected
                    key = ((u64)__float_as_uint(d) << 32) | (unsigned)sidx[j];
                    adm = cnt < k || key < maxkey;
                }
            }
            const u64 ball = __builtin_amdgcn_ballot_w64(adm);
            if (ball == 0) continue;
            const unsigned slice = (unsigned)(ball >> (qi * SUB)) & 0xFFu;
            if (slice == 0) continue;
            const int nh = __popc(slice);
            if (cnt + nh <= k) {
                if (adm) kept[cnt + __popc(slice & below)] = key;
                cnt += nh;
                if (cnt == k) rescan_max();
            } else {
                for (int t = 0; t < SUB; ++t) {
                    if (!((slice >> t) & 1u)) continue;
                    const u64 kt = shfl_u64(key, qi * SUB + t);
                    if (cnt < k) {
                        if (sub == 0) kept[cnt] = kt;
                        if (++cnt == k) rescan_max();
                    } else if (kt < maxkey) {
                        if (sub == 0) kept[maxpos] = kt;
                        rescan_max();
                    }
                }
            }
        }
    };

    const bool active = p < n && h.npts > 0 && qx == qx && qy == qy && qz == qz; // NaN queries select nothing
    if (active) {
        const float edge = 1.0f / h.inv_h;
        const int cx = min(max(cell_coord(qx, h.minx, h.inv_h, h.gx), 0), h.gx - 1);
        const int cy = min(max(cell_coord(qy, h.miny, h.inv_h, h.gy), 0), h.gy - 1);
        const int cz = min(max(cell_coord(qz, h.minz, h.inv_h, h.gz), 0), h.gz - 1);
        const int rmax = max(max(max(cx, h.gx - 1 - cx), max(cy, h.gy - 1 - cy)), max(cz, h.gz - 1 - cz));
        for (int R = 1;; ++R) {
            const int xa = max(cx - R, 0), xb = min(cx + R, h.gx - 1);
            for (int z = max(cz - R, 0); z <= min(cz + R, h.gz - 1); ++z)
                for (int y = max(cy - R, 0); y <= min(cy + R, h.gy - 1); ++y) {
                    const int rowc = h.gx * (y + h.gy * z);
                    const bool face = R == 1 || z == cz - R || z == cz + R || y == cy - R || y == cy + R;
                    if (face) { // the whole x-extent of this row belongs to shell R (for R = 1: the full 3^3 block)
                        scan_run(cs[rowc + xa], cs[rowc + xb + 1]);
                    } else {    // inner row: only the two end cells are new
                        if (cx - R >= 0) scan_run(cs[rowc + cx - R], cs[rowc + cx - R + 1]);
                        if (cx + R <= h.gx - 1) scan_run(cs[rowc + cx + R], cs[rowc + cx + R + 1]);
                    }
                }
            if (R >= rmax) break; // the block covers the grid
            if (cnt == k) {
                const float cover = (float)R * edge * 0.999f;
                if (__uint_as_float((unsigned)(maxkey >> 32)) < cover * cover) break;
            }
        }
    }
    // rank sort (keys are distinct: the index is part of the key)
    for (int e = sub; e < cnt; e += SUB) {
        const u64 ve = kept[e];
        int rank = 0;
        for (int f = 0; f < cnt; ++f) rank += kept[f] < ve ? 1 : 0;
        outk[rank] = ve;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (p < n) {
        const size_t base = ((size_t)b * n + p) * k;
        const int first = cnt > 0 ? (int)(unsigned)outk[0] : 0;
        for (int j = sub; j < k; j += SUB) {
            float d = INFINITY;
            int id = 0;
            if (j < cnt) {
                const u64 key = outk[j];
                d = __uint_as_float((unsigned)(key >> 32));
                id = (int)(unsigned)key;
            }
            if (MODE == 1) {
                d = sqrtf(d);
                if (d > radius && radius >= 0.0f) id = first;
            }
            dist_out[base + j] = d;
            idx_out[base + j] = id;
        }
    }
}

} // namespace ogc_grid

using namespace ogc_grid;

int ogc_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                        int *idx, hipStream_t s, const GridHdr **dense_hdrs, void **workspace) {
    const size_t lds = (size_t)2 * QPW * nsample * sizeof(int);
    // the cell-ordered traversal needs the centres to BE the points (ball_query(pc, pc), the reference's only live
    // use: losses/seg_loss_unsup.py:151, losses/flow_loss_unsup.py:84); other centre sets use the all-pairs scan
    const bool same = (new_xyz == xyz) && (m == n);
    if (!same || n < 1024 || lds > 64 * 1024 || !(radius > 0.0f) || !(radius < 3.0e38f)) return OGC_ERR_UNSUPPORTED;
    const int stride_cells = GRID_MAX_CELLS + 1;
    const size_t bytes_hdr = (sizeof(GridHdr) * b + 255) / 256 * 256;
    const size_t bytes_cs = (sizeof(int) * (size_t)b * stride_cells + 255) / 256 * 256;
    const size_t bytes_idx = (sizeof(int) * (size_t)b * n + 255) / 256 * 256;
    const size_t bytes_xyz = sizeof(float) * (size_t)b * n * 3;
    char *ws = nullptr;
    if (hipMallocAsync((void **)&ws, bytes_hdr + bytes_cs + bytes_idx + bytes_xyz, s) != hipSuccess || !ws) {
        (void)hipGetLastError();
        return OGC_ERR_UNSUPPORTED;
    }
    GridHdr *hdrs = reinterpret_cast<GridHdr *>(ws);
    int *cell_start = reinterpret_cast<int *>(ws + bytes_hdr);
    int *sorted_idx = reinterpret_cast<int *>(ws + bytes_hdr + bytes_cs);
    float *sorted_xyz = reinterpret_cast<float *>(ws + bytes_hdr + bytes_cs + bytes_idx);
    hipLaunchKernelGGL(grid_build_kernel, dim3(b), dim3(BUILD_THREADS), 0, s, n, radius, 0, stride_cells, xyz, hdrs,
                       cell_start, sorted_idx, sorted_xyz);
    hipLaunchKernelGGL(ball_query_grid_kernel, dim3(ogc_divup(n, QPW), b), dim3(OGC_WAVE), lds, s, n, m,
                       radius * radius, nsample, stride_cells, hdrs, cell_start, sorted_idx, sorted_xyz, idx);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFreeAsync(ws, s);
        ogc_set_error("ogc_ball_query (grid): launch failed: %s", hipGetErrorString(e));
        return OGC_ERR_LAUNCH;
    }
    *dense_hdrs = hdrs;
    *workspace = ws;
    return OGC_OK;
}

void ogc_ball_query_grid_release(void *workspace, hipStream_t s) { (void)hipFreeAsync(workspace, s); }

// k-NN over cell lists.  Returns OGC_OK after queueing build + query, or OGC_ERR_UNSUPPORTED (caller: all-pairs scan).
int ogc_knn_grid(int mode, int b, int n, int m, int k, float radius, const float *unknown, const float *known,
                 float *dist, int *idx, hipStream_t s) {
    const size_t lds = (size_t)2 * QPW * k * sizeof(u64);
    if (m < 1024 || m <= 4 * k || lds > 64 * 1024) return OGC_ERR_UNSUPPORTED;
    const int stride_cells = GRID_MAX_CELLS + 1;
    const size_t bytes_hdr = (sizeof(GridHdr) * b + 255) / 256 * 256;
    const size_t bytes_cs = (sizeof(int) * (size_t)b * stride_cells + 255) / 256 * 256;
    const size_t bytes_idx = (sizeof(int) * (size_t)b * m + 255) / 256 * 256;
    const size_t bytes_xyz = sizeof(float) * (size_t)b * m * 3;
    char *ws = nullptr;
    if (hipMallocAsync((void **)&ws, bytes_hdr + bytes_cs + bytes_idx + bytes_xyz, s) != hipSuccess || !ws) {
        (void)hipGetLastError();
        return OGC_ERR_UNSUPPORTED;
    }
    GridHdr *hdrs = reinterpret_cast<GridHdr *>(ws);
    int *cell_start = reinterpret_cast<int *>(ws + bytes_hdr);
    int *sorted_idx = reinterpret_cast<int *>(ws + bytes_hdr + bytes_cs);
    float *sorted_xyz = reinterpret_cast<float *>(ws + bytes_hdr + bytes_cs + bytes_idx);
    hipLaunchKernelGGL(grid_build_kernel, dim3(b), dim3(BUILD_THREADS), 0, s, m, 0.0f, k, stride_cells, known, hdrs,
                       cell_start, sorted_idx, sorted_xyz);
    dim3 grid(ogc_divup(n, QPW), b);
    if (mode == 1)
        hipLaunchKernelGGL(knn_grid_kernel<1>, grid, dim3(OGC_WAVE), lds, s, n, m, k, radius, stride_cells, unknown, hdrs,
                           cell_start, sorted_idx, sorted_xyz, dist, idx);
    else
        hipLaunchKernelGGL(knn_grid_kernel<0>, grid, dim3(OGC_WAVE), lds, s, n, m, k, radius, stride_cells, unknown, hdrs,
                           cell_start, sorted_idx, sorted_xyz, dist, idx);
    const hipError_t e = hipGetLastError();
    (void)hipFreeAsync(ws, s);
    if (e != hipSuccess) {
        ogc_set_error("ogc_knn (grid): launch failed: %s", hipGetErrorString(e));
        return OGC_ERR_LAUNCH;
    }
    return OGC_OK;
}
