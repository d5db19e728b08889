// grid.hip — uniform-grid (cell-list) acceleration of the fixed-radius query and of k-NN, with results identical to
// the brute-force scans.
//
// The reference's ball query (ball_query_gpu.cu:9-45) tests every centre against every point: 8*N*M flop for
// ~2.3 MB of input/output per 8192-point cloud, i.e. compute-bound by two orders of magnitude.  A radius query only
// needs the points of the 27 cells around the centre when the cell edge is >= the radius.  Per call:
//   grid_build_kernel   one workgroup per cloud, the cloud held in registers: bounding box -> cell edge h >= 1.01 r
//                       (enlarged until the grid has <= GRID_MAX_CELLS cells; density-based for k-NN) -> LDS
//                       histogram -> scan -> scatter: cell_start[], and the points re-ordered by cell as 16-byte
//                       records (x, y, z, index), so a query reads a candidate with one load from a contiguous run;
//   ball_query_grid     one LANE PER CANDIDATE: a wavefront takes eight centres consecutive in cell order, deals the
//                       candidates of the union of their neighbourhoods (nine contiguous runs) to its 64 lanes and
//                       tests every centre against all lanes at once (centre coordinates as scalars, two centres per
//                       packed instruction); ballots turn hits into list slots; eight lanes per centre rank-sort the
//                       list by point index -> first nsample, padded with the first: the row the reference produces
//                       by scanning in index order and stopping after nsample hits;
//   knn_grid            eight lanes per query scan the cells shell by shell until the k-th distance is covered.
// Exactness: a hit satisfies |dx| < r in every axis, the cell coordinate is floor((x - min) / h) with h >= 1.01 r, so
// the cell coordinates of a centre and any of its hits differ by at most one even with fp32 rounding of the
// quotient (relative error 1e-7 * up to 16384 cells << 0.01); points with non-finite coordinates can never be
// hits (their distance is inf/NaN) and are left out of the grid.
#include <stdlib.h>

#include "ogc_common.h"
#include "grid.h"

namespace ogc_grid {

constexpr int GRID_MAX_CELLS = 16384;
constexpr int BUILD_THREADS = 1024;

// Development probe (tools/bq_probe.hip compiles this file with OGC_GRID_PROBE): cycle stamps of the build's phases
// (workgroup 0) and per-phase cycle sums over all wavefronts of the query.
#ifdef OGC_GRID_PROBE
__device__ unsigned long long ogc_grid_probe[64];
#define OGC_PROBE_BUILD(i) \
    if (blockIdx.x == 0 && threadIdx.x == 0) ogc_grid_probe[i] = __builtin_amdgcn_s_memtime()
#define OGC_PROBE_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define OGC_PROBE_ADD(i, a, b) \
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicAdd(&ogc_grid_probe[i], (b) - (a))
#else
#define OGC_PROBE_BUILD(i)
#define OGC_PROBE_T(var)
#define OGC_PROBE_ADD(i, a, b)
#endif

__device__ __forceinline__ int cell_coord(float x, float mn, float inv_h, int g) {
    // floor((x - mn) * inv_h) clamped to [-2, g + 1]; NaN -> -2 (outside every neighbourhood)
    const float f = floorf((x - mn) * inv_h);
    if (!(f >= -2.0f)) return -2;
    if (f > (float)(g + 1)) return g + 1;
    return (int)f;
}

// max(floor((x - mn) * inv_h), 0) as an int, for a FINITE x: the cell coordinate before the clamp to the grid's upper edge
// (the median keeps the conversion in range; a NaN — a centre that is no point of the grid — gives 0)
__device__ __forceinline__ int cell_floor(float x, float mn, float inv_h) {
    return (int)__builtin_amdgcn_fmed3f(floorf((x - mn) * inv_h), 0.0f, 1.0e9f);
}

// min(max(floor((x - mn) * inv_h), 0), g - 1) in four instructions: subtract, multiply (the same two roundings as cell_floor),
// convert with floor rounding (saturating; NaN -> 0) and an integer median.  Equal to min(cell_floor(x, mn, inv_h), g - 1).
__device__ __forceinline__ int cell_clamped(float x, float mn, float inv_h, int g) {
    const float q = (x - mn) * inv_h;
    int c, r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(c) : "v"(q));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(c), "v"(g - 1));
    return r;
}

// a point's coordinates along the grid's (fast, mid, slow) axes (GridHdr::fast; wave-uniform selects)
#define OGC_GRID_AXES(H, X, Y, Z, FX, FY, FZ)                                                    \
    const float FX = (H).fast == 0 ? (X) : ((H).fast == 1 ? (Y) : (Z)), FY = (H).fast == 0 ? (Y) : (X), \
                FZ = (H).fast == 2 ? (Y) : (Z)

// v^(1/dims) for the cell edge
__device__ __forceinline__ float dims_root(float v, int dims) { return dims == 1 ? v : (dims == 2 ? sqrtf(v) : cbrtf(v)); }

// Grid parameters from the bounding box, by ONE lane per workgroup, in single precision: the edge only steers which candidates
// a search meets.  What the searches rely on holds for ANY origin, edge and cell counts: cell = min(floor((x - min) / h), g - 1)
// clamped at 0 is monotone in x, so two points closer than r <= h / 1.01 along an axis are at most one cell apart (a 1 % margin
// against the 1e-7 relative error of the fp32 quotient and of r * 1.01f itself).  (This used to be double-precision pow / cbrt /
// division chains: ~2200 cycles on the one lane everybody waits for; now ~a quarter.)
__device__ __forceinline__ GridHdr grid_header(const float (&lo)[3], const float (&hi)[3], int n, float radius,
                                               int knn_k, float knn_div, int prefer_cells) {
    GridHdr h;
    bool cells_ok = false;
    const bool any = lo[0] <= hi[0];
    float ext[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) ext[a] = any ? fminf(hi[a] - lo[a], 3.0e38f) : 0.0f; // (a difference of finite numbers may overflow)
    float edge = radius * 1.01f;
    const float maxext = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
    if (knn_k > 0) {
        // k-NN mode: pick the edge from the mean density so that the 3^d block around a query holds ~2.5 k points
        // (d = number of axes with a non-negligible extent: flat or linear clouds get fewer cells per block)
        int dims = 0;
        float vol = 1.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (ext[a] > 1e-3f * maxext && ext[a] > 0.0f) { ++dims; vol *= ext[a]; }
        // cell edge = HALF the expected distance of the k-th neighbour (rho h^d = k / (V_d 2^d), V_d the unit ball): the
        // search then ends after the 5^d block (R = 2), ~3.7 k candidates in 3-D, where an edge of 0.73 r_k (the former
        // 2.5 k points per 3^d block) also needed R = 2 but scanned 11.6 k
        const float inv_n = 1.0f / (float)max(n, 1);
        const float per_cell = fmaxf((float)knn_k / (dims == 3 ? knn_div : (dims == 2 ? 12.6f : 4.0f)), 0.5f);
        edge = dims > 0 ? dims_root(vol * per_cell * inv_n, dims) : 1.0f;
        if (knn_div < 0.0f && dims > 0) {
            // knn_wave_kernel: the edge that puts ~(-knn_div) points into the query's block of five cells per axis CLIPPED to the
            // bounding box — a road scene 4 m high is two or three cells thick whatever the edge, so the block is a slab and its
            // cells may be much longer than the mean density of the box suggests (which counts a ball that the slab cuts off)
            const float target = -knn_div;
            const float rho = (float)max(n, 1) / vol;
            edge = dims_root(target / (rho * (dims == 3 ? 125.0f : (dims == 2 ? 25.0f : 5.0f))), dims);
            for (int it = 0; it < 3; ++it) {
                float fixed = 1.0f;
                int freed = 0;
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    if (ext[a] > 1e-3f * maxext && ext[a] > 0.0f) {
                        if (5.0f * edge >= ext[a]) fixed *= ext[a]; else ++freed;
                    }
                if (freed == 0) break;
                edge = dims_root(target / (rho * fixed * (freed == 3 ? 125.0f : (freed == 2 ? 25.0f : 5.0f))), freed);
            }
        }
        // radius-clamped search (ogc_knn_clamped): neighbours beyond `radius` are replaced by the nearest one anyway,
        // so the search may stop once the scanned block covers the radius.  When the radius is SHORTER than the
        // density-based edge, cells of edge 1.01 r make that one shell of far fewer candidates (but never less than
        // ~one point per cell: the nearest neighbour of a query in an empty region must still be found by shells).
        if (radius > 0.0f && radius < 3.0e38f && dims > 0) {
            const float one_per_cell = dims_root(vol * inv_n, dims);
            const float limited = fmaxf(radius * 1.01f, one_per_cell);
            if (limited < edge) edge = limited;
            // A radius-limited search of the cloud in itself (prefer_cells): when a ball holds few points (mean <= 18 at the
            // mean density) knn_cells_kernel finds them all in the 27 cells of edge 1.01 r around a query and sorts them in
            // registers — the ball query's grid, far fewer candidates than the shells of the density-based one.
            if (prefer_cells && radius < 1.0e12f) {
                const float r = radius;
                const float ball = dims == 3 ? 4.18879f * r * r * r : (dims == 2 ? 3.14159f * r * r : 2.0f * r);
                if ((float)n * ball <= 18.0f * vol) {
                    edge = fmaxf(edge, r * 1.01f); // (no finer than the density asks for: the build's cost grows with the cell count)
                    cells_ok = true;
                }
            }
        }
    }
    if (!(edge > 0.0f) || !isfinite(edge)) edge = fmaxf(maxext, 1.0f);    // degenerate: one cell per axis
    edge = fmaxf(edge, maxext * 1e-6f);                                   // keep the quotient well inside int range
    float g0 = 1.0f, g1 = 1.0f, g2 = 1.0f;
    for (int it = 0; it < 64; ++it) {
        const float inv = 1.0f / edge;
        g0 = floorf(ext[0] * inv) + 1.0f; g1 = floorf(ext[1] * inv) + 1.0f; g2 = floorf(ext[2] * inv) + 1.0f;
        const float total = g0 * g1 * g2;
        if (total <= (float)GRID_MAX_CELLS) break;
        edge *= cbrtf(total * (1.0f / (float)GRID_MAX_CELLS)) * 1.02f;
    }
    // Cell order.  A radius search visits the cells (x - 1 .. x + 1, y - 1 .. y + 1, z - 1 .. z + 1): nine runs of the
    // cell-sorted array when x runs fastest.  When one axis has at most two cells (a road scene a few metres high searched with
    // r = 2 m) and THAT axis runs fastest, the cells (all of it, y - 1 .. y + 1) of one z are contiguous: three runs, three
    // times as long — the query kernels' cost is per run, not per candidate.  Otherwise x stays the fastest axis.
    int fast = 0;
    if (knn_k == 0 || prefer_cells != 0) {
        if (g0 > 2.0f && g1 <= 2.0f && g1 <= g2) fast = 1;
        else if (g0 > 2.0f && g2 <= 2.0f) fast = 2;
    }
    // (fast, mid, slow) = (x, y, z) | (y, x, z) | (z, x, y)
    const float l0 = any ? lo[0] : 0.f, l1 = any ? lo[1] : 0.f, l2 = any ? lo[2] : 0.f;
    h.minx = fast == 0 ? l0 : (fast == 1 ? l1 : l2);
    h.miny = fast == 0 ? l1 : l0;
    h.minz = fast == 2 ? l1 : l2;
    h.inv_h = 1.0f / edge;
    h.gx = (int)(fast == 0 ? g0 : (fast == 1 ? g1 : g2));
    h.gy = (int)(fast == 0 ? g1 : g0);
    h.gz = (int)(fast == 2 ? g1 : g2);
    h.fast = fast;
    h.slab = h.gx <= 2 ? 1 : 0;
    h.npts = 0;
    h.dense = 0;
    h.heavy = 0;
    h.knn_general = (cells_ok || prefer_cells == 2) ? 0 : 1;   // (2: a grid shared by several searches, built for their largest radius)
    h.pending = 0;
    return h;
}

// One workgroup per cloud.  PPT > 0: every thread keeps its PPT points (and their cells) in registers, so the cloud
// is read from memory once; PPT == 0: any size, the three passes re-read the cloud.  Five barriers in all: the
// bounding box and the cell-count scan are wave-level (DPP / shuffles) with one 16-entry exchange through LDS each.
template <int PPT>
__global__ __launch_bounds__(BUILD_THREADS) void grid_build_kernel(float knn_div, int n, float radius, int knn_k, int prefer_cells,
                                                                   int stride_cells,
                                                                   const float *__restrict__ xyz,
                                                                   GridHdr *__restrict__ hdrs,
                                                                   int *__restrict__ cell_start,
                                                                   float4 *__restrict__ sorted_pts) {
    __shared__ int s_cnt[GRID_MAX_CELLS]; // histogram -> exclusive starts -> scatter cursors
    __shared__ float s_red[6][BUILD_THREADS / 64];
    __shared__ int s_wave[BUILD_THREADS / 64];
    __shared__ int s_tail; // cursor for points left out of the grid (non-finite coordinates)
    __shared__ GridHdr s_hdr;
    constexpr int R = PPT > 0 ? PPT : 1;
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;
    const float *pts = xyz + (size_t)b * n * 3;
    float px[R], py[R], pz[R];
    int cell[R];

    OGC_PROBE_BUILD(0);
    // 1. load + bounding box of the finite points
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto widen = [&](float x, float y, float z) {
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
            mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
        }
    };
    // (PPT > 0: thread t owns the points t * PPT .. t * PPT + PPT - 1 — 3 PPT consecutive floats, read as 16-byte
    // loads when the cloud's size and base allow it: 6 loads instead of 24 strided ones for PPT = 8)
    if (PPT > 0) {
        const bool wide = PPT % 4 == 0 && (n & 3) == 0 && (((size_t)(const void *)pts) & 15) == 0;
        if (wide && (t + 1) * R <= n) {
            float f[3 * R];
            const float4 *p4 = reinterpret_cast<const float4 *>(pts + (size_t)t * R * 3);
#pragma unroll
            for (int i = 0; i < 3 * R / 4; ++i) {
                const float4 v = p4[i];
                f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) { px[i] = f[3 * i]; py[i] = f[3 * i + 1]; pz[i] = f[3 * i + 2]; }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int k = t * R + i;
                px[i] = py[i] = pz[i] = NAN;
                if (k < n) { px[i] = pts[k * 3]; py[i] = pts[k * 3 + 1]; pz[i] = pts[k * 3 + 2]; }
            }
        }
    }
    for (int c = t; c < GRID_MAX_CELLS; c += BUILD_THREADS) s_cnt[c] = 0; // overlaps the loads
    if (PPT > 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) widen(px[i], py[i], pz[i]);
    } else {
        for (int k = t; k < n; k += BUILD_THREADS) widen(pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = -ogc_wave_max_f32(-mn[a]), hi = ogc_wave_max_f32(mx[a]);
        if (lane == 0) { s_red[a][wave] = lo; s_red[3 + a][wave] = hi; }
    }
    OGC_PROBE_BUILD(1);
    __syncthreads();
    OGC_PROBE_BUILD(2);
    // the grid parameters are derived ONCE (double-precision pow / cbrt / floor loops: hundreds of instructions that
    // used to run on all 1024 threads of the one CU a cloud gets) and handed to the others through LDS
    if (wave == 0) {
        float lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = lane < BUILD_THREADS / 64 ? s_red[a][lane] : INFINITY;
            const float u = lane < BUILD_THREADS / 64 ? s_red[3 + a][lane] : -INFINITY;
            lo[a] = -ogc_wave_max_f32(-l);
            hi[a] = ogc_wave_max_f32(u);
        }
        if (lane == 0) s_hdr = grid_header(lo, hi, n, radius, knn_k, knn_div, prefer_cells);
    }
    __syncthreads();
    OGC_PROBE_BUILD(3);
    GridHdr h = s_hdr;
    const int ncell = h.gx * h.gy * h.gz;
    // (a finite coordinate: the float -> int conversion saturates where cell_coord clamps to [-2, g + 1], and the clamp to
    // the grid follows either way — the same cell, without cell_coord's two branches per axis)
    auto cell_of = [&](float x, float y, float z) -> int {
        OGC_GRID_AXES(h, x, y, z, fx, fy, fz);
        const int cx = min(cell_floor(fx, h.minx, h.inv_h), h.gx - 1);
        const int cy = min(cell_floor(fy, h.miny, h.inv_h), h.gy - 1);
        const int cz = min(cell_floor(fz, h.minz, h.inv_h), h.gz - 1);
        return (isfinite(x) && isfinite(y) && isfinite(z)) ? cx + h.gx * (cy + h.gy * cz) : -1;
    };

    // 2. histogram (LDS atomics)
    if (PPT > 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            cell[i] = (t * R + i < n) ? cell_of(px[i], py[i], pz[i]) : -2;
            if (cell[i] >= 0) atomicAdd(&s_cnt[cell[i]], 1);
        }
    } else {
        for (int k = t; k < n; k += BUILD_THREADS) {
            const int c = cell_of(pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2]);
            if (c >= 0) atomicAdd(&s_cnt[c], 1);
        }
    }
    __syncthreads();
    OGC_PROBE_BUILD(4);

    // 3. exclusive scan of s_cnt[0..ncell): a contiguous chunk per thread, wave scan of the chunk sums, wave totals
    const int per = (ncell + BUILD_THREADS - 1) / BUILD_THREADS;
    const int c0 = min(t * per, ncell), c1 = min(c0 + per, ncell);
    int sum = 0;
    for (int c = c0; c < c1; ++c) sum += s_cnt[c];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    OGC_PROBE_BUILD(5);
    int before = 0, npts = 0;
#pragma unroll
    for (int w = 0; w < BUILD_THREADS / 64; ++w) {
        const int v = s_wave[w];
        if (w < wave) before += v;
        npts += v;
    }
    int run = before + incl - sum;
    int *cs = cell_start + (size_t)b * stride_cells;
    for (int c = c0; c < c1; ++c) {
        const int cnt = s_cnt[c];
        s_cnt[c] = run; // becomes the scatter cursor
        cs[c] = run;
        run += cnt;
    }
    if (t == 0) {
        cs[ncell] = npts;
        s_tail = npts;
        h.npts = npts;
        // mean number of candidates a centre would test (27 cells at the mean occupancy).  When that is a large share
        // of the cloud the cell lists buy nothing, and rows saturate early, which an index-ordered scan exploits (it
        // stops after nsample hits) while a cell-ordered scan cannot.
        const float per_query = 27.0f * (float)npts / (float)ncell;
        h.dense = per_query > 0.25f * (float)n ? 1 : 0;
        // ball_query_cells_kernel sorts lists of up to 32 hits: with full cells around it a centre has ~0.15 * per_query hits
        // (ball / 27 cells), so beyond a mean of ~18 too many wavefronts would have to repeat their work in the general body
        h.heavy = per_query > 120.0f ? 1 : 0;
        // knn_cells_kernel (radius-limited search over the 27 cells around a query) needs cells at least as long as the radius —
        // the same bound knn_grid_kernel stops its shells with — and lists that fit its register sort
        if (!(1.0f * (1.0f / h.inv_h) * 0.999f >= radius) || per_query > 120.0f) h.knn_general = 1;
        hdrs[b] = h;
    }
    __syncthreads();
    OGC_PROBE_BUILD(6);

    // 4. scatter (order inside a cell is arbitrary; the queries order their results themselves).  One 16-byte record
    //    per point: x, y, z and the point's index (bit pattern), so a query reads a candidate with a single load.
    //    Points with non-finite coordinates are in no cell; they are listed after the cells so that a same-set query
    //    still emits their (empty) rows.
    float4 *sp = sorted_pts + (size_t)b * n;
    if (PPT > 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int k = t * R + i;
            if (cell[i] >= 0) sp[atomicAdd(&s_cnt[cell[i]], 1)] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
            else if (cell[i] == -1) sp[atomicAdd(&s_tail, 1)] = make_float4(NAN, NAN, NAN, __int_as_float(k));
        }
    } else {
        for (int k = t; k < n; k += BUILD_THREADS) {
            const float x = pts[k * 3], y = pts[k * 3 + 1], z = pts[k * 3 + 2];
            const int c = cell_of(x, y, z);
            if (c >= 0) sp[atomicAdd(&s_cnt[c], 1)] = make_float4(x, y, z, __int_as_float(k));
            else sp[atomicAdd(&s_tail, 1)] = make_float4(NAN, NAN, NAN, __int_as_float(k));
        }
    }
    OGC_PROBE_BUILD(7);
}

// The same build by SEVERAL workgroups per cloud, with nothing exchanged between them.  One workgroup per cloud is one CU per
// cloud: 16 of 256 CUs for a batch of 16, each pushing 8192 16-byte records and the cell starts through its own store path
// (the launch ends when those drain), its histogram and scatter serialised on one LDS.  Here every workgroup of a cloud reads
// the WHOLE cloud (L2-resident after the first reader), derives the same bounding box and header, and computes every point's
// cell — but owns only a contiguous range of cells [c_lo, c_hi): it counts the points below its range (its base offset), builds
// the histogram / scan / cursors of its own range in LDS and scatters only the points that fall into it.  No grid barrier, no
// flag: redundant arithmetic instead of communication.  Results: the same cell starts; the order of the points inside a cell
// is arbitrary in both kernels (LDS atomics), which no query depends on.
constexpr int SPLIT_MIN = 8;                                // parts per cloud (at least): a part owns <= GRID_MAX_CELLS / 8 cells
constexpr int SPLIT_CELLS = GRID_MAX_CELLS / SPLIT_MIN;

template <int PPT>
__global__ __launch_bounds__(BUILD_THREADS) void grid_build_split_kernel(float knn_div, int nb, int split, int n, float radius,
                                                                         int knn_k, int prefer_cells, int stride_cells,
                                                                         const float *__restrict__ xyz,
                                                                         GridHdr *__restrict__ hdrs,
                                                                         int *__restrict__ cell_start,
                                                                         float4 *__restrict__ sorted_pts) {
    __shared__ int s_cnt[SPLIT_CELLS]; // histogram -> scatter cursors of the cells [c_lo, c_hi)
    __shared__ float4 s_stage[BUILD_THREADS / 64 * 64 * 6]; // 6 KiB per wavefront: the transposition of the coalesced loads
    __shared__ float s_red[6][BUILD_THREADS / 64];
    __shared__ int s_wave[BUILD_THREADS / 64], s_counts[BUILD_THREADS / 64];
    __shared__ int s_tail;
    __shared__ GridHdr s_hdr;
    static_assert(PPT > 0 && PPT % 8 == 0, "the split build keeps the cloud in registers, eight points per thread and pass");
    const int t = threadIdx.x, b = blockIdx.x % nb, part = blockIdx.x / nb, lane = t & 63, wave = t >> 6;
    const float *pts = xyz + (size_t)b * n * 3;
    float px[PPT], py[PPT], pz[PPT];
    int cell[PPT];
    // thread t owns the points 8192 * pass + 8 t + i (i < 8) of pass = 0 .. PPT / 8 - 1
    auto point_index = [&](int i) { return (i >> 3) * (8 * BUILD_THREADS) + t * 8 + (i & 7); };

    OGC_PROBE_BUILD(0);
    // 1. load + bounding box of the finite points.  A wavefront's 512 points of a pass are 6 KiB of contiguous memory: it
    //    reads them as six fully coalesced 16-byte loads per lane (lane L takes the 16-byte pieces L, L + 64, ...), parks
    //    them in its own LDS strip and reads back the 96 contiguous bytes of ITS eight points.  (Reading those 96 bytes
    //    straight from memory — lanes 96 bytes apart, 48 cache lines per load instruction, every line visited by six
    //    instructions — took 8.7 k cycles for the cloud, i.e. most of the kernel.)
    const bool wide = (n & 3) == 0 && (((size_t)(const void *)pts) & 15) == 0;
    if (wide) {
        float4 *strip = s_stage + wave * (64 * 6);
        const int n4 = n * 3 / 4; // 16-byte pieces of the cloud
#pragma unroll
        for (int pass = 0; pass < PPT / 8; ++pass) {
            const int base4 = (pass * (8 * BUILD_THREADS) + wave * 512) * 3 / 4; // first piece of the wavefront's 512 points
            const float4 *p4 = reinterpret_cast<const float4 *>(pts) + base4;
            float4 v[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                v[j] = make_float4(NAN, NAN, NAN, NAN);
                if (base4 + lane + 64 * j < n4) v[j] = p4[lane + 64 * j];
            }
            if (pass > 0) __builtin_amdgcn_wave_barrier(); // (the strip is read by this wavefront only)
#pragma unroll
            for (int j = 0; j < 6; ++j) strip[lane + 64 * j] = v[j];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float f[24];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float4 w = strip[lane * 6 + j];
                f[4 * j] = w.x; f[4 * j + 1] = w.y; f[4 * j + 2] = w.z; f[4 * j + 3] = w.w;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { px[pass * 8 + i] = f[3 * i]; py[pass * 8 + i] = f[3 * i + 1]; pz[pass * 8 + i] = f[3 * i + 2]; }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int k = point_index(i);
            px[i] = py[i] = pz[i] = NAN;
            if (k < n) { px[i] = pts[k * 3]; py[i] = pts[k * 3 + 1]; pz[i] = pts[k * 3 + 2]; }
        }
    }
    for (int c = t; c < SPLIT_CELLS; c += BUILD_THREADS) s_cnt[c] = 0;
    // bounding box: minima / maxima of ALL coordinates first (v_min3 / v_max3 ignore NaNs) — three instructions per point
    // instead of twelve — next to a running sum of the coordinates' magnitudes, which is finite iff every coordinate is (or
    // overflows: a false alarm).  A point with a non-finite coordinate is in no cell, and must not lend its other coordinates
    // to the box either (grid_build_kernel takes the box over fully finite points; one stray (NaN, 1e30, 0) would stretch this
    // one until the grid is a single cell): when the sum is not finite the box is taken again over the finite points only
    // (once, all workgroups of the cloud alike).
    bool filtered = false;
    for (;;) {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool odd_one = false; // some coordinate of mine is NaN or infinite
        if (!filtered) {
            float mag0 = 0.0f, mag1 = 0.0f;
#pragma unroll
            for (int i = 0; i < PPT; i += 2) {
                mn[0] = ogc_min3_f32(mn[0], px[i], px[i + 1]); mx[0] = ogc_max3_f32(mx[0], px[i], px[i + 1]);
                mn[1] = ogc_min3_f32(mn[1], py[i], py[i + 1]); mx[1] = ogc_max3_f32(mx[1], py[i], py[i + 1]);
                mn[2] = ogc_min3_f32(mn[2], pz[i], pz[i + 1]); mx[2] = ogc_max3_f32(mx[2], pz[i], pz[i + 1]);
            }
            if (n >= PPT * BUILD_THREADS) { // (wave-uniform: every slot of mine is a point of the cloud)
#pragma unroll
                for (int i = 0; i < PPT; i += 2) {
                    mag0 += (fabsf(px[i]) + fabsf(py[i])) + fabsf(pz[i]);
                    mag1 += (fabsf(px[i + 1]) + fabsf(py[i + 1])) + fabsf(pz[i + 1]);
                }
            } else { // slots beyond n hold NaN fillers of the loads above: they must not raise the alarm
#pragma unroll
                for (int i = 0; i < PPT; ++i)
                    if (point_index(i) < n) mag0 += (fabsf(px[i]) + fabsf(py[i])) + fabsf(pz[i]);
            }
            odd_one = !(mag0 + mag1 < INFINITY);
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const float x = px[i], y = py[i], z = pz[i];
                if (isfinite(x) && isfinite(y) && isfinite(z)) {
                    mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
                    mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
                }
            }
        }
        const bool wave_odd = __builtin_amdgcn_ballot_w64(odd_one) != 0ull;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = -ogc_wave_max_f32(-mn[a]), hi = ogc_wave_max_f32(mx[a]);
            if (lane == 0) { s_red[a][wave] = lo; s_red[3 + a][wave] = hi; }
        }
        if (lane == 0) s_wave[wave] = wave_odd ? 1 : 0; // (s_wave is free until the scan)
        OGC_PROBE_BUILD(1);
        __syncthreads();
        OGC_PROBE_BUILD(2);
        if (wave == 0) {
            float lo[3], hi[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float l = lane < BUILD_THREADS / 64 ? s_red[a][lane] : INFINITY;
                const float u = lane < BUILD_THREADS / 64 ? s_red[3 + a][lane] : -INFINITY;
                lo[a] = -ogc_wave_max_f32(-l);
                hi[a] = ogc_wave_max_f32(u);
            }
            const bool any_odd = __builtin_amdgcn_ballot_w64(lane < BUILD_THREADS / 64 && s_wave[lane] != 0) != 0ull;
            if (lane == 0) {
                s_hdr = grid_header(lo, hi, n, radius, knn_k, knn_div, prefer_cells);
                s_hdr.pending = (any_odd && !filtered) ? 1 : 0; // (borrowed as the "take the box again" flag; 0 when the loop ends)
            }
        }
        __syncthreads();
        if (s_hdr.pending == 0) break;
        filtered = true;
        __syncthreads(); // everybody has read the flag before lane 0 writes the header again
    }
    // (the loop ends behind a barrier: the header is visible)
    OGC_PROBE_BUILD(3);
    GridHdr h = s_hdr;
    const int ncell = h.gx * h.gy * h.gz;
    const int per = (ncell + split - 1) / split;                 // <= SPLIT_CELLS: split >= SPLIT_MIN
    const int c_lo = min(part * per, ncell), c_hi = min(c_lo + per, ncell);

    // 2. cells of ALL points; histogram of my range; how many points lie below it / in the grid at all
    int counts = 0; // valid | below << 16   (n <= 16384: 15 bits each)
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const float x = px[i], y = py[i], z = pz[i];
        OGC_GRID_AXES(h, x, y, z, fx, fy, fz);
        const int cx = cell_clamped(fx, h.minx, h.inv_h, h.gx);
        const int cy = cell_clamped(fy, h.miny, h.inv_h, h.gy);
        const int cz = cell_clamped(fz, h.minz, h.inv_h, h.gz);
        const bool fin = fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY;
        const int c = (point_index(i) < n) ? (fin ? cx + h.gx * (cy + h.gy * cz) : -1) : -2;
        cell[i] = c;
        counts += (c >= 0 ? 1 : 0) + ((c >= 0 && c < c_lo) ? 0x10000 : 0);
        if (c >= c_lo && c < c_hi) atomicAdd(&s_cnt[c - c_lo], 1);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) counts += __shfl_xor(counts, off, 64);
    if (lane == 0) s_counts[wave] = counts;
    __syncthreads();
    OGC_PROBE_BUILD(4);

    // 3. exclusive scan of my range's counts, offset by the points below the range
    const int nloc = c_hi - c_lo;
    const int chunk = (nloc + BUILD_THREADS - 1) / BUILD_THREADS;
    const int c0 = min(t * chunk, nloc), c1 = min(c0 + chunk, nloc);
    int sum = 0;
    for (int c = c0; c < c1; ++c) sum += s_cnt[c];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    OGC_PROBE_BUILD(5);
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BUILD_THREADS / 64; ++w) {
        before += w < wave ? s_wave[w] : 0;
        total += s_counts[w];
    }
    const int npts = total & 0xFFFF, below = total >> 16;
    int run = below + before + incl - sum;
    int *cs = cell_start + (size_t)b * stride_cells;
    for (int c = c0; c < c1; ++c) {
        const int cnt = s_cnt[c];
        s_cnt[c] = run; // becomes the scatter cursor
        cs[c_lo + c] = run;
        run += cnt;
    }
    if (t == 0 && part == 0) {
        cs[ncell] = npts;
        h.npts = npts;
        // (the flags of grid_build_kernel: see there)
        const float per_query = 27.0f * (float)npts / (float)ncell;
        h.dense = per_query > 0.25f * (float)n ? 1 : 0;
        h.heavy = per_query > 120.0f ? 1 : 0;
        if (!(1.0f * (1.0f / h.inv_h) * 0.999f >= radius) || per_query > 120.0f) h.knn_general = 1;
        hdrs[b] = h;
    }
    if (t == 0) s_tail = npts;
    __syncthreads();
    OGC_PROBE_BUILD(6);

    // 4. scatter the points of my range (part 0: also the points outside the grid, behind all cells)
    float4 *sp = sorted_pts + (size_t)b * n;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int k = point_index(i), c = cell[i];
        if (c >= c_lo && c < c_hi) sp[atomicAdd(&s_cnt[c - c_lo], 1)] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
        else if (c == -1 && part == 0) sp[atomicAdd(&s_tail, 1)] = make_float4(NAN, NAN, NAN, __int_as_float(k));
    }
    OGC_PROBE_BUILD(7);
}

// parts per cloud of the split build (0: one workgroup per cloud).  OGC_GRID_SPLIT in the environment overrides (A/B runs).
static int grid_build_parts(int n) {
    static const int forced = [] { const char *e = getenv("OGC_GRID_SPLIT"); return e ? atoi(e) : -1; }();
    if (n > 16 * BUILD_THREADS) return 0;
    if (forced >= 0) return forced == 0 ? 0 : (forced < SPLIT_MIN ? SPLIT_MIN : (forced > 32 ? 32 : forced));
    return n <= 8 * BUILD_THREADS ? 8 : 16;
}

static void launch_grid_build(int b, int n, float radius, int knn_k, int stride_cells, const float *xyz, GridHdr *hdrs,
                              int *cell_start, float4 *sorted_pts, hipStream_t s, int prefer_cells = 0,
                              float knn_div = 33.5f /* points per cell = k / knn_div; 33.5: cell edge = half the expected k-th neighbour distance */) {
    const int parts = grid_build_parts(n);
    if (parts > 0 && n <= 8 * BUILD_THREADS)
        hipLaunchKernelGGL(grid_build_split_kernel<8>, dim3(b * parts), dim3(BUILD_THREADS), 0, s, knn_div, b, parts, n, radius, knn_k,
                           prefer_cells, stride_cells, xyz, hdrs, cell_start, sorted_pts);
    else if (parts > 0)
        hipLaunchKernelGGL(grid_build_split_kernel<16>, dim3(b * parts), dim3(BUILD_THREADS), 0, s, knn_div, b, parts, n, radius, knn_k,
                           prefer_cells, stride_cells, xyz, hdrs, cell_start, sorted_pts);
    else if (n <= 8 * BUILD_THREADS)
        hipLaunchKernelGGL(grid_build_kernel<8>, dim3(b), dim3(BUILD_THREADS), 0, s, knn_div, n, radius, knn_k, prefer_cells, stride_cells, xyz,
                           hdrs, cell_start, sorted_pts);
    else if (n <= 16 * BUILD_THREADS)
        hipLaunchKernelGGL(grid_build_kernel<16>, dim3(b), dim3(BUILD_THREADS), 0, s, knn_div, n, radius, knn_k, prefer_cells, stride_cells,
                           xyz, hdrs, cell_start, sorted_pts);
    else
        hipLaunchKernelGGL(grid_build_kernel<0>, dim3(b), dim3(BUILD_THREADS), 0, s, knn_div, n, radius, knn_k, prefer_cells, stride_cells, xyz,
                           hdrs, cell_start, sorted_pts);
}

constexpr int SUB = 8;               // lanes cooperating on one query in the finishing steps
constexpr int QPW = OGC_WAVE / SUB;  // queries (centres) per wavefront

__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// The candidate runs of a box of cells [xlo, xhi] x (y0-1 .. y0+1) x (z0-1 .. z0+1): the cells of one (y, z) row are
// contiguous in the cell-sorted array, so the box is NINE runs.  Lane r < 9 fetches run r; the nine (start, offset)
// pairs are then broadcast to scalars so that every lane can map a flat candidate number to an array position.
// (A macro, not a struct: the eighteen scalars must stay in SGPRs — as members of an object passed by reference the
// compiler put them in scratch memory and indexed them per candidate.)
// (SLAB — GridHdr::slab, the fast axis has at most two cells: THREE runs, the cells (any x, XLO .. XHI) of the rows z - 1 ..
// z + 1, where XLO .. XHI is then a range of y; the other six runs are empty)
#define OGC_BOX_SETUP(SLAB, XLO, XHI, Y0, Z0)                                                        \
    {                                                                                                \
        int lo_ = 0, len_ = 0;                                                                       \
        if ((SLAB) && lane < 3) {                                                                    \
            const int z_ = (Z0) + lane - 1;                                                          \
            if (z_ >= 0 && z_ < h.gz && (XLO) <= (XHI)) {                                            \
                lo_ = cs[h.gx * ((XLO) + h.gy * z_)];                                                \
                len_ = cs[h.gx * ((XHI) + h.gy * z_) + h.gx] - lo_;                                  \
            }                                                                                        \
        } else if (!(SLAB) && lane < 9) {                                                            \
            const int y_ = (Y0) + (lane % 3) - 1, z_ = (Z0) + (lane / 3) - 1;                        \
            if (y_ >= 0 && y_ < h.gy && z_ >= 0 && z_ < h.gz && (XLO) <= (XHI)) {                    \
                const int rowc_ = h.gx * (y_ + h.gy * z_);                                           \
                lo_ = cs[rowc_ + (XLO)];                                                             \
                len_ = cs[rowc_ + (XHI) + 1] - lo_;                                                  \
            }                                                                                        \
        }                                                                                            \
        int incl_ = len_;                                                                            \
        _Pragma("unroll") for (int off_ = 1; off_ < 16; off_ <<= 1) {                                \
            const int up_ = __shfl_up(incl_, off_, 64);                                              \
            if (lane >= off_) incl_ += up_;                                                          \
        }                                                                                            \
        const int excl_ = incl_ - len_;                                                              \
        box_total = lane_bcast(incl_, 8);                                                            \
        b0 = lane_bcast(lo_ - excl_, 0);                                                             \
        s1 = lane_bcast(excl_, 1); b1 = lane_bcast(lo_ - excl_, 1);                                  \
        s2 = lane_bcast(excl_, 2); b2 = lane_bcast(lo_ - excl_, 2);                                  \
        s3 = lane_bcast(excl_, 3); b3 = lane_bcast(lo_ - excl_, 3);                                  \
        s4 = lane_bcast(excl_, 4); b4 = lane_bcast(lo_ - excl_, 4);                                  \
        s5 = lane_bcast(excl_, 5); b5 = lane_bcast(lo_ - excl_, 5);                                  \
        s6 = lane_bcast(excl_, 6); b6 = lane_bcast(lo_ - excl_, 6);                                  \
        s7 = lane_bcast(excl_, 7); b7 = lane_bcast(lo_ - excl_, 7);                                  \
        s8 = lane_bcast(excl_, 8); b8 = lane_bcast(lo_ - excl_, 8);                                  \
    }
// the LAST run whose start is <= f (empty runs share their start with the next one)
#define OGC_BOX_POSITION(F)                                                                           \
    ((F) + ((F) >= s8 ? b8 : (F) >= s7 ? b7 : (F) >= s6 ? b6 : (F) >= s5 ? b5 : (F) >= s4 ? b4       \
                     : (F) >= s3 ? b3 : (F) >= s2 ? b2 : (F) >= s1 ? b1 : b0))

// squared distances of ONE candidate to TWO centres, packed (v_pk_*_f32): the reference's fp32 expression per half
__device__ __forceinline__ ogc_v2f sqdist_pair(ogc_v2f qx, ogc_v2f qy, ogc_v2f qz, float x, float y, float z) {
#pragma clang fp contract(off)
    const ogc_v2f cx2 = {x, x}, cy2 = {y, y}, cz2 = {z, z};
    const ogc_v2f dx = qx - cx2, dy = qy - cy2, dz = qz - cz2;
    return ogc_sqsum3(dx, dy, dz);
}

// Ball query of a cloud against itself over the cell lists: ONE LANE PER CANDIDATE.
// A wavefront takes eight centres that are consecutive in cell order.  Centres on the same (y, z) row of cells form a
// batch (usually the whole wavefront is one batch); the union of their 27-cell neighbourhoods is a box of nine
// contiguous runs, whose candidates are dealt to the 64 lanes (16-byte records, one load each).  Each centre of the
// batch is then tested by all lanes at once — its coordinates are wave-uniform scalars, two centres per packed
// instruction, the squared distance is the reference's fp32 expression — and one ballot turns the hits into
// consecutive slots of the centre's hit list (hit counts live in scalar registers).  Testing a candidate outside a
// centre's own 27 cells is harmless (the distance decides), so the box needs no per-centre bookkeeping.
// Finish: eight lanes per centre rank-sort the hit list by point index (indices are distinct), keep the first
// nsample, pad with the smallest, 16-byte stores — the row the reference produces by scanning in index order and
// stopping after nsample hits.  A centre with more hits than the list holds is redone through an LDS bitmap over
// point indices (set a bit per hit, read the first nsample set bits), also exact.
// Clouds flagged dense by the build (the 27 cells hold a large share of the cloud, so cell lists buy nothing and
// rows saturate early) are scanned in INDEX order instead, by the same wavefronts: hits then arrive in the order
// of the output and a wavefront stops as soon as its eight rows are full.
// (the body of the kernel: ball_query_cells_kernel below runs it too, for the wavefronts its short lists cannot hold)
__device__ __forceinline__ void ball_query_grid_body(int lane, int first_centre, int *gq_smem, int n, int m, float radius2, int nsample,
                                                     int hit_cap, int stride_cells, const float *__restrict__ xyz,
                                                     const GridHdr *__restrict__ hdrs, const int *__restrict__ cell_start,
                                                     const float4 *__restrict__ sorted_pts, int *__restrict__ idx_out) {
    const int b = blockIdx.y;
    OGC_PROBE_T(pt0);
    const GridHdr h = hdrs[b];
    int *hits = gq_smem;                                       // [QPW][hit_cap]
    int *outr = gq_smem + QPW * hit_cap;                       // [QPW][nsample] sorted rows
    unsigned *bitmap = reinterpret_cast<unsigned *>(outr + QPW * nsample); // [ceil(n / 32)], overflow path only
    const int *cs = cell_start + (size_t)b * stride_cells;
    const float4 *pts = sorted_pts + (size_t)b * n;

    // every group of eight lanes holds the eight centres (lane & 7), so 8-lane butterflies see the whole set
    const int pc = first_centre + (lane & (QPW - 1));
    float4 me = make_float4(NAN, NAN, NAN, __int_as_float(-1));
    if (pc < n) me = pts[pc]; // positions >= h.npts hold the non-finite points: no hits, an all-zero row
    const bool live = pc < h.npts;
    int cnt_s[QPW]; // wave-uniform hit counts (may exceed hit_cap)
#pragma unroll
    for (int c = 0; c < QPW; ++c) cnt_s[c] = 0;
    unsigned sorted_rows = 0; // rows written in ascending order already (index-order scan, bitmap path)
    int box_total, b0, s1, b1, s2, b2, s3, b3, s4, b4, s5, b5, s6, b6, s7, b7, s8, b8;
    const unsigned live_mask = (unsigned)__builtin_amdgcn_ballot_w64(live) & 0xFFu;

    if (h.dense) {
        // ---- index-order scan of the whole cloud: rows come out sorted, stop when all rows are full
        const float *src = xyz + (size_t)b * n * 3;
        sorted_rows = 0xFFu;
        for (int f0 = 0; f0 < n; f0 += OGC_WAVE) {
            const int f = f0 + lane;
            float x = NAN, y = NAN, z = NAN;
            if (f < n) { x = src[f * 3]; y = src[f * 3 + 1]; z = src[f * 3 + 2]; }
            bool all_full = true;
#pragma unroll
            for (int c = 0; c < QPW; c += 2) {
                if (!((live_mask >> c) & 3u)) continue;
                const ogc_v2f qx = {lane_bcast(me.x, c), lane_bcast(me.x, c + 1)};
                const ogc_v2f qy = {lane_bcast(me.y, c), lane_bcast(me.y, c + 1)};
                const ogc_v2f qz = {lane_bcast(me.z, c), lane_bcast(me.z, c + 1)};
                const ogc_v2f d = sqdist_pair(qx, qy, qz, x, y, z);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (cnt_s[c + u] >= nsample) continue;
                    const bool hit = (u == 0 ? d.x : d.y) < radius2;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                    if (mask != 0) {
                        if (hit) {
                            const int slot = cnt_s[c + u] + (int)__builtin_amdgcn_mbcnt_hi(
                                (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            if (slot < nsample) outr[(c + u) * nsample + slot] = f;
                        }
                        cnt_s[c + u] += __popcll(mask);
                    }
                    if (cnt_s[c + u] < nsample && ((live_mask >> (c + u)) & 1u)) all_full = false;
                }
            }
            if (all_full) break;
        }
    } else {
        OGC_GRID_AXES(h, me.x, me.y, me.z, fx, fy, fz);
        const int cx = cell_coord(fx, h.minx, h.inv_h, h.gx);
        const int cy = cell_coord(fy, h.miny, h.inv_h, h.gy);
        const int cz = cell_coord(fz, h.minz, h.inv_h, h.gz);
        const bool slab = h.slab != 0;
        const int cr = slab ? cy : cx, gr = slab ? h.gy : h.gx; // the coordinate a batch's box ranges over
        unsigned todo = live_mask;
        while (todo != 0) {
            const int c0 = __ffs(todo) - 1;
            const int y0 = lane_bcast(cy, c0), z0 = lane_bcast(cz, c0);
            const bool mine = live && (slab || cy == y0) && cz == z0; // (slab: a batch is the centres of one z)
            const unsigned batch = (unsigned)__builtin_amdgcn_ballot_w64(mine) & todo;
            todo &= ~batch;
            int xlo = mine ? cr : 0x7fffffff, xhi = mine ? cr : -1;
#pragma unroll
            for (int off = 1; off < QPW; off <<= 1) {
                xlo = min(xlo, __shfl_xor(xlo, off, 64));
                xhi = max(xhi, __shfl_xor(xhi, off, 64));
            }
            const int bx0 = max(lane_bcast(xlo, 0) - 1, 0), bx1 = min(lane_bcast(xhi, 0) + 1, gr - 1);
            OGC_BOX_SETUP(slab, bx0, bx1, y0, z0)
            const float4 nothing = make_float4(NAN, NAN, NAN, 0.0f); // NaN: never a hit
            float4 ahead = nothing; // the next round's candidate is in flight while this round is tested
            if (lane < box_total) ahead = pts[OGC_BOX_POSITION(lane)];
            for (int f0 = 0; f0 < box_total; f0 += OGC_WAVE) {
                const float4 cand = ahead;
                const int fn = f0 + OGC_WAVE + lane;
                ahead = nothing;
                if (fn < box_total) ahead = pts[OGC_BOX_POSITION(fn)];
                const int v = __float_as_int(cand.w);
#pragma unroll
                for (int c = 0; c < QPW; c += 2) {
                    if (!((batch >> c) & 3u)) continue; // wave-uniform
                    const ogc_v2f qx = {lane_bcast(me.x, c), lane_bcast(me.x, c + 1)};
                    const ogc_v2f qy = {lane_bcast(me.y, c), lane_bcast(me.y, c + 1)};
                    const ogc_v2f qz = {lane_bcast(me.z, c), lane_bcast(me.z, c + 1)};
                    const ogc_v2f d = sqdist_pair(qx, qy, qz, cand.x, cand.y, cand.z);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (!((batch >> (c + u)) & 1u)) continue;
                        const bool hit = (u == 0 ? d.x : d.y) < radius2;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                        if (mask == 0) continue;
                        if (hit) {
                            const int slot = cnt_s[c + u] + (int)__builtin_amdgcn_mbcnt_hi(
                                (unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            if (slot < hit_cap) hits[(c + u) * hit_cap + slot] = v;
                        }
                        cnt_s[c + u] += __popcll(mask);
                    }
                }
            }
        }
    }
    OGC_PROBE_T(pt1);
    // hit counts: lane c < 8 holds the count of centre c
    int cnt = 0;
#pragma unroll
    for (int c = 0; c < QPW; ++c) cnt = lane == c ? cnt_s[c] : cnt;
    if (h.dense) cnt = min(cnt, nsample);

    // centres whose hit list overflowed: exact redo through a bitmap over point indices
    unsigned over = h.dense ? 0u : (unsigned)__builtin_amdgcn_ballot_w64(lane < QPW && cnt > hit_cap);
    while (over != 0) {
        const int c = __ffs(over) - 1;
        over &= over - 1;
        sorted_rows |= 1u << c;
        const int words = (n + 31) >> 5;
        for (int w = lane; w < words; w += OGC_WAVE) bitmap[w] = 0u;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const float qx = lane_bcast(me.x, c), qy = lane_bcast(me.y, c), qz = lane_bcast(me.z, c);
        OGC_GRID_AXES(h, qx, qy, qz, gfx_, gfy_, gfz_);
        const int ccx = cell_coord(gfx_, h.minx, h.inv_h, h.gx);
        const int ccy = cell_coord(gfy_, h.miny, h.inv_h, h.gy);
        const int ccz = cell_coord(gfz_, h.minz, h.inv_h, h.gz);
        const bool slab_o = h.slab != 0;
        const int bx0 = max((slab_o ? ccy : ccx) - 1, 0), bx1 = min((slab_o ? ccy : ccx) + 1, (slab_o ? h.gy : h.gx) - 1);
        OGC_BOX_SETUP(slab_o, bx0, bx1, ccy, ccz)
        for (int f0 = 0; f0 < box_total; f0 += OGC_WAVE) {
            const int f = f0 + lane;
            if (f < box_total) {
                const float4 cand = pts[OGC_BOX_POSITION(f)];
                if (ogc_sqdist(qx, qy, qz, cand.x, cand.y, cand.z) < radius2) {
                    const unsigned v = (unsigned)__float_as_int(cand.w);
                    atomicOr(&bitmap[v >> 5], 1u << (v & 31u));
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        int found = 0;
        for (int w0 = 0; w0 < words && found < nsample; w0 += OGC_WAVE) {
            const int w = w0 + lane;
            unsigned bits = w < words ? bitmap[w] : 0u;
            const int pcn = __popc(bits);
            int incl = pcn;
#pragma unroll
            for (int off = 1; off < OGC_WAVE; off <<= 1) {
                const int up = __shfl_up(incl, off, 64);
                if (lane >= off) incl += up;
            }
            int pos = found + incl - pcn;
            while (bits != 0u && pos < nsample) {
                outr[c * nsample + pos] = (w << 5) + (__ffs(bits) - 1);
                bits &= bits - 1u;
                ++pos;
            }
            found += lane_bcast(incl, OGC_WAVE - 1);
        }
        if (lane == c) cnt = min(found, nsample);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();

    OGC_PROBE_T(pt2);
    // finish: eight lanes per centre
    const int sub = lane & (SUB - 1), qi = lane >> 3;
    const int total_hits = __shfl(cnt, qi, 64);
    const int q = __float_as_int(__shfl(me.w, qi, 64));
    int *row = outr + qi * nsample;
    if (!((sorted_rows >> qi) & 1u)) {
        // rank sort (the indices are distinct): element e goes to position #{f : hits[f] < hits[e]}
        const int *mine_hits = hits + qi * hit_cap;
        for (int e = sub; e < total_hits; e += SUB) {
            const int ve = mine_hits[e];
            int rank = 0;
            for (int f = 0; f < total_hits; ++f) rank += mine_hits[f] < ve ? 1 : 0;
            if (rank < nsample) row[rank] = ve;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    OGC_PROBE_T(pt3);
    if (q >= 0) {
        const int kept = min(total_hits, nsample);
        const int first = kept > 0 ? row[0] : 0;
        int *o = idx_out + ((size_t)b * m + q) * nsample;
        if ((nsample & 3) == 0) { // 16-byte stores
            for (int j = sub * 4; j < nsample; j += SUB * 4) {
                int4 val;
                val.x = j < kept ? row[j] : first;
                val.y = j + 1 < kept ? row[j + 1] : first;
                val.z = j + 2 < kept ? row[j + 2] : first;
                val.w = j + 3 < kept ? row[j + 3] : first;
                *reinterpret_cast<int4 *>(o + j) = val;
            }
        } else {
            for (int j = sub; j < nsample; j += SUB) o[j] = j < kept ? row[j] : first;
        }
    }
    OGC_PROBE_T(pt4);
    OGC_PROBE_ADD(16, pt0, pt1);
    OGC_PROBE_ADD(17, pt1, pt2);
    OGC_PROBE_ADD(18, pt2, pt3);
    OGC_PROBE_ADD(19, pt3, pt4);
}

__global__ __launch_bounds__(OGC_WAVE) void ball_query_grid_kernel(int n, int m, float radius2, int nsample,
                                                                   int hit_cap, int stride_cells,
                                                                   const float *__restrict__ xyz,
                                                                   const GridHdr *__restrict__ hdrs,
                                                                   const int *__restrict__ cell_start,
                                                                   const float4 *__restrict__ sorted_pts,
                                                                   int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) int gq_smem[];
    ball_query_grid_body(threadIdx.x, blockIdx.x * QPW, gq_smem, n, m, radius2, nsample, hit_cap, stride_cells, xyz, hdrs, cell_start,
                         sorted_pts, idx_out);
}

// ---- the same query with FOUR lanes per centre, for sparse neighbourhoods -----------------------------------------------
// ball_query_grid_kernel deals the candidates of eight centres' common box to the 64 lanes and tests every centre against
// every lane: with the ~60 candidates and 12 hits per centre of the loss's shape (8192 points in 60 x 4 x 80, r = 2) most
// of its ~780 vector instructions per wavefront are bookkeeping — nine-run position lookups, a ballot, a scalar branch and
// a slot computation per (centre, round), scalar broadcasts of the centres, a quadratic rank sort through LDS.  Here a
// wavefront takes SIXTEEN centres consecutive in cell order and each gets four lanes, which walk the centre's own nine
// runs (the three cells around it in x of each of the 3 x 3 rows): lane s tests candidates s and s + 4 of a run with one
// packed distance, a hit's list slot is a population count over the group's bits of the two ballots, and a miss is
// stored to a spare slot instead of branching.  Lists of up to BQ_FAST hits are then sorted IN REGISTERS by a bitonic
// network over 4 lanes x 8 keys (exchanges at distance < 8 inside a lane, the others by quad permutations) and the
// rows leave straight from the registers — no loop, no LDS round trip after two reads.  The rows are those of the
// reference's index-ordered scan, as with the other kernel.  A wavefront with a longer list, and every wavefront of a
// cloud the build flagged dense or heavy, runs the general body above (twice: eight centres each).
#ifndef OGC_BQ_STRIP_PAD
#define OGC_BQ_STRIP_PAD 0
#endif
constexpr int CL = 4;                 // lanes per centre
constexpr int CPW = OGC_WAVE / CL;    // centres per wavefront
constexpr int BQ_FAST = 32;           // hits per centre the register sort holds
constexpr int BQ_CAP = 64;            // hit slots per centre (slot BQ_CAP takes the misses; also the general body's lists)
constexpr int BQ_SEG = 20;            // ints per lane of a centre's LDS strip: 16 private hit slots, slot 16 takes misses / overflow
// ints per centre (the compacted list of up to BQ_CAP hits + sentinels lives in the same strip) + BQ_STRIP_PAD.  With 80 ints per
// centre the sixteen strips of a wavefront start in two banks only (80 mod 32 = 16); padding the strips to 84 spreads them over
// eight start banks but costs the eighth wavefront per SIMD (5376 bytes per wavefront: 18.1 against 16.9 us) — the pad stays 0 and
// the MISSES, which are most of the stores, go to one of the four spare slots of a lane's segment by centre pair instead
constexpr int BQ_STRIP_PAD = OGC_BQ_STRIP_PAD;
constexpr int BQ_LIST = CL * BQ_SEG + BQ_STRIP_PAD;
constexpr int BQ_RUN = 128;           // longest run the slab walk takes (a longer one sends the wavefront to the general body)
constexpr int BQ_PAD = BQ_RUN + 32;   // records readable past the end of the cell-sorted array (lanes whose run has ended read on)

// 16-byte store of an output row piece, non-temporal: the rows are 33 MB that nobody in this launch reads again; as ordinary
// stores they sit dirty in the L2s until the end-of-kernel write-back (1.6 us of the operator at the C4 loss shape).
__device__ __forceinline__ void store_row16(int *p, int4 v) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    const v4i_ vv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(vv, reinterpret_cast<v4i_ *>(p));
}

template <int R>
__device__ __forceinline__ int quad_bcast(int v) { // lane R of every group of four lanes
    return __builtin_amdgcn_update_dpp(0, v, R * 0x55, 0xF, 0xF, true);
}

#ifndef OGC_BQ_MINWAVES
#define OGC_BQ_MINWAVES 8   // wavefronts per SIMD the register budget leaves room for (tools/bq_probe.hip builds variants)
#endif
template <int NS, int WPB>
__global__ __launch_bounds__(OGC_WAVE * WPB, OGC_BQ_MINWAVES) void ball_query_cells_kernel(int n, int m, float radius2, int stride_cells, int lds_ints,
                                                                       const float *__restrict__ xyz,
                                                                       const GridHdr *__restrict__ hdrs,
                                                                       const int *__restrict__ cell_start,
                                                                       const float4 *__restrict__ sorted_pts,
                                                                       int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) int gq_smem_all[];
    // WPB independent wavefronts per workgroup (nothing is shared between them: a workgroup is only the unit of dispatch)
    const int lane = threadIdx.x & (OGC_WAVE - 1), wave_in_block = threadIdx.x >> 6;
    const int grp = blockIdx.x * WPB + wave_in_block; // sixteen centres
    int *gq_smem = gq_smem_all + wave_in_block * lds_ints;
    const int b = blockIdx.y, sub = lane & (CL - 1), g = lane >> 2;
    OGC_PROBE_T(pt0);
    const GridHdr h = hdrs[b];
    bool general = h.dense != 0 || h.heavy != 0;
    if (!general) {
        const int *cs = cell_start + (size_t)b * stride_cells;
        const float4 *pts = sorted_pts + (size_t)b * n;
        const int pc = grp * CPW + g;
        float4 me = make_float4(NAN, NAN, NAN, __int_as_float(-1));
        if (pc < n) me = pts[pc]; // positions >= h.npts hold the non-finite points: no hits, an all-zero row
        const bool live = pc < h.npts;
        int *mine = gq_smem + g * BQ_LIST; // the centre's strip
        int *seg = mine + sub * BQ_SEG;    // my own hit slots
        {   // every slot starts as +inf: the sort reads the first eight of each lane whatever was found
            const int4 inf4 = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
            int4 *l4 = reinterpret_cast<int4 *>(seg);
#pragma unroll
            for (int i = 0; i < 4; ++i) l4[i] = inf4;
        }
        // the centre's cell, as the build computed it (a live centre is finite: the conversion saturates where cell_coord
        // clamps, and the clamp to the grid follows either way)
        OGC_GRID_AXES(h, me.x, me.y, me.z, gfx, gfy, gfz);
        const int cx = min(cell_floor(gfx, h.minx, h.inv_h), h.gx - 1);
        const int cy = min(cell_floor(gfy, h.miny, h.inv_h), h.gy - 1);
        const int cz = min(cell_floor(gfz, h.minz, h.inv_h), h.gz - 1);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.gx - 1);
        const bool slab = h.slab != 0; // (wave-uniform)
        // run r = the cells x0 .. x1 of row (cy + r % 3 - 1, cz + r / 3 - 1): lane s fetches runs s and s + 4, all fetch run 8
        // (no branch around the loads and all six in flight together: rows outside the grid read a clamped row and get
        // length 0 afterwards).  Slab grids: run r < 3 = the cells (any x, cy - 1 .. cy + 1) of z = cz + r - 1, lane s fetches run s.
        auto row_of = [&](int r, bool &inside) {
            const int r3 = r / 3;
            const int y = cy + (r - 3 * r3) - 1, z = cz + r3 - 1;
            inside = live && y >= 0 && y < h.gy && z >= 0 && z < h.gz;
            return h.gx * (min(max(y, 0), h.gy - 1) + h.gy * min(max(z, 0), h.gz - 1));
        };
        bool in_a, in_b, in_c;
        int row_a = row_of(sub, in_a), row_b = row_of(sub + 4, in_b), row_c = row_of(8, in_c);
        int first_a = row_a + x0, last_a = row_a + x1 + 1;
        if (slab) {
            const int z = cz + sub - 1;
            in_a = live && sub < 3 && z >= 0 && z < h.gz;
            const int zc = min(max(z, 0), h.gz - 1);
            first_a = h.gx * (max(cy - 1, 0) + h.gy * zc);
            last_a = h.gx * (min(cy + 1, h.gy - 1) + h.gy * zc) + h.gx;
            row_b = row_c = first_a - x0; // (their loads repeat lane s's first one; the runs do not exist)
            in_b = in_c = false;
        }
        int lo_a = cs[first_a], end_a = cs[last_a];
        int lo_b = cs[row_b + x0], end_b = cs[row_b + x1 + 1];
        int lo_c = cs[row_c + x0], end_c = cs[row_c + x1 + 1];
        asm volatile("" : "+v"(lo_a), "+v"(end_a), "+v"(lo_b), "+v"(end_b), "+v"(lo_c), "+v"(end_c));
        const int len_a = in_a ? end_a - lo_a : 0, len_b = in_b ? end_b - lo_b : 0, len_c = in_c ? end_c - lo_c : 0;

        // Every lane appends ITS hits to ITS sixteen slots — no ballot, no slot arithmetic across the group (that was ~13 of
        // the ~26 vector instructions a tested candidate cost); a miss is stored to slot 16 instead of branching, and so is
        // the seventeenth hit of a lane (the count goes on: such a wavefront is redone by the general body).
        int cnt_l = 0; // my hits
        bool crowded = false; // (wave-uniform) a single centre's candidates do not fit the LDS strip: the general body takes over
        // (slots 16 .. 19 of a segment are spare: a miss goes to 16 + (centre pair mod 4), so that the eight centres whose strips
        // start in the same bank spread their — frequent — miss stores over four banks instead of one)
        const int miss = 16 + ((g >> 1) & 3);
        auto slots = [&](bool has_a, bool near_a, bool has_b, bool near_b, int ia, int ib) {
            const bool hit_a = has_a && near_a, hit_b = has_b && near_b;
            seg[hit_a ? min(cnt_l, 16) : miss] = ia;
            cnt_l += hit_a ? 1 : 0;
            seg[hit_b ? min(cnt_l, 16) : miss] = ib;
            cnt_l += hit_b ? 1 : 0;
        };
        const char *pts_bytes = reinterpret_cast<const char *>(pts);
        auto record = [&](int position) { // (positions past the end of a run are read — the array is padded — and discarded)
            return *reinterpret_cast<const float4 *>(pts_bytes + ((unsigned)position << 4));
        };
        if (slab) {
            // three long runs, walked side by side: step t tests the candidates 8 t .. 8 t + 7 of each (six loads in flight per
            // lane).  Positions are kept as byte offsets; a lane whose runs have ended keeps reading while others in the wavefront
            // go on — at most BQ_RUN records past a run's end: the next cloud's records or the padding behind the last cloud.
            crowded = __builtin_amdgcn_ballot_w64(len_a > BQ_RUN) != 0ull;
            const int b0 = quad_bcast<0>(in_a ? lo_a : 0), b1 = quad_bcast<1>(in_a ? lo_a : 0), b2 = quad_bcast<2>(in_a ? lo_a : 0);
            const unsigned e0 = (unsigned)(b0 + quad_bcast<0>(len_a)) << 4, e1 = (unsigned)(b1 + quad_bcast<1>(len_a)) << 4,
                           e2 = (unsigned)(b2 + quad_bcast<2>(len_a)) << 4;
            unsigned q0 = (unsigned)(b0 + sub) << 4, q1 = (unsigned)(b1 + sub) << 4, q2 = (unsigned)(b2 + sub) << 4;
            auto rec = [&](unsigned byte_offset) { return *reinterpret_cast<const float4 *>(pts_bytes + byte_offset); };
            constexpr unsigned NEXT = CL * 16u; // my second candidate of a step
            // (Software-pipelining this loop — the loads of step t + 1 issued before step t is tested, to shorten the wavefront's chain
            // of dependent round trips — needs 24 more registers than the 64 that eight wavefronts per SIMD leave: 116-140 bytes of
            // scratch per lane and 46 us instead of 17 for the kernel.  Measured at the end of round 5 and dropped.)
            if (!crowded)
                for (;;) {
                    const float4 a0 = rec(q0), c0 = rec(q0 + NEXT);
                    const float4 a1 = rec(q1), c1 = rec(q1 + NEXT);
                    const float4 a2 = rec(q2), c2 = rec(q2 + NEXT);
                    __builtin_amdgcn_sched_barrier(0);
                    const ogc_v2f d0 = sqdist_pair(ogc_v2f{a0.x, c0.x}, ogc_v2f{a0.y, c0.y}, ogc_v2f{a0.z, c0.z}, me.x, me.y, me.z);
                    slots(q0 < e0, d0.x < radius2, q0 + NEXT < e0, d0.y < radius2, __float_as_int(a0.w), __float_as_int(c0.w));
                    const ogc_v2f d1 = sqdist_pair(ogc_v2f{a1.x, c1.x}, ogc_v2f{a1.y, c1.y}, ogc_v2f{a1.z, c1.z}, me.x, me.y, me.z);
                    slots(q1 < e1, d1.x < radius2, q1 + NEXT < e1, d1.y < radius2, __float_as_int(a1.w), __float_as_int(c1.w));
                    const ogc_v2f d2 = sqdist_pair(ogc_v2f{a2.x, c2.x}, ogc_v2f{a2.y, c2.y}, ogc_v2f{a2.z, c2.z}, me.x, me.y, me.z);
                    slots(q2 < e2, d2.x < radius2, q2 + NEXT < e2, d2.y < radius2, __float_as_int(a2.w), __float_as_int(c2.w));
                    q0 += 2 * NEXT; q1 += 2 * NEXT; q2 += 2 * NEXT;
                    if (__builtin_amdgcn_ballot_w64(q0 < e0 || q1 < e1 || q2 < e2) == 0ull) break;
                }
        } else
        // three runs at a time: six candidate loads in flight per lane
#pragma unroll
        for (int r0 = 0; r0 < 9; r0 += 3) {
            int lo[3], hi[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int r = r0 + i;
                const int l = r == 0 ? quad_bcast<0>(lo_a) : r == 1 ? quad_bcast<1>(lo_a) : r == 2 ? quad_bcast<2>(lo_a)
                            : r == 3 ? quad_bcast<3>(lo_a) : r == 4 ? quad_bcast<0>(lo_b) : r == 5 ? quad_bcast<1>(lo_b)
                            : r == 6 ? quad_bcast<2>(lo_b) : r == 7 ? quad_bcast<3>(lo_b) : lo_c;
                const int w = r == 0 ? quad_bcast<0>(len_a) : r == 1 ? quad_bcast<1>(len_a) : r == 2 ? quad_bcast<2>(len_a)
                            : r == 3 ? quad_bcast<3>(len_a) : r == 4 ? quad_bcast<0>(len_b) : r == 5 ? quad_bcast<1>(len_b)
                            : r == 6 ? quad_bcast<2>(len_b) : r == 7 ? quad_bcast<3>(len_b) : len_c;
                lo[i] = l;
                hi[i] = l + w;
            }
            float4 ca[3], cb[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ca[i] = record(lo[i] + sub);
                cb[i] = record(lo[i] + sub + CL);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const ogc_v2f d = sqdist_pair(ogc_v2f{ca[i].x, cb[i].x}, ogc_v2f{ca[i].y, cb[i].y}, ogc_v2f{ca[i].z, cb[i].z},
                                              me.x, me.y, me.z);
                const int p = lo[i] + sub;
                slots(p < hi[i], d.x < radius2, p + CL < hi[i], d.y < radius2, __float_as_int(ca[i].w), __float_as_int(cb[i].w));
                // a run longer than eight candidates (wave-uniform test)
                int pp = p + 2 * CL;
                while (__builtin_amdgcn_ballot_w64(pp < hi[i]) != 0ull) {
                    const float4 a = record(min(pp, n - 1)), c2 = record(min(pp + CL, n - 1));
                    const ogc_v2f d2 = sqdist_pair(ogc_v2f{a.x, c2.x}, ogc_v2f{a.y, c2.y}, ogc_v2f{a.z, c2.z}, me.x, me.y, me.z);
                    slots(pp < hi[i], d2.x < radius2, pp + CL < hi[i], d2.y < radius2, __float_as_int(a.w), __float_as_int(c2.w));
                    pp += 2 * CL;
                }
            }
        }
        OGC_PROBE_T(pt1);
        // hits of my centre (the same number in its four lanes)
        int cnt = cnt_l + __builtin_amdgcn_update_dpp(0, cnt_l, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
        cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xF, 0xF, true);               // quad_perm [2,3,0,1]
        general = crowded || __builtin_amdgcn_ballot_w64(cnt > BQ_CAP || cnt_l > 16) != 0ull;
        if (!general) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            int4 k0 = *reinterpret_cast<const int4 *>(seg);
            int4 k1 = *reinterpret_cast<const int4 *>(seg + 4);
            // The sort below takes eight keys per lane.  A lane with more than eight hits (one wavefront in ~8 on the C4 scenes),
            // or a centre with more than BQ_FAST: the four lanes' hits are first moved to the front of the centre's strip, one
            // after the other — the list the long-list code further down expects — and read back eight per lane.
            if (__builtin_amdgcn_ballot_w64(cnt_l > 8 || cnt > BQ_FAST) != 0ull) {
                const int4 k2 = *reinterpret_cast<const int4 *>(seg + 8);
                const int4 k3 = *reinterpret_cast<const int4 *>(seg + 12);
                const int own[16] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, k2.x, k2.y, k2.z, k2.w, k3.x, k3.y, k3.z, k3.w};
                const int c0 = quad_bcast<0>(cnt_l), c1 = quad_bcast<1>(cnt_l), c2 = quad_bcast<2>(cnt_l);
                const int before = (sub > 0 ? c0 : 0) + (sub > 1 ? c1 : 0) + (sub > 2 ? c2 : 0);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier(); // everybody has read its slots
                int *spare = mine + BQ_CAP + sub;  // (a write nobody reads)
#pragma unroll
                for (int r = 0; r < 16; ++r) *(r < cnt_l ? mine + before + r : spare) = own[r];
#pragma unroll
                for (int r = 0; r < 8; ++r) *(sub * 8 + r >= cnt ? mine + sub * 8 + r : spare) = 0x7fffffff;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                k0 = *reinterpret_cast<const int4 *>(mine + sub * 8);
                k1 = *reinterpret_cast<const int4 *>(mine + sub * 8 + 4);
            }
            int x[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
            // bitonic network, element e = 8 * lane + register, every exchange ascending (each merge starts with the
            // "flip" e <-> e ^ (k - 1), then half-cleaners e <-> e ^ j)
#define OGC_BQ_INTRA(MASK)                                                  \
            _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_)                \
                if ((r_ ^ (MASK)) > r_) {                                   \
                    const int lo_ = min(x[r_], x[r_ ^ (MASK)]);             \
                    x[r_ ^ (MASK)] = max(x[r_], x[r_ ^ (MASK)]);            \
                    x[r_] = lo_;                                            \
                }
            // partner = register (r ^ RMASK) of the lane given by the quad permutation QP; the lower lane keeps the minimum
#define OGC_BQ_INTER(QP, RMASK, UPPER)                                                                  \
            {                                                                                           \
                int p_[8];                                                                              \
                _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_)                                        \
                    p_[r_] = __builtin_amdgcn_update_dpp(0, x[r_ ^ (RMASK)], QP, 0xF, 0xF, true);       \
                _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_)                                        \
                    x[r_] = (UPPER) ? max(x[r_], p_[r_]) : min(x[r_], p_[r_]);                          \
            }
            const bool odd = (sub & 1) != 0, high = (sub & 2) != 0;
            OGC_BQ_INTRA(1)                                                     // runs of 2
            OGC_BQ_INTRA(3) OGC_BQ_INTRA(1)                                     // 4
            OGC_BQ_INTRA(7) OGC_BQ_INTRA(2) OGC_BQ_INTRA(1)                     // 8
            OGC_BQ_INTER(0xB1, 7, odd) OGC_BQ_INTRA(4) OGC_BQ_INTRA(2) OGC_BQ_INTRA(1)                              // 16: lane ^ 1
            OGC_BQ_INTER(0x1B, 7, high) OGC_BQ_INTER(0xB1, 0, odd) OGC_BQ_INTRA(4) OGC_BQ_INTRA(2) OGC_BQ_INTRA(1)  // 32: lane ^ 3, ^ 1
#undef OGC_BQ_INTRA
#undef OGC_BQ_INTER
            const int q = __float_as_int(me.w);
            const int kept = min(cnt, NS);
            const int first = cnt > 0 ? quad_bcast<0>(x[0]) : 0;
            int *o = idx_out + ((size_t)b * m + max(q, 0)) * NS;
            OGC_PROBE_T(pf2);
            if (cnt <= BQ_FAST && q >= 0) {
                // lane L holds the sorted entries 8 L .. 8 L + 7.  Stores in which the group's four lanes cover 64
                // CONTIGUOUS bytes need lane L to write entries 4 L .. 4 L + 3 (then 16 + 4 L ..): an exchange inside the
                // quad (a store instruction whose lanes write every other 16 bytes leaves half-written lines everywhere)
                int v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = sub * 8 + r < kept ? x[r] : first;
                int s1[4], s2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a1 = __builtin_amdgcn_update_dpp(0, v[r], 0x50, 0xF, 0xF, true);     // quad_perm [0,0,1,1]
                    const int b1 = __builtin_amdgcn_update_dpp(0, v[r + 4], 0x50, 0xF, 0xF, true);
                    const int a2 = __builtin_amdgcn_update_dpp(0, v[r], 0xFA, 0xF, 0xF, true);     // quad_perm [2,2,3,3]
                    const int b2 = __builtin_amdgcn_update_dpp(0, v[r + 4], 0xFA, 0xF, 0xF, true);
                    s1[r] = odd ? b1 : a1;
                    s2[r] = odd ? b2 : a2;
                }
                const int j0 = sub * 4;
                if (j0 < NS) store_row16(o + j0, make_int4(s1[0], s1[1], s1[2], s1[3]));
                if (16 + j0 < NS) store_row16(o + 16 + j0, make_int4(s2[0], s2[1], s2[2], s2[3]));
                const int4 pad = make_int4(first, first, first, first);
#pragma unroll
                for (int j = BQ_FAST; j < NS; j += 16) store_row16(o + j + j0, pad);
            }
            // lists of 33 .. BQ_CAP hits (rare where this kernel is used), one at a time by the WHOLE wavefront: lane e takes
            // element e, its rank is #{f : hits[f] < hits[e]} (distinct indices) from broadcast 16-byte reads of the list,
            // entries are stored one by one.  (Four lanes doing this for their own centre take ~10 us, and so does sending
            // the wavefront through the general body: the kernel ends with its slowest wavefront.)
            unsigned long long big = __builtin_amdgcn_ballot_w64(sub == 0 && cnt > BQ_FAST);
            while (big != 0ull) {
                const int src = __ffsll((long long)big) - 1;
                big &= big - 1ull;
                const int cc = lane_bcast(cnt, src), qq = lane_bcast(q, src);
                int *list = gq_smem + (src >> 2) * BQ_LIST;
                if (lane < 4) list[cc + lane] = 0x7fffffff; // sentinels: the 16-byte reads run past the end
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int ve = lane < cc ? list[lane] : 0x7fffffff;
                int rank = 0;
                for (int f = 0; f < cc; f += 4) {
                    const int4 w = *reinterpret_cast<const int4 *>(list + f);
                    rank += (w.x < ve ? 1 : 0) + (w.y < ve ? 1 : 0) + (w.z < ve ? 1 : 0) + (w.w < ve ? 1 : 0);
                }
                const unsigned long long zero = __builtin_amdgcn_ballot_w64(lane < cc && rank == 0);
                const int lowest = lane_bcast(ve, __ffsll((long long)zero) - 1);
                if (qq >= 0) {
                    int *oc = idx_out + ((size_t)b * m + qq) * NS;
                    if (lane < cc && rank < NS) oc[rank] = ve;
                    if (cc + lane < NS) oc[cc + lane] = lowest;
                }
            }
            OGC_PROBE_T(pf3);
            OGC_PROBE_ADD(16, pt0, pt1);
            OGC_PROBE_ADD(18, pt1, pf2);
            OGC_PROBE_ADD(19, pf2, pf3);
            return;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier(); // the general body reuses the LDS
    }
#pragma unroll 1
    for (int half = 0; half < CPW / QPW; ++half)
        ball_query_grid_body(lane, grp * CPW + half * QPW, gq_smem, n, m, radius2, NS, BQ_CAP, stride_cells, xyz, hdrs,
                             cell_start, sorted_pts, idx_out);
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
// ---- the first block of a plain k-NN as ONE sorting network ------------------------------------------------------------------
// 8 NK keys of a query (NK per lane of its 8-lane group, element e = lane * NK + register) sorted ascending by a bitonic
// network: compare-exchange distances below NK stay inside a lane (register pairs, compile-time indices), the others are
// lane exchanges (ds_swizzle xor 1 / 2 / 4).  All eight groups of the wavefront run the same instruction stream, so the
// cost is per wavefront, not per admitted candidate as with the insert-and-rescan of `consider` (which serialises over
// the candidates of all eight queries: ~60 % of the kernel's instructions at k = 32).
template <int X>
__device__ __forceinline__ u64 knn_xor_lane(u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)v, (X << 10) | 0x1F);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_swizzle((int)(unsigned)(v >> 32), (X << 10) | 0x1F);
    return ((u64)hi << 32) | lo;
}

template <int NK, int SUBT>
__device__ __forceinline__ void knn_sort_keys(u64 (&key)[NK], int sub) {
    constexpr int N = SUBT * NK;
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int d = size >> 1; d >= 1; d >>= 1) {
            if (d >= NK) {
                const int lx = d / NK;
                const bool lower = (sub & lx) == 0;
#pragma unroll
                for (int t = 0; t < NK; ++t) {
                    const u64 other = lx == 1 ? knn_xor_lane<1>(key[t]) : (lx == 2 ? knn_xor_lane<2>(key[t]) : (lx == 4 ? knn_xor_lane<4>(key[t]) : (lx == 8 ? knn_xor_lane<8>(key[t]) : knn_xor_lane<16>(key[t]))));
                    const bool up = ((sub * NK + t) & size) == 0;
                    const bool take = (up == lower) ? other < key[t] : other > key[t];
                    key[t] = take ? other : key[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < NK; ++t) {
                    if ((t & d) == 0) {
                        const bool up = ((sub * NK + t) & size) == 0;
                        const u64 a = key[t], c = key[t | d];
                        const bool sw = up ? c < a : a < c;
                        key[t] = sw ? c : a;
                        key[t | d] = sw ? a : c;
                    }
                }
            }
        }
    }
}

// keys of the flat candidate list `mine[0 .. total)` (total <= 8 NK), sorted; the k smallest go to kept[] in ascending order.
// Returns the number kept.
template <int NK, int SUBT>
__device__ __forceinline__ int knn_first_block(const float4 *__restrict__ pts, const int *mine, int total, float qx, float qy,
                                               float qz, int sub, int k, u64 *kept, int have) {
    // slots 0 .. have - 1: the keys kept so far (a later shell merges into them); then the `total` new candidates
    constexpr int SUB = SUBT; // lanes per query (shadows the file's constant)
    u64 key[NK];
    float4 cand[NK];
#pragma unroll
    for (int t = 0; t < NK; ++t) { // all loads in flight; slot t * 8 + sub (any assignment will do: everything is sorted)
        const int f = t * SUB + sub - have;
        cand[t] = make_float4(NAN, NAN, NAN, 0.f);
        if (f >= 0 && f < total) cand[t] = pts[mine[f]];
    }
    int valid = 0;
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        const float d = ogc_sqdist(qx, qy, qz, cand[t].x, cand[t].y, cand[t].z);
        bool ok = d < INFINITY; // NaN / inf are never selected (empty slots hold NaN)
        key[t] = ok ? (((u64)__float_as_uint(d) << 32) | (unsigned)__float_as_int(cand[t].w)) : ~0ull;
        if (t * SUB + sub < have) { key[t] = kept[t * SUB + sub]; ok = true; }
        valid += ok ? 1 : 0;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier(); // kept[] is rewritten below
    valid += __builtin_amdgcn_ds_swizzle(valid, (1 << 10) | 0x1F);
    valid += __builtin_amdgcn_ds_swizzle(valid, (2 << 10) | 0x1F);
    valid += __builtin_amdgcn_ds_swizzle(valid, (4 << 10) | 0x1F);
    if constexpr (SUBT >= 16) valid += __builtin_amdgcn_ds_swizzle(valid, (8 << 10) | 0x1F);
    if constexpr (SUBT == 32) valid += __builtin_amdgcn_ds_swizzle(valid, (16 << 10) | 0x1F);
    knn_sort_keys<NK, SUBT>(key, sub);
    const int keep = min(valid, k);
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        const int e = sub * NK + t;
        if (e < keep) kept[e] = key[t];
    }
    return keep;
}

// MODE 0: squared distances (ogc_knn).  MODE 1: sqrt + radius clamp of the indices (ogc_knn_clamped).
constexpr int KNN_FLAT_CAP = 192; // positions of the first shell kept as one flat list per query (else: run by run)
template <int MODE, int SUBT = 8>
__global__ __launch_bounds__(OGC_WAVE, 4) void knn_grid_kernel(int n, int m, int k, float radius, float lim2, int stride_cells,
                                                            int deferred, const float *__restrict__ unknown,
                                                            const GridHdr *__restrict__ hdrs,
                                                            const int *__restrict__ cell_start,
                                                            const float4 *__restrict__ sorted_pts,
                                                            float *__restrict__ dist_out, int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) u64 kq_smem[];
    // SUBT lanes per query: 8 (eight queries per wavefront), or 16 (four) for launches that leave most of the chip idle — few
    // queries of one or two clouds, FlowStep3D at B = 1 — where a wavefront's serial work per query, not the number of wavefronts, is the time
    constexpr int SUB = SUBT, QPW = OGC_WAVE / SUBT;
    const int lane = threadIdx.x, b = blockIdx.y;
    const int sub = lane & (SUB - 1), qi = lane / SUB;
    u64 *kept = kq_smem + (size_t)qi * k;           // [QPW][k]
    u64 *outk = kq_smem + (size_t)(QPW + qi) * k;   // [QPW][k]
    int *flat = reinterpret_cast<int *>(kq_smem + (size_t)2 * QPW * k); // [QPW][KNN_FLAT_CAP] positions of the first shell
    const GridHdr h = hdrs[b];
    // deferred: knn_cells_kernel ran first.  It did every row of a cloud it could take except the rows it marked with
    // idx[row][0] = -1 (a list longer than its register sort), and nothing of a cloud flagged knn_general.
    // (deferred == 2: knn_wave_kernel ran first on EVERY cloud — whatever knn_general says — and marked the rows it left)
    if (deferred && (deferred == 2 || !h.knn_general) && !h.pending) return;
    int p = blockIdx.x * QPW + qi;
    if (deferred && (deferred == 2 || !h.knn_general) && p < n && idx_out[((size_t)b * n + p) * k] != -1) p = n; // done already: no work, no output
    const int *cs = cell_start + (size_t)b * stride_cells;
    const float4 *pts = sorted_pts + (size_t)b * m;
    const unsigned below = (1u << sub) - 1u;

    float qx = NAN, qy = NAN, qz = NAN;
    if (p < n) {
        const float *u = unknown + ((size_t)b * n + p) * 3;
        qx = u[0]; qy = u[1]; qz = u[2];
    }
    int cnt = 0, maxpos = 0;
    u64 maxkey = 0;
    // Radius-limited search (MODE 1 with a radius): a neighbour beyond the radius is replaced by the nearest one in the
    // output whatever it is, so only candidates WITHIN the radius are kept (in C4's smoothness term ~2 of the ~27 a
    // block holds — the kept set, its maximum tracking and the final rank sort shrink accordingly); the nearest
    // candidate of all is tracked on the side for the rows that have nobody within the radius.
    // within  <=>  sqrtf(d2) <= radius  <=>  d2 <= lim2, lim2 = the largest float whose (correctly rounded) root is
    // <= radius (sqrtf is monotone), found among the neighbours of radius^2 by the host (knn_radius_limit2).
    const bool limited = MODE == 1 && radius >= 0.0f;
    u64 best_any = ~0ull; // per lane: the smallest key this lane has seen (limited mode)
    bool kept_sorted = false; // kept[0 .. cnt) is in ascending order (straight from the sorting network of the first block)
    int total_scanned = 0;
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
    // one round of the scan: the group's lane `sub` holds candidate `cand` (valid or not)
    auto consider = [&](bool valid, const float4 cand) {
        bool adm = false;
        u64 key = 0;
        if (valid) {
            const float d = ogc_sqdist(qx, qy, qz, cand.x, cand.y, cand.z);
            if (d < INFINITY) { // NaN / inf are never selected
                key = ((u64)__float_as_uint(d) << 32) | (unsigned)__float_as_int(cand.w);
                adm = cnt < k || key < maxkey;
                if (limited) {
                    best_any = key < best_any ? key : best_any;
                    adm = adm && d <= lim2;
                }
            }
        }
        const u64 ball = __builtin_amdgcn_ballot_w64(adm);
        if (ball == 0) return;
        const unsigned slice = (unsigned)(ball >> (qi * SUB)) & (SUB == 32 ? 0xFFFFFFFFu : ((1u << (SUB & 31)) - 1u));
        if (slice == 0) return;
        const int nh = __popc(slice);
        kept_sorted = false;
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
    };
    // scan the run [j0, j1) of the cell-sorted arrays with the 8 lanes of the group
    auto scan_run = [&](int j0, int j1) {
        for (int j = j0 + sub; __builtin_amdgcn_ballot_w64(j < j1) != 0; j += SUB) {
            float4 cand = make_float4(NAN, NAN, NAN, 0.f);
            if (j < j1) cand = pts[j];
            consider(j < j1, cand);
        }
    };

    const bool active = p < n && h.npts > 0 && qx == qx && qy == qy && qz == qz; // NaN queries select nothing
    if (active) {
        const float edge = 1.0f / h.inv_h;
        OGC_GRID_AXES(h, qx, qy, qz, fx, fy, fz);
        const int cx = min(max(cell_coord(fx, h.minx, h.inv_h, h.gx), 0), h.gx - 1);
        const int cy = min(max(cell_coord(fy, h.miny, h.inv_h, h.gy), 0), h.gy - 1);
        const int cz = min(max(cell_coord(fz, h.minz, h.inv_h, h.gz), 0), h.gz - 1);
        const int rmax = max(max(max(cx, h.gx - 1 - cx), max(cy, h.gy - 1 - cy)), max(cz, h.gz - 1 - cz));
        const int R0 = limited ? 1 : 2; // radius (in cells) of the block scanned first
        for (int R = R0;; ++R) {
            const int xa = max(cx - R, 0), xb = min(cx + R, h.gx - 1);
            // Shell R as ONE flat list of record positions (LDS), scanned eight candidates at a time with the next load in
            // flight.  A shell is (2R + 1)^2 rows of cells; a face row (or any row of the first block) contributes its
            // whole x-extent as one run of the cell-sorted array, an inner row its two end cells.  Walking the runs one
            // after the other is a dependent (bounds -> records) round trip per run with most of the eight lanes idle (a
            // run holds a handful of points); here the lanes fetch the bounds of all runs (two passes: lengths, then
            // positions), and the records are then read back to back.
            bool done_flat = false;
            {
                const int side = 2 * R + 1, nrows = side * side;
                const float inv_side = 1.0f / (float)side;
                auto row_runs = [&](int r, int &s0, int &l0, int &s1, int &l1) {
                    s0 = l0 = s1 = l1 = 0;
                    const int rz = (int)(((float)r + 0.5f) * inv_side); // r / side without an integer division (r < 2^12)
                    const int z = cz + rz - R, y = cy + (r - rz * side) - R;
                    if (r >= nrows || z < 0 || z >= h.gz || y < 0 || y >= h.gy) return;
                    const int rowc = h.gx * (y + h.gy * z);
                    const bool face = R == R0 || z == cz - R || z == cz + R || y == cy - R || y == cy + R;
                    if (face) {
                        s0 = cs[rowc + xa];
                        l0 = cs[rowc + xb + 1] - s0;
                    } else {
                        if (cx - R >= 0) { s0 = cs[rowc + cx - R]; l0 = cs[rowc + cx - R + 1] - s0; }
                        if (cx + R <= h.gx - 1) { s1 = cs[rowc + cx + R]; l1 = cs[rowc + cx + R + 1] - s1; }
                    }
                };
                int mine_total = 0;
                for (int r = sub; r < nrows; r += SUB) {
                    int s0, l0, s1, l1;
                    row_runs(r, s0, l0, s1, l1);
                    mine_total += l0 + l1;
                }
                int incl = mine_total;
#pragma unroll
                for (int off = 1; off < SUB; off <<= 1) {
                    const int up = __shfl_up(incl, off, SUB);
                    if (sub >= off) incl += up;
                }
                const int total = __shfl(incl, qi * SUB + SUB - 1, 64);
                if (total <= KNN_FLAT_CAP) {
                    int *mine = flat + qi * KNN_FLAT_CAP;
                    int w = incl - mine_total;
                    for (int r = sub; r < nrows; r += SUB) {
                        int s0, l0, s1, l1;
                        row_runs(r, s0, l0, s1, l1);
                        for (int i = 0; i < l0; ++i) mine[w + i] = s0 + i;
                        w += l0;
                        for (int i = 0; i < l1; ++i) mine[w + i] = s1 + i;
                        w += l1;
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    if (!limited && R == R0 && total <= 128) {
                        // plain k-NN, first block: select by sorting instead of insert-and-rescan (later shells admit few
                        // candidates — the kept maximum filters them — and a merge network per shell measured slower)
                        cnt = knn_first_block<128 / SUB, SUB>(pts, mine, total, qx, qy, qz, sub, k, kept, 0);
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_wave_barrier();
                        kept_sorted = true;
                        if (cnt == k) { maxkey = kept[k - 1]; maxpos = k - 1; }
                        total_scanned = -1; // (marks: nothing left for the loop below)
                    }
                    const float4 nothing = make_float4(NAN, NAN, NAN, 0.f);
                    int f = total_scanned < 0 ? total : sub;
                    total_scanned = 0;
                    float4 cur = nothing;
                    if (f < total) cur = pts[mine[f]];
                    while (__builtin_amdgcn_ballot_w64(f < total) != 0) {
                        const int fn = f + SUB;
                        float4 nxt = nothing;
                        if (fn < total) nxt = pts[mine[fn]];
                        consider(f < total, cur);
                        cur = nxt;
                        f = fn;
                    }
                    __builtin_amdgcn_wave_barrier(); // the list is rewritten by the next shell
                    done_flat = true;
                }
            }
            if (!done_flat) {
                for (int z = max(cz - R, 0); z <= min(cz + R, h.gz - 1); ++z)
                    for (int y = max(cy - R, 0); y <= min(cy + R, h.gy - 1); ++y) {
                        const int rowc = h.gx * (y + h.gy * z);
                        const bool face = R == R0 || z == cz - R || z == cz + R || y == cy - R || y == cy + R;
                        if (face) { // the whole x-extent of this row belongs to shell R (for R = R0: the full first block)
                            scan_run(cs[rowc + xa], cs[rowc + xb + 1]);
                        } else {    // inner row: only the two end cells are new
                            if (cx - R >= 0) scan_run(cs[rowc + cx - R], cs[rowc + cx - R + 1]);
                            if (cx + R <= h.gx - 1) scan_run(cs[rowc + cx + R], cs[rowc + cx + R + 1]);
                        }
                    }
            }
            if (R >= rmax) break; // the block covers the grid
            const float cover = (float)R * edge * 0.999f;
            if (cnt == k) {
                if (__uint_as_float((unsigned)(maxkey >> 32)) < cover * cover) break;
            }
            if (limited && cover >= radius) {
                // every point within the radius has been seen (unseen points are farther than R * edge >= 1.001 r).
                // Entry 0 must still be the true nearest neighbour: stop only when the nearest seen candidate lies
                // inside the covered ball (always the case when somebody is within the radius).
                u64 best = best_any;
#pragma unroll
                for (int off = 1; off < SUB; off <<= 1) {
                    const u64 ob = shfl_xor_u64(best, off);
                    best = ob < best ? ob : best;
                }
                if (best != ~0ull && __uint_as_float((unsigned)(best >> 32)) < cover * cover) break;
            }
        }
    }
    // rank sort (keys are distinct: the index is part of the key) — unless the kept set is still the sorted output of
    // the first block's network
    if (kept_sorted) {
        outk = kept;
    } else {
        for (int e = sub; e < cnt; e += SUB) {
            const u64 ve = kept[e];
            int rank = 0;
            for (int f = 0; f < cnt; ++f) rank += kept[f] < ve ? 1 : 0;
            outk[rank] = ve;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (p < n) {
        const size_t base = ((size_t)b * n + p) * k;
        int first = cnt > 0 ? (int)(unsigned)outk[0] : 0;
        if (limited && cnt == 0) { // nobody within the radius: every entry is the nearest neighbour of all
            u64 best = best_any;
#pragma unroll
            for (int off = 1; off < SUB; off <<= 1) {
                const u64 ob = shfl_xor_u64(best, off);
                best = ob < best ? ob : best;
            }
            first = best != ~0ull ? (int)(unsigned)best : 0;
        }
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
                if (d > radius && radius >= 0.0f) { id = first; d = INFINITY; } // clamped entries carry dist = +inf
            }
            dist_out[base + j] = d;
            idx_out[base + j] = id;
        }
    }
}

// ---- plain k-NN (k <= 32) with the WHOLE WAVEFRONT on one query at a time --------------------------------------------------
// knn_grid_kernel gives a query eight lanes: the candidates of its first block (5^3 cells, ~120 points) are sorted by a 128-key
// network of 64-bit compare-exchanges, five instructions each, and a query whose k-th neighbour lies outside that block — about
// half of them: the cells are half the EXPECTED k-th distance — walks a second shell by insertion while the other seven queries of
// the wavefront wait: ~5400 vector instructions per eight queries, 0.30 of the issue peak at 16 x 8192 x 8192, k = 32.
// Selecting k of a few hundred candidates does not need them sorted.  Here a wavefront takes its queries one after the other:
//   * cells of ~2 points (k / 16 per cell), block = 5 x 5 x 5 cells = 25 runs of the cell-sorted array, ~250 candidates covering
//     1.28 x the expected k-th distance; lanes 0 .. 24 fetch the runs' bounds, a wave scan numbers the candidates, the lanes
//     write their runs' positions into a flat LDS list and every lane then loads up to four candidates: one round trip each;
//   * a THRESHOLD on the squared distance is moved until between k and 64 candidates lie at or below it: each trial is four
//     compares whose masks are counted by the scalar unit; the first guess comes from the cell edge (the density), the next ones
//     from count ~ T^(3/2) — two or three trials;
//   * those <= 64 candidates are compacted into one (distance, index) key per lane and sorted by a 21-stage bitonic network
//     ACROSS THE LANES (one 64-bit compare-exchange per lane and stage); lanes 0 .. k - 1 then hold the row, in order, and
//     store it as two coalesced pieces.
// Exact by the argument of knn_grid_kernel: everything outside the selection is farther than everything inside, keys are
// distinct, and the row is accepted only when the k-th distance lies inside the ball the block is known to cover (or the block
// covers the grid).  A query whose block does not (sparse regions, more than 256 candidates, more than 64 ties at the
// threshold) is marked idx[row][0] = -1 for knn_grid_kernel, launched afterwards in `deferred == 2` mode.
constexpr int KW_PER_LANE = 12;                      // candidates per lane, at most
constexpr int KW_CAND = KW_PER_LANE * OGC_WAVE;      // per query and block
// (cells: grid_header with knn_div = -(candidates wanted in the clipped block) = -min(7 k, 230))

__device__ __forceinline__ u64 kw_xor_lane(u64 v, int d) {
    const unsigned lo = __shfl_xor((unsigned)v, d, 64), hi = __shfl_xor((unsigned)(v >> 32), d, 64);
    return ((u64)hi << 32) | lo;
}

template <int MODE>
__global__ __launch_bounds__(OGC_WAVE, 8) void knn_wave_kernel(int n, int m, int k, float radius, int stride_cells, int qpw,
                                                               const float *__restrict__ unknown, GridHdr *__restrict__ hdrs,
                                                               const int *__restrict__ cell_start,
                                                               const float4 *__restrict__ sorted_pts,
                                                               float *__restrict__ dist_out, int *__restrict__ idx_out) {
    __shared__ int flat[KW_CAND];
    __shared__ u64 slots[OGC_WAVE];
    const int lane = threadIdx.x, b = blockIdx.y;
    const GridHdr h = hdrs[b];
    const int *cs = cell_start + (size_t)b * stride_cells;
    const float4 *pts = sorted_pts + (size_t)b * m;
    const float edge = 1.0f / h.inv_h;
    // first threshold: the block (five cells per axis) was sized for ~7 k candidates; a ball holding `want` ~ 1.45 k of them at
    // that density has the volume fraction want / (7 k) of the 125-cell cube (the count model of the trials below corrects it)
    const float want = fminf(1.45f * (float)k, 48.0f);
    const float rk = edge * cbrtf(125.0f * want / (4.18879f * fminf(7.0f * (float)k, 230.0f)));
    const float t_first = rk * rk;
    // row of the block a lane fetches the bounds of: (2R + 1)^2 rows, R = 2 (25 lanes) and R = 3 (49 lanes)
    const int ry2 = lane % 5, rz2 = lane / 5, ry3 = lane % 7, rz3 = lane / 7;
    bool any_left = false;
    // the wavefront's queries (qpw <= 8), one per lane: coordinates and the cell of the projection into the grid, computed once
    // side by side instead of once per query on every lane
    float mqx = NAN, mqy = NAN, mqz = NAN;
    {
        const int pl = blockIdx.x * qpw + lane;
        if (lane < qpw && pl < n) {
            const float *u = unknown + ((size_t)b * n + pl) * 3;
            mqx = u[0]; mqy = u[1]; mqz = u[2];
        }
    }
    int mcx, mcy, mcz;
    {
        OGC_GRID_AXES(h, mqx, mqy, mqz, fx, fy, fz);
        mcx = min(max(cell_coord(fx, h.minx, h.inv_h, h.gx), 0), h.gx - 1);
        mcy = min(max(cell_coord(fy, h.miny, h.inv_h, h.gy), 0), h.gy - 1);
        mcz = min(max(cell_coord(fz, h.minz, h.inv_h, h.gz), 0), h.gz - 1);
    }
    for (int qn = 0; qn < qpw; ++qn) {
        const int p = (blockIdx.x * qpw + qn);
        if (p >= n) break;
        const float qx = lane_bcast(mqx, qn), qy = lane_bcast(mqy, qn), qz = lane_bcast(mqz, qn);
        const size_t base = ((size_t)b * n + p) * k;
        const bool active = h.npts > 0 && qx == qx && qy == qy && qz == qz; // NaN queries select nothing
        int cnt = 0;
        u64 key = ~0ull;
        bool accept = true;
        if (active) {
            const int cx = lane_bcast(mcx, qn), cy = lane_bcast(mcy, qn), cz = lane_bcast(mcz, qn);
            const int rmax = max(max(max(cx, h.gx - 1 - cx), max(cy, h.gy - 1 - cy)), max(cz, h.gz - 1 - cz));
            float T = t_first;
            // the block of (2R + 1)^3 cells, R = 2; a query whose k-th neighbour is not inside the ball that block covers tries
            // R = 3 (the WHOLE block again: the wavefront re-reads ~250 records it had, instead of carrying them)
            for (int R = 2; R <= 3; ++R) {
                accept = true;
                const int xa = max(cx - R, 0), xb = min(cx + R, h.gx - 1);
                const int side = 2 * R + 1;
                int s_r = 0, l_r = 0;
                {
                    const int y = cy + (R == 2 ? ry2 : ry3) - R, z = cz + (R == 2 ? rz2 : rz3) - R;
                    if (lane < side * side && y >= 0 && y < h.gy && z >= 0 && z < h.gz) {
                        const int rowc = h.gx * (y + h.gy * z);
                        s_r = cs[rowc + xa];
                        l_r = cs[rowc + xb + 1] - s_r;
                    }
                }
                // inclusive scan over the wavefront in six DPP steps (row shifts, then the row broadcasts)
                int incl = l_r;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, true);   // row_bcast:15 into rows 1, 3
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, true);   // row_bcast:31 into rows 2, 3
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (total > KW_CAND) { accept = false; break; }     // a crowded block: left to knn_grid_kernel
                const int w = incl - l_r;
                for (int i = 0; __builtin_amdgcn_ballot_w64(i < l_r) != 0ull; ++i)
                    if (i < l_r) flat[w + i] = s_r + i;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int nch = (total + OGC_WAVE - 1) / OGC_WAVE;  // candidates per lane (wave-uniform)
                float d[KW_PER_LANE];
                int valid = 0;
#pragma unroll
                for (int j = 0; j < KW_PER_LANE; ++j) {
                    d[j] = INFINITY;
                    if (j < nch) {
                        const int f = lane + OGC_WAVE * j;
                        float4 rec = make_float4(NAN, NAN, NAN, 0.f);
                        if (f < total) {
                            rec = pts[flat[f]];
                            flat[f] = __float_as_int(rec.w);     // (only this lane reads slot f: the list now holds the point's index)
                        }
                        const float dj = ogc_sqdist(qx, qy, qz, rec.x, rec.y, rec.z);
                        const bool ok = dj < INFINITY;        // NaN / inf are never selected (empty slots hold NaN)
                        d[j] = ok ? dj : INFINITY;
                        valid += __popcll(__builtin_amdgcn_ballot_w64(ok));
                    }
                }
                // the threshold: between min(k, valid) and 64 candidates at or below it
                int below = 0;
                if (valid <= k) {
                    T = __int_as_float(0x7f7fffff);            // no more candidates than the row holds: all of them (every finite distance)
                    below = valid;
                } else {
                    float lo_t = 0.0f, hi_t = 3.0e38f;         // count(lo_t) < k, count(hi_t) > 64 (once tried)
                    bool found = false;
                    if (!(T < 3.0e38f)) T = t_first;
                    for (int it = 0; it < 24; ++it) {
                        int c = 0;
#pragma unroll
                        for (int j = 0; j < KW_PER_LANE; ++j)
                            if (j < nch) c += __popcll(__builtin_amdgcn_ballot_w64(d[j] <= T));
                        if (c >= k && c <= OGC_WAVE) { below = c; found = true; break; }
                        if (c < k) lo_t = T; else hi_t = T;
                        // next trial: count ~ T^(3/2), kept strictly inside the bracket; bisection once the model stalls
                        float next = T * __powf(want / fmaxf((float)c, 0.5f), 2.0f / 3.0f);
                        if (it >= 6 || !(next > lo_t) || !(next < hi_t)) next = hi_t < 3.0e38f ? 0.5f * (lo_t + hi_t) : 2.0f * fmaxf(T, 1.0e-30f);
                        if (!(next > lo_t) || !(next < hi_t)) break; // the bracket has no float left: ties
                        T = next;
                    }
                    if (!found) { accept = false; break; }      // more than 64 - k ties at the k-th distance: knn_grid_kernel
                }
                int slot_base = 0;
#pragma unroll
                for (int j = 0; j < KW_PER_LANE; ++j) {
                    if (j < nch) {
                        const bool sel = d[j] <= T;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(sel);
                        if (mask != 0ull) {
                            const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)slot_base));
                            if (sel) slots[slot] = ((u64)__float_as_uint(d[j]) << 32) | (unsigned)flat[lane + OGC_WAVE * j];
                            slot_base += __popcll(mask);
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                key = lane < below ? slots[lane] : ~0ull;
                // bitonic network over the 64 lanes, ascending: per stage ONE 64-bit compare; which lanes keep the smaller key is a
                // constant of the stage (ascending block == lower lane of the pair), so "take the partner's key" is the compare's
                // mask xor that constant — a scalar instruction — fed to the two selects (the keys are distinct, sentinels apart,
                // which may go either way); partners by ds_swizzle (xor 1 .. 16) or a lane permutation (xor 32)
#pragma unroll
                for (int size = 2; size <= OGC_WAVE; size <<= 1) {
#pragma unroll
                    for (int dd = size >> 1; dd >= 1; dd >>= 1) {
                        const u64 other = dd == 1 ? knn_xor_lane<1>(key) : dd == 2 ? knn_xor_lane<2>(key) : dd == 4 ? knn_xor_lane<4>(key)
                                        : dd == 8 ? knn_xor_lane<8>(key) : dd == 16 ? knn_xor_lane<16>(key) : kw_xor_lane(key, 32);
                        u64 keep_max = 0ull; // lanes that keep the LARGER key in this stage (compile-time constant)
#pragma unroll
                        for (int l = 0; l < OGC_WAVE; ++l)
                            if ((((l & size) == 0) || size == OGC_WAVE) != ((l & dd) == 0)) keep_max |= 1ull << l;
                        const u64 take = __builtin_amdgcn_ballot_w64(other < key) ^ keep_max;
                        unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
                        asm("v_cndmask_b32 %0, %0, %2, %4\n\tv_cndmask_b32 %1, %1, %3, %4"
                            : "+v"(lo), "+v"(hi) : "v"((unsigned)other), "v"((unsigned)(other >> 32)), "s"(take));
                        key = ((u64)hi << 32) | lo;
                    }
                }
                cnt = min(below, k);
                __builtin_amdgcn_wave_barrier(); // (flat / slots are rewritten by the next block or query)
                // the k-th neighbour must lie inside the ball the block covers — unless the block is the whole grid
                const float cover = (float)R * edge * 0.999f;
                const unsigned kth_hi = (unsigned)(__shfl(key, max(cnt - 1, 0), 64) >> 32);
                if (rmax <= R || (cnt == k && __uint_as_float(kth_hi) < cover * cover)) break;
                accept = false;                                 // (R = 3 did not cover it either: knn_grid_kernel's shells)
            }
        }
        if (!accept) {
            if (lane == 0) idx_out[base] = -1;
            any_left = true;
            continue;
        }
        const int first = cnt > 0 ? (int)(unsigned)__shfl(key, 0, 64) : 0;
        if (lane < k) {
            float dv = INFINITY;
            int iv = 0;
            if (lane < cnt) {
                dv = __uint_as_float((unsigned)(key >> 32));
                iv = (int)(unsigned)key;
            }
            if (MODE == 1) {
                dv = sqrtf(dv);
                if (dv > radius && radius >= 0.0f) { iv = first; dv = INFINITY; } // clamped entries carry dist = +inf
            }
            dist_out[base + lane] = dv;
            idx_out[base + lane] = iv;
        }
    }
    if (any_left && lane == 0) hdrs[b].pending = 1;
}


// ---- three nearest neighbours over the cell lists: ONE LANE PER QUERY ------------------------------------------------------------
// three_nn (interpolate_gpu.cu:81-124: the feature-propagation modules' inverse-distance weights, 8192 targets against the 2048
// centres of the level above) as an all-pairs scan tests every target against every centre.  Here a lane takes one target and
// walks the cells around it shell by shell — rows of the (2R + 1)^3 block, a row's x-extent being one run of the cell-sorted
// array — keeping the three smallest (distance, index) keys in registers; it stops when the third distance lies inside the
// ball the scanned block is known to cover (R h, as knn_grid_kernel), or the block covers the grid.  The reference keeps
// the EARLIER index among equal distances (strict '<' while scanning in index order): the smallest keys, whatever the order
// in which candidates are met.  The grid holds ~1.5 points per cell, so the first block (27 cells, ~40 candidates against
// 2048) ends ~95 % of the searches.  Lanes of a wavefront are unrelated targets: every loop runs to its longest lane.
__global__ __launch_bounds__(OGC_WAVE, 8) void three_nn_grid_kernel(int n, int m, int stride_cells, const float *__restrict__ unknown,
                                                                    const GridHdr *__restrict__ hdrs,
                                                                    const int *__restrict__ cell_start,
                                                                    const float4 *__restrict__ sorted_pts,
                                                                    float *__restrict__ dist2, int *__restrict__ idx) {
    const int lane = threadIdx.x, b = blockIdx.y, q = blockIdx.x * OGC_WAVE + lane;
    const GridHdr h = hdrs[b];
    const int *cs = cell_start + (size_t)b * stride_cells;
    const float4 *pts = sorted_pts + (size_t)b * m;
    float qx = NAN, qy = NAN, qz = NAN;
    if (q < n) {
        const float *u = unknown + ((size_t)b * n + q) * 3;
        qx = u[0]; qy = u[1]; qz = u[2];
    }
    const u64 none = (u64)0x7f800000u << 32; // (+inf, index 0): what the reference's rows hold where nothing was found
    u64 k1 = none, k2 = none, k3 = none;
    const float edge = 1.0f / h.inv_h;
    OGC_GRID_AXES(h, qx, qy, qz, fx, fy, fz);
    const int cx = min(max(cell_coord(fx, h.minx, h.inv_h, h.gx), 0), h.gx - 1);
    const int cy = min(max(cell_coord(fy, h.miny, h.inv_h, h.gy), 0), h.gy - 1);
    const int cz = min(max(cell_coord(fz, h.minz, h.inv_h, h.gz), 0), h.gz - 1);
    const int rmax = max(max(max(cx, h.gx - 1 - cx), max(cy, h.gy - 1 - cy)), max(cz, h.gz - 1 - cz));
    bool open = q < n && h.npts > 0 && qx == qx && qy == qy && qz == qz; // (a NaN target selects nothing)
    auto scan_run = [&](int j0, int j1) {
        for (int j = j0; j < j1; ++j) {
            const float4 c = pts[j];
            const float d = ogc_sqdist(qx, qy, qz, c.x, c.y, c.z);
            const u64 key = ((u64)__float_as_uint(d) << 32) | (unsigned)__float_as_int(c.w);
            const bool c1 = key < k1, c2 = key < k2, c3 = key < k3;
            k3 = c2 ? k2 : (c3 ? key : k3);
            k2 = c1 ? k1 : (c2 ? key : k2);
            k1 = c1 ? key : k1;
        }
    };
    for (int R = 1; __builtin_amdgcn_ballot_w64(open) != 0ull; ++R) {
        if (open) {
            const int xa = max(cx - R, 0), xb = min(cx + R, h.gx - 1);
            for (int z = max(cz - R, 0); z <= min(cz + R, h.gz - 1); ++z)
                for (int y = max(cy - R, 0); y <= min(cy + R, h.gy - 1); ++y) {
                    const int rowc = h.gx * (y + h.gy * z);
                    const bool face = R == 1 || z == cz - R || z == cz + R || y == cy - R || y == cy + R;
                    if (face) { // the row's whole x-extent belongs to shell R (R = 1: the full first block)
                        scan_run(cs[rowc + xa], cs[rowc + xb + 1]);
                    } else {    // inner row: only the two end cells are new
                        if (cx - R >= 0) scan_run(cs[rowc + cx - R], cs[rowc + cx - R + 1]);
                        if (cx + R <= h.gx - 1) scan_run(cs[rowc + cx + R], cs[rowc + cx + R + 1]);
                    }
                }
            const float cover = (float)R * edge * 0.999f;
            if (R >= rmax || __uint_as_float((unsigned)(k3 >> 32)) < cover * cover) open = false;
        }
    }
    if (q < n) {
        float *o = dist2 + ((size_t)b * n + q) * 3;
        int *oi = idx + ((size_t)b * n + q) * 3;
        o[0] = __uint_as_float((unsigned)(k1 >> 32)); o[1] = __uint_as_float((unsigned)(k2 >> 32)); o[2] = __uint_as_float((unsigned)(k3 >> 32));
        oi[0] = (int)(unsigned)k1; oi[1] = (int)(unsigned)k2; oi[2] = (int)(unsigned)k3;
    }
}

// ---- radius-limited k-NN of a cloud in itself with FOUR lanes per query ---------------------------------------------------
// ogc_knn_clamped(pc, pc) with a radius (the smoothness term's lists, losses/seg_loss_unsup.py:150: k = 32 within 1 m — about 3
// of the ~10 candidates the 27 cells hold): a neighbour beyond the radius is replaced by the nearest one whatever it is, so the
// row is "the points within the radius, ascending by (distance, index), first K of them, the rest = the first".  That is the
// ball query's traversal with another sort key: the structure of ball_query_cells_kernel — sixteen queries (= points, in cell
// order) per wavefront, four lanes each walking the query's nine runs, slots from the ballots' bits, lists of up to 32 keys
// sorted in registers (64-bit keys here) and stored straight from the registers.  knn_grid_kernel, launched after it in
// `deferred` mode, does what is left: rows marked idx[row][0] = -1 (more than 32 points within the radius) and whole clouds
// the build flagged knn_general (cells shorter than the radius, crowded cells).  Same results as knn_grid_kernel alone.
constexpr int KQ_LIST = BQ_FAST + 4; // keys per list (slot BQ_FAST takes the misses)

template <int K>
__global__ __launch_bounds__(OGC_WAVE, 8) void knn_cells_kernel(int n, float lim2, int stride_cells, GridHdr *__restrict__ hdrs,
                                                                const int *__restrict__ cell_start,
                                                                const float4 *__restrict__ sorted_pts,
                                                                float *__restrict__ dist_out, int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) u64 kc_smem[];
    const int lane = threadIdx.x, b = blockIdx.y, sub = lane & (CL - 1), g = lane >> 2;
    const GridHdr h = hdrs[b];
    if (h.knn_general) return;
    const int *cs = cell_start + (size_t)b * stride_cells;
    const float4 *pts = sorted_pts + (size_t)b * n;
    const int pc = blockIdx.x * CPW + g;
    float4 me = make_float4(NAN, NAN, NAN, __int_as_float(-1));
    if (pc < n) me = pts[pc]; // positions >= h.npts hold the non-finite points: nobody within the radius
    const bool live = pc < h.npts;
    u64 *mine = kc_smem + g * KQ_LIST;
    {   // every list starts as BQ_FAST +inf keys: the sort reads all of them
        const int4 inf4 = make_int4(-1, -1, -1, -1);
        int4 *l4 = reinterpret_cast<int4 *>(mine + sub * (BQ_FAST / CL));
#pragma unroll
        for (int i = 0; i < BQ_FAST / CL / 2; ++i) l4[i] = inf4;
    }
    OGC_GRID_AXES(h, me.x, me.y, me.z, gfx, gfy, gfz);
    const int cx = min(cell_floor(gfx, h.minx, h.inv_h), h.gx - 1);
    const int cy = min(cell_floor(gfy, h.miny, h.inv_h), h.gy - 1);
    const int cz = min(cell_floor(gfz, h.minz, h.inv_h), h.gz - 1);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.gx - 1);
    const bool slab = h.slab != 0; // (wave-uniform; see ball_query_cells_kernel)
    auto row_of = [&](int r, bool &inside) {
        const int r3 = r / 3;
        const int y = cy + (r - 3 * r3) - 1, z = cz + r3 - 1;
        inside = live && y >= 0 && y < h.gy && z >= 0 && z < h.gz;
        return h.gx * (min(max(y, 0), h.gy - 1) + h.gy * min(max(z, 0), h.gz - 1));
    };
    bool in_a, in_b, in_c;
    int row_a = row_of(sub, in_a), row_b = row_of(sub + 4, in_b), row_c = row_of(8, in_c);
    int first_a = row_a + x0, last_a = row_a + x1 + 1;
    if (slab) {
        const int z = cz + sub - 1;
        in_a = live && sub < 3 && z >= 0 && z < h.gz;
        const int zc = min(max(z, 0), h.gz - 1);
        first_a = h.gx * (max(cy - 1, 0) + h.gy * zc);
        last_a = h.gx * (min(cy + 1, h.gy - 1) + h.gy * zc) + h.gx;
        row_b = row_c = first_a - x0;
        in_b = in_c = false;
    }
    int lo_a = cs[first_a], end_a = cs[last_a];
    int lo_b = cs[row_b + x0], end_b = cs[row_b + x1 + 1];
    int lo_c = cs[row_c + x0], end_c = cs[row_c + x1 + 1];
    asm volatile("" : "+v"(lo_a), "+v"(end_a), "+v"(lo_b), "+v"(end_b), "+v"(lo_c), "+v"(end_c));
    const int len_a = in_a ? end_a - lo_a : 0, len_b = in_b ? end_b - lo_b : 0, len_c = in_c ? end_c - lo_c : 0;

    int cnt = 0; // points within the radius of my query (the same number in its four lanes)
    const unsigned below_a = (1u << sub) - 1u, below_b = 0xFu | (below_a << 4);
    const int shift = CL * g;
    auto slots = [&](bool has_a, bool near_a, bool has_b, bool near_b, u64 ka, u64 kb) {
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(has_a) & __builtin_amdgcn_ballot_w64(near_a);
        const unsigned long long mb = __builtin_amdgcn_ballot_w64(has_b) & __builtin_amdgcn_ballot_w64(near_b);
        const unsigned bits = ((unsigned)(ma >> shift) & 0xFu) | (((unsigned)(mb >> shift) & 0xFu) << 4);
        const int sa = cnt + __popc(bits & below_a), sb = cnt + __popc(bits & below_b);
        mine[(has_a && near_a) ? min(sa, BQ_FAST) : BQ_FAST] = ka;
        mine[(has_b && near_b) ? min(sb, BQ_FAST) : BQ_FAST] = kb;
        cnt += __popc(bits);
    };
    const char *pts_bytes = reinterpret_cast<const char *>(pts);
    auto record = [&](int position) { // (positions past the end of a run are read — the array is padded — and discarded)
        return *reinterpret_cast<const float4 *>(pts_bytes + ((unsigned)position << 4));
    };
    auto key_of = [](float d, float w) { return ((u64)__float_as_uint(d) << 32) | (unsigned)__float_as_int(w); };
    if (slab) {
        int p0 = quad_bcast<0>(lo_a), p1 = quad_bcast<1>(lo_a), p2 = quad_bcast<2>(lo_a);
        const int hi0 = p0 + quad_bcast<0>(len_a), hi1 = p1 + quad_bcast<1>(len_a), hi2 = p2 + quad_bcast<2>(len_a);
        p0 += sub; p1 += sub; p2 += sub;
        const int last = n - 1;
        for (;;) {
            const float4 a0 = record(min(p0, last)), b0 = record(min(p0 + CL, last));
            const float4 a1 = record(min(p1, last)), b1 = record(min(p1 + CL, last));
            const float4 a2 = record(min(p2, last)), b2 = record(min(p2 + CL, last));
            __builtin_amdgcn_sched_barrier(0);
            const ogc_v2f d0 = sqdist_pair(ogc_v2f{a0.x, b0.x}, ogc_v2f{a0.y, b0.y}, ogc_v2f{a0.z, b0.z}, me.x, me.y, me.z);
            slots(p0 < hi0, d0.x <= lim2, p0 + CL < hi0, d0.y <= lim2, key_of(d0.x, a0.w), key_of(d0.y, b0.w));
            const ogc_v2f d1 = sqdist_pair(ogc_v2f{a1.x, b1.x}, ogc_v2f{a1.y, b1.y}, ogc_v2f{a1.z, b1.z}, me.x, me.y, me.z);
            slots(p1 < hi1, d1.x <= lim2, p1 + CL < hi1, d1.y <= lim2, key_of(d1.x, a1.w), key_of(d1.y, b1.w));
            const ogc_v2f d2 = sqdist_pair(ogc_v2f{a2.x, b2.x}, ogc_v2f{a2.y, b2.y}, ogc_v2f{a2.z, b2.z}, me.x, me.y, me.z);
            slots(p2 < hi2, d2.x <= lim2, p2 + CL < hi2, d2.y <= lim2, key_of(d2.x, a2.w), key_of(d2.y, b2.w));
            p0 += 2 * CL; p1 += 2 * CL; p2 += 2 * CL;
            if (__builtin_amdgcn_ballot_w64(p0 < hi0 || p1 < hi1 || p2 < hi2) == 0ull) break;
        }
    } else
#pragma unroll
    for (int r0 = 0; r0 < 9; r0 += 3) {
        int lo[3], hi[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int r = r0 + i;
            const int l = r == 0 ? quad_bcast<0>(lo_a) : r == 1 ? quad_bcast<1>(lo_a) : r == 2 ? quad_bcast<2>(lo_a)
                        : r == 3 ? quad_bcast<3>(lo_a) : r == 4 ? quad_bcast<0>(lo_b) : r == 5 ? quad_bcast<1>(lo_b)
                        : r == 6 ? quad_bcast<2>(lo_b) : r == 7 ? quad_bcast<3>(lo_b) : lo_c;
            const int w = r == 0 ? quad_bcast<0>(len_a) : r == 1 ? quad_bcast<1>(len_a) : r == 2 ? quad_bcast<2>(len_a)
                        : r == 3 ? quad_bcast<3>(len_a) : r == 4 ? quad_bcast<0>(len_b) : r == 5 ? quad_bcast<1>(len_b)
                        : r == 6 ? quad_bcast<2>(len_b) : r == 7 ? quad_bcast<3>(len_b) : len_c;
            lo[i] = l;
            hi[i] = l + w;
        }
        float4 ca[3], cb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ca[i] = record(lo[i] + sub);
            cb[i] = record(lo[i] + sub + CL);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // (query - candidate) squared, summed as (x + y) + z: the expression of knn_grid_kernel / the reference, per half
            const ogc_v2f d = sqdist_pair(ogc_v2f{ca[i].x, cb[i].x}, ogc_v2f{ca[i].y, cb[i].y}, ogc_v2f{ca[i].z, cb[i].z},
                                          me.x, me.y, me.z);
            const int p = lo[i] + sub;
            slots(p < hi[i], d.x <= lim2, p + CL < hi[i], d.y <= lim2, key_of(d.x, ca[i].w), key_of(d.y, cb[i].w));
            int pp = p + 2 * CL;
            while (__builtin_amdgcn_ballot_w64(pp < hi[i]) != 0ull) { // a run longer than eight candidates
                const float4 a = record(min(pp, n - 1)), c2 = record(min(pp + CL, n - 1));
                const ogc_v2f d2 = sqdist_pair(ogc_v2f{a.x, c2.x}, ogc_v2f{a.y, c2.y}, ogc_v2f{a.z, c2.z}, me.x, me.y, me.z);
                slots(pp < hi[i], d2.x <= lim2, pp + CL < hi[i], d2.y <= lim2, key_of(d2.x, a.w), key_of(d2.y, c2.w));
                pp += 2 * CL;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int q = __float_as_int(me.w);
    if (cnt > BQ_FAST) { // more keys than the register sort holds: the row goes to knn_grid_kernel
        if (sub == 0 && q >= 0) {
            idx_out[((size_t)b * n + q) * K] = -1;
            hdrs[b].pending = 1;
        }
    }
    u64 x[8];
    {
        const int4 *l4 = reinterpret_cast<const int4 *>(mine + sub * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int4 v = l4[i];
            x[2 * i] = ((u64)(unsigned)v.y << 32) | (unsigned)v.x;
            x[2 * i + 1] = ((u64)(unsigned)v.w << 32) | (unsigned)v.z;
        }
    }
    // bitonic network over 4 lanes x 8 keys, element e = 8 * lane + register, every exchange ascending (see ball_query_cells_kernel)
#define OGC_KQ_INTRA(MASK)                                                  \
    _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_)                        \
        if ((r_ ^ (MASK)) > r_) {                                           \
            const u64 a_ = x[r_], b_ = x[r_ ^ (MASK)];                      \
            x[r_] = a_ < b_ ? a_ : b_;                                      \
            x[r_ ^ (MASK)] = a_ < b_ ? b_ : a_;                             \
        }
#define OGC_KQ_INTER(QP, RMASK, UPPER)                                                                              \
    {                                                                                                               \
        u64 p_[8];                                                                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) {                                                          \
            const u64 v_ = x[r_ ^ (RMASK)];                                                                         \
            const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v_, QP, 0xF, 0xF, true);   \
            const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v_ >> 32), QP, 0xF, 0xF, true); \
            p_[r_] = ((u64)hi_ << 32) | lo_;                                                                        \
        }                                                                                                           \
        _Pragma("unroll") for (int r_ = 0; r_ < 8; ++r_) {                                                          \
            const bool mine_less_ = x[r_] < p_[r_];                                                                 \
            x[r_] = (mine_less_ != (UPPER)) ? x[r_] : p_[r_];                                                       \
        }                                                                                                           \
    }
    const bool odd = (sub & 1) != 0, high = (sub & 2) != 0;
    OGC_KQ_INTRA(1)
    OGC_KQ_INTRA(3) OGC_KQ_INTRA(1)
    OGC_KQ_INTRA(7) OGC_KQ_INTRA(2) OGC_KQ_INTRA(1)
    OGC_KQ_INTER(0xB1, 7, odd) OGC_KQ_INTRA(4) OGC_KQ_INTRA(2) OGC_KQ_INTRA(1)
    OGC_KQ_INTER(0x1B, 7, high) OGC_KQ_INTER(0xB1, 0, odd) OGC_KQ_INTRA(4) OGC_KQ_INTRA(2) OGC_KQ_INTRA(1)
#undef OGC_KQ_INTRA
#undef OGC_KQ_INTER
    const int kept = min(cnt, K);
    const int first = cnt > 0 ? quad_bcast<0>((int)(unsigned)x[0]) : 0;
    // entry j: (sqrt(d2), index) for j < kept, else (+inf, first).  Lane L holds entries 8 L .. 8 L + 7; it writes entries
    // 4 L .. 4 L + 3 and 16 + 4 L .. (64 contiguous bytes per row and store): an exchange inside the quad.
    int vi[8];
    float vd[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool real = sub * 8 + r < kept;
        vi[r] = real ? (int)(unsigned)x[r] : first;
        vd[r] = real ? sqrtf(__uint_as_float((unsigned)(x[r] >> 32))) : INFINITY;
    }
    int i1[4], i2[4];
    float d1[4], d2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ia1 = __builtin_amdgcn_update_dpp(0, vi[r], 0x50, 0xF, 0xF, true), ib1 = __builtin_amdgcn_update_dpp(0, vi[r + 4], 0x50, 0xF, 0xF, true);
        const int ia2 = __builtin_amdgcn_update_dpp(0, vi[r], 0xFA, 0xF, 0xF, true), ib2 = __builtin_amdgcn_update_dpp(0, vi[r + 4], 0xFA, 0xF, 0xF, true);
        const int da1 = __builtin_amdgcn_update_dpp(0, __float_as_int(vd[r]), 0x50, 0xF, 0xF, true);
        const int db1 = __builtin_amdgcn_update_dpp(0, __float_as_int(vd[r + 4]), 0x50, 0xF, 0xF, true);
        const int da2 = __builtin_amdgcn_update_dpp(0, __float_as_int(vd[r]), 0xFA, 0xF, 0xF, true);
        const int db2 = __builtin_amdgcn_update_dpp(0, __float_as_int(vd[r + 4]), 0xFA, 0xF, 0xF, true);
        i1[r] = odd ? ib1 : ia1;
        i2[r] = odd ? ib2 : ia2;
        d1[r] = __int_as_float(odd ? db1 : da1);
        d2[r] = __int_as_float(odd ? db2 : da2);
    }
    if (q >= 0 && cnt <= BQ_FAST) {
        const size_t base = ((size_t)b * n + q) * K;
        const int j0 = sub * 4;
        if (j0 < K) {
            *reinterpret_cast<int4 *>(idx_out + base + j0) = make_int4(i1[0], i1[1], i1[2], i1[3]);
            *reinterpret_cast<float4 *>(dist_out + base + j0) = make_float4(d1[0], d1[1], d1[2], d1[3]);
        }
        if (16 + j0 < K) {
            *reinterpret_cast<int4 *>(idx_out + base + 16 + j0) = make_int4(i2[0], i2[1], i2[2], i2[3]);
            *reinterpret_cast<float4 *>(dist_out + base + 16 + j0) = make_float4(d2[0], d2[1], d2[2], d2[3]);
        }
    }
}

} // namespace ogc_grid

using namespace ogc_grid;

// OGC_KNN_CELLS=0 in the environment: knn_grid_kernel alone (A/B runs, tests of both paths)
static bool ogc_knn_cells_enabled() {
    const char *e = getenv("OGC_KNN_CELLS");
    return !(e && e[0] == '0');
}

// OGC_BQ_CELLS=0 in the environment: the general kernel for every row length (A/B runs, tests of both kernels)
static bool ogc_bq_cells_enabled() {
    const char *e = getenv("OGC_BQ_CELLS");
    return !(e && e[0] == '0');
}

namespace {
struct GridLayout { // one buffer: headers | cell starts | cell-sorted records (+ BQ_PAD readable records behind them)
    size_t bytes_hdr, bytes_cs, bytes_pts;
    GridLayout(int b, int n)
        : bytes_hdr((sizeof(GridHdr) * b + 255) / 256 * 256),
          bytes_cs((sizeof(int) * (size_t)b * (GRID_MAX_CELLS + 1) + 255) / 256 * 256),
          bytes_pts(sizeof(float4) * ((size_t)b * n + BQ_PAD)) {}
    size_t total() const { return bytes_hdr + bytes_cs + bytes_pts; }
    GridHdr *hdrs(void *p) const { return reinterpret_cast<GridHdr *>(p); }
    int *cell_start(void *p) const { return reinterpret_cast<int *>(static_cast<char *>(p) + bytes_hdr); }
    float4 *sorted_pts(void *p) const { return reinterpret_cast<float4 *>(static_cast<char *>(p) + bytes_hdr + bytes_cs); }
};
constexpr int STRIDE_CELLS = GRID_MAX_CELLS + 1;

// the query kernels of ogc_ball_query on a built grid (four lanes per centre for the usual row lengths; the general kernel —
// eight centres per wavefront — otherwise)
int launch_ball_query(const GridLayout &L, void *grid, int b, int n, int m, float radius, int nsample, const float *xyz, int *idx,
                      hipStream_t s) {
    GridHdr *hdrs = L.hdrs(grid);
    int *cell_start = L.cell_start(grid);
    float4 *sorted_pts = L.sorted_pts(grid);
    const int stride_cells = STRIDE_CELLS;
    // hit slots per centre: the smallest list that holds a full row keeps the LDS footprint at ~5 KiB per wavefront,
    // i.e. the full eight wavefronts per SIMD; a centre with more hits takes the bitmap path
    const int hit_cap = nsample > 64 ? nsample : 64;
    const size_t lds = ((size_t)QPW * (hit_cap + nsample) + (size_t)(n + 31) / 32) * sizeof(int);
    const size_t lds_body = ((size_t)QPW * (BQ_CAP + nsample) + (size_t)(n + 31) / 32) * sizeof(int);
    const size_t lds_cells = sizeof(int) * CPW * BQ_LIST;
    const size_t lds4 = lds_body > lds_cells ? lds_body : lds_cells;
    // waves per workgroup of the four-lane kernel (a workgroup is only its unit of dispatch): OGC_BQ_WPB = 1 | 2 | 4
    static const int wpb = [] { const char *e = getenv("OGC_BQ_WPB"); const int v = e ? atoi(e) : 1; return v == 2 || v == 4 ? v : 1; }();
    const int lds4_ints = (int)((lds4 + 15) / 16 * 4);
    const dim3 grid4(ogc_divup(ogc_divup(n, CPW), wpb), b);
#define OGC_BQ_CELLS_W(NS, W)                                                                                         \
    hipLaunchKernelGGL((ball_query_cells_kernel<NS, W>), grid4, dim3(OGC_WAVE * W), (size_t)lds4_ints * 4 * W, s, n, m, \
                       radius * radius, stride_cells, lds4_ints, xyz, hdrs, cell_start, sorted_pts, idx)
#define OGC_BQ_CELLS(NS)                                                                                              \
    {                                                                                                                 \
        if (wpb == 4) OGC_BQ_CELLS_W(NS, 4);                                                                          \
        else if (wpb == 2) OGC_BQ_CELLS_W(NS, 2);                                                                     \
        else OGC_BQ_CELLS_W(NS, 1);                                                                                   \
    }
    const bool cells = ogc_bq_cells_enabled();
    if (nsample == 64 && cells) OGC_BQ_CELLS(64)
    else if (nsample == 32 && cells) OGC_BQ_CELLS(32)
    else if (nsample == 16 && cells) OGC_BQ_CELLS(16)
    else
        hipLaunchKernelGGL(ball_query_grid_kernel, dim3(ogc_divup(n, QPW), b), dim3(OGC_WAVE), lds, s, n, m,
                           radius * radius, nsample, hit_cap, stride_cells, xyz, hdrs, cell_start, sorted_pts, idx);
#undef OGC_BQ_CELLS_W
#undef OGC_BQ_CELLS
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ogc_set_error("ogc_ball_query (grid): launch failed: %s", hipGetErrorString(e));
        return OGC_ERR_LAUNCH;
    }
    return OGC_OK;
}

bool ball_query_grid_applies(int n, int nsample, float radius) {
    const int hit_cap = nsample > 64 ? nsample : 64;
    const size_t lds = ((size_t)QPW * (hit_cap + nsample) + (size_t)(n + 31) / 32) * sizeof(int);
    return n >= 1024 && lds <= 64 * 1024 && radius > 0.0f && radius < 3.0e38f;
}

// d2 <= lim2  <=>  sqrtf(d2) <= radius: the largest float whose correctly rounded root does not exceed the radius
float knn_radius_limit2(int mode, float radius) {
    float lim2 = INFINITY;
    if (mode == 1 && radius >= 0.0f) {
        lim2 = radius * radius;
        while (lim2 > 0.0f && sqrtf(lim2) > radius) lim2 = nextafterf(lim2, 0.0f);
        for (int it = 0; it < 4; ++it) {
            const float up = nextafterf(lim2, INFINITY);
            if (up < INFINITY && sqrtf(up) <= radius) lim2 = up;
        }
    }
    return lim2;
}

// the query kernels of ogc_knn / ogc_knn_clamped on a built grid.  cells: four lanes per query over the 27 cells around it first;
// knn_grid_kernel afterwards only does what that kernel left (marked rows, clouds flagged knn_general)
int launch_knn(const GridLayout &L, void *grid, int mode, int b, int n, int m, int k, float radius, bool cells, const float *unknown,
               float *dist, int *idx, hipStream_t s, bool wave = false) {
    GridHdr *hdrs = L.hdrs(grid);
    int *cell_start = L.cell_start(grid);
    float4 *sorted_pts = L.sorted_pts(grid);
    const int stride_cells = STRIDE_CELLS;
    const size_t lds = (size_t)2 * QPW * k * sizeof(u64) + (size_t)QPW * KNN_FLAT_CAP * sizeof(int);
    dim3 grid8(ogc_divup(n, QPW), b);
    const float lim2 = knn_radius_limit2(mode, radius);
    int deferred = 0;
    if (cells) {
        const dim3 grid4(ogc_divup(n, CPW), b);
        const size_t lds4 = sizeof(u64) * CPW * KQ_LIST;
#define OGC_KNN_CELLS(K)                                                                                              \
    hipLaunchKernelGGL(knn_cells_kernel<K>, grid4, dim3(OGC_WAVE), lds4, s, n, lim2, stride_cells, hdrs, cell_start, \
                       sorted_pts, dist, idx)
        if (k == 32) OGC_KNN_CELLS(32);          // the row lengths of the configs' smoothness terms (4 / 8: flow losses, OGC-DR)
        else if (k == 16) OGC_KNN_CELLS(16);
        else if (k == 8) OGC_KNN_CELLS(8);
        else OGC_KNN_CELLS(4);
#undef OGC_KNN_CELLS
        deferred = 1;
    } else if (wave) {
        // the whole wavefront on one query at a time (k <= 32); knn_grid_kernel afterwards does the rows it marked
        const long long queries = (long long)b * n;
        int qpw = (int)(queries / 8192);
        qpw = qpw < 1 ? 1 : (qpw > 8 ? 8 : qpw);
        const dim3 gridw(ogc_divup(n, qpw), b);
        if (mode == 1)
            hipLaunchKernelGGL(knn_wave_kernel<1>, gridw, dim3(OGC_WAVE), 0, s, n, m, k, radius, stride_cells, qpw, unknown, hdrs,
                               cell_start, sorted_pts, dist, idx);
        else
            hipLaunchKernelGGL(knn_wave_kernel<0>, gridw, dim3(OGC_WAVE), 0, s, n, m, k, radius, stride_cells, qpw, unknown, hdrs,
                               cell_start, sorted_pts, dist, idx);
        deferred = 2;
        static const bool only = [] { const char *e = getenv("OGC_KNN_WAVE_ONLY"); return e && e[0] == '1'; }(); // (development:
        if (only) return OGC_OK;                                  // rows left to knn_grid_kernel keep idx[row][0] = -1)
    }
    // sixteen lanes per query (four queries per wavefront) when eight would leave most SIMDs without a wavefront: the launch's time
    // is then one wavefront's serial work (FlowStep3D at B = 1: 4096 queries = 512 wavefronts of ~48 us; forward 7.45 -> 7.07 ms).
    // OGC_KNN_LANES=8|16 forces.
    static const int forced_lanes = [] { const char *e = getenv("OGC_KNN_LANES"); return e ? atoi(e) : 0; }();
    // (measured, tools/bench_ops.py --ops knn,knnc, 8 -> 16 lanes: 1 x 8192 x 8192, k = 32 0.079 -> 0.056 ms; 16 x 2048 <- 8192, k = 64
    // 0.264 -> 0.238; 16 x 512 <- 1024, k = 64 0.190 -> 0.116; but 16 x 8192 x 8192 0.240 -> 0.249, and the radius-limited searches,
    // which keep a handful of candidates, lose from 2048 wavefronts on: 16 x 1024 <- 2048 0.037 -> 0.042)
    const bool limited = mode == 1 && radius >= 0.0f;
    const long long waves8 = (long long)b * ogc_divup(n, QPW);
    const bool wide = forced_lanes == 16 || (forced_lanes != 8 && waves8 <= (limited ? 1024 : 4096));
    dim3 grid16(ogc_divup(n, OGC_WAVE / 16), b), grid32(ogc_divup(n, OGC_WAVE / 32), b);
    // ... and thirty-two (two queries per wavefront) for the smallest launches (FlowStep3D's 2048-point levels at B = 1)
    const bool wider = forced_lanes == 32 || (forced_lanes == 0 && waves8 <= (limited ? 256 : 1024)); // (1 x 8192 x 8192: 0.056 -> 0.050 ms)
    if (mode == 1 && wider)
        hipLaunchKernelGGL((knn_grid_kernel<1, 32>), grid32, dim3(OGC_WAVE), lds, s, n, m, k, radius, lim2, stride_cells, deferred, unknown,
                           hdrs, cell_start, sorted_pts, dist, idx);
    else if (wider)
        hipLaunchKernelGGL((knn_grid_kernel<0, 32>), grid32, dim3(OGC_WAVE), lds, s, n, m, k, radius, lim2, stride_cells, deferred, unknown,
                           hdrs, cell_start, sorted_pts, dist, idx);
    else if (mode == 1 && wide)
        hipLaunchKernelGGL((knn_grid_kernel<1, 16>), grid16, dim3(OGC_WAVE), lds, s, n, m, k, radius, lim2, stride_cells, deferred, unknown,
                           hdrs, cell_start, sorted_pts, dist, idx);
    else if (mode == 1)
        hipLaunchKernelGGL((knn_grid_kernel<1, 8>), grid8, dim3(OGC_WAVE), lds, s, n, m, k, radius, lim2, stride_cells, deferred, unknown,
                           hdrs, cell_start, sorted_pts, dist, idx);
    else if (wide)
        hipLaunchKernelGGL((knn_grid_kernel<0, 16>), grid16, dim3(OGC_WAVE), lds, s, n, m, k, radius, lim2, stride_cells, deferred, unknown,
                           hdrs, cell_start, sorted_pts, dist, idx);
    else
        hipLaunchKernelGGL((knn_grid_kernel<0, 8>), grid8, dim3(OGC_WAVE), lds, s, n, m, k, radius, lim2, stride_cells, deferred, unknown,
                           hdrs, cell_start, sorted_pts, dist, idx);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ogc_set_error("ogc_knn (grid): launch failed: %s", hipGetErrorString(e));
        return OGC_ERR_LAUNCH;
    }
    return OGC_OK;
}

bool knn_cells_applies(int mode, int n, int m, int k, float radius, bool same) {
    return mode == 1 && radius > 0.0f && radius < 1.0e18f && same && n == m && (k == 4 || k == 8 || k == 16 || k == 32) &&
           ogc_knn_cells_enabled();
}
} // namespace

int ogc_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                        int *idx, hipStream_t s) {
    // the cell-ordered traversal needs the centres to BE the points (ball_query(pc, pc), the reference's only live
    // use: losses/seg_loss_unsup.py:151, losses/flow_loss_unsup.py:84); other centre sets use the all-pairs scan
    const bool same = (new_xyz == xyz) && (m == n);
    if (!same || !ball_query_grid_applies(n, nsample, radius)) return OGC_ERR_UNSUPPORTED;
    const GridLayout L(b, n);
    void *ws = ogc_workspace(s, L.total());
    if (!ws) return OGC_ERR_UNSUPPORTED;
    launch_grid_build(b, n, radius, 0, STRIDE_CELLS, xyz, L.hdrs(ws), L.cell_start(ws), L.sorted_pts(ws), s);
    return launch_ball_query(L, ws, b, n, m, radius, nsample, xyz, idx, s);
}

// k-NN over cell lists.  Returns OGC_OK after queueing build + query, or OGC_ERR_UNSUPPORTED (caller: all-pairs scan).
int ogc_knn_grid(int mode, int b, int n, int m, int k, float radius, const float *unknown, const float *known,
                 float *dist, int *idx, hipStream_t s) {
    const size_t lds = (size_t)2 * QPW * k * sizeof(u64) + (size_t)QPW * KNN_FLAT_CAP * sizeof(int);
    // smallest cloud searched through cells (OGC_KNN_GRID_MIN in the environment, A/B runs; 1024 until round 4): below it the
    // all-pairs scan, one lane per query — at B = 1 a 512-point level of FlowStep3D is 16 wavefronts scanning for 160 us, against
    // ~60 us of build + search here (forward 7.70 -> 7.55 ms)
    static const int min_m = [] { const char *e = getenv("OGC_KNN_GRID_MIN"); const int v = e ? atoi(e) : 256; return v < 64 ? 64 : v; }();
    if (m < min_m || m <= 4 * k || lds > 64 * 1024) return OGC_ERR_UNSUPPORTED;
    const GridLayout L(b, m);
    void *ws = ogc_workspace(s, L.total());
    if (!ws) return OGC_ERR_UNSUPPORTED;
    // radius-limited search of a cloud in itself (the smoothness term's neighbour lists): the build then prefers cells of edge
    // 1.01 r when balls are sparsely filled
    const bool cells = knn_cells_applies(mode, n, m, k, radius, unknown == known);
    // points per cell = k / knn_div (OGC_KNN_DIV in the environment: A/B runs)
    // 33.5 = cell edge of half the expected k-th neighbour distance.  The first block's 128-key sort wants ~120 candidates in its 125
    // cells; where the cloud is denser than its bounding box suggests (scenes: ground, objects) a block holds more and the query falls
    // back to insertion, so full launches on scene-like clouds want SMALLER cells, while a launch that leaves the chip under-filled
    // (one wavefront's latency) wants fewer, fuller cells.  Measured (ms; uniform slab / synthetic scene, 16 x 8192 x 8192):
    //   k = 32: div 28 0.219 / 0.374, 33.5 0.240 / 0.305, 40 0.263 / 0.279;  k = 64: 33.5 - / 1.234, 40 - / 1.026, 48 - / 0.872
    //   (16 x 2048 <- 8192, k = 64: 33.5 0.239 / 0.453, 48 0.205 / 0.346);  1 x 8192 x 8192, k = 32: 28 0.044 / 0.043, 33.5 0.050 / 0.050
    static const float forced_div = [] { const char *e = getenv("OGC_KNN_DIV"); const float v = e ? (float)atof(e) : 0.0f; return v > 1.0f ? v : 0.0f; }();
    const bool small_launch = (long long)b * ogc_divup(n, QPW) <= 1024;
    float knn_div = 33.5f;
    if (forced_div > 0.0f) knn_div = forced_div;
    else if (small_launch) knn_div = (k >= 24 && k <= 40) ? 28.0f : 33.5f;
    else if (k >= 56) knn_div = m >= 4096 ? 48.0f : 33.5f;
    // (k = 24..40 on full launches stays at 33.5: 40 trades 0.305 -> 0.281 on scenes for 0.240 -> 0.264 on uniform clouds and 0.257 -> 0.309
    // at 8 x 16384 x 16384)
    // k <= 32 outside the radius-limited self search: a wavefront per query over cells of k / 16 points (knn_wave_kernel);
    // OGC_KNN_WAVE=0 in the environment: knn_grid_kernel alone (A/B runs, tests of both kernels)
    static const bool wave_on = [] { const char *e = getenv("OGC_KNN_WAVE"); return !(e && e[0] == '0'); }();
    const bool wave = wave_on && !cells && k <= 32 && forced_div == 0.0f;
    if (wave) knn_div = -(7.0f * (float)k < 230.0f ? 7.0f * (float)k : 230.0f);
    launch_grid_build(b, m, mode == 1 ? radius : 0.0f, k, STRIDE_CELLS, known, L.hdrs(ws), L.cell_start(ws), L.sorted_pts(ws), s,
                      cells ? 1 : 0, knn_div);
    return launch_knn(L, ws, mode, b, n, m, k, radius, cells, unknown, dist, idx, s, wave);
}

// ---- one grid for several radius searches of a batch of clouds in themselves (fused extension, include/ogc_ops.h) ------------
extern "C" long long ogc_cell_grid_bytes(int b, int n) {
    if (b < 0 || n < 0) return -1;
    return (long long)GridLayout(b, n).total();
}

extern "C" int ogc_cell_grid_build(int b, int n, float radius, const float *xyz, void *grid, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0, "ogc_cell_grid_build: negative dimension");
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(xyz && grid, "ogc_cell_grid_build: null pointer");
    if (!(radius > 0.0f) || !(radius < 3.0e38f) || n < 1024) {
        ogc_set_error("ogc_cell_grid_build: needs a finite positive radius and clouds of at least 1024 points (n=%d, r=%g)", n,
                      (double)radius);
        return OGC_ERR_UNSUPPORTED;
    }
    const GridLayout L(b, n);
    launch_grid_build(b, n, radius, 0, STRIDE_CELLS, xyz, L.hdrs(grid), L.cell_start(grid), L.sorted_pts(grid), (hipStream_t)stream, 2);
    OGC_CHECK_LAUNCH("ogc_cell_grid_build");
    return OGC_OK;
}

extern "C" int ogc_ball_query_cells(int b, int n, float radius, int nsample, const float *xyz, const void *grid, float grid_radius,
                                    int *idx, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && nsample >= 0, "ogc_ball_query_cells: negative dimension");
    if (b == 0 || n == 0 || nsample == 0) return OGC_OK;
    OGC_REQUIRE(xyz && grid && idx, "ogc_ball_query_cells: null pointer");
    OGC_REQUIRE((long long)b * n * nsample < (1ll << 31), "ogc_ball_query_cells: idx exceeds 32-bit indexing");
    // cells are 1.01 x the radius the grid was built for: a query of up to that radius finds its hits in the 27 cells around it
    if (!(radius <= grid_radius) || !ball_query_grid_applies(n, nsample, radius)) {
        ogc_set_error("ogc_ball_query_cells: radius %g exceeds the grid's (%g), or a shape the cell lists do not take (n=%d, "
                      "nsample=%d)", (double)radius, (double)grid_radius, n, nsample);
        return OGC_ERR_UNSUPPORTED;
    }
    return launch_ball_query(GridLayout(b, n), const_cast<void *>(grid), b, n, n, radius, nsample, xyz, idx, (hipStream_t)stream);
}

extern "C" int ogc_knn_clamped_cells(int b, int n, int k, float radius, const float *xyz, void *grid, float grid_radius, float *dist,
                                     int *idx, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && k >= 1, "ogc_knn_clamped_cells: bad dimension");
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(xyz && grid && dist && idx, "ogc_knn_clamped_cells: null pointer");
    const size_t lds = (size_t)2 * QPW * k * sizeof(u64) + (size_t)QPW * KNN_FLAT_CAP * sizeof(int);
    if (!(radius > 0.0f) || !(radius <= grid_radius) || n < 1024 || n <= 4 * k || lds > 64 * 1024) {
        ogc_set_error("ogc_knn_clamped_cells: needs 0 < radius <= the grid's radius (%g vs %g), n >= 1024, n > 4 k", (double)radius,
                      (double)grid_radius);
        return OGC_ERR_UNSUPPORTED;
    }
    return launch_knn(GridLayout(b, n), grid, 1, b, n, n, k, radius, knn_cells_applies(1, n, n, k, radius, true), xyz, dist, idx,
                      (hipStream_t)stream);
}

// three_nn over cell lists.  OGC_OK after queueing build + query, OGC_ERR_UNSUPPORTED when the caller should run its scan
// (few known points: the scan is as fast; OGC_THREE_NN_GRID=0 in the environment).
int ogc_three_nn_grid(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, hipStream_t s) {
    static const bool on = [] { const char *e = getenv("OGC_THREE_NN_GRID"); return !(e && e[0] == '0'); }();
    if (!on || m < 1024) return OGC_ERR_UNSUPPORTED;
    const GridLayout L(b, m);
    void *ws = ogc_workspace(s, L.total());
    if (!ws) return OGC_ERR_UNSUPPORTED;
    // density: 3 / 2 = 1.5 points per cell — the ball of radius h around a target (what the first block covers) then holds
    // ~6 of them, three or more for ~95 % of the targets
    launch_grid_build(b, m, 0.0f, 3, STRIDE_CELLS, known, L.hdrs(ws), L.cell_start(ws), L.sorted_pts(ws), s, 0, 2.0f);
    hipLaunchKernelGGL(three_nn_grid_kernel, dim3(ogc_divup(n, OGC_WAVE), b), dim3(OGC_WAVE), 0, s, n, m, STRIDE_CELLS, unknown,
                       L.hdrs(ws), L.cell_start(ws), L.sorted_pts(ws), dist2, idx);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ogc_set_error("ogc_three_nn (grid): launch failed: %s", hipGetErrorString(e));
        return OGC_ERR_LAUNCH;
    }
    return OGC_OK;
}
