// ogc_common.h — shared device helpers for libogc_ops.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ogc_ops.h"

#define OGC_WAVE 64

// ---- error plumbing (no exit(): SURVEY.md §5 "failure detection") -------------------------
void ogc_set_error(const char *fmt, ...);
// persistent stream-ordered scratch buffer of at least `bytes` for work queued on `stream` (api.hip); nullptr on failure
void *ogc_workspace(hipStream_t stream, size_t bytes);

#define OGC_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            ogc_set_error(__VA_ARGS__);   \
            return OGC_ERR_INVALID_ARG;   \
        }                                 \
    } while (0)

#define OGC_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            ogc_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return OGC_ERR_LAUNCH;                                               \
        }                                                                        \
    } while (0)

static inline int ogc_divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- the deterministic-gradient mode (det.hip; ogc_set_deterministic) -------------------------------------------------------
bool ogc_deterministic();
void *ogc_det_scratch(hipStream_t stream, size_t bytes); // grow-only scratch per (device, stream), apart from ogc_workspace
// dst[e] (+)= part[0 * n + e] + part[1 * n + e] + ... in slot order, one thread per element
hipError_t ogc_det_reduce_f32(float *dst, const float *part, int slots, long long n, int accumulate, hipStream_t s);
hipError_t ogc_det_reduce_f64(double *dst, const double *part, int slots, long long n, int accumulate, hipStream_t s);
// per sample the positions t (ascending) with idx[b, t] == j: start (b, n + 1), pos (b, T), in the stream's deterministic scratch
int ogc_det_lists(const char *name, int b, int n, long long T, const int *idx, const int **start_out, const int **pos_out,
                  size_t extra, void **extra_ptr, hipStream_t s);
// out (b, c, n) (+)= sum over t with idx[b, t] == j of grad_out[b, ch, t] (interp: grad_out[b, ch, t / 3] * weight[b, t]), ascending t
int ogc_det_scatter_add(const char *name, int b, int c, int n, long long T, const int *idx, const float *grad_out,
                        long long go_bstride, const float *weight, int interp, float *out, int accumulate, hipStream_t s);

// ---- zero-fill as a KERNEL ------------------------------------------------------------------------------------------
// Not hipMemsetAsync: captured into a HIP graph (graph_step.py, utils/subgraph.py) a memset node of this stack zeroes
// correctly on the first replay and fills with a few stale low bits on the later ones (tools/memset_probe.py: a buffer set
// to 1e30 comes back as ~3e-41 where it should be 0) — harmless-looking, until the stale bits are large.  Every accumulator of
// this library that kernels add into with atomics is zeroed by this launch instead.  `bytes` and `ptr` multiples of 4.
namespace {
__global__ void ogc_zero_kernel(uint32_t *__restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) p[i] = 0u;
}
} // namespace
// (api.hip) true when [ptr, ptr + bytes) lies in the region ogc_zero_arena_begin() zeroed on `stream` and handed out once
bool ogc_zero_arena_covers(const void *ptr, size_t bytes, hipStream_t stream);
static inline hipError_t ogc_zero_async(void *ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (ogc_zero_arena_covers(ptr, bytes, stream)) return hipSuccess; // one fill per step instead of one launch per accumulator
    if ((bytes & 3) != 0 || ((uintptr_t)ptr & 3) != 0) return hipErrorInvalidValue;
    const size_t words = bytes >> 2;
    size_t blocks = (words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ogc_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<uint32_t *>(ptr), words);
    return hipGetLastError();
}

// ---- arithmetic pinned to the reference's source expression ---------------------------------
// d = (ux-x)*(ux-x) + (uy-y)*(uy-y) + (uz-z)*(uz-z), fp32, left to right, one rounding per
// operation, never contracted into FMA (interpolate_gpu.cu:40, ball_query_gpu.cu:33,
// sampling_gpu.cu:133).  The *_rn intrinsics are immune to -ffp-contract.
//
// OGC_FMAD (a compile-time variant, never the default): the same expression as `nvcc --fmad=true` — the reference's build —
// contracts it, fma(dz, dz, fma(dy, dy, dx * dx)): two roundings fewer.  ogc_amd/csrc/build.py links the search kernels compiled
// that way into libogc_ops_fmad.so, loaded instead of libogc_ops.so when OGC_FMAD=1 is in the environment, for a user who
// holds index tensors of the reference's CUDA binary on tie-heavy data and wants to compare against them; its results equal
// the oracle's under oracle_set_fmad(1) (tests/test_fmad_mode_gpu.py).  Every pruning argument of the kernels (cell lists,
// bucket boxes) only uses that the expression is monotone in |dx|, |dy|, |dz| with IEEE rounding, which holds for both forms.
typedef float ogc_v2f_ __attribute__((ext_vector_type(2)));
#ifdef OGC_FMAD
__device__ __forceinline__ float ogc_sqsum3(float dx, float dy, float dz) {
    return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}
__device__ __forceinline__ ogc_v2f_ ogc_sqsum3(ogc_v2f_ dx, ogc_v2f_ dy, ogc_v2f_ dz) {
#pragma clang fp contract(off)
    const ogc_v2f_ xx = dx * dx;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, xx));
}
#else
__device__ __forceinline__ float ogc_sqsum3(float dx, float dy, float dz) {
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
__device__ __forceinline__ ogc_v2f_ ogc_sqsum3(ogc_v2f_ dx, ogc_v2f_ dy, ogc_v2f_ dz) {
#pragma clang fp contract(off)
    dx = dx * dx;
    dy = dy * dy;
    dz = dz * dz;
    return (dx + dy) + dz;
}
#endif
__device__ __forceinline__ float ogc_sqdist(float ax, float ay, float az, float bx, float by,
                                            float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return ogc_sqsum3(dx, dy, dz);
}

// IEEE minNum in ONE instruction: hipcc expands fminf() to a canonicalising v_max + v_min pair because it cannot
// prove its operands are not signalling NaNs; v_min_f32 in the kernel's default IEEE mode already returns the
// non-NaN operand (CUDA's min(d, temp[k]) at sampling_gpu.cu:134 has the same semantics).
__device__ __forceinline__ float ogc_min_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float ogc_max3_f32(float a, float b, float c) { // NaN operands are ignored, as by v_min3
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float ogc_min3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Smallest of eight squared distances (NaNs are ignored unless all eight are NaN).
__device__ __forceinline__ float ogc_min8_f32(const float (&d)[8]) {
    return ogc_min_f32(ogc_min3_f32(ogc_min3_f32(d[0], d[1], d[2]), d[3], d[4]), ogc_min3_f32(d[5], d[6], d[7]));
}

// Wave-level masks of "d[u] < thr" for the eight candidates of a group: eight independent v_cmp (no serial
// dependence), after which the per-candidate slow paths are guarded by cheap scalar branches.
__device__ __forceinline__ void ogc_masks8(const float (&d)[8], float thr, unsigned long long (&mk)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) mk[u] = __builtin_amdgcn_ballot_w64(d[u] < thr);
}

// ---- all-pairs scan engine shared by kNN / three-NN / ball query ---------------------------------------------
// Streams the `n` candidate points of a cloud (AoS xyz) past the lane's query point (qx, qy, qz).  One wavefront
// per workgroup, so everything below is wave-synchronous (LDS operations of one wave complete in issue order; no
// barriers).  What the PMC counters say about these kernels (rocprofv3, ball query): a wave issues one instruction
// per 4 cycles whatever its kind, a dependent VALU op issues every 8, and LDS footprints keep occupancy at
// 1-2 waves per SIMD — so the scan is bounded by INSTRUCTIONS PER CANDIDATE, not by memory.  Hence:
//   * candidates are staged through a 6 KiB LDS tile stored as SoA (x[512] | y[512] | z[512]); the wave fetches the
//     NEXT tile from global memory (12-byte loads, one point per lane, coalesced) while it scans the current one;
//   * a group of eight candidates = six broadcast ds_read_b128 (every lane reads the same address); thanks to the
//     SoA layout consecutive registers hold the same coordinate of neighbouring candidates, so the distance
//     arithmetic runs as packed fp32 (v_pk_add/v_pk_mul, two candidates per instruction, same IEEE results);
//   * the eight distances are independent chains, the next group's reads are issued before the current group is
//     evaluated, and the whole group is tested with ONE wave-level branch.
// Slots past the end of the cloud are filled with +inf coordinates: their distance is +inf (or NaN) and can never
// satisfy a strict `<` test, so callers need no tail handling.
//   group(d, base) : called with the 8 squared distances of candidates base..base+7; returns true to stop.
constexpr int OGC_TILE = 512;                 // candidates per tile
constexpr int OGC_TILE_FLOATS = OGC_TILE * 3; // 1536 floats = 6 KiB

typedef float ogc_v2f __attribute__((ext_vector_type(2)));

struct OgcGroup { // eight candidates, SoA
    float4 x0, x1, y0, y1, z0, z1;
};

__device__ __forceinline__ void ogc_group_read(const float *tile, int g, OgcGroup &c) {
    const float4 *tx = reinterpret_cast<const float4 *>(tile) + 2 * g;
    const float4 *ty = reinterpret_cast<const float4 *>(tile + OGC_TILE) + 2 * g;
    const float4 *tz = reinterpret_cast<const float4 *>(tile + 2 * OGC_TILE) + 2 * g;
    c.x0 = tx[0]; c.x1 = tx[1];
    c.y0 = ty[0]; c.y1 = ty[1];
    c.z0 = tz[0]; c.z1 = tz[1];
}

__device__ __forceinline__ void ogc_dist8(const OgcGroup &c, float qx, float qy, float qz, float (&d)[8]) {
#pragma clang fp contract(off)
    const ogc_v2f q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
    const ogc_v2f X[4] = {{c.x0.x, c.x0.y}, {c.x0.z, c.x0.w}, {c.x1.x, c.x1.y}, {c.x1.z, c.x1.w}};
    const ogc_v2f Y[4] = {{c.y0.x, c.y0.y}, {c.y0.z, c.y0.w}, {c.y1.x, c.y1.y}, {c.y1.z, c.y1.w}};
    const ogc_v2f Z[4] = {{c.z0.x, c.z0.y}, {c.z0.z, c.z0.w}, {c.z1.x, c.z1.y}, {c.z1.z, c.z1.w}};
    ogc_v2f dx[4], dy[4], dz[4], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) dx[u] = q2x - X[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) dy[u] = q2y - Y[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) dz[u] = q2z - Z[u];
#ifdef OGC_FMAD
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = ogc_sqsum3(dx[u], dy[u], dz[u]);
#else
#pragma unroll
    for (int u = 0; u < 4; ++u) dx[u] = dx[u] * dx[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) dy[u] = dy[u] * dy[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) dz[u] = dz[u] * dz[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = dx[u] + dy[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = s[u] + dz[u];
#endif
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        d[2 * u] = s[u].x;
        d[2 * u + 1] = s[u].y;
    }
}

// Fetch tile `t` into registers: lane l holds points t*512 + i*64 + l, i = 0..7 (+inf beyond the cloud).
__device__ __forceinline__ void ogc_tile_fetch(const float *__restrict__ pts, int n, int t, int lane,
                                               float (&r)[8][3]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = t * OGC_TILE + i * OGC_WAVE + lane;
        if (p < n) {
            const float *q = pts + (size_t)p * 3;
            r[i][0] = q[0]; r[i][1] = q[1]; r[i][2] = q[2];
        } else {
            r[i][0] = r[i][1] = r[i][2] = INFINITY;
        }
    }
}

template <class Group>
__device__ __forceinline__ void ogc_scan_candidates(const float *__restrict__ pts, int n, float qx, float qy,
                                                    float qz, float *tile, int lane, Group &&group) {
    const int ntiles = (n + OGC_TILE - 1) / OGC_TILE;
    float pre[8][3];
    if (ntiles > 0) ogc_tile_fetch(pts, n, 0, lane, pre);
    bool stop = false;
    for (int t = 0; t < ntiles && !stop; ++t) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            tile[i * OGC_WAVE + lane] = pre[i][0];
            tile[OGC_TILE + i * OGC_WAVE + lane] = pre[i][1];
            tile[2 * OGC_TILE + i * OGC_WAVE + lane] = pre[i][2];
        }
        if (t + 1 < ntiles) ogc_tile_fetch(pts, n, t + 1, lane, pre); // in flight during the scan
        const int cnt = min(OGC_TILE, n - t * OGC_TILE);
        const int ngroups = (cnt + 7) >> 3;   // the last group may contain +inf padding
        const int base = t * OGC_TILE;
        OgcGroup ga, gb;
        ogc_group_read(tile, 0, ga);
        // two groups per iteration with ping-pong registers: no copies, next group's reads ahead of the math
        for (int g = 0; g < ngroups && !stop; g += 2) {
            float d[8];
            ogc_group_read(tile, min(g + 1, OGC_TILE / 8 - 1), gb);
            ogc_dist8(ga, qx, qy, qz, d);
            stop = group(d, base + g * 8);
            if (g + 1 < ngroups && !stop) {
                ogc_group_read(tile, min(g + 2, OGC_TILE / 8 - 1), ga);
                ogc_dist8(gb, qx, qy, qz, d);
                stop = group(d, base + g * 8 + 8);
            }
        }
    }
}

// ---- DPP cross-lane helpers (wave64) ---------------------------------------------------------
// dpp_ctrl encodings (LLVM SIDefines.h): quad_perm 0x00-0xFF, row_shr:n 0x110+n,
// row_mirror 0x140, row_half_mirror 0x141.
template <int CTRL>
__device__ __forceinline__ float ogc_dpp_f32(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned ogc_dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// Max over the 64 lanes of a wave; every lane receives the result.
// quad xor1, quad xor2, half-row mirror, row mirror -> each 16-lane row uniform; then 4 readlanes.
__device__ __forceinline__ float ogc_wave_max_f32(float v) {
    v = fmaxf(v, ogc_dpp_f32<0xB1>(v));
    v = fmaxf(v, ogc_dpp_f32<0x4E>(v));
    v = fmaxf(v, ogc_dpp_f32<0x141>(v));
    v = fmaxf(v, ogc_dpp_f32<0x140>(v));
    const int iv = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
