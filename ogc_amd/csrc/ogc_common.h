// ogc_common.h — shared device helpers for libogc_ops.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ogc_ops.h"

#define OGC_WAVE 64

// ---- error plumbing (no exit(): SURVEY.md §5 "failure detection") -------------------------
void ogc_set_error(const char *fmt, ...);

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

// ---- arithmetic pinned to the reference's source expression ---------------------------------
// d = (ux-x)*(ux-x) + (uy-y)*(uy-y) + (uz-z)*(uz-z), fp32, left to right, one rounding per
// operation, never contracted into FMA (interpolate_gpu.cu:40, ball_query_gpu.cu:33,
// sampling_gpu.cu:133).  The *_rn intrinsics are immune to -ffp-contract.
__device__ __forceinline__ float ogc_sqdist(float ax, float ay, float az, float bx, float by,
                                            float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// IEEE minNum in ONE instruction: hipcc expands fminf() to a canonicalising v_max + v_min pair because it cannot
// prove its operands are not signalling NaNs; v_min_f32 in the kernel's default IEEE mode already returns the
// non-NaN operand (CUDA's min(d, temp[k]) at sampling_gpu.cu:134 has the same semantics).
__device__ __forceinline__ float ogc_min_f32(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
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

// ---- all-pairs scan engine shared by kNN / three-NN / ball query ---------------------------------------------
// Streams the `n` candidate points of a cloud (AoS xyz) past the lane's query point (qx, qy, qz).  One wavefront
// per workgroup, so everything below is wave-synchronous (LDS operations of one wave complete in issue order; no
// barriers).
//   * Candidates are staged through a 6 KiB LDS tile (OGC_TILE points): the wave loads the NEXT tile from global
//     memory into registers with coalesced 16-byte loads while it scans the current one, so HBM/L2 latency is
//     off the critical path.  (A first version fed the candidates through scalar loads / SGPR operands; one wave
//     can only keep one scalar request in flight and the ~800-cycle round trip per 8 candidates bounded the scan.)
//   * Every lane reads the same LDS address (broadcast ds_read_b128, conflict-free), eight candidates = 96 B at a
//     time, and the next group's reads are issued before the current group is evaluated.
//   * The eight squared distances are evaluated as eight independent chains in stage-major order: a dependent
//     VALU op issues only every ~8 cycles per wave on gfx950 (measured, tools/clock_probe.hip) and these kernels
//     run at 1-2 waves per SIMD, so instruction-level parallelism is what fills the pipe.
//   group(d, base)  : called with the 8 squared distances of candidates base..base+7; returns true to stop.
//   single(d, idx)  : called for the < 8 tail candidates; returns true to stop.
// Wave-level masks of "d[u] < thr" for the eight candidates of a group: eight independent v_cmp (no serial
// dependence), after which the per-candidate slow paths are guarded by cheap scalar branches.
__device__ __forceinline__ void ogc_masks8(const float (&d)[8], float thr, unsigned long long (&mk)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) mk[u] = __builtin_amdgcn_ballot_w64(d[u] < thr);
}

constexpr int OGC_TILE = 512;                 // candidates per tile
constexpr int OGC_TILE_FLOATS = OGC_TILE * 3; // 1536 floats = 6 KiB

__device__ __forceinline__ void ogc_dist8(const float4 (&c)[6], float qx, float qy, float qz, float (&d)[8]) {
    const float x[8] = {c[0].x, c[0].w, c[1].z, c[2].y, c[3].x, c[3].w, c[4].z, c[5].y};
    const float y[8] = {c[0].y, c[1].x, c[1].w, c[2].z, c[3].y, c[4].x, c[4].w, c[5].z};
    const float z[8] = {c[0].z, c[1].y, c[2].x, c[2].w, c[3].z, c[4].y, c[5].x, c[5].w};
    float dx[8], dy[8], dz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) dx[u] = __fsub_rn(qx, x[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) dy[u] = __fsub_rn(qy, y[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) dz[u] = __fsub_rn(qz, z[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) dx[u] = __fmul_rn(dx[u], dx[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) dy[u] = __fmul_rn(dy[u], dy[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) dz[u] = __fmul_rn(dz[u], dz[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = __fadd_rn(dx[u], dy[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = __fadd_rn(d[u], dz[u]);
}

// Loads tile `t` (floats [t*1536, t*1536+1536) of the cloud, clipped to nfloats) into registers:
// lane l holds float4 #(i*64 + l), i = 0..5.  ALIGNED: the cloud base is 16-byte aligned.
template <bool ALIGNED>
__device__ __forceinline__ void ogc_tile_fetch(const float *__restrict__ pts, int nfloats, int t, int lane,
                                               float4 (&r)[6]) {
    const int base = t * OGC_TILE_FLOATS;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int f = base + (i * OGC_WAVE + lane) * 4;
        if (ALIGNED && f + 3 < nfloats) {
            r[i] = *reinterpret_cast<const float4 *>(pts + f);
        } else {
            r[i].x = f + 0 < nfloats ? pts[f + 0] : 0.0f;
            r[i].y = f + 1 < nfloats ? pts[f + 1] : 0.0f;
            r[i].z = f + 2 < nfloats ? pts[f + 2] : 0.0f;
            r[i].w = f + 3 < nfloats ? pts[f + 3] : 0.0f;
        }
    }
}

template <bool ALIGNED, class Group, class Single>
__device__ __forceinline__ void ogc_scan_tiles(const float *__restrict__ pts, int n, float qx, float qy, float qz,
                                               float *tile, int lane, Group &&group, Single &&single) {
    const int nfloats = n * 3;
    const int ntiles = (n + OGC_TILE - 1) / OGC_TILE;
    float4 *tile4 = reinterpret_cast<float4 *>(tile);
    float4 pre[6];
    if (ntiles > 0) ogc_tile_fetch<ALIGNED>(pts, nfloats, 0, lane, pre);
    bool stop = false;
    for (int t = 0; t < ntiles && !stop; ++t) {
#pragma unroll
        for (int i = 0; i < 6; ++i) tile4[i * OGC_WAVE + lane] = pre[i];
        if (t + 1 < ntiles) ogc_tile_fetch<ALIGNED>(pts, nfloats, t + 1, lane, pre); // in flight during the scan
        const int cnt = min(OGC_TILE, n - t * OGC_TILE);
        const int ngroups = cnt >> 3;
        const int base = t * OGC_TILE;
        float4 c[6], nx[6];
        if (ngroups > 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) c[i] = tile4[i];
        }
        for (int g = 0; g < ngroups && !stop; ++g) {
            const int gn = min(g + 1, ngroups - 1) * 6; // clamped LDS prefetch of the next group
#pragma unroll
            for (int i = 0; i < 6; ++i) nx[i] = tile4[gn + i];
            float d[8];
            ogc_dist8(c, qx, qy, qz, d);
            stop = group(d, base + g * 8);
#pragma unroll
            for (int i = 0; i < 6; ++i) c[i] = nx[i];
        }
        for (int k = ngroups * 8; k < cnt && !stop; ++k)
            stop = single(ogc_sqdist(qx, qy, qz, tile[3 * k], tile[3 * k + 1], tile[3 * k + 2]), base + k);
    }
}

template <class Group, class Single>
__device__ __forceinline__ void ogc_scan_candidates(const float *__restrict__ pts, int n, float qx, float qy,
                                                    float qz, float *tile, int lane, Group &&group,
                                                    Single &&single) {
    if ((reinterpret_cast<uintptr_t>(pts) & 15) == 0)
        ogc_scan_tiles<true>(pts, n, qx, qy, qz, tile, lane, group, single);
    else
        ogc_scan_tiles<false>(pts, n, qx, qy, qz, tile, lane, group, single);
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
