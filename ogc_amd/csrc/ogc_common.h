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
