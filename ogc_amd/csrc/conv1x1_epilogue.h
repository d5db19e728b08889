// conv1x1_epilogue.h — what the forward / input-gradient GEMM kernels of conv1x1.hip (register tile per wavefront, fp32 or 16-bit
// tensors) and conv1x1_h.hip (persistent kernel for 16-bit tensors) share: constants, the pooled epilogue (PoolOut, see "POOL" at
// conv1x1_gemm_kernel) and its one-instruction maxima.
#pragma once
#include "conv1x1_shared.h"

namespace {

constexpr int FW_KQ_MAX = 40; // K <= 160 channels held in registers (40 float4 per lane)
constexpr int GN_SLOTS = 16;  // spread of the fused GroupNorm statistics over cache lines (see ogc_conv1x1_gemm_gnstats)

// max(x, x of the DPP partner) as ONE instruction.  (fmaxf on a DPP-moved value costs three: the move, a canonicalising
// v_max of the moved value — the compiler cannot know it is not a signalling NaN — and the maximum.)  The two wait states a
// DPP read needs after a VALU write of its source are inside the asm: the hazard recogniser does not look into it.
template <int CTRL>
__device__ __forceinline__ float ogc_max_dpp_f32(float x) {
    float r;
    if constexpr (CTRL == 0xB1)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else if constexpr (CTRL == 0x4E)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else if constexpr (CTRL == 0x141)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float ogc_max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float ogc_max2_f32(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct PoolOut {
    float *yext;          // (b, M, centres) largest raw output where sign[m] >= 0, smallest where sign[m] < 0
    int *aext;            // its neighbour index
    const float *sign;    // (M) the scale of the GroupNorm that follows (only its sign is used)
    int s;
};

// The extremes of the neighbourhoods in a wave's 64-row x 64-position tile (see POOL above).  acc[a][c][r]: row a * 16 + kk * 4 + r,
// position p0 + 4 j + c.  SEG = lanes per neighbourhood (4, 8, 16 for 16, 32, 64 neighbours).  sgn: +-1 per row of the tile (LDS).
template <int SEG, typename ACC>
__device__ __forceinline__ void ogc_pool_extremes_epilogue(const ACC (&acc)[4][4], int nblk, const float *sgn, int j, int kk,
                                                           int m0, int M, int centres, int pool_s, float *ye, int *ae) {
    const bool writer = (j & (SEG - 1)) == 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (a < nblk) {
            const float4 sg4 = *reinterpret_cast<const float4 *>(sgn + a * 16 + kk * 4);
            const float sga[4] = {sg4.x, sg4.y, sg4.z, sg4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + a * 16 + kk * 4 + r;
                const float sg = sga[r]; // exact: +-1 * v
                const float v0 = sg * acc[a][0][r], v1 = sg * acc[a][1][r];
                const float v2 = sg * acc[a][2][r], v3 = sg * acc[a][3][r];
                float hi = ogc_max3_f32(v0, v1, ogc_max2_f32(v2, v3));
                hi = ogc_max_dpp_f32<0xB1>(hi);
                hi = ogc_max_dpp_f32<0x4E>(hi);
                if (SEG >= 8) hi = ogc_max_dpp_f32<0x141>(hi);
                if (SEG >= 16) hi = ogc_max_dpp_f32<0x140>(hi);
                // first position of the neighbourhood that attains it (64: none in this lane) — selects, no branches
                unsigned idx = v3 == hi ? 4u * j + 3u : 64u;
                idx = v2 == hi ? 4u * j + 2u : idx;
                idx = v1 == hi ? 4u * j + 1u : idx;
                idx = v0 == hi ? 4u * j : idx;
                idx = min(idx, ogc_dpp_u32<0xB1>(idx));
                idx = min(idx, ogc_dpp_u32<0x4E>(idx));
                if (SEG >= 8) idx = min(idx, ogc_dpp_u32<0x141>(idx));
                if (SEG >= 16) idx = min(idx, ogc_dpp_u32<0x140>(idx));
                if (writer && m < M) {
                    const int o = (a * 16 + r) * centres;
                    ye[o] = sg * hi;
                    ae[o] = (int)(idx & (unsigned)(pool_s - 1)); // index inside the neighbourhood
                }
            }
        }
    }
}


} // namespace

// (conv1x1_h.hip) the persistent kernel for 16-bit tensors; false when the shape is not its (the caller launches the tile kernel)
bool ogc_gemm16_launch(bool transpose_a, bool stats_on, bool pro, bool pool_on, int b, int M, int K, int hw, int groups,
                       const float *w, const unsigned short *in, unsigned short *out, double *stats, const float *pa,
                       const float *pb, int pro_relu, hipStream_t s, const float *pool_sign, float *pool_yext, int *pool_aext,
                       int pool_s);
// (conv1x1_h.hip) the persistent kernel for fp32 tensors with <= 64 reduction channels; false when the shape is not its
bool ogc_gemm32_launch(bool transpose_a, bool stats_on, bool pro, bool pool_on, int b, int M, int K, int hw, int groups,
                       const float *w, const float *in, float *out, double *stats, const float *pa, const float *pb, int pro_relu,
                       hipStream_t s, const float *pool_sign, float *pool_yext, int *pool_aext, int pool_s);
