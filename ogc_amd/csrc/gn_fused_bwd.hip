// gn_fused_bwd.hip — backward of  y = conv( relu( GroupNorm(y_prev) ) )  without the GroupNorm backward passes.
//
// Inside a SharedMLP (reference utils/nn_util.py:45-85: Conv2d 1x1 -> GroupNorm -> ReLU, repeated) the normalised
// activation z = relu(a y_prev + bb) (a, bb per sample and channel: the GroupNorm folded into an affine map) is never
// stored: the convolution and its weight gradient recompute it while loading y_prev (conv1x1.hip).  The backward pass
// of one such layer used to be
//     dW   = g_y z^T                      (weight gradient, reads y_prev and g_y)
//     g_z  = W^T g_y                      (input gradient, writes g_z)
//     S1, S2 per (sample, group)          (GroupNorm sums, reads y_prev and g_z)
//     g_prev = alpha mask g_z + c2 y_prev + c3      (GroupNorm adjoint, reads y_prev and g_z, writes g_prev)
// — nine passes over a tensor of the activation's size, five of them in the two GroupNorm kernels (3.1 ms of a 14 ms
// C4 step, all of it re-reading).  Both GroupNorm sums follow from two MOMENT MATRICES the weight-gradient kernel can
// accumulate next to each other:   with mask = [a y_prev + bb > 0],
//     H  = g_y mask^T,   H2 = g_y (mask . y_prev)^T              (cout x cin each, per sample)
//     dW = sum_b  H2 diag(a_b) + H diag(bb_b)
//     T1[k] = sum_pos g_z[k] mask[k]          = sum_m W[m,k] H[m,k]
//     T2[k] = sum_pos g_z[k] mask[k] y_prev[k] = sum_m W[m,k] H2[m,k]
// so the sums are known BEFORE the input gradient is formed, and the input-gradient GEMM applies the adjoint in its
// epilogue (it holds the g_z tile; it reads the matching y_prev tile) and writes g_prev directly: g_z never exists.
// Five passes instead of nine: weight gradient 2 reads, input gradient 2 reads + 1 write.
//   ogc_conv1x1_wgrad_moments   H, H2          (v_mfma_f32_16x16x4_f32, two accumulator sets)
//   ogc_gn_moments_combine      dW, dgamma, dbeta, and alpha / c2 / c3 per (sample, channel)
//   ogc_conv1x1_dgrad_adjoint   g_prev
#include "ogc_common.h"
#include "conv_stage.h"
#include "act_io.h"

extern int ogc_g_matmul_bf16; // (conv1x1.hip) operand precision switch
// (conv1x1_h.hip) the dense adjoint input gradient for 16-bit tensors on the persistent kernel; false: not its shape
bool ogc_gemm16_adjoint_launch(int b, int M, int K, int hw, int relu, const float *w, const unsigned short *gy,
                               const unsigned short *yprev, const float *pa, const float *pb, const float *coef,
                               unsigned short *out, hipStream_t s);

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int WG_WAVES = 4;

// ---- moment matrices ------------------------------------------------------------------------------------------------
// Same operand layout as conv1x1_wgrad_kernel: for a step of 16 positions lane (i = l & 15, k = l >> 4) loads ONE float4 =
// row (c0 + i), positions pb + 4k .. 4k+3; MFMA k-slot k of sub-step s is position pb + 4k + s for both operands.
// A workgroup stays inside one sample (blockIdx.x = sample * chunks + chunk): H / H2 are per sample.
// POOLED: g_y is the gradient of a max-pooled GroupNorm in sparse form (ogc_group_norm_maxpool_bwd_sparse): `dy` is the
// convolution's OUTPUT y, and g_y[row, pos] = fmaf(c2, y, c3) + (pos % S == arg ? ag : 0) is rebuilt from coef2[b, row] =
// (c2, c3) and inj[b, row, pos / S] = (ag, arg) — a step's 16 positions lie inside one neighbourhood (S = 16, 32, 64).
// AT: element type of x and dy (float / ogc_bf16: act_io.h).
template <int COB, int CIB, bool POOLED, typename AT = float>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void wgrad_moments_kernel(int cin, int cout, int hw, int chunks,
                                                                           int steps_per_wave, int relu,
                                                                           const AT *__restrict__ x,   // y_prev (B, cin, hw)
                                                                           const AT *__restrict__ dy,  // g_y (B, cout, hw)
                                                                           const float *__restrict__ aff_a,
                                                                           const float *__restrict__ aff_b,
                                                                           const float2 *__restrict__ coef2, // (B, cout)
                                                                           const float2 *__restrict__ inj,   // (B, cout, hw >> s_shift)
                                                                           int s_shift,
                                                                           float *__restrict__ hm) {      // (B, 2, cout, cin)
    __shared__ float red[WG_WAVES][COB * CIB * 256]; // one slab per wave (see conv1x1_wgrad_kernel), H then H2
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, k = lane >> 4;
    const int img = blockIdx.x / chunks, chunk = blockIdx.x - img * chunks;
    const int co0 = blockIdx.y * (16 * COB), ci0 = blockIdx.z * (16 * CIB);
    const int steps_per_img = hw >> 4;
    const int first = (chunk * WG_WAVES + wave) * steps_per_wave;
    const int mine = max(0, min(steps_per_wave, steps_per_img - first)); // this wavefront's steps: [first, first + mine)

    v4f acc1[COB][CIB], acc2[COB][CIB];
#pragma unroll
    for (int a = 0; a < COB; ++a)
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            acc1[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
            acc2[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    int yrow[COB], xrow[CIB], jrow[COB];
    float ca[CIB], cb[CIB], pc2[COB], pc3[COB];
    const int centres = hw >> s_shift, smask = (1 << s_shift) - 1;
#pragma unroll
    for (int a = 0; a < COB; ++a) {
        const int row = min(co0 + a * 16 + i, cout - 1);
        yrow[a] = row * hw;
        if (POOLED) {
            const float2 cc = coef2[(size_t)img * cout + row];
            pc2[a] = cc.x;
            pc3[a] = cc.y;
            jrow[a] = row * centres;
        }
    }
    const float2 *jb_ = POOLED ? inj + (size_t)img * cout * centres : nullptr;
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
        const int ch = min(ci0 + c * 16 + i, cin - 1);
        xrow[c] = ch * hw;
        ca[c] = aff_a[(size_t)img * cin + ch];
        cb[c] = aff_b[(size_t)img * cin + ch];
    }
    const AT *yb_ = dy + (size_t)img * cout * hw + 4 * k;
    const AT *xb_ = x + (size_t)img * cin * hw + 4 * k;
    int cur = min(first, steps_per_img - 1);
    const int stop = min(first + mine, steps_per_img) - 1; // the walk stays on the wave's last step once it is reached
    // Unconditional loads of one shape (see conv1x1_wgrad_kernel: a load under a per-lane condition makes the compiler wait
    // for ALL outstanding loads wherever it needs one, which serialises the ping-pong): rows beyond the tensors are clamped
    // onto the last row — they only reach rows / columns of H, H2 that are never stored — and steps beyond the wave's
    // share re-read its last step and are not computed with.
    auto load = [&](float4(&yv)[COB], float4(&xv)[CIB], float2(&jv)[COB], int &jpos) { // step `cur`, then advance
        const int pb = cur * 16;
#pragma unroll
        for (int a = 0; a < COB; ++a) yv[a] = ogc_ld4(yb_ + yrow[a] + pb);
#pragma unroll
        for (int c = 0; c < CIB; ++c) xv[c] = ogc_ld4(xb_ + xrow[c] + pb);
        if (POOLED) {
#pragma unroll
            for (int a = 0; a < COB; ++a) jv[a] = jb_[jrow[a] + (pb >> s_shift)];
            jpos = (pb & smask) + 4 * k; // this lane's first position inside the neighbourhood
        }
        cur = min(cur + 1, max(stop, 0));
    };
    auto fma16 = [&](float4(&yv)[COB], const float4(&xraw)[CIB], const float2(&jv)[COB], int jpos) {
        if (POOLED) {
#pragma unroll
            for (int a = 0; a < COB; ++a) { // the expression of gn_maxpool_bwd_dx_kernel, bit for bit
                const int rel = __float_as_int(jv[a].y) - jpos;
                const float ag = jv[a].x;
                yv[a].x = fmaf(pc2[a], yv[a].x, pc3[a]) + (rel == 0 ? ag : 0.f);
                yv[a].y = fmaf(pc2[a], yv[a].y, pc3[a]) + (rel == 1 ? ag : 0.f);
                yv[a].z = fmaf(pc2[a], yv[a].z, pc3[a]) + (rel == 2 ? ag : 0.f);
                yv[a].w = fmaf(pc2[a], yv[a].w, pc3[a]) + (rel == 3 ? ag : 0.f);
            }
        }
        float4 m1[CIB], m2[CIB];
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            const float4 v = xraw[c];
            // mask = [a y + bb > 0] (relu); without relu: all ones
            const bool bx = relu ? fmaf(ca[c], v.x, cb[c]) > 0.f : true;
            const bool by = relu ? fmaf(ca[c], v.y, cb[c]) > 0.f : true;
            const bool bz = relu ? fmaf(ca[c], v.z, cb[c]) > 0.f : true;
            const bool bw = relu ? fmaf(ca[c], v.w, cb[c]) > 0.f : true;
            m1[c] = make_float4(bx ? 1.f : 0.f, by ? 1.f : 0.f, bz ? 1.f : 0.f, bw ? 1.f : 0.f);
            m2[c] = make_float4(bx ? v.x : 0.f, by ? v.y : 0.f, bz ? v.z : 0.f, bw ? v.w : 0.f);
        }
#pragma unroll
        for (int a = 0; a < COB; ++a)
#pragma unroll
            for (int c = 0; c < CIB; ++c) {
                acc1[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].x, m1[c].x, acc1[a][c], 0, 0, 0);
                acc2[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].x, m2[c].x, acc2[a][c], 0, 0, 0);
                acc1[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].y, m1[c].y, acc1[a][c], 0, 0, 0);
                acc2[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].y, m2[c].y, acc2[a][c], 0, 0, 0);
                acc1[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].z, m1[c].z, acc1[a][c], 0, 0, 0);
                acc2[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].z, m2[c].z, acc2[a][c], 0, 0, 0);
                acc1[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].w, m1[c].w, acc1[a][c], 0, 0, 0);
                acc2[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].w, m2[c].w, acc2[a][c], 0, 0, 0);
            }
    };

    float4 ya[COB], xa[CIB], yb[COB], xb[CIB];
    float2 ja[COB], jb[COB];
    int pa_ = 0, pb_ = 0;
    load(ya, xa, ja, pa_);
    int s = 0;
    for (; s + 1 < mine; s += 2) { // ping-pong registers: next step's loads fly during the MFMAs (no branch in the body)
        load(yb, xb, jb, pb_);
        fma16(ya, xa, ja, pa_);
        load(ya, xa, ja, pa_);
        fma16(yb, xb, jb, pb_);
    }
    if (s < mine) fma16(ya, xa, ja, pa_);

    float *dst = hm + (size_t)img * 2 * cout * cin;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (which) __syncthreads();
#pragma unroll
        for (int a = 0; a < COB; ++a)
#pragma unroll
            for (int c = 0; c < CIB; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) // C/D layout: lane l holds rows (l >> 4) * 4 + r of column l & 15
                    red[wave][(a * CIB + c) * 256 + (k * 4 + r) * 16 + i] = which ? acc2[a][c][r] : acc1[a][c][r];
        __syncthreads();
        for (int t = threadIdx.x; t < COB * CIB * 256; t += WG_WAVES * OGC_WAVE) {
            const int blk = t >> 8, a = blk / CIB, c = blk % CIB;
            const int row = co0 + a * 16 + ((t & 255) >> 4), col = ci0 + c * 16 + (t & 15);
            const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
            if (row < cout && col < cin && v != 0.0f) unsafeAtomicAdd(dst + ((size_t)which * cout + row) * cin + col, v);
        }
    }
}

// ---- the moment matrices from 16-bit tensors --------------------------------------------------------------------------------
// With x and dy stored as bf16 the kernel above moves half the bytes and is no faster: per 16 positions it still issues one load
// per row block (16 rows x 32 bytes each: the vector memory pipe works per row, not per byte) and 128 fp32 MFMAs (0.44 ms of
// matrix time for the 64 x 64 layers of C2's first level, twice what the halved bytes cost).  Here a step is 32 positions — lane
// (i, k) loads SIXTEEN bytes: row i, positions pb + 8 k .. 8 k + 7 — and the products run on gfx950's v_mfma_f32_16x16x32_bf16: a
// lane's eight consecutive positions are its eight k-slots (one MFMA per lane load and accumulator; ogc_mfma_bf16_k32, act_io.h).
// Nothing is rounded that was not already: dy (dense form) is used as loaded, the mask is 0 / 1, mask . x is x or 0; only the
// POOLED form rebuilds g_y in fp32 and rounds it (bf16 operands, as everywhere under ogc_set_matmul_precision(1)).
typedef short v4s16 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ogc_unpack8(const uint4 &u, float (&f)[8]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}

template <int COB, int CIB, bool POOLED>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void wgrad_moments16_kernel(int cin, int cout, int hw, int chunks,
                                                                             int steps_per_wave, int relu,
                                                                             const ogc_bf16 *__restrict__ x,   // y_prev (B, cin, hw)
                                                                             const ogc_bf16 *__restrict__ dy,  // g_y or y (B, cout, hw)
                                                                             const float *__restrict__ aff_a,
                                                                             const float *__restrict__ aff_b,
                                                                             const float2 *__restrict__ coef2, // (B, cout)
                                                                             const float2 *__restrict__ inj,   // (B, cout, hw >> s_shift)
                                                                             int s_shift,
                                                                             float *__restrict__ hm) {         // (B, 2, cout, cin)
    __shared__ float red[WG_WAVES][COB * CIB * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, k = lane >> 4;
    const int img = blockIdx.x / chunks, chunk = blockIdx.x - img * chunks;
    const int co0 = blockIdx.y * (16 * COB), ci0 = blockIdx.z * (16 * CIB);
    const int steps_per_img = hw >> 5;
    const int first = (chunk * WG_WAVES + wave) * steps_per_wave;
    const int mine = max(0, min(steps_per_wave, steps_per_img - first));

    v4f acc1[COB][CIB], acc2[COB][CIB];
#pragma unroll
    for (int a = 0; a < COB; ++a)
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            acc1[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
            acc2[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    int yrow[COB], xrow[CIB], jrow[COB];
    float ca[CIB], cb[CIB], pc2[COB], pc3[COB];
    const int centres = hw >> s_shift, smask = (1 << s_shift) - 1;
#pragma unroll
    for (int a = 0; a < COB; ++a) {
        const int row = min(co0 + a * 16 + i, cout - 1);
        yrow[a] = row * hw;
        if (POOLED) {
            const float2 cc = coef2[(size_t)img * cout + row];
            pc2[a] = cc.x;
            pc3[a] = cc.y;
            jrow[a] = row * centres;
        }
    }
    const float2 *jb_ = POOLED ? inj + (size_t)img * cout * centres : nullptr;
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
        const int ch = min(ci0 + c * 16 + i, cin - 1);
        xrow[c] = ch * hw;
        ca[c] = aff_a[(size_t)img * cin + ch];
        cb[c] = aff_b[(size_t)img * cin + ch];
    }
    const ogc_bf16 *yb_ = dy + (size_t)img * cout * hw + 8 * k;
    const ogc_bf16 *xb_ = x + (size_t)img * cin * hw + 8 * k;
    int cur = min(first, steps_per_img - 1);
    const int stop = min(first + mine, steps_per_img) - 1;
    // (unconditional loads of one shape, clamped rows and steps: see wgrad_moments_kernel)
    auto load = [&](uint4(&yv)[COB], uint4(&xv)[CIB], float2(&jv)[COB], int &jpos) { // step `cur`, then advance
        const int pb = cur * 32;
#pragma unroll
        for (int a = 0; a < COB; ++a) yv[a] = *reinterpret_cast<const uint4 *>(yb_ + yrow[a] + pb);
#pragma unroll
        for (int c = 0; c < CIB; ++c) xv[c] = *reinterpret_cast<const uint4 *>(xb_ + xrow[c] + pb);
        if (POOLED) { // a lane's eight positions lie inside one neighbourhood (S >= 16)
#pragma unroll
            for (int a = 0; a < COB; ++a) jv[a] = jb_[jrow[a] + ((pb + 8 * k) >> s_shift)];
            jpos = (pb + 8 * k) & smask;
        }
        cur = min(cur + 1, max(stop, 0));
    };
    auto fma32 = [&](const uint4(&yraw)[COB], const uint4(&xraw)[CIB], const float2(&jv)[COB], int jpos) {
        v4s16 y0[COB], y1[COB];
#pragma unroll
        for (int a = 0; a < COB; ++a) {
            if (POOLED) { // the expression of gn_maxpool_bwd_dx_kernel on the stored y, rounded to the operand precision
                float f[8];
                ogc_unpack8(yraw[a], f);
                const int rel = __float_as_int(jv[a].y) - jpos;
                const float ag = jv[a].x;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaf(pc2[a], f[e], pc3[a]) + (rel == e ? ag : 0.f);
                y0[a] = ogc_pack_bf16_rr(f[0], f[1], f[2], f[3]);
                y1[a] = ogc_pack_bf16_rr(f[4], f[5], f[6], f[7]);
            } else {
                y0[a] = __builtin_bit_cast(v4s16, make_uint2(yraw[a].x, yraw[a].y));
                y1[a] = __builtin_bit_cast(v4s16, make_uint2(yraw[a].z, yraw[a].w));
            }
        }
        v4s16 m1a[CIB], m1b[CIB], m2a[CIB], m2b[CIB];
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            float f[8];
            ogc_unpack8(xraw[c], f);
            unsigned keep[4]; // 0xFFFF per 16-bit half whose position passes the ReLU
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool lo = relu ? fmaf(ca[c], f[2 * e], cb[c]) > 0.f : true;
                const bool hi = relu ? fmaf(ca[c], f[2 * e + 1], cb[c]) > 0.f : true;
                keep[e] = (lo ? 0x0000FFFFu : 0u) | (hi ? 0xFFFF0000u : 0u);
            }
            const unsigned xs[4] = {xraw[c].x, xraw[c].y, xraw[c].z, xraw[c].w};
            m1a[c] = __builtin_bit_cast(v4s16, make_uint2(keep[0] & 0x3F803F80u, keep[1] & 0x3F803F80u)); // bf16 1.0 = 0x3F80
            m1b[c] = __builtin_bit_cast(v4s16, make_uint2(keep[2] & 0x3F803F80u, keep[3] & 0x3F803F80u));
            m2a[c] = __builtin_bit_cast(v4s16, make_uint2(keep[0] & xs[0], keep[1] & xs[1]));
            m2b[c] = __builtin_bit_cast(v4s16, make_uint2(keep[2] & xs[2], keep[3] & xs[3]));
        }
#pragma unroll
        for (int a = 0; a < COB; ++a)
#pragma unroll
            for (int c = 0; c < CIB; ++c) {
                // one v_mfma_f32_16x16x32_bf16 per accumulator: the lane's eight positions are its eight k-slots
                acc1[a][c] = ogc_mfma_bf16_k32(y0[a], y1[a], m1a[c], m1b[c], acc1[a][c]);
                acc2[a][c] = ogc_mfma_bf16_k32(y0[a], y1[a], m2a[c], m2b[c], acc2[a][c]);
            }
    };

    uint4 ya[COB], xa[CIB], yb[COB], xb[CIB];
    float2 ja[COB], jb[COB];
    int pa_ = 0, pb_ = 0;
    load(ya, xa, ja, pa_);
    int s = 0;
    for (; s + 1 < mine; s += 2) { // ping-pong registers: next step's loads fly during the MFMAs (no branch in the body)
        load(yb, xb, jb, pb_);
        fma32(ya, xa, ja, pa_);
        load(ya, xa, ja, pa_);
        fma32(yb, xb, jb, pb_);
    }
    if (s < mine) fma32(ya, xa, ja, pa_);

    float *dst = hm + (size_t)img * 2 * cout * cin;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (which) __syncthreads();
#pragma unroll
        for (int a = 0; a < COB; ++a)
#pragma unroll
            for (int c = 0; c < CIB; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[wave][(a * CIB + c) * 256 + (k * 4 + r) * 16 + i] = which ? acc2[a][c][r] : acc1[a][c][r];
        __syncthreads();
        for (int t = threadIdx.x; t < COB * CIB * 256; t += WG_WAVES * OGC_WAVE) {
            const int blk = t >> 8, a = blk / CIB, c = blk % CIB;
            const int row = co0 + a * 16 + ((t & 255) >> 4), col = ci0 + c * 16 + (t & 15);
            const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
            if (row < cout && col < cin && v != 0.0f) unsafeAtomicAdd(dst + ((size_t)which * cout + row) * cin + col, v);
        }
    }
}


template <int COB, int CIB, typename AT>
void moments_launch(int b, int cin, int cout, int hw, int relu, const AT *x, const AT *dy, const float *pa,
                    const float *pb, const float *coef2, const float *inj, int s_shift, float *hm, hipStream_t s) {
    const int steps_per_img = sizeof(AT) == 2 ? hw >> 5 : hw >> 4; // (16-bit tensors: 32 positions per step)
    const int tiles = ogc_divup(cout, 16 * COB) * ogc_divup(cin, 16 * CIB);
    long long waves = (2048 / tiles) / b;            // wavefronts per sample and tile pair: ~2048 over the chip
    if (waves < 4) waves = 4;
    int spw = (int)((steps_per_img + waves - 1) / waves);
    if (spw < 8) spw = 8;
    spw = (spw + 1) / 2 * 2;
    const int chunks = ogc_divup(steps_per_img, spw * WG_WAVES);
    dim3 grid(b * chunks, ogc_divup(cout, 16 * COB), ogc_divup(cin, 16 * CIB));
    const float2 *c2 = reinterpret_cast<const float2 *>(coef2), *ij = reinterpret_cast<const float2 *>(inj);
    if constexpr (sizeof(AT) == 2) {
        if (inj)
            hipLaunchKernelGGL((wgrad_moments16_kernel<COB, CIB, true>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, cin, cout, hw,
                               chunks, spw, relu, x, dy, pa, pb, c2, ij, s_shift, hm);
        else
            hipLaunchKernelGGL((wgrad_moments16_kernel<COB, CIB, false>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, cin, cout, hw,
                               chunks, spw, relu, x, dy, pa, pb, c2, ij, 0, hm);
    } else {
        if (inj)
            hipLaunchKernelGGL((wgrad_moments_kernel<COB, CIB, true, AT>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, cin, cout, hw, chunks,
                               spw, relu, x, dy, pa, pb, c2, ij, s_shift, hm);
        else
            hipLaunchKernelGGL((wgrad_moments_kernel<COB, CIB, false, AT>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, cin, cout, hw, chunks,
                               spw, relu, x, dy, pa, pb, c2, ij, 0, hm);
    }
}

// ---- dW, the GroupNorm parameter gradients and the coefficients of the adjoint ----------------------------------------------
// blocks [0, b): one per sample — T1, T2 per channel, the two sums per group, alpha / c2 / c3 per channel, and the sample's
// share of dgamma / dbeta; the other blocks: one element of dW per thread.
__global__ __launch_bounds__(256) void moments_combine_kernel(int b, int cin, int cout, int hw, int groups,
                                                              const float *__restrict__ hm, const float *__restrict__ w,
                                                              const float *__restrict__ pa, const float *__restrict__ pb,
                                                              const float *__restrict__ mean, const float *__restrict__ rstd,
                                                              const float *__restrict__ gamma, float *__restrict__ dw,
                                                              float *__restrict__ coef, // (b, cin, 3): alpha, c2, c3
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ double s1[32], s2[32];
    const int cpg = cin / groups;
    if ((int)blockIdx.x < b) {
        const int img = blockIdx.x;
        __shared__ double tt1[512], tt2[512]; // T1, T2 per channel (cin <= 512, checked by the entry point)
        for (int t = threadIdx.x; t < 512; t += blockDim.x) { tt1[t] = 0.0; tt2[t] = 0.0; }
        if (threadIdx.x < 32) { s1[threadIdx.x] = 0.0; s2[threadIdx.x] = 0.0; }
        __syncthreads();
        const float *h1 = hm + (size_t)img * 2 * cout * cin, *h2 = h1 + (size_t)cout * cin;
        // work item = (channel k, row lane ml of 8): consecutive threads take consecutive channels (coalesced rows of H)
        for (int it = threadIdx.x; it < cin * 8; it += blockDim.x) {
            const int ml = it / cin, k = it - ml * cin;
            double t1 = 0.0, t2 = 0.0;
            for (int m = ml; m < cout; m += 8) {
                const double wv = w[(size_t)m * cin + k];
                t1 += wv * h1[(size_t)m * cin + k];
                t2 += wv * h2[(size_t)m * cin + k];
            }
            atomicAdd(&tt1[k], t1);
            atomicAdd(&tt2[k], t2);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < cin; k += blockDim.x) {
            const int g = k / cpg;
            const double mu = mean[img * groups + g], r = rstd[img * groups + g], gm = gamma[k];
            atomicAdd(&s1[g], gm * tt1[k]);                       // S1 = sum_c gamma_c sum_pos g'
            atomicAdd(&s2[g], gm * (tt2[k] - mu * tt1[k]) * r);   // S2 = sum_c gamma_c sum_pos g' xhat
        }
        __syncthreads();
        const double n = (double)cpg * (double)hw;
        for (int k = threadIdx.x; k < cin; k += blockDim.x) {
            const int g = k / cpg;
            const double mu = mean[img * groups + g], r = rstd[img * groups + g];
            const double c2 = -r * r * s2[g] / n;
            const double c3 = -r * s1[g] / n - c2 * mu;
            float *o = coef + ((size_t)img * cin + k) * 3;
            o[0] = (float)((double)gamma[k] * r);
            o[1] = (float)c2;
            o[2] = (float)c3;
        }
        // this sample's share of the GroupNorm parameter gradients (fp32 atomics over the b samples; zeroed by the entry point)
        for (int k = threadIdx.x; k < cin; k += blockDim.x) {
            const int g = k / cpg;
            const double mu = mean[img * groups + g], r = rstd[img * groups + g];
            unsafeAtomicAdd(dgamma + k, (float)((tt2[k] - mu * tt1[k]) * r));
            unsafeAtomicAdd(dbeta + k, (float)tt1[k]);
        }
    } else {
        // one element of dW per thread, summed over the samples in a fixed order
        const long long e = ((long long)blockIdx.x - b) * blockDim.x + threadIdx.x;
        if (e < (long long)cout * cin) {
            const int k = (int)(e % cin);
            float acc = 0.f;
            for (int img = 0; img < b; ++img) {
                const float *h1 = hm + (size_t)img * 2 * cout * cin, *h2 = h1 + (size_t)cout * cin;
                acc += pa[(size_t)img * cin + k] * h2[e] + pb[(size_t)img * cin + k] * h1[e];
            }
            dw[e] = acc;
        }
    }
}

// ---- input gradient with the GroupNorm adjoint in the epilogue -----------------------------------------------------------------
// g_prev[b, m, p] = alpha[b, m] mask (sum_k w[k, m] g_y[b, k, p]) + c2[b, m] y_prev[b, m, p] + c3[b, m]
// The GEMM of conv1x1_gemm_kernel<TRANSPOSE_A = true> (M = cin, K = cout): a wave owns 64 positions, the whole K x 64 tile
// of g_y in registers, A = w^T staged through LDS 64 rows at a time; the epilogue holds four consecutive positions of
// an output row per lane and reads the same four of y_prev.
// POOLED: g_y in the sparse form of ogc_group_norm_maxpool_bwd_sparse, rebuilt from the convolution's output (`gy` = y) as
// in wgrad_moments_kernel; a lane's four positions lie inside one neighbourhood (S >= 4).
// AT: element type of gy, yprev and out (float / ogc_bf16: act_io.h).
// BF: operands rounded to bf16 on v_mfma_f32_16x16x16_bf16, staged as in conv1x1_gemm_kernel (the 16-bit instantiation: at 64 x 64
// channels the fp32 MFMAs of a tile take 3.4 us per wavefront, twice what its halved bytes cost).
template <int KQ, bool POOLED, typename AT = float, bool BF = false>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void dgrad_adjoint_kernel(int M, int K, int hw, int relu,
                                                                           const float *__restrict__ w,     // (K, M)
                                                                           const AT *__restrict__ gy,    // (B, K, hw)
                                                                           const AT *__restrict__ yprev, // (B, M, hw)
                                                                           const float *__restrict__ pa,
                                                                           const float *__restrict__ pb,
                                                                           const float *__restrict__ coef,  // (B, M, 3)
                                                                           const float2 *__restrict__ coef2, // (B, K)
                                                                           const float2 *__restrict__ inj,   // (B, K, hw >> s_shift)
                                                                           int s_shift,
                                                                           AT *__restrict__ out) {       // (B, M, hw)
    extern __shared__ __attribute__((aligned(16))) float a_lds[]; // [64][ogc_a_ld(Kq)] weights (conv_stage.h), then [64][5] coefficients
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * WG_WAVES + wave) * 64;
    const int Kq = (K + 3) >> 2;
    const bool live = p0 < hw;
    const AT *inb = gy + (size_t)b * K * hw;
    AT *outb = out + (size_t)b * M * hw;
    const AT *yb = yprev + (size_t)b * M * hw;
    const int a_ld = ogc_a_ld(Kq);
    float *cf = a_lds + (size_t)64 * a_ld;

    float4 xin[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int row = q * 4 + kk;
        xin[q] = (live && q < Kq && row < K) ? ogc_ld4(inb + (size_t)row * hw + p0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (POOLED) {
        // (c2, c3, ag, arg) of the wave's K rows x (64 >> s_shift) neighbourhoods through a wave-private LDS table: read back one
        // 16-byte entry per row right where it is used (held in registers next to the K x 64 tile they cost a wavefront per SIMD)
        const int centres = hw >> s_shift, cpw = 64 >> s_shift;
        float4 *tab = reinterpret_cast<float4 *>(cf + 64 * 5) + (size_t)wave * K * cpw;
        for (int e = lane; e < K * cpw; e += OGC_WAVE) {
            const int row = e / cpw, centre = (p0 >> s_shift) + (e - row * cpw);
            const float2 cc = coef2[(size_t)b * K + row];
            const float2 jj = (live && centre < centres) ? inj[((size_t)b * K + row) * centres + centre]
                                                         : make_float2(0.f, __int_as_float(-1));
            tab[e] = make_float4(cc.x, cc.y, jj.x, jj.y);
        }
        __syncthreads();
        const int mine = (4 * j) >> s_shift, jpos = (p0 + 4 * j) & ((1 << s_shift) - 1);
#pragma unroll
        for (int q = 0; q < KQ; ++q) { // the expression of gn_maxpool_bwd_dx_kernel, bit for bit (rows beyond K stay zero)
            const int row = q * 4 + kk;
            if (q < Kq && row < K) {
                const float4 t = tab[row * cpw + mine];
                const int rel = __float_as_int(t.w) - jpos;
                xin[q].x = fmaf(t.x, xin[q].x, t.y) + (rel == 0 ? t.z : 0.f);
                xin[q].y = fmaf(t.x, xin[q].y, t.y) + (rel == 1 ? t.z : 0.f);
                xin[q].z = fmaf(t.x, xin[q].z, t.y) + (rel == 2 ? t.z : 0.f);
                xin[q].w = fmaf(t.x, xin[q].w, t.y) + (rel == 3 ? t.z : 0.f);
            }
        }
    }
    typedef short v4s __attribute__((ext_vector_type(4)));
    constexpr int GQ = (KQ + 3) / 4;
    v4s xb[BF ? GQ : 1][4]; // BF: the tile as packed bf16 operands (k-slot i of lane group kk = row 4 (4 g + i) + kk)
    if constexpr (BF) {
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            float4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = 4 * g + i < KQ ? xin[4 * g + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            xb[g][0] = ogc_pack_bf16_rr(r[0].x, r[1].x, r[2].x, r[3].x);
            xb[g][1] = ogc_pack_bf16_rr(r[0].y, r[1].y, r[2].y, r[3].y);
            xb[g][2] = ogc_pack_bf16_rr(r[0].z, r[1].z, r[2].z, r[3].z);
            xb[g][3] = ogc_pack_bf16_rr(r[0].w, r[1].w, r[2].w, r[3].w);
        }
    }
    for (int m0 = 0; m0 < M; m0 += 64) {
        __syncthreads(); // previous tile fully consumed
        if constexpr (BF) {
            ogc_stage_weight_tile_bf16<true, WG_WAVES, GQ>(a_lds, w, m0, M, K, Kq); // packed operands of w^T (conv_stage.h)
        } else {
            ogc_stage_weight_tile<true, WG_WAVES>(a_lds, w, m0, M, K, Kq);
        }
        for (int t = threadIdx.x; t < 64; t += WG_WAVES * OGC_WAVE) {
            const int m = m0 + t;
            const bool in = m < M;
            cf[t * 5 + 0] = in ? pa[(size_t)b * M + m] : 0.f;
            cf[t * 5 + 1] = in ? pb[(size_t)b * M + m] : 0.f;
            cf[t * 5 + 2] = in ? coef[((size_t)b * M + m) * 3] : 0.f;
            cf[t * 5 + 3] = in ? coef[((size_t)b * M + m) * 3 + 1] : 0.f;
            cf[t * 5 + 4] = in ? coef[((size_t)b * M + m) * 3 + 2] : 0.f;
        }
        __syncthreads();
        const int nblk = min(4, (M - m0 + 15) >> 4);
        // One 16-row block of the tile at a time (unlike conv1x1_gemm_kernel, which runs the four blocks side by side): the
        // block's accumulators are 16 registers instead of 64, its y_prev values (requested before its MFMA loop, consumed
        // after it) another 16 — the epilogue needs both in VGPRs, and holding the whole tile there costs two wavefronts
        // per SIMD.  Four independent accumulators (the four column blocks) keep the MFMA pipe issuing back to back.
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a < nblk) {
                float4 yv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + a * 16 + kk * 4 + r;
                    yv[r] = (live && m < M) ? ogc_ld4(yb + (size_t)m * hw + p0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                v4f acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = (v4f){0.f, 0.f, 0.f, 0.f};
                if constexpr (BF) {
                    // pairs of groups on v_mfma_f32_16x16x32_bf16, a single last group on the 16x16x16 form — the order of
                    // conv1x1_gemm16_kernel's adjoint form (conv1x1_h.hip), so that the two kernels agree bit for bit
                    const v4s *a_bf = reinterpret_cast<const v4s *>(a_lds);
#pragma unroll
                    for (int g = 0; g + 1 < GQ; g += 2) {
                        if (4 * (g + 1) < Kq) {
                            const v4s av0 = a_bf[(g * 64 + a * 16 + j) * 4 + kk], av1 = a_bf[((g + 1) * 64 + a * 16 + j) * 4 + kk];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = ogc_mfma_bf16_k32(av0, av1, xb[g][c], xb[g + 1][c], acc[c]);
                        } else if (4 * g < Kq) {
                            const v4s av = a_bf[(g * 64 + a * 16 + j) * 4 + kk];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, xb[g][c], acc[c], 0, 0, 0);
                        }
                    }
                    if constexpr (GQ % 2 == 1) {
                        constexpr int g = GQ - 1;
                        if (4 * g < Kq) {
                            const v4s av = a_bf[(g * 64 + a * 16 + j) * 4 + kk];
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, xb[g][c], acc[c], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    if (q < Kq) {
                        const float av = a_lds[(a * 16 + j) * a_ld + q * 4 + kk]; // A[m0+16a+j][4q+kk]
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xin[q].x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xin[q].y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xin[q].z, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xin[q].w, acc[3], 0, 0, 0);
                    }
                }
                }
                if (live) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mi = a * 16 + kk * 4 + r, m = m0 + mi; // C/D layout: row (l >> 4) * 4 + r, column l & 15
                        if (m < M) {
                            const float fa = cf[mi * 5], fb = cf[mi * 5 + 1], al = cf[mi * 5 + 2], c2 = cf[mi * 5 + 3], c3 = cf[mi * 5 + 4];
                            const float4 y = yv[r];
                            float4 g = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                            if (relu) {
                                g.x = fmaf(fa, y.x, fb) > 0.f ? g.x : 0.f; g.y = fmaf(fa, y.y, fb) > 0.f ? g.y : 0.f;
                                g.z = fmaf(fa, y.z, fb) > 0.f ? g.z : 0.f; g.w = fmaf(fa, y.w, fb) > 0.f ? g.w : 0.f;
                            }
                            float4 o;
                            o.x = fmaf(al, g.x, fmaf(c2, y.x, c3)); o.y = fmaf(al, g.y, fmaf(c2, y.y, c3));
                            o.z = fmaf(al, g.z, fmaf(c2, y.z, c3)); o.w = fmaf(al, g.w, fmaf(c2, y.w, c3));
                            ogc_st4(outb + (size_t)m * hw + p0 + 4 * j, o);
                        }
                    }
                }
            }
        }
    }
}

} // namespace

namespace {
int nsample_shift(int nsample) { return nsample == 16 ? 4 : nsample == 32 ? 5 : nsample == 64 ? 6 : -1; }

template <typename AT>
int wgrad_moments_impl(const char *name, int b, int cin, int cout, int hw, int relu, const AT *y_prev, const float *pa,
                       const float *pb, const AT *grad_y, const float *coef2, const float *inj, int s_shift,
                       float *moments, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && hw >= 1, "%s: bad shape", name);
    OGC_REQUIRE(y_prev && pa && pb && grad_y && moments, "%s: null pointer", name);
    if ((hw & (sizeof(AT) == 2 ? 31 : 15)) != 0 || (((uintptr_t)y_prev | (uintptr_t)grad_y) & 15) != 0) {
        ogc_set_error("%s: hw=%d must be a multiple of 16 (32 for 16-bit tensors) and the tensors 16-byte aligned", name, hw);
        return OGC_ERR_UNSUPPORTED;
    }
    if (sizeof(AT) == 2 && !ogc_g_matmul_bf16) {
        ogc_set_error("%s: 16-bit activations need ogc_set_matmul_precision(1)", name);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE((long long)cin * hw < (1ll << 31) && (long long)cout * hw < (1ll << 31) && b <= 32768,
                "%s: one sample exceeds 32-bit indexing", name);
    hipStream_t s = (hipStream_t)stream;
    if (b == 0) return OGC_OK;
    if (ogc_zero_async(moments, sizeof(float) * 2 * (size_t)b * cin * cout, s) != hipSuccess) {
        ogc_set_error("%s: memset failed", name);
        return OGC_ERR_LAUNCH;
    }
#define OGC_ML(A, C) moments_launch<A, C, AT>(b, cin, cout, hw, relu, y_prev, grad_y, pa, pb, coef2, inj, s_shift, moments, s)
    if (cout <= 16 && cin <= 16) OGC_ML(1, 1);
    else if (cout <= 32 && cin <= 16) OGC_ML(2, 1);
    else if (cout <= 32 && cin <= 32) OGC_ML(2, 2);
    else if (cin <= 16) OGC_ML(4, 1);
    else if (cin <= 32) OGC_ML(4, 2);
    else if (cout <= 32) OGC_ML(2, 4);
    else if (inj && sizeof(AT) == 4) OGC_ML(4, 2); // the pooled fp32 <4, 4> instantiation runs out of registers (256 + 34): two column tiles instead
    else OGC_ML(4, 4);
#undef OGC_ML
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

extern "C" int ogc_conv1x1_wgrad_moments(int b, int cin, int cout, int hw, int relu, const float *y_prev, const float *pa,
                                         const float *pb, const float *grad_y, float *moments, ogc_stream_t stream) {
    return wgrad_moments_impl<float>("ogc_conv1x1_wgrad_moments", b, cin, cout, hw, relu, y_prev, pa, pb, grad_y, nullptr, nullptr, 0,
                                     moments, stream);
}

extern "C" int ogc_conv1x1_wgrad_moments_h(int b, int cin, int cout, int hw, int relu, const ogc_bf16_t *y_prev, const float *pa,
                                           const float *pb, const ogc_bf16_t *grad_y, float *moments, ogc_stream_t stream) {
    return wgrad_moments_impl<ogc_bf16>("ogc_conv1x1_wgrad_moments_h", b, cin, cout, hw, relu, y_prev, pa, pb, grad_y, nullptr,
                                        nullptr, 0, moments, stream);
}

// The same with grad_y in the sparse form of ogc_group_norm_maxpool_bwd_sparse: y (B, cout, hw) is the convolution's output,
// hw = centres * nsample (nsample 16, 32 or 64).
namespace {
template <typename AT>
int wgrad_moments_pooled_impl(const char *name, int b, int cin, int cout, int hw, int relu, int nsample, const AT *y_prev,
                              const float *pa, const float *pb, const AT *y, const float *coef2, const float *inj,
                              float *moments, ogc_stream_t stream) {
    const int sh = nsample_shift(nsample);
    if (sh < 0 || hw % nsample != 0 || (((uintptr_t)coef2 | (uintptr_t)inj) & 7) != 0) {
        ogc_set_error("%s: nsample=%d must be 16, 32 or 64 and divide hw=%d", name, nsample, hw);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE(b == 0 || (coef2 && inj), "%s: null pointer", name);
    return wgrad_moments_impl<AT>(name, b, cin, cout, hw, relu, y_prev, pa, pb, y, coef2, inj, sh, moments, stream);
}
} // namespace

extern "C" int ogc_conv1x1_wgrad_moments_pooled(int b, int cin, int cout, int hw, int relu, int nsample, const float *y_prev,
                                                const float *pa, const float *pb, const float *y, const float *coef2,
                                                const float *inj, float *moments, ogc_stream_t stream) {
    return wgrad_moments_pooled_impl<float>("ogc_conv1x1_wgrad_moments_pooled", b, cin, cout, hw, relu, nsample, y_prev, pa, pb, y,
                                            coef2, inj, moments, stream);
}

extern "C" int ogc_conv1x1_wgrad_moments_pooled_h(int b, int cin, int cout, int hw, int relu, int nsample,
                                                  const ogc_bf16_t *y_prev, const float *pa, const float *pb, const ogc_bf16_t *y,
                                                  const float *coef2, const float *inj, float *moments, ogc_stream_t stream) {
    return wgrad_moments_pooled_impl<ogc_bf16>("ogc_conv1x1_wgrad_moments_pooled_h", b, cin, cout, hw, relu, nsample, y_prev, pa,
                                               pb, y, coef2, inj, moments, stream);
}

extern "C" int ogc_gn_moments_combine(int b, int cin, int cout, int hw, int groups, const float *moments, const float *w,
                                      const float *pa, const float *pb, const float *mean, const float *rstd,
                                      const float *gamma, float *grad_w, float *coef, float *grad_gamma,
                                      float *grad_beta, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 1 && cin >= 1 && cin <= 512 && cout >= 1 && hw >= 1 && groups >= 1 && groups <= 32 && cin % groups == 0,
                "ogc_gn_moments_combine: bad shape (cin <= 512, groups <= 32 dividing cin)");
    OGC_REQUIRE(moments && w && pa && pb && mean && rstd && gamma && grad_w && coef && grad_gamma && grad_beta,
                "ogc_gn_moments_combine: null pointer");
    OGC_REQUIRE(grad_beta == grad_gamma + cin, "ogc_gn_moments_combine: grad_beta must follow grad_gamma (one 2 x cin buffer)");
    if (ogc_zero_async(grad_gamma, sizeof(float) * 2 * (size_t)cin, (hipStream_t)stream) != hipSuccess) {
        ogc_set_error("ogc_gn_moments_combine: memset failed");
        return OGC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(moments_combine_kernel, dim3(b + ogc_divup((long long)cout * cin, 256)), dim3(256), 0,
                       (hipStream_t)stream, b, cin, cout, hw, groups, moments, w, pa, pb, mean, rstd, gamma, grad_w, coef,
                       grad_gamma, grad_beta);
    OGC_CHECK_LAUNCH("ogc_gn_moments_combine");
    return OGC_OK;
}

namespace {
template <typename AT>
int dgrad_adjoint_impl(const char *name, int b, int cin, int cout, int hw, int relu, const float *w, const AT *grad_y,
                       const AT *y_prev, const float *pa, const float *pb, const float *coef, const float *coef2,
                       const float *inj, int s_shift, AT *grad_prev, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && hw >= 1, "%s: bad shape", name);
    OGC_REQUIRE(w && grad_y && y_prev && pa && pb && coef && grad_prev, "%s: null pointer", name);
    if ((hw & 63) != 0 || cout > 160 || (((uintptr_t)grad_y | (uintptr_t)y_prev | (uintptr_t)grad_prev) & ogc_act_mask<AT>()) != 0) {
        ogc_set_error("%s: needs hw %% 64 == 0, cout <= 160 and 16-byte aligned tensors (hw=%d, cout=%d)", name, hw, cout);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE((long long)cin * hw < (1ll << 31) && (long long)cout * hw < (1ll << 31) && b <= 65535,
                "%s: one sample exceeds 32-bit indexing", name);
    if (b == 0) return OGC_OK;
    const int M = cin, K = cout, Kq = (K + 3) / 4;
    const size_t lds = ((size_t)64 * ogc_a_ld(Kq) + 64 * 5) * sizeof(float) +
                       (inj ? (size_t)WG_WAVES * K * (64 >> s_shift) * sizeof(float4) : 0);
    if (sizeof(AT) == 2 && !ogc_g_matmul_bf16) {
        ogc_set_error("%s: 16-bit activations need ogc_set_matmul_precision(1)", name);
        return OGC_ERR_UNSUPPORTED;
    }
    if (lds > 64 * 1024) { // (the kernels keep the default dynamic-LDS limit)
        ogc_set_error("%s: the weight tile and the pooled table need %zu bytes of LDS (> 64 KiB) at cout=%d, nsample=%d", name,
                      lds, cout, 64 >> (6 - s_shift));
        return OGC_ERR_UNSUPPORTED;
    }
    dim3 grid(ogc_divup(hw, 64 * WG_WAVES), b);
    hipStream_t s = (hipStream_t)stream;
    if constexpr (sizeof(AT) == 2) { // dense form, enough tiles: the persistent kernel (the same expressions)
        if (!inj && ogc_gemm16_adjoint_launch(b, M, K, hw, relu, w, grad_y, y_prev, pa, pb, coef, grad_prev, s)) {
            OGC_CHECK_LAUNCH(name);
            return OGC_OK;
        }
    }
    const float2 *c2 = reinterpret_cast<const float2 *>(coef2), *ij = reinterpret_cast<const float2 *>(inj);
#define OGC_DGA(KQV)                                                                                                        \
    do {                                                                                                                    \
        if (inj)                                                                                                            \
            hipLaunchKernelGGL((dgrad_adjoint_kernel<KQV, true, AT, sizeof(AT) == 2>), grid, dim3(WG_WAVES * OGC_WAVE), lds, s, M, K, hw, relu, w,  \
                               grad_y, y_prev, pa, pb, coef, c2, ij, s_shift, grad_prev);                                  \
        else                                                                                                                \
            hipLaunchKernelGGL((dgrad_adjoint_kernel<KQV, false, AT, sizeof(AT) == 2>), grid, dim3(WG_WAVES * OGC_WAVE), lds, s, M, K, hw, relu, w, \
                               grad_y, y_prev, pa, pb, coef, c2, ij, 0, grad_prev);                                        \
    } while (0)
    if (Kq <= 8) OGC_DGA(8);
    else if (Kq <= 16) OGC_DGA(16);
    else if (Kq <= 25) OGC_DGA(25);
    else if (Kq <= 33) OGC_DGA(33);
    else OGC_DGA(40);
#undef OGC_DGA
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

extern "C" int ogc_conv1x1_dgrad_adjoint(int b, int cin, int cout, int hw, int relu, const float *w, const float *grad_y,
                                         const float *y_prev, const float *pa, const float *pb, const float *coef,
                                         float *grad_prev, ogc_stream_t stream) {
    return dgrad_adjoint_impl<float>("ogc_conv1x1_dgrad_adjoint", b, cin, cout, hw, relu, w, grad_y, y_prev, pa, pb, coef, nullptr,
                                     nullptr, 0, grad_prev, stream);
}

extern "C" int ogc_conv1x1_dgrad_adjoint_h(int b, int cin, int cout, int hw, int relu, const float *w, const ogc_bf16_t *grad_y,
                                           const ogc_bf16_t *y_prev, const float *pa, const float *pb, const float *coef,
                                           ogc_bf16_t *grad_prev, ogc_stream_t stream) {
    return dgrad_adjoint_impl<ogc_bf16>("ogc_conv1x1_dgrad_adjoint_h", b, cin, cout, hw, relu, w, grad_y, y_prev, pa, pb, coef,
                                        nullptr, nullptr, 0, grad_prev, stream);
}

// The same with grad_y in the sparse form of ogc_group_norm_maxpool_bwd_sparse (y: the convolution's output).
namespace {
template <typename AT>
int dgrad_adjoint_pooled_impl(const char *name, int b, int cin, int cout, int hw, int relu, int nsample, const float *w,
                              const AT *y, const float *coef2, const float *inj, const AT *y_prev, const float *pa,
                              const float *pb, const float *coef, AT *grad_prev, ogc_stream_t stream) {
    const int sh = nsample_shift(nsample);
    if (sh < 0 || hw % nsample != 0 || (((uintptr_t)coef2 | (uintptr_t)inj) & 7) != 0) {
        ogc_set_error("%s: nsample=%d must be 16, 32 or 64 and divide hw=%d", name, nsample, hw);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE(b == 0 || (coef2 && inj), "%s: null pointer", name);
    return dgrad_adjoint_impl<AT>(name, b, cin, cout, hw, relu, w, y, y_prev, pa, pb, coef, coef2, inj, sh, grad_prev, stream);
}
} // namespace

extern "C" int ogc_conv1x1_dgrad_adjoint_pooled(int b, int cin, int cout, int hw, int relu, int nsample, const float *w,
                                                const float *y, const float *coef2, const float *inj, const float *y_prev,
                                                const float *pa, const float *pb, const float *coef, float *grad_prev,
                                                ogc_stream_t stream) {
    return dgrad_adjoint_pooled_impl<float>("ogc_conv1x1_dgrad_adjoint_pooled", b, cin, cout, hw, relu, nsample, w, y, coef2, inj,
                                            y_prev, pa, pb, coef, grad_prev, stream);
}

extern "C" int ogc_conv1x1_dgrad_adjoint_pooled_h(int b, int cin, int cout, int hw, int relu, int nsample, const float *w,
                                                  const ogc_bf16_t *y, const float *coef2, const float *inj,
                                                  const ogc_bf16_t *y_prev, const float *pa, const float *pb, const float *coef,
                                                  ogc_bf16_t *grad_prev, ogc_stream_t stream) {
    return dgrad_adjoint_pooled_impl<ogc_bf16>("ogc_conv1x1_dgrad_adjoint_pooled_h", b, cin, cout, hw, relu, nsample, w, y, coef2,
                                               inj, y_prev, pa, pb, coef, grad_prev, stream);
}
