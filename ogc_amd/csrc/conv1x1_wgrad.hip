// conv1x1.hip — weight gradient of the per-point (1x1) convolutions of the shared MLPs, on the fp32 MFMA pipe.
//
//     dW[co, ci] = sum_b sum_p dY[b, co, p] * X[b, ci, p]          X (B, Cin, HW), dY (B, Cout, HW), fp32, NCHW
//
// Replaces the weight-gradient half of the nn.Conv2d(kernel 1x1, bias=False) layers of SharedMLP
// (reference: utils/nn_util.py:45-85, :155-172).  MIOpen serves this shape with an NHWC implicit-GEMM kernel wrapped
// in two full-tensor NCHW->NHWC transposes (x and dy): on the C4 training step that is ~3 ms of transposes plus ~2 ms
// of igemm per step.  Here the tensors are read once, in place:
//   * GEMM view: M = Cout, N = Cin, K = B*HW (millions) — a tiny output and a huge reduction, i.e. a streaming,
//     HBM-bound kernel; v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, 157 TF peak) keeps the math off the critical path;
//   * operand layout without any shuffle: for a step of 16 positions, lane (i = l & 15, k = l >> 4) loads ONE float4
//     = row (c0 + i), positions pb + 4k .. 4k+3.  MFMA k-slot k of sub-step s is position pb + 4k + s for BOTH
//     operands, so component s of the two float4s are directly the A and B operands of sub-step s;
//   * a wave owns a (16*COB) x (16*CIB) tile of dW in registers and strides over the positions; the next step's
//     loads are issued before the current step's MFMAs; the four waves of a workgroup are reduced through LDS,
//     workgroups through fp32 atomics into dW (zeroed by this entry point).
#include "conv1x1_shared.h"
#include "act_io.h"

namespace {

// PRO: the layer's input was never materialised — x holds the previous layer's raw convolution output and the operand
// is act(pa[b, ci] * x + pb[b, ci]), recomputed while loading (see conv1x1_gemm_kernel).
// POOLED: dy is not stored — `dy` holds y, the convolution's raw output, and the gradient of the pooled GroupNorm behind it is
// rebuilt while y is loaded:  g_y[row, pos] = fmaf(c2, y, c3) + (pos % S == arg ? ag : 0)  with (c2, c3) = coef2[b, row] and
// (ag, arg) = inj[b, row, pos / S] (ogc_group_norm_maxpool_bwd_sparse; a step's 16 positions lie inside one neighbourhood,
// S = 16, 32, 64) — the expression of gn_maxpool_bwd_dx_kernel, bit for bit.
// XT / YT: element types of x and dy (float / ogc_bf16: act_io.h).
// A workgroup's share of dW[row][col].  Ordinarily one fp32 atomic per element and workgroup (the workgroups of a tile split the
// positions among them: blockIdx.x); with slab != 0 (the deterministic mode, det.hip) a plain store into slab blockIdx.x of a
// scratch buffer instead — zeros included — and ogc_det_reduce_f32 adds the slabs to dW in blockIdx.x order afterwards.
__device__ __forceinline__ void wgrad_emit(float *__restrict__ dw, long long slab, int row, int col, int cout, int cin, float v) {
    if (row < cout && col < cin) {
        if (slab) dw[(size_t)blockIdx.x * slab + (size_t)row * cin + col] = v;
        else if (v != 0.0f) unsafeAtomicAdd(dw + (size_t)row * cin + col, v);
    }
}

template <int COB, int CIB, bool PRO, bool BF, bool POOLED = false, typename XT = float, typename YT = float>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void conv1x1_wgrad_kernel(int batch, int cin, int cout, int hw,
                                                                           int steps_per_wave,
                                                                           const XT *__restrict__ x,
                                                                           const YT *__restrict__ dy,
                                                                           float *__restrict__ dw,
                                                                           const float *__restrict__ aff_a,
                                                                           const float *__restrict__ aff_b, int pro_relu,
                                                                           const float2 *__restrict__ coef2 = nullptr,
                                                                           const float2 *__restrict__ inj = nullptr,
                                                                           int s_shift = 0, long long slab = 0) {
    // the four waves' partial tiles, one slab each (plain stores: ds_add_f32 sustains well under one lane per cycle), summed
    // by the threads that send them on
    __shared__ float red[WG_WAVES][COB * CIB * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, k = lane >> 4;
    const int co0 = blockIdx.y * (16 * COB), ci0 = blockIdx.z * (16 * CIB);
    const int steps_per_img = hw >> 4;
    const long long nsteps = (long long)batch * steps_per_img;
    const long long first = ((long long)blockIdx.x * WG_WAVES + wave) * steps_per_wave;
    // this wavefront's steps: [first, first + mine) — everything about the walk is wave-uniform (SALU, scalar branches)
    const int mine = (int)max(0LL, min((long long)steps_per_wave, nsteps - first));

    v4f acc[COB][CIB];
#pragma unroll
    for (int a = 0; a < COB; ++a)
#pragma unroll
        for (int c = 0; c < CIB; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};

    // The wave walks consecutive steps, so (image, position) advance incrementally: one division per wave instead of a
    // 64-bit division per step (which cost more issue slots than the step's 64 MFMAs).
    const long long start = min(first, nsteps - 1);
    int cur_b = (int)(start / steps_per_img);
    int cur_off = (int)(start - (long long)cur_b * steps_per_img);
    int left = mine - 1; // steps after the current one; the walk stays on the last step once they are used up
    int yrow[COB], xrow[CIB], coef[CIB]; // one sample's activation fits 32-bit offsets (checked by the entry point)
    constexpr int NP = POOLED ? COB : 1;
    int orow[NP];
    const int centres = hw >> s_shift, smask = (1 << s_shift) - 1;
#pragma unroll
    for (int a = 0; a < COB; ++a) {
        yrow[a] = min(co0 + a * 16 + i, cout - 1) * hw;
        if (POOLED) orow[a] = min(co0 + a * 16 + i, cout - 1);
    }
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
        coef[c] = min(ci0 + c * 16 + i, cin - 1);
        xrow[c] = coef[c] * hw;
    }
    // EVERY load of a step is unconditional and of the same shape: rows beyond the tensors are clamped onto the last row
    // (they only feed rows / columns of the tile that are never stored), steps beyond the wave's share re-read its last
    // step and are not computed with.  A load under a per-lane condition (`row < cout ? *p : 0`) becomes its own
    // exec-masked block and the compiler then waits for ALL outstanding loads wherever it needs one — which serialised
    // the ping-pong below: the next step's loads were waited for before the current step's MFMAs (50 % of the time of
    // this kernel at one wavefront per SIMD).  PRO: the affine map of this lane's input channels travels WITH the step
    // (eight cached dword loads more) for the same reason: loaded only at image changes, its wait was a vmcnt(0).
    auto load = [&](float4(&yv)[COB], float4(&xv)[CIB], float(&fa)[CIB], float(&fb)[CIB], float2(&cc)[NP], float2(&jv)[NP],
                    int &jpos) { // current step, then advance
        const int pb = cur_off * 16 + 4 * k;
        const YT *yb_ = dy + (size_t)cur_b * cout * hw + pb;
        const XT *xb_ = x + (size_t)cur_b * cin * hw + pb;
#pragma unroll
        for (int a = 0; a < COB; ++a) yv[a] = ogc_ld4(yb_ + yrow[a]);
        if constexpr (POOLED) { // (travels with the step like the affine map below: the sample may change from step to step)
#pragma unroll
            for (int a = 0; a < COB; ++a) {
                const size_t r = (size_t)cur_b * cout + orow[a];
                cc[a] = coef2[r];
                jv[a] = inj[r * centres + (pb >> s_shift)];
            }
            jpos = pb & smask; // this lane's first position inside the neighbourhood
        }
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            xv[c] = ogc_ld4(xb_ + xrow[c]);
            if (PRO) {
                fa[c] = aff_a[(size_t)cur_b * cin + coef[c]];
                fb[c] = aff_b[(size_t)cur_b * cin + coef[c]];
            }
        }
        if (left > 0) {
            --left;
            if (++cur_off == steps_per_img) { cur_off = 0; ++cur_b; }
        }
    };
    auto fma16 = [&](const float4(&yraw)[COB], const float4(&xraw)[CIB], const float(&fa)[CIB], const float(&fb)[CIB],
                     const float2(&cc)[NP], const float2(&jv)[NP], int jpos) {
        float4 yv[COB];
#pragma unroll
        for (int a = 0; a < COB; ++a) {
            yv[a] = yraw[a];
            if constexpr (POOLED) {
                const int rel = __float_as_int(jv[a].y) - jpos;
                const float ag = jv[a].x;
                yv[a].x = fmaf(cc[a].x, yraw[a].x, cc[a].y) + (rel == 0 ? ag : 0.f);
                yv[a].y = fmaf(cc[a].x, yraw[a].y, cc[a].y) + (rel == 1 ? ag : 0.f);
                yv[a].z = fmaf(cc[a].x, yraw[a].z, cc[a].y) + (rel == 2 ? ag : 0.f);
                yv[a].w = fmaf(cc[a].x, yraw[a].w, cc[a].y) + (rel == 3 ? ag : 0.f);
            }
        }
        float4 xv[CIB];
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            xv[c] = xraw[c];
            if (PRO) {
                float4 v = xraw[c];
                v.x = fmaf(fa[c], v.x, fb[c]); v.y = fmaf(fa[c], v.y, fb[c]);
                v.z = fmaf(fa[c], v.z, fb[c]); v.w = fmaf(fa[c], v.w, fb[c]);
                if (pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                xv[c] = v;
            }
        }
        if constexpr (BF) {
            // the lane's four consecutive positions are the four k-slots 4k .. 4k+3 of ONE 16x16x16 MFMA
            v4s yb16[COB], xb16[CIB];
#pragma unroll
            for (int a = 0; a < COB; ++a) yb16[a] = ogc_pack_bf16(yv[a].x, yv[a].y, yv[a].z, yv[a].w);
#pragma unroll
            for (int c = 0; c < CIB; ++c) xb16[c] = ogc_pack_bf16(xv[c].x, xv[c].y, xv[c].z, xv[c].w);
#pragma unroll
            for (int a = 0; a < COB; ++a)
#pragma unroll
                for (int c = 0; c < CIB; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(yb16[a], xb16[c], acc[a][c], 0, 0, 0);
        } else {
#pragma unroll
            for (int a = 0; a < COB; ++a)
#pragma unroll
                for (int c = 0; c < CIB; ++c) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].x, xv[c].x, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].y, xv[c].y, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].z, xv[c].z, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].w, xv[c].w, acc[a][c], 0, 0, 0);
                }
        }
    };

    float4 ya[COB], xa[CIB], yb[COB], xb[CIB];
    float faa[CIB], fba[CIB], fab[CIB], fbb[CIB];
    float2 cca[NP], ccb[NP], jva[NP], jvb[NP];
    int jpa = 0, jpb = 0;
    load(ya, xa, faa, fba, cca, jva, jpa);
    int s = 0;
    for (; s + 1 < mine; s += 2) {               // ping-pong registers: next step's loads fly during the MFMAs.  No branch inside
        load(yb, xb, fab, fbb, ccb, jvb, jpb);   // the loop body: with one, the accumulators travelled AGPR -> VGPR -> AGPR every round
        fma16(ya, xa, faa, fba, cca, jva, jpa);
        load(ya, xa, faa, fba, cca, jva, jpa);
        fma16(yb, xb, fab, fbb, ccb, jvb, jpb);
    }
    if (s < mine) fma16(ya, xa, faa, fba, cca, jva, jpa);

    // C/D layout of 16x16x4: lane l holds rows (l >> 4) * 4 + r (r = 0..3) of column l & 15
#pragma unroll
    for (int a = 0; a < COB; ++a)
#pragma unroll
        for (int c = 0; c < CIB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(a * CIB + c) * 256 + (k * 4 + r) * 16 + i] = acc[a][c][r];
    __syncthreads();
    for (int t = threadIdx.x; t < COB * CIB * 256; t += WG_WAVES * OGC_WAVE) {
        const int blk = t >> 8, a = blk / CIB, c = blk % CIB;
        const int row = co0 + a * 16 + ((t & 255) >> 4), col = ci0 + c * 16 + (t & 15);
        const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        wgrad_emit(dw, slab, row, col, cout, cin, v);
    }
}

// ---- the register-tile weight gradient from 16-bit gradients ------------------------------------------------------------------
// conv1x1_wgrad_kernel with dy stored as bf16 moves half the bytes and is no faster: per 16 positions it issues one load per row
// block — sixteen rows x 32 bytes: the vector memory pipe works per row, not per byte — plus the per-row coefficient loads.  Here
// a step is 32 positions: lane (i, k) loads row i, positions pb + 8 k .. 8 k + 7 (sixteen bytes of a bf16 tensor, two float4 of
// an fp32 x), and a lane's eight positions are the eight k-slots of ONE v_mfma_f32_16x16x32_bf16 (gfx950; bf16 operands only: the
// 16-bit entry points require ogc_set_matmul_precision(1)).  PRO / POOLED as in conv1x1_wgrad_kernel, the same expressions.
__device__ __forceinline__ void ogc_unpack8(const uint4 &u, float (&f)[8]) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xFFFF0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xFFFF0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xFFFF0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
struct Raw8 { uint4 a, b; }; // eight positions as loaded: bf16 -> a; fp32 -> a, b
__device__ __forceinline__ void ogc_ld8(const ogc_bf16 *p, Raw8 &r) { r.a = *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ void ogc_ld8(const float *p, Raw8 &r) {
    r.a = *reinterpret_cast<const uint4 *>(p);
    r.b = *reinterpret_cast<const uint4 *>(p + 4);
}
template <typename T>
__device__ __forceinline__ void ogc_widen8(const Raw8 &r, float (&f)[8]) {
    if constexpr (sizeof(T) == 2) {
        ogc_unpack8(r.a, f);
    } else {
        f[0] = __uint_as_float(r.a.x); f[1] = __uint_as_float(r.a.y); f[2] = __uint_as_float(r.a.z); f[3] = __uint_as_float(r.a.w);
        f[4] = __uint_as_float(r.b.x); f[5] = __uint_as_float(r.b.y); f[6] = __uint_as_float(r.b.z); f[7] = __uint_as_float(r.b.w);
    }
}

template <int COB, int CIB, bool PRO, bool POOLED, typename XT>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void conv1x1_wgrad16_kernel(int batch, int cin, int cout, int hw,
                                                                             int steps_per_wave, const XT *__restrict__ x,
                                                                             const ogc_bf16 *__restrict__ dy,
                                                                             float *__restrict__ dw,
                                                                             const float *__restrict__ aff_a,
                                                                             const float *__restrict__ aff_b, int pro_relu,
                                                                             const float2 *__restrict__ coef2,
                                                                             const float2 *__restrict__ inj, int s_shift,
                                                                             long long slab) {
    __shared__ float red[WG_WAVES][COB * CIB * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, k = lane >> 4;
    const int co0 = blockIdx.y * (16 * COB), ci0 = blockIdx.z * (16 * CIB);
    const int steps_per_img = hw >> 5;
    const long long nsteps = (long long)batch * steps_per_img;
    const long long first = ((long long)blockIdx.x * WG_WAVES + wave) * steps_per_wave;
    const int mine = (int)max(0LL, min((long long)steps_per_wave, nsteps - first));

    v4f acc[COB][CIB];
#pragma unroll
    for (int a = 0; a < COB; ++a)
#pragma unroll
        for (int c = 0; c < CIB; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
    const long long start = min(first, nsteps - 1);
    int cur_b = (int)(start / steps_per_img);
    int cur_off = (int)(start - (long long)cur_b * steps_per_img);
    int left = mine - 1;
    int yrow[COB], xrow[CIB], coef[CIB];
    constexpr int NP = POOLED ? COB : 1;
    int orow[NP];
    const int centres = hw >> s_shift, smask = (1 << s_shift) - 1;
#pragma unroll
    for (int a = 0; a < COB; ++a) {
        yrow[a] = min(co0 + a * 16 + i, cout - 1) * hw;
        if (POOLED) orow[a] = min(co0 + a * 16 + i, cout - 1);
    }
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
        coef[c] = min(ci0 + c * 16 + i, cin - 1);
        xrow[c] = coef[c] * hw;
    }
    // (unconditional loads of one shape; the affine map and the sparse gradient's entries travel with the step: see
    // conv1x1_wgrad_kernel)
    auto load = [&](Raw8(&yv)[COB], Raw8(&xv)[CIB], float(&fa)[CIB], float(&fb)[CIB], float2(&cc)[NP], float2(&jv)[NP], int &jpos) {
        const int pb = cur_off * 32 + 8 * k;
        const ogc_bf16 *yb_ = dy + (size_t)cur_b * cout * hw + pb;
        const XT *xb_ = x + (size_t)cur_b * cin * hw + pb;
#pragma unroll
        for (int a = 0; a < COB; ++a) ogc_ld8(yb_ + yrow[a], yv[a]);
        if constexpr (POOLED) { // a lane's eight positions lie inside one neighbourhood (S >= 16)
#pragma unroll
            for (int a = 0; a < COB; ++a) {
                const size_t r = (size_t)cur_b * cout + orow[a];
                cc[a] = coef2[r];
                jv[a] = inj[r * centres + (pb >> s_shift)];
            }
            jpos = pb & smask;
        }
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            ogc_ld8(xb_ + xrow[c], xv[c]);
            if (PRO) {
                fa[c] = aff_a[(size_t)cur_b * cin + coef[c]];
                fb[c] = aff_b[(size_t)cur_b * cin + coef[c]];
            }
        }
        if (left > 0) {
            --left;
            if (++cur_off == steps_per_img) { cur_off = 0; ++cur_b; }
        }
    };
    auto fma32 = [&](const Raw8(&yraw)[COB], const Raw8(&xraw)[CIB], const float(&fa)[CIB], const float(&fb)[CIB],
                     const float2(&cc)[NP], const float2(&jv)[NP], int jpos) {
        v4s y0[COB], y1[COB], x0[CIB], x1[CIB];
#pragma unroll
        for (int a = 0; a < COB; ++a) {
            if constexpr (POOLED) {
                float f[8];
                ogc_unpack8(yraw[a].a, f);
                const int rel = __float_as_int(jv[a].y) - jpos;
                const float ag = jv[a].x;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaf(cc[a].x, f[e], cc[a].y) + (rel == e ? ag : 0.f);
                y0[a] = ogc_pack_bf16_rr(f[0], f[1], f[2], f[3]);
                y1[a] = ogc_pack_bf16_rr(f[4], f[5], f[6], f[7]);
            } else { // dense gradient: the stored bits are the operand
                y0[a] = __builtin_bit_cast(v4s, make_uint2(yraw[a].a.x, yraw[a].a.y));
                y1[a] = __builtin_bit_cast(v4s, make_uint2(yraw[a].a.z, yraw[a].a.w));
            }
        }
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
            if constexpr (!PRO && sizeof(XT) == 2) {
                x0[c] = __builtin_bit_cast(v4s, make_uint2(xraw[c].a.x, xraw[c].a.y));
                x1[c] = __builtin_bit_cast(v4s, make_uint2(xraw[c].a.z, xraw[c].a.w));
            } else {
                float f[8];
                ogc_widen8<XT>(xraw[c], f);
                if (PRO) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = fmaf(fa[c], f[e], fb[c]);
                        if (pro_relu) f[e] = fmaxf(f[e], 0.f);
                    }
                }
                x0[c] = ogc_pack_bf16_rr(f[0], f[1], f[2], f[3]);
                x1[c] = ogc_pack_bf16_rr(f[4], f[5], f[6], f[7]);
            }
        }
#pragma unroll
        for (int a = 0; a < COB; ++a)
#pragma unroll
            for (int c = 0; c < CIB; ++c) {
                acc[a][c] = ogc_mfma_bf16_k32(y0[a], y1[a], x0[c], x1[c], acc[a][c]); // v_mfma_f32_16x16x32_bf16 (gfx950)
            }
    };

    Raw8 ya[COB], xa[CIB], yb[COB], xb[CIB];
    float faa[CIB], fba[CIB], fab[CIB], fbb[CIB];
    float2 cca[NP], ccb[NP], jva[NP], jvb[NP];
    int jpa = 0, jpb = 0;
    load(ya, xa, faa, fba, cca, jva, jpa);
    int s = 0;
    for (; s + 1 < mine; s += 2) {
        load(yb, xb, fab, fbb, ccb, jvb, jpb);
        fma32(ya, xa, faa, fba, cca, jva, jpa);
        load(ya, xa, faa, fba, cca, jva, jpa);
        fma32(yb, xb, fab, fbb, ccb, jvb, jpb);
    }
    if (s < mine) fma32(ya, xa, faa, fba, cca, jva, jpa);

#pragma unroll
    for (int a = 0; a < COB; ++a)
#pragma unroll
        for (int c = 0; c < CIB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(a * CIB + c) * 256 + (k * 4 + r) * 16 + i] = acc[a][c][r];
    __syncthreads();
    for (int t = threadIdx.x; t < COB * CIB * 256; t += WG_WAVES * OGC_WAVE) {
        const int blk = t >> 8, a = blk / CIB, c = blk % CIB;
        const int row = co0 + a * 16 + ((t & 255) >> 4), col = ci0 + c * 16 + (t & 15);
        const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        wgrad_emit(dw, slab, row, col, cout, cin, v);
    }
}

// Deterministic mode: `splits` zeroed slabs of cout x cin floats in the stream's scratch (a workgroup without positions leaves
// its slab alone); nullptr + g_wgrad_det_failed when the scratch cannot be had.  Otherwise dw itself and slab 0.
thread_local bool g_wgrad_det_failed = false;
float *wgrad_det_slabs(float *dw, int splits, int cin, int cout, long long &slab, hipStream_t s) {
    slab = 0;
    if (!ogc_deterministic()) return dw;
    const size_t words = (size_t)splits * cout * cin;
    float *part = static_cast<float *>(ogc_det_scratch(s, words * sizeof(float)));
    if (!part) {
        g_wgrad_det_failed = true;
        return nullptr;
    }
    const size_t blocks = (words + 255) / 256;
    hipLaunchKernelGGL(ogc_zero_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s,
                       reinterpret_cast<uint32_t *>(part), words);
    slab = (long long)cout * cin;
    return part;
}
void wgrad_det_finish(float *dw, const float *part, int splits, long long slab, hipStream_t s) {
    if (slab && ogc_det_reduce_f32(dw, part, splits, slab, 1, s) != hipSuccess) g_wgrad_det_failed = true;
}

template <int COB, int CIB, typename XT = float, typename YT = float>
void wgrad_launch(int b, int cin, int cout, int hw, const XT *x, const YT *dy, float *dw, const float *pa,
                  const float *pb, int pro_relu, hipStream_t s, const float2 *coef2 = nullptr, const float2 *inj = nullptr,
                  int s_shift = 0) {
    constexpr bool Y16 = sizeof(YT) == 2;
    const bool wide = Y16 && (hw & 31) == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0; // 32-position steps
    const long long nsteps = (long long)b * (wide ? hw >> 5 : hw >> 4);
    const int tiles = ogc_divup(cout, 16 * COB) * ogc_divup(cin, 16 * CIB);
    // ~2048 waves over the chip per tile pair, but at least 8 steps (128 positions) per wave
    // Wavefronts over the chip per tile pair.  The 64x64 tiles (COB = CIB = 4) run best with ~1024 wavefronts in total —
    // one workgroup per CU, twice the positions per wavefront, half as many LDS reductions and atomic epilogues
    // (64 -> 64: 0.169 -> 0.150 ms, 128 -> 128: 0.258 -> 0.222 ms) — unless the tile grid is ragged (131 -> 128: six
    // tiles, two of them nearly empty), where the finer split balances better; the small tiles keep ~2048.
    long long waves = ((COB * CIB == 16 && tiles <= 4) ? 1024 : 2048) / tiles;
    if (waves < 256) waves = 256;
    long long spw = (nsteps + waves - 1) / waves;
    if (spw < 8) spw = 8;
    spw = (spw + 1) / 2 * 2;
    const int wgs = (int)((nsteps + spw * WG_WAVES - 1) / (spw * WG_WAVES));
    dim3 grid(wgs, ogc_divup(cout, 16 * COB), ogc_divup(cin, 16 * CIB));
    long long slab = 0;
    float *const dst = wgrad_det_slabs(dw, wgs, cin, cout, slab, s); // (dw itself unless the deterministic mode is on)
    if (!dst) return;
    if constexpr (Y16) {
        if (wide) {
#define OGC_WGRAD16(PROV, POOLV)                                                                                               \
    hipLaunchKernelGGL((conv1x1_wgrad16_kernel<COB, CIB, PROV, POOLV, XT>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, b, cin, cout, hw, \
                       (int)spw, x, dy, dst, pa, pb, pro_relu, coef2, inj, s_shift, slab)
            if (inj) OGC_WGRAD16(true, true);
            else if (pa) OGC_WGRAD16(true, false);
            else OGC_WGRAD16(false, false);
#undef OGC_WGRAD16
            wgrad_det_finish(dw, dst, wgs, slab, s);
            return;
        }
    }
#define OGC_WGRAD(PROV, BFV)                                                                                          \
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<COB, CIB, PROV, BFV, false, XT, YT>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, b, cin, cout, \
                       hw, (int)spw, x, dy, dst, pa, pb, pro_relu, nullptr, nullptr, 0, slab)
    if (inj) { // pooled form of dy (fp32 operands, previous layer's norm folded in): see POOLED
        // (fp32 operands for fp32 tensors whatever the precision switch says; 16-bit tensors: bf16 operands)
        hipLaunchKernelGGL((conv1x1_wgrad_kernel<COB, CIB, true, sizeof(YT) == 2, true, XT, YT>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, b, cin, cout,
                           hw, (int)spw, x, dy, dst, pa, pb, pro_relu, coef2, inj, s_shift, slab);
    } else if (g_matmul_bf16 || sizeof(XT) == 2 || sizeof(YT) == 2) { // (16-bit tensors: checked to come with bf16 operands)
        if (pa) OGC_WGRAD(true, true);
        else OGC_WGRAD(false, true);
    } else {
        if constexpr (sizeof(XT) == 4 && sizeof(YT) == 4) {
            if (pa) OGC_WGRAD(true, false);
            else OGC_WGRAD(false, false);
        }
    }
#undef OGC_WGRAD
    wgrad_det_finish(dw, dst, wgs, slab, s);
}


// ---- the same weight gradient for WIDE layers (cin, cout >= 128): a 128 x 128 tile of dW per workgroup, operands shared through LDS
// conv1x1_wgrad_kernel gives every wavefront a 64 x 64 tile and its own operand loads: at 128 -> 256 channels that is 8 tile
// pairs, every dy element fetched by two of them and every x element by four — 2.1 GB through the L2s for 0.8 GB of tensors, at
// 81 TFLOP/s.  Here the four wavefronts of a workgroup take the four 64 x 64 quarters of one 128 x 128 tile and walk the SAME
// positions: a stage is 32 positions of 128 dy rows and 128 x rows (32 KiB), loaded ONCE by the workgroup (thread t: the 16-byte
// piece t & 7 of rows (t >> 3) + 32 j — full 128-byte lines), transformed once (PRO: the previous layer's GroupNorm + ReLU on x;
// POOLED: the pooled GroupNorm's gradient rebuilt from y, as in the kernel above, bit for bit) and parked in LDS, from where
// every wavefront reads its 64 + 64 rows in the MFMA operand layout of the kernel above (lane (i, k): row i, positions 4k .. 4k + 3).
// Stages are double-buffered: the loads of stage s + 1 are issued before the 128 MFMAs of stage s, their transform and LDS
// write sit between the stage's two 16-position halves, one barrier per stage.  Two workgroups per CU (72 KiB of LDS each), so
// one's barrier is the other's matrix time.  dy is read once per 128 input channels, x once per 128 output channels.
constexpr int WS_POS = 32;              // positions per stage
constexpr int WS_LD = WS_POS + 4;       // row stride in LDS (floats): 144 bytes, 16-byte aligned, rows spread over the banks

// (Measured, 16 x 32768 positions: 128 -> 256 channels 0.480 -> 0.367 ms = 94 TFLOP/s, 256 -> 128 0.439 -> 0.354, 128 -> 128
// 0.195 -> 0.198, with the previous layer's norm folded in 0.226 -> 0.202.  A 256 x 128 tile on eight wavefronts — both tensors
// read from memory exactly once — gave the same 0.367 ms: the kernel is no longer bound by the operand traffic.)
// BF: operands rounded to bf16 on v_mfma_f32_16x16x16_bf16 (ogc_set_matmul_precision), as in conv1x1_wgrad_kernel.
template <bool PRO, bool POOLED, bool BF = false, typename XT = float, typename YT = float>
__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_shared_kernel(int batch, int cin, int cout, int hw, int stages_per_wg,
                                                                      const XT *__restrict__ x, const YT *__restrict__ dy,
                                                                      float *__restrict__ dw, const float *__restrict__ aff_a,
                                                                      const float *__restrict__ aff_b, int pro_relu,
                                                                      const float2 *__restrict__ coef2,
                                                                      const float2 *__restrict__ inj, int s_shift, long long slab) {
    constexpr int YROWS = 128, WS_ROWS = YROWS + 128, YJ = 4, XJ = 4, NJ = YJ + XJ; // pieces per thread: dy, x
    constexpr int RSTEP = 32;                                        // rows between a thread's pieces
    extern __shared__ __attribute__((aligned(16))) float ws_lds[]; // [2][WS_ROWS][WS_LD]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = lane & 15, k = lane >> 4;
    const int co0 = blockIdx.y * YROWS, ci0 = blockIdx.z * 128;
    const int half_r = wave >> 1, half_c = wave & 1;                 // this wavefront's 64 x 64 part of the tile
    const int stages_per_img = hw / WS_POS;
    const long long nstages = (long long)batch * stages_per_img;
    const long long first = (long long)blockIdx.x * stages_per_wg;
    const int mine = (int)max(0LL, min((long long)stages_per_wg, nstages - first));
    if (mine == 0) return;                                           // (workgroup-uniform)

    // my pieces of a stage: piece q of rows rr + RSTEP j (j < YJ: dy rows, then XJ x rows), clamped onto the tensors
    const int q = t & 7, rr = t >> 3;
    int grow[NJ], lrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = rr + RSTEP * (j < YJ ? j : j - YJ);
        grow[j] = j < YJ ? min(co0 + r, cout - 1) : min(ci0 + r, cin - 1);
        lrow[j] = (j < YJ ? r : YROWS + r) * WS_LD + 4 * q;
    }
    const int centres = POOLED ? hw >> s_shift : 0;
    int cur_b = (int)(first / stages_per_img);
    int cur_off = (int)(first - (long long)cur_b * stages_per_img);

    float4 raw[NJ];
    float fa[XJ], fb[XJ];
    float2 cc[YJ], jv[YJ];
    int jpos = 0;
    auto fetch = [&]() { // the current stage into registers, then advance
        const int pos = cur_off * WS_POS + 4 * q;
        const YT *yb_ = dy + (size_t)cur_b * cout * hw + pos;
        const XT *xb_ = x + (size_t)cur_b * cin * hw + pos;
#pragma unroll
        for (int j = 0; j < YJ; ++j) raw[j] = ogc_ld4(yb_ + (size_t)grow[j] * hw);
#pragma unroll
        for (int j = YJ; j < NJ; ++j) raw[j] = ogc_ld4(xb_ + (size_t)grow[j] * hw);
        if constexpr (POOLED) {
#pragma unroll
            for (int j = 0; j < YJ; ++j) {
                const size_t r = (size_t)cur_b * cout + grow[j];
                cc[j] = coef2[r];
                jv[j] = inj[r * centres + (pos >> s_shift)];
            }
            jpos = pos & ((1 << s_shift) - 1);
        }
        if constexpr (PRO) {
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                fa[j] = aff_a[(size_t)cur_b * cin + grow[YJ + j]];
                fb[j] = aff_b[(size_t)cur_b * cin + grow[YJ + j]];
            }
        }
        if (++cur_off == stages_per_img) { cur_off = 0; ++cur_b; }
    };
    auto park = [&](float *buf) { // transform (the expressions of conv1x1_wgrad_kernel) and store to LDS
#pragma unroll
        for (int j = 0; j < YJ; ++j) {
            float4 v = raw[j];
            if constexpr (POOLED) {
                const int rel = __float_as_int(jv[j].y) - jpos;
                const float ag = jv[j].x;
                v.x = fmaf(cc[j].x, v.x, cc[j].y) + (rel == 0 ? ag : 0.f);
                v.y = fmaf(cc[j].x, v.y, cc[j].y) + (rel == 1 ? ag : 0.f);
                v.z = fmaf(cc[j].x, v.z, cc[j].y) + (rel == 2 ? ag : 0.f);
                v.w = fmaf(cc[j].x, v.w, cc[j].y) + (rel == 3 ? ag : 0.f);
            }
            *reinterpret_cast<float4 *>(buf + lrow[j]) = v;
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            float4 v = raw[YJ + j];
            if constexpr (PRO) {
                v.x = fmaf(fa[j], v.x, fb[j]); v.y = fmaf(fa[j], v.y, fb[j]);
                v.z = fmaf(fa[j], v.z, fb[j]); v.w = fmaf(fa[j], v.w, fb[j]);
                if (pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            *reinterpret_cast<float4 *>(buf + lrow[YJ + j]) = v;
        }
    };

    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int yoff = (half_r * 64 + i) * WS_LD + 4 * k, xoff = (YROWS + half_c * 64 + i) * WS_LD + 4 * k;
    auto half_stage = [&](const float *buf, int h) { // 16 positions: 64 MFMAs
        float4 yv[4], xv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) yv[a] = *reinterpret_cast<const float4 *>(buf + yoff + a * 16 * WS_LD + h * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[c] = *reinterpret_cast<const float4 *>(buf + xoff + c * 16 * WS_LD + h * 16);
        if constexpr (BF) {
            // the lane's four consecutive positions are the four k-slots 4k .. 4k+3 of ONE 16x16x16 MFMA
            v4s yb16[4], xb16[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) yb16[a] = ogc_pack_bf16(yv[a].x, yv[a].y, yv[a].z, yv[a].w);
#pragma unroll
            for (int c = 0; c < 4; ++c) xb16[c] = ogc_pack_bf16(xv[c].x, xv[c].y, xv[c].z, xv[c].w);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(yb16[a], xb16[c], acc[a][c], 0, 0, 0);
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].x, xv[c].x, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].y, xv[c].y, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].z, xv[c].z, acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a].w, xv[c].w, acc[a][c], 0, 0, 0);
                }
        }
    };

    float *buf0 = ws_lds, *buf1 = ws_lds + WS_ROWS * WS_LD;
    fetch();
    park(buf0);
    __syncthreads();
    for (int st = 0; st < mine; ++st) {
        const bool more = st + 1 < mine; // (workgroup-uniform)
        if (more) fetch();
        half_stage(buf0, 0);
        if (more) park(buf1);            // (buf1 was last read before the barrier that ended the previous stage)
        half_stage(buf0, 1);
        __syncthreads();
        float *tmp = buf0; buf0 = buf1; buf1 = tmp;
    }
    // C/D layout of 16x16x4: lane l holds rows (l >> 4) * 4 + r (r = 0..3) of column l & 15; workgroups add with fp32 atomics
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = co0 + half_r * 64 + a * 16 + k * 4 + r, col = ci0 + half_c * 64 + c * 16 + i;
                const float v = acc[a][c][r];
                wgrad_emit(dw, slab, row, col, cout, cin, v);
            }
}

// ---- the shared 128 x 128 tile for 16-bit tensors -----------------------------------------------------------------------------------
// conv1x1_wgrad_shared_kernel with bf16 tensors still moves 8 bytes per lane and load, parks fp32 in LDS and lets every wavefront
// convert what it reads (each value twice): 128 -> 256 pooled at C2 0.71 ms for 1.6 GB.  Here a stage is 64 positions: a thread
// loads SIXTEEN bytes (8 positions of one row; 8 pieces x 32 rows per pass, four passes for the 128 + 128 rows), transforms once
// (PRO / POOLED: the expressions of the kernel above) and parks the values as PACKED bf16 (row stride 72 elements = 144 bytes: the
// 8-byte operand reads of a 16-lane row group fall into distinct bank pairs); a wavefront reads its operands ready-made — lane
// (i, k): row i, positions 8 k .. 8 k + 7 of a 32-position half: one v_mfma_f32_16x16x32_bf16 per accumulator and half stage.
constexpr int W16_POS = 64;            // positions per stage
constexpr int W16_LD = W16_POS + 8;    // row stride in LDS (bf16 elements)

template <bool PRO, bool POOLED>
__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_shared16_kernel(int batch, int cin, int cout, int hw, int stages_per_wg,
                                                                        const ogc_bf16 *__restrict__ x, const ogc_bf16 *__restrict__ dy,
                                                                        float *__restrict__ dw, const float *__restrict__ aff_a,
                                                                        const float *__restrict__ aff_b, int pro_relu,
                                                                        const float2 *__restrict__ coef2,
                                                                        const float2 *__restrict__ inj, int s_shift, long long slab) {
    constexpr int YROWS = 128, W_ROWS = YROWS + 128, YJ = 4, XJ = 4, NJ = YJ + XJ; // pieces per thread: dy, x
    constexpr int RSTEP = 32;                                                    // rows between a thread's pieces
    extern __shared__ __attribute__((aligned(16))) ogc_bf16 w16_lds[]; // [2][W_ROWS][W16_LD]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = lane & 15, k = lane >> 4;
    const int co0 = blockIdx.y * YROWS, ci0 = blockIdx.z * 128;
    const int half_r = wave >> 1, half_c = wave & 1;
    const int stages_per_img = hw / W16_POS;
    const long long nstages = (long long)batch * stages_per_img;
    const long long first = (long long)blockIdx.x * stages_per_wg;
    const int mine = (int)max(0LL, min((long long)stages_per_wg, nstages - first));
    if (mine == 0) return;

    const int q = t & 7, rr = t >> 3; // piece q (8 positions) of rows rr + RSTEP j
    int grow[NJ], lrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = rr + RSTEP * (j < YJ ? j : j - YJ);
        grow[j] = j < YJ ? min(co0 + r, cout - 1) : min(ci0 + r, cin - 1);
        lrow[j] = (j < YJ ? r : YROWS + r) * W16_LD + 8 * q;
    }
    const int centres = POOLED ? hw >> s_shift : 0;
    int cur_b = (int)(first / stages_per_img);
    int cur_off = (int)(first - (long long)cur_b * stages_per_img);

    uint4 raw[NJ];
    float fa[XJ], fb[XJ];
    float2 cc[YJ], jv[YJ];
    int jpos = 0;
    auto fetch = [&]() {
        const int pos = cur_off * W16_POS + 8 * q;
        const ogc_bf16 *yb_ = dy + (size_t)cur_b * cout * hw + pos;
        const ogc_bf16 *xb_ = x + (size_t)cur_b * cin * hw + pos;
#pragma unroll
        for (int j = 0; j < YJ; ++j) raw[j] = *reinterpret_cast<const uint4 *>(yb_ + (size_t)grow[j] * hw);
#pragma unroll
        for (int j = YJ; j < NJ; ++j) raw[j] = *reinterpret_cast<const uint4 *>(xb_ + (size_t)grow[j] * hw);
        if constexpr (POOLED) { // a piece's eight positions lie inside one neighbourhood (S >= 16)
#pragma unroll
            for (int j = 0; j < YJ; ++j) {
                const size_t r = (size_t)cur_b * cout + grow[j];
                cc[j] = coef2[r];
                jv[j] = inj[r * centres + (pos >> s_shift)];
            }
            jpos = pos & ((1 << s_shift) - 1);
        }
        if constexpr (PRO) {
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                fa[j] = aff_a[(size_t)cur_b * cin + grow[YJ + j]];
                fb[j] = aff_b[(size_t)cur_b * cin + grow[YJ + j]];
            }
        }
        if (++cur_off == stages_per_img) { cur_off = 0; ++cur_b; }
    };
    auto park = [&](ogc_bf16 *buf) {
#pragma unroll
        for (int j = 0; j < YJ; ++j) {
            uint4 o = raw[j];
            if constexpr (POOLED) {
                float f[8];
                ogc_unpack8(raw[j], f);
                const int rel = __float_as_int(jv[j].y) - jpos;
                const float ag = jv[j].x;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaf(cc[j].x, f[e], cc[j].y) + (rel == e ? ag : 0.f);
                const v4s lo = ogc_pack_bf16(f[0], f[1], f[2], f[3]), hi = ogc_pack_bf16(f[4], f[5], f[6], f[7]);
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                o = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
            *reinterpret_cast<uint4 *>(buf + lrow[j]) = o;
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            uint4 o = raw[YJ + j];
            if constexpr (PRO) {
                float f[8];
                ogc_unpack8(raw[YJ + j], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] = fmaf(fa[j], f[e], fb[j]);
                    if (pro_relu) f[e] = fmaxf(f[e], 0.f);
                }
                const v4s lo = ogc_pack_bf16(f[0], f[1], f[2], f[3]), hi = ogc_pack_bf16(f[4], f[5], f[6], f[7]);
                const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                o = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
            *reinterpret_cast<uint4 *>(buf + lrow[YJ + j]) = o;
        }
    };

    v4f acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int yoff = (half_r * 64 + i) * W16_LD + 8 * k, xoff = (YROWS + half_c * 64 + i) * W16_LD + 8 * k;
    auto half_stage = [&](const ogc_bf16 *buf, int h) { // 32 positions: one k32 MFMA per accumulator
        uint4 yv[4], xv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) yv[a] = *reinterpret_cast<const uint4 *>(buf + yoff + a * 16 * W16_LD + h * 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[c] = *reinterpret_cast<const uint4 *>(buf + xoff + c * 16 * W16_LD + h * 32);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[a][c] = ogc_mfma_bf16_k32(__builtin_bit_cast(v4s, make_uint2(yv[a].x, yv[a].y)),
                                              __builtin_bit_cast(v4s, make_uint2(yv[a].z, yv[a].w)),
                                              __builtin_bit_cast(v4s, make_uint2(xv[c].x, xv[c].y)),
                                              __builtin_bit_cast(v4s, make_uint2(xv[c].z, xv[c].w)), acc[a][c]);
    };

    ogc_bf16 *buf0 = w16_lds, *buf1 = w16_lds + W_ROWS * W16_LD;
    fetch();
    park(buf0);
    __syncthreads();
    for (int st = 0; st < mine; ++st) {
        const bool more = st + 1 < mine;
        if (more) fetch();
        half_stage(buf0, 0);
        if (more) park(buf1);
        half_stage(buf0, 1);
        __syncthreads();
        ogc_bf16 *tmp = buf0; buf0 = buf1; buf1 = tmp;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = co0 + half_r * 64 + a * 16 + k * 4 + r, col = ci0 + half_c * 64 + c * 16 + i;
                const float v = acc[a][c][r];
                wgrad_emit(dw, slab, row, col, cout, cin, v);
            }
}

// true when the launch was made (both tensors bf16, hw a multiple of 64, 16-byte aligned)
bool wgrad_shared16_launch(int b, int cin, int cout, int hw, const ogc_bf16 *x, const ogc_bf16 *dy, float *dw, const float *pa,
                           const float *pb, int pro_relu, hipStream_t s, const float2 *coef2, const float2 *inj, int s_shift) {
    static const bool off = [] { const char *e = getenv("OGC_WGRAD_SHARED16"); return e && e[0] == '0'; }();
    if (off || cin < 128 || cout < 128 || (hw % W16_POS) != 0 || (((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return false;
    if (inj && ((1 << s_shift) < 8 || !pa)) return false;
    const size_t lds = sizeof(ogc_bf16) * 2 * 256 * W16_LD;
    const int tiles = ogc_divup(cout, 128) * ogc_divup(cin, 128);
    const long long nstages = (long long)b * (hw / W16_POS);
    long long wgs = 512 / tiles;
    if (wgs < 1) wgs = 1;
    long long spw = (nstages + wgs - 1) / wgs;
    if (spw < 8) spw = 8;
    const int gx = (int)((nstages + spw - 1) / spw);
    dim3 grid(gx, ogc_divup(cout, 128), ogc_divup(cin, 128));
    long long slab = 0;
    float *const dst = wgrad_det_slabs(dw, gx, cin, cout, slab, s);
    if (!dst) return true; // (the failure is reported by wgrad_impl)
#define OGC_WGS16(PROV, POOLV)                                                                                                 \
    {                                                                                                                          \
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_wgrad_shared16_kernel<PROV, POOLV>),   \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;         \
        if (!ok) { (void)hipGetLastError(); return false; }                                                                    \
        hipLaunchKernelGGL((conv1x1_wgrad_shared16_kernel<PROV, POOLV>), grid, dim3(256), lds, s, b, cin, cout, hw, (int)spw, x, \
                           dy, dst, pa, pb, pro_relu, coef2, inj, s_shift, slab);                                               \
    }
    if (inj) OGC_WGS16(true, true)
    else if (pa) OGC_WGS16(true, false)
    else OGC_WGS16(false, false)
#undef OGC_WGS16
    wgrad_det_finish(dw, dst, gx, slab, s);
    return true;
}

// OGC_WGRAD_SHARED=0 in the environment: the 64 x 64 register tiles for every width (A/B runs, tests of both kernels)
bool wgrad_shared_enabled() {
    static const bool on = [] { const char *e = getenv("OGC_WGRAD_SHARED"); return !(e && e[0] == '0'); }();
    return on;
}

// true when the launch was made
template <typename XT, typename YT>
bool wgrad_shared_launch(int b, int cin, int cout, int hw, const XT *x, const YT *dy, float *dw, const float *pa,
                         const float *pb, int pro_relu, hipStream_t s, const float2 *coef2, const float2 *inj, int s_shift) {
    if (!wgrad_shared_enabled() || cin < 128 || cout < 128 || (hw % WS_POS) != 0) return false;
    if (inj && (1 << s_shift) < 4) return false;
    const size_t lds = sizeof(float) * 2 * 256 * WS_LD;
    const int tiles = ogc_divup(cout, 128) * ogc_divup(cin, 128);
    const long long nstages = (long long)b * (hw / WS_POS);
    // two workgroups per CU over all tiles, at least 8 stages each
    long long wgs = 512 / tiles;
    if (wgs < 1) wgs = 1;
    long long spw = (nstages + wgs - 1) / wgs;
    if (spw < 8) spw = 8;
    const int gx = (int)((nstages + spw - 1) / spw);
    dim3 grid(gx, ogc_divup(cout, 128), ogc_divup(cin, 128));
    constexpr bool F32 = sizeof(XT) == 4 && sizeof(YT) == 4;
    if (inj && !pa) return false;
    long long slab = 0;
    float *const dst = wgrad_det_slabs(dw, gx, cin, cout, slab, s);
    if (!dst) return true; // (the failure is reported by wgrad_impl)
#define OGC_WGS(PROV, POOLV, BFV)                                                                                            \
    {                                                                                                                        \
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_wgrad_shared_kernel<PROV, POOLV, BFV, XT, YT>), \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;       \
        if (!ok) { (void)hipGetLastError(); return false; }                                                                  \
        hipLaunchKernelGGL((conv1x1_wgrad_shared_kernel<PROV, POOLV, BFV, XT, YT>), grid, dim3(256), lds, s, b, cin, cout, hw, (int)spw, \
                           x, dy, dst, pa, pb, pro_relu, coef2, inj, s_shift, slab);                                          \
    }
    // (the pooled form keeps fp32 operands whatever the precision switch says, as with the register tiles)
    if (inj) { OGC_WGS(true, true, !F32) }
    else if (g_matmul_bf16 || !F32) { if (pa) OGC_WGS(true, false, true) else OGC_WGS(false, false, true) }
    else if constexpr (F32) { if (pa) OGC_WGS(true, false, false) else OGC_WGS(false, false, false) }
#undef OGC_WGS
    wgrad_det_finish(dw, dst, gx, slab, s);
    return true;
}

} // namespace

namespace {
template <typename XT, typename YT>
int wgrad_impl(const char *name, int b, int cin, int cout, int hw, const XT *x, const YT *dy, float *dw,
               const float *pa, const float *pb, int pro_relu, ogc_stream_t stream, const float2 *coef2 = nullptr,
               const float2 *inj = nullptr, int s_shift = 0) {
    OGC_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && hw >= 1, "%s: bad shape", name);
    OGC_REQUIRE(x && dy && dw, "%s: null pointer", name);
    if ((hw & 15) != 0 || ((uintptr_t)x & ogc_act_mask<XT>()) != 0 || ((uintptr_t)dy & ogc_act_mask<YT>()) != 0) {
        ogc_set_error("%s: hw=%d must be a multiple of 16 and x/dy 16-byte aligned", name, hw);
        return OGC_ERR_UNSUPPORTED;
    }
    if ((sizeof(XT) == 2 || sizeof(YT) == 2) && !g_matmul_bf16) {
        ogc_set_error("%s: 16-bit activations need ogc_set_matmul_precision(1)", name);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE((long long)cin * hw < (1ll << 31) && (long long)cout * hw < (1ll << 31),
                "%s: one sample exceeds 32-bit indexing", name);
    hipStream_t s = (hipStream_t)stream;
    if (ogc_zero_async(dw, sizeof(float) * (size_t)cin * cout, s) != hipSuccess) {
        ogc_set_error("%s: memset failed", name);
        return OGC_ERR_LAUNCH;
    }
    if (b == 0) return OGC_OK;
    g_wgrad_det_failed = false;
    struct DetCheck { // (deterministic mode: a launcher that could not get its slabs or its ordered pass says so here)
        const char *name;
        int status() const {
            if (!g_wgrad_det_failed) return OGC_OK;
            ogc_set_error("%s (deterministic): no scratch memory for the partial tiles, or their ordered pass failed", name);
            return OGC_ERR_LAUNCH;
        }
    } det{name};
    // layers of 128 channels and more on both sides: a 128 x 128 tile per workgroup, operands shared through LDS
    if constexpr (sizeof(XT) == 2 && sizeof(YT) == 2) { // both tensors in 16 bits: 64-position stages of packed operands
        if (wgrad_shared16_launch(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s, coef2, inj, s_shift)) {
            OGC_CHECK_LAUNCH(name);
            return det.status();
        }
    }
    if (wgrad_shared_launch(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s, coef2, inj, s_shift)) {
        OGC_CHECK_LAUNCH(name);
        return det.status();
    }
    // register tile per wave: (16*COB) x (16*CIB) outputs.  Small channel counts use small tiles so that no MFMA
    // work is spent on padding; wide layers use 64x64 tiles (16 accumulators) and split the rest over the grid.
    if (inj) { // (the pooled form is offered for the wide tails only: 64-row tiles)
        if (cin <= 32) wgrad_launch<4, 2, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s, coef2, inj, s_shift);
        else wgrad_launch<4, 4, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s, coef2, inj, s_shift);
        OGC_CHECK_LAUNCH(name);
        return det.status();
    }
    if (cout <= 16 && cin <= 16) wgrad_launch<1, 1, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cout <= 32 && cin <= 16) wgrad_launch<2, 1, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cout <= 32 && cin <= 32) wgrad_launch<2, 2, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cin <= 16) wgrad_launch<4, 1, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cin <= 32) wgrad_launch<4, 2, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cout <= 32) wgrad_launch<2, 4, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else wgrad_launch<4, 4, XT, YT>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    OGC_CHECK_LAUNCH(name);
    return det.status();
}
} // namespace

extern "C" int ogc_conv1x1_wgrad(int b, int cin, int cout, int hw, const float *x, const float *dy, float *dw,
                                 ogc_stream_t stream) {
    return wgrad_impl("ogc_conv1x1_wgrad", b, cin, cout, hw, x, dy, dw, nullptr, nullptr, 0, stream);
}

// 16-bit forms (act_io.h; bf16 operands: need ogc_set_matmul_precision(1)).  _xf: x fp32, dy bf16 (the three coordinate columns of
// a grouped first layer: x = the relative coordinates); _h: both bf16.
extern "C" int ogc_conv1x1_wgrad_xf_h(int b, int cin, int cout, int hw, const float *x, const ogc_bf16_t *dy, float *dw,
                                      ogc_stream_t stream) {
    return wgrad_impl<float, ogc_bf16>("ogc_conv1x1_wgrad_xf_h", b, cin, cout, hw, x, dy, dw, nullptr, nullptr, 0, stream);
}

extern "C" int ogc_conv1x1_wgrad_affine(int b, int cin, int cout, int hw, int relu, const float *x, const float *pa,
                                        const float *pb, const float *dy, float *dw, ogc_stream_t stream) {
    OGC_REQUIRE(pa && pb, "ogc_conv1x1_wgrad_affine: null pointer");
    return wgrad_impl("ogc_conv1x1_wgrad_affine", b, cin, cout, hw, x, dy, dw, pa, pb, relu, stream);
}

extern "C" int ogc_conv1x1_wgrad_affine_h(int b, int cin, int cout, int hw, int relu, const ogc_bf16_t *x, const float *pa,
                                          const float *pb, const ogc_bf16_t *dy, float *dw, ogc_stream_t stream) {
    OGC_REQUIRE(pa && pb, "ogc_conv1x1_wgrad_affine_h: null pointer");
    return wgrad_impl<ogc_bf16, ogc_bf16>("ogc_conv1x1_wgrad_affine_h", b, cin, cout, hw, x, dy, dw, pa, pb, relu, stream);
}

// ogc_conv1x1_wgrad_affine with dy in the sparse form of ogc_group_norm_maxpool_bwd_sparse: y is the convolution's raw output
// (b, cout, hw) and g_y is rebuilt from (y, coef2, inj) while y is loaded (see POOLED at conv1x1_wgrad_kernel) — the weight
// gradient of the LAST layer of a set-abstraction MLP without the dense gradient of its pooled GroupNorm.  fp32 operands.
namespace {
template <typename AT>
int wgrad_affine_pooled_impl(const char *name, int b, int cin, int cout, int hw, int relu, int nsample, const AT *x,
                             const float *pa, const float *pb, const AT *y, const float *coef2, const float *inj, float *dw,
                             ogc_stream_t stream) {
    OGC_REQUIRE(pa && pb && coef2 && inj, "%s: null pointer", name);
    const int sh = nsample == 16 ? 4 : nsample == 32 ? 5 : nsample == 64 ? 6 : -1;
    if (sh < 0 || hw % nsample != 0 || (((uintptr_t)coef2 | (uintptr_t)inj) & 7) != 0) {
        ogc_set_error("%s: nsample=%d must be 16, 32 or 64 and divide hw=%d", name, nsample, hw);
        return OGC_ERR_UNSUPPORTED;
    }
    return wgrad_impl<AT, AT>(name, b, cin, cout, hw, x, y, dw, pa, pb, relu, stream, reinterpret_cast<const float2 *>(coef2),
                              reinterpret_cast<const float2 *>(inj), sh);
}
} // namespace

extern "C" int ogc_conv1x1_wgrad_affine_pooled(int b, int cin, int cout, int hw, int relu, int nsample, const float *x,
                                               const float *pa, const float *pb, const float *y, const float *coef2,
                                               const float *inj, float *dw, ogc_stream_t stream) {
    return wgrad_affine_pooled_impl<float>("ogc_conv1x1_wgrad_affine_pooled", b, cin, cout, hw, relu, nsample, x, pa, pb, y, coef2,
                                           inj, dw, stream);
}

extern "C" int ogc_conv1x1_wgrad_affine_pooled_h(int b, int cin, int cout, int hw, int relu, int nsample, const ogc_bf16_t *x,
                                                 const float *pa, const float *pb, const ogc_bf16_t *y, const float *coef2,
                                                 const float *inj, float *dw, ogc_stream_t stream) {
    return wgrad_affine_pooled_impl<ogc_bf16>("ogc_conv1x1_wgrad_affine_pooled_h", b, cin, cout, hw, relu, nsample, x, pa, pb, y,
                                              coef2, inj, dw, stream);
}
