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
#include <stdlib.h>

#include "ogc_common.h"
#include "conv_stage.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));  // four bf16 MFMA operand values

constexpr int WG_WAVES = 4;

// Operand precision of the convolution kernels (ogc_set_matmul_precision): 0 = fp32 operands on
// v_mfma_f32_16x16x4_f32 (default; results are exact fp32 FMA chains), 1 = operands rounded to bf16 (nearest even) on
// v_mfma_f32_16x16x16_bf16, fp32 accumulation, tensors in memory stay fp32 — what autocast(bfloat16) gives the
// reference's Conv2d layers (BASELINE config "OGC-DR ..., bf16").  Layers of 128 channels and more sit on the fp32
// MFMA roof (157 TFLOP/s); with bf16 operands (2.5 PFLOP/s) they fall back onto the HBM roof.
int g_matmul_bf16 = 0;

__device__ __forceinline__ v4s ogc_pack_bf16(float a, float b, float c, float d) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    union { v2bf h[2]; v4s s; } u;
    u.h[0] = __builtin_convertvector((v2f){a, b}, v2bf); // one v_cvt_pk_bf16_f32 per pair
    u.h[1] = __builtin_convertvector((v2f){c, d}, v2bf);
    return u.s;
}

// PRO: the layer's input was never materialised — x holds the previous layer's raw convolution output and the operand
// is act(pa[b, ci] * x + pb[b, ci]), recomputed while loading (see conv1x1_gemm_kernel).
// POOLED: dy is not stored — `dy` holds y, the convolution's raw output, and the gradient of the pooled GroupNorm behind it is
// rebuilt while y is loaded:  g_y[row, pos] = fmaf(c2, y, c3) + (pos % S == arg ? ag : 0)  with (c2, c3) = coef2[b, row] and
// (ag, arg) = inj[b, row, pos / S] (ogc_group_norm_maxpool_bwd_sparse; a step's 16 positions lie inside one neighbourhood,
// S = 16, 32, 64) — the expression of gn_maxpool_bwd_dx_kernel, bit for bit.
template <int COB, int CIB, bool PRO, bool BF, bool POOLED = false>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void conv1x1_wgrad_kernel(int batch, int cin, int cout, int hw,
                                                                           int steps_per_wave,
                                                                           const float *__restrict__ x,
                                                                           const float *__restrict__ dy,
                                                                           float *__restrict__ dw,
                                                                           const float *__restrict__ aff_a,
                                                                           const float *__restrict__ aff_b, int pro_relu,
                                                                           const float2 *__restrict__ coef2 = nullptr,
                                                                           const float2 *__restrict__ inj = nullptr,
                                                                           int s_shift = 0) {
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
        const float *yb_ = dy + (size_t)cur_b * cout * hw + pb;
        const float *xb_ = x + (size_t)cur_b * cin * hw + pb;
#pragma unroll
        for (int a = 0; a < COB; ++a) yv[a] = *reinterpret_cast<const float4 *>(yb_ + yrow[a]);
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
            xv[c] = *reinterpret_cast<const float4 *>(xb_ + xrow[c]);
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
        if (row < cout && col < cin && v != 0.0f) unsafeAtomicAdd(dw + (size_t)row * cin + col, v);
    }
}

template <int COB, int CIB>
void wgrad_launch(int b, int cin, int cout, int hw, const float *x, const float *dy, float *dw, const float *pa,
                  const float *pb, int pro_relu, hipStream_t s, const float2 *coef2 = nullptr, const float2 *inj = nullptr,
                  int s_shift = 0) {
    const long long nsteps = (long long)b * (hw >> 4);
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
#define OGC_WGRAD(PROV, BFV)                                                                                          \
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<COB, CIB, PROV, BFV>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, b, cin, cout, \
                       hw, (int)spw, x, dy, dw, pa, pb, pro_relu)
    if (inj) { // pooled form of dy (fp32 operands, previous layer's norm folded in): see POOLED
        hipLaunchKernelGGL((conv1x1_wgrad_kernel<COB, CIB, true, false, true>), grid, dim3(WG_WAVES * OGC_WAVE), 0, s, b, cin, cout,
                           hw, (int)spw, x, dy, dw, pa, pb, pro_relu, coef2, inj, s_shift);
    } else if (g_matmul_bf16) {
        if (pa) OGC_WGRAD(true, true);
        else OGC_WGRAD(false, true);
    } else {
        if (pa) OGC_WGRAD(true, false);
        else OGC_WGRAD(false, false);
    }
#undef OGC_WGRAD
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
template <bool PRO, bool POOLED, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_shared_kernel(int batch, int cin, int cout, int hw, int stages_per_wg,
                                                                      const float *__restrict__ x, const float *__restrict__ dy,
                                                                      float *__restrict__ dw, const float *__restrict__ aff_a,
                                                                      const float *__restrict__ aff_b, int pro_relu,
                                                                      const float2 *__restrict__ coef2,
                                                                      const float2 *__restrict__ inj, int s_shift) {
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
        const float *yb_ = dy + (size_t)cur_b * cout * hw + pos;
        const float *xb_ = x + (size_t)cur_b * cin * hw + pos;
#pragma unroll
        for (int j = 0; j < YJ; ++j) raw[j] = *reinterpret_cast<const float4 *>(yb_ + (size_t)grow[j] * hw);
#pragma unroll
        for (int j = YJ; j < NJ; ++j) raw[j] = *reinterpret_cast<const float4 *>(xb_ + (size_t)grow[j] * hw);
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
                if (row < cout && col < cin && v != 0.0f) unsafeAtomicAdd(dw + (size_t)row * cin + col, v);
            }
}

// OGC_WGRAD_SHARED=0 in the environment: the 64 x 64 register tiles for every width (A/B runs, tests of both kernels)
bool wgrad_shared_enabled() {
    static const bool on = [] { const char *e = getenv("OGC_WGRAD_SHARED"); return !(e && e[0] == '0'); }();
    return on;
}

// true when the launch was made
bool wgrad_shared_launch(int b, int cin, int cout, int hw, const float *x, const float *dy, float *dw, const float *pa,
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
#define OGC_WGS(PROV, POOLV, BFV)                                                                                            \
    {                                                                                                                        \
        static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_wgrad_shared_kernel<PROV, POOLV, BFV>), \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;       \
        if (!ok) { (void)hipGetLastError(); return false; }                                                                  \
        hipLaunchKernelGGL((conv1x1_wgrad_shared_kernel<PROV, POOLV, BFV>), grid, dim3(256), lds, s, b, cin, cout, hw, (int)spw, \
                           x, dy, dw, pa, pb, pro_relu, coef2, inj, s_shift);                                                 \
    }
    // (the pooled form keeps fp32 operands whatever the precision switch says, as with the register tiles)
    if (inj) { if (pa) OGC_WGS(true, true, false) else return false; }
    else if (g_matmul_bf16) { if (pa) OGC_WGS(true, false, true) else OGC_WGS(false, false, true) }
    else if (pa) OGC_WGS(true, false, false)
    else OGC_WGS(false, false, false)
#undef OGC_WGS
    return true;
}

} // namespace

namespace {

// ---- forward / input-gradient GEMM:  OUT[b, m, p] = sum_k A[m, k] * IN[b, k, p] -----------------------------------
// forward: A = W (M = Cout, K = Cin, IN = x);  input gradient: A = W^T (M = Cin, K = Cout, IN = dy).
// A wave owns 64 positions.  Lane (k' = l >> 4, j = l & 15) loads the float4 IN[k0 + k'][p0 + 4j .. 4j+3]; MFMA column
// block c (c = 0..3) takes component c of it, i.e. column j of block c is position p0 + 4j + c.  The same lane then
// holds the four consecutive positions of an output row in its four column-block accumulators, so results leave as
// float4 stores — no shuffles on either side.  The whole IN tile (K x 64) stays in registers (read from HBM exactly
// once); A is staged through LDS in [k/4][m][4] order (conflict-free ds_read_b32 of the A operand) one 64-row tile at
// a time and shared by the four waves of the workgroup.
constexpr int FW_KQ_MAX = 40; // K <= 160 channels held in registers (40 float4 per lane)
constexpr int GN_SLOTS = 16;  // spread of the fused GroupNorm statistics over cache lines (see ogc_conv1x1_gemm_gnstats)

// KQ: compile-time bound on ceil(K / 4) — the IN tile costs KQ float4 registers per lane, so narrow layers get a small
// register footprint and more wavefronts per SIMD.  STATS: also accumulate, per (batch, GroupNorm group), the sum and
// the sum of squares of the outputs (the first pass of the GroupNorm that follows every one of these convolutions):
// row sums are reduced over the 16 lanes of a DPP row, then over the workgroup in LDS (fp64), then one fp64 atomic
// per (group, statistic) and workgroup into one of GN_SLOTS copies of the accumulator.
// PRO: the input is act(pa[b, k] * in + pb[b, k]) — the GroupNorm (+ ReLU) of the PREVIOUS layer applied while its raw
// convolution output is loaded, so that the normalised activation is never written to memory.
// BF: bf16 operands.  Sixteen input rows (four register quads q .. q+3) feed one 16x16x16 MFMA: k-slot 4*kk + i of
// lane group kk is input row 4*(q+i) + kk for BOTH operands (a permutation of the sixteen rows, which a sum over k
// does not see), so the register-resident tile needs no shuffle; the weights are staged in LDS already packed that way.
// POOL: the positions are (centre, neighbour) pairs with pool_s in {16, 32, 64} neighbours per centre, and the kernel
// also writes, per (batch, output row, centre), the extreme raw output over the neighbourhood and where it sits
// (smallest neighbour index among equals): everything the max-pool over act(GroupNorm(out)) that ends a set-abstraction
// MLP needs — the norm is a monotone map per (batch, row), rising or falling with the sign of its scale gamma[row],
// which is known before the convolution runs, so one extreme per row suffices (the largest value for gamma >= 0, the
// smallest for gamma < 0: the largest of the negated values) — and that pass never reads `out`
// (ogc_group_norm_pool_extremes).  A wave's 64 positions hold whole neighbourhoods; values are reduced over the 4
// accumulator columns of a lane and the pool_s / 4 lanes of a DPP row first, then the smallest index attaining them.
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

template <bool TRANSPOSE_A, int KQ, bool STATS, bool PRO, bool BF, bool POOL = false>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void conv1x1_gemm_kernel(int M, int K, int hw, int groups,
                                                                          const float *__restrict__ w, // (Cout, Cin)
                                                                          const float *__restrict__ in,
                                                                          float *__restrict__ out,
                                                                          double *__restrict__ stats,
                                                                          const float *__restrict__ pa,
                                                                          const float *__restrict__ pb, int pro_relu,
                                                                          PoolOut pool = PoolOut()) {
    extern __shared__ __attribute__((aligned(16))) float a_lds[]; // [64][ogc_a_ld(Kq)]: the current 64-row tile of A (conv_stage.h)
    __shared__ double s_stats[STATS ? 64 : 1];                    // [groups][2]
    __shared__ __attribute__((aligned(16))) float s_sign[POOL ? 64 : 4]; // POOL: -1 for the tile's rows whose next scale is negative
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * WG_WAVES + wave) * 64;
    const int Kq = (K + 3) >> 2, a_ld = ogc_a_ld(Kq);
    const bool live = p0 < hw; // hw is a multiple of 64 for every wave that is live
    const float *inb = in + (size_t)b * K * hw;
    float *outb = out + (size_t)b * M * hw;
    if (STATS && threadIdx.x < 2 * groups) s_stats[threadIdx.x] = 0.0;

    float4 xin[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int row = q * 4 + kk;
        xin[q] = (live && q < Kq && row < K) ? *reinterpret_cast<const float4 *>(inb + (size_t)row * hw + p0 + 4 * j)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (PRO) { // a second pass, so that all loads above are in flight before the first one is waited for; in chunks
               // of eight rows so that the coefficients stay transient in wide layers (no spare registers there)
        constexpr int CH = 8;
#pragma unroll
        for (int q0 = 0; q0 < KQ; q0 += CH) {
            float ca[CH], cb[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int q = q0 + u, row = q * 4 + kk;
                const bool have = q < KQ && live && q < Kq && row < K; // padding rows stay exact zeros: act(0 * 0 + 0)
                ca[u] = have ? pa[(size_t)b * K + row] : 0.f;
                cb[u] = have ? pb[(size_t)b * K + row] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int q = q0 + u;
                if (q < KQ) {
                    float4 v = xin[q];
                    v.x = fmaf(ca[u], v.x, cb[u]); v.y = fmaf(ca[u], v.y, cb[u]);
                    v.z = fmaf(ca[u], v.z, cb[u]); v.w = fmaf(ca[u], v.w, cb[u]);
                    if (pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    xin[q] = v;
                }
            }
        }
    }
    constexpr int GQ = (KQ + 3) / 4;
    v4s xb[BF ? GQ : 1][4]; // BF: the tile as packed bf16 operands (half the registers of the fp32 tile, which dies here)
    if constexpr (BF) {
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            float4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = 4 * g + i < KQ ? xin[4 * g + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            xb[g][0] = ogc_pack_bf16(r[0].x, r[1].x, r[2].x, r[3].x);
            xb[g][1] = ogc_pack_bf16(r[0].y, r[1].y, r[2].y, r[3].y);
            xb[g][2] = ogc_pack_bf16(r[0].z, r[1].z, r[2].z, r[3].z);
            xb[g][3] = ogc_pack_bf16(r[0].w, r[1].w, r[2].w, r[3].w);
        }
    }
    const int cpg = STATS ? M / groups : 1; // channels per group, a multiple of 4 on this path
    for (int m0 = 0; m0 < M; m0 += 64) {
        __syncthreads(); // previous tile fully consumed
        if constexpr (BF) {
            // a_bf[(g * 64 + mi) * 4 + kr] = bf16 x 4 of A[m0 + mi][4 * (4g + i) + kr], i = 0..3
            v4s *a_bf = reinterpret_cast<v4s *>(a_lds);
            const int Gq = (Kq + 3) >> 2;
            for (int t = threadIdx.x; t < Gq * 256; t += WG_WAVES * OGC_WAVE) {
                const int kr = t & 3, mi = (t >> 2) & 63, g = t >> 8;
                const int m = m0 + mi;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 4 * (4 * g + i) + kr;
                    v[i] = (m < M && k < K) ? (TRANSPOSE_A ? w[(size_t)k * M + m] : w[(size_t)m * K + k]) : 0.f;
                }
                a_bf[t] = ogc_pack_bf16(v[0], v[1], v[2], v[3]);
            }
        } else {
        ogc_stage_weight_tile<TRANSPOSE_A, WG_WAVES>(a_lds, w, m0, M, K, Kq); // a_lds[mi * LD + k] = A[m0 + mi][k]
        }
        if constexpr (POOL) {
            if (threadIdx.x < 64) s_sign[threadIdx.x] = (m0 + (int)threadIdx.x < M && pool.sign[m0 + threadIdx.x] < 0.f) ? -1.f : 1.f;
        }
        __syncthreads();
        const int nblk = min(4, (M - m0 + 15) >> 4); // 16-row blocks of this tile that hold real rows (uniform)
        v4f acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
            const v4s *a_bf = reinterpret_cast<const v4s *>(a_lds);
#pragma unroll
            for (int g = 0; g < GQ; ++g) {
                if (4 * g < Kq) {
                    v4s av[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) av[a] = a_bf[(g * 64 + a * 16 + j) * 4 + kk];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (a < nblk) {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[a], xb[g][c], acc[a][c], 0, 0, 0);
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            if (q < Kq) {
                float av[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = a_lds[(a * 16 + j) * a_ld + q * 4 + kk]; // A[m0+16a+j][4q+kk]
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < nblk) { // no MFMA work on padding rows (M = 32: half of the tile)
                        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], xin[q].x, acc[a][0], 0, 0, 0);
                        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], xin[q].y, acc[a][1], 0, 0, 0);
                        acc[a][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], xin[q].z, acc[a][2], 0, 0, 0);
                        acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], xin[q].w, acc[a][3], 0, 0, 0);
                    }
                }
            }
        }
        }
        if (live) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + a * 16 + kk * 4 + r; // C/D layout: row (l >> 4) * 4 + r, column l & 15
                    if (m < M) {
                        const float4 o = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
                        *reinterpret_cast<float4 *>(outb + (size_t)m * hw + p0 + 4 * j) = o;
                    }
                }
            if constexpr (POOL) {
                const int centres = hw / pool.s;
                const int centre = (p0 + 4 * j) / pool.s;    // of this lane's four positions
                // outputs of row m0 + kk * 4 of this lane's centre; row a * 16 + r is (a * 16 + r) * centres further on
                const size_t o0 = ((size_t)b * M + m0 + kk * 4) * centres + centre;
                if (pool.s == 64) ogc_pool_extremes_epilogue<16>(acc, nblk, s_sign, j, kk, m0, M, centres, 64, pool.yext + o0, pool.aext + o0);
                else if (pool.s == 32) ogc_pool_extremes_epilogue<8>(acc, nblk, s_sign, j, kk, m0, M, centres, 32, pool.yext + o0, pool.aext + o0);
                else ogc_pool_extremes_epilogue<4>(acc, nblk, s_sign, j, kk, m0, M, centres, 16, pool.yext + o0, pool.aext + o0);
            }
            if (STATS) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < nblk) {
                        // the lane's 4 rows (m .. m+3) x 4 positions; rows >= M are exact zeros
                        float sm = 0.f, sq = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float v = acc[a][c][r];
                                sm += v;
                                sq += v * v;
                            }
                        // sum over the 16 lanes of the DPP row (same rows, the other 60 positions)
                        sm += ogc_dpp_f32<0xB1>(sm); sq += ogc_dpp_f32<0xB1>(sq);
                        sm += ogc_dpp_f32<0x4E>(sm); sq += ogc_dpp_f32<0x4E>(sq);
                        sm += ogc_dpp_f32<0x141>(sm); sq += ogc_dpp_f32<0x141>(sq);
                        sm += ogc_dpp_f32<0x140>(sm); sq += ogc_dpp_f32<0x140>(sq);
                        const int m = m0 + a * 16 + kk * 4;
                        if (j == 0 && m < M) {
                            const int g = m / cpg;
                            atomicAdd(&s_stats[2 * g], (double)sm);
                            atomicAdd(&s_stats[2 * g + 1], (double)sq);
                        }
                    }
                }
            }
        }
    }
    if (STATS) {
        __syncthreads();
        if (threadIdx.x < 2 * groups) {
            const int slot = blockIdx.x % GN_SLOTS;
            double *dst = stats + ((size_t)slot * gridDim.y + b) * 2 * groups;
            atomicAdd(dst + threadIdx.x, s_stats[threadIdx.x]);
        }
    }
}

// ---- streaming variant for the wide forward layers (K > 100): load, compute and store OVERLAP ---------------------------------
// conv1x1_gemm_kernel gives a wave ONE 64-position tile: load K x 64 (32 KB at K = 128), ~1000 MFMAs, store.  At 128 -> 128 the
// memory time of a tile (load + store, every CU at once: ~13 us each at the HBM rate) equals its MFMA time (~27 us per pair of
// co-resident waves), and the two waves a SIMD holds start together and stay in lockstep — both load, then both compute —
// so the phases ADD: 216 us = 54 (load) + 109 (MFMA) + 54 (store) instead of max(108, 109).  Here a persistent wave walks over
// tiles with two register sets: the next tile's loads are issued before the current tile's MFMAs and its stores drain behind
// them; the whole weight matrix is staged in LDS once per workgroup (one workgroup of four waves per CU).
// EXACT: K fills the KQ register quads (Kq == KQ) and M is a multiple of 64 — no wave-uniform branch is left between the
// MFMAs of a tile (a scalar compare-and-branch between two MFMAs costs issue slots the matrix pipe cannot fill).
// STATS: the sums and sums of squares of the outputs per (sample, GroupNorm group) as well (see conv1x1_gemm_kernel): a lane's
// four rows x four positions, summed over the sixteen lanes of its DPP row, go straight to the fp64 accumulators in global
// memory (copy blockIdx.x % GN_SLOTS): 32 atomics per wavefront and 64-row tile.
// POOL: the extreme of every neighbourhood as well (PoolOut; see conv1x1_gemm_kernel), for the max-pool that ends the MLP.
// DUAL: two workgroups per CU (two wavefronts per SIMD), ONE register set per wavefront.  With one wavefront per SIMD the
// compiler's wait-count pass drains every load in flight — the prefetched next tile included — in front of each tile's MFMAs
// (the run-time loop over row tiles loses its bookkeeping; unrolled, the kernel spills), so load, MFMA and store phases add
// up.  Two wavefronts that each load, compute and store in turn fall out of step by themselves: one computes while the other
// waits for memory.  Needs the weights of both workgroups in LDS (2 x ~70 KB at 128 x 128) and <= 256 registers.
// TRANS: the A operand is w^T (the input gradient of the layer: out = w^T . in), staged transposed; everything else is the same.
template <int KQ, bool PRO, bool EXACT, bool STATS = false, bool POOL = false, bool DUAL = false, bool TRANS = false>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE, DUAL ? 2 : 1) void conv1x1_gemm_stream_kernel(int M, int K, int hw, int ntiles,
                                                                                 const float *__restrict__ w,
                                                                                 const float *__restrict__ in,
                                                                                 float *__restrict__ out,
                                                                                 const float *__restrict__ pa,
                                                                                 const float *__restrict__ pb, int pro_relu,
                                                                                 int groups = 1, int nbatch = 1,
                                                                                 double *__restrict__ stats = nullptr,
                                                                                 PoolOut pool = PoolOut()) {
    extern __shared__ __attribute__((aligned(16))) float a_lds[]; // [M / 64 tiles][64][ogc_a_ld(Kq)], then PRO: [wave][2][KQ * 4]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kk = lane >> 4;
    const int Kq = (K + 3) >> 2, Mt = (M + 63) >> 6, a_ld = ogc_a_ld(Kq);
    const int tiles_per_img = hw >> 6;
    for (int mt = 0; mt < Mt; ++mt)
        ogc_stage_weight_tile<TRANS, WG_WAVES>(a_lds + (size_t)mt * 64 * a_ld, w, mt * 64, M, K, Kq);
    // POOL: -1 for the rows whose next scale is negative, after the other strips (16-byte aligned: every strip is)
    float *sgn_all = a_lds + (size_t)Mt * 64 * a_ld + (PRO ? WG_WAVES * 2 * FW_KQ_MAX * 4 : 0) + WG_WAVES * 4 * 16 * 2 * 2;
    if constexpr (POOL) {
        for (int t = threadIdx.x; t < Mt * 64; t += WG_WAVES * OGC_WAVE) sgn_all[t] = (t < M && pool.sign[t] < 0.f) ? -1.f : 1.f;
    }
    __syncthreads();
    const int nw = gridDim.x * WG_WAVES;
    // The loads of a tile are unconditional and of one shape (a load under a per-lane condition becomes an exec-masked block
    // of its own, and the compiler then waits for ALL outstanding loads — the prefetched next tile included — wherever it
    // needs one): padding rows (k >= K) re-read the last row, their weights in LDS are zeros.
    // Addresses: a wave-uniform base per row quad (SGPRs) + one of two 32-bit lane offsets — 33 64-bit lane addresses kept
    // across the loop would cost 66 registers of the 512.
    const unsigned off_main = (unsigned)(kk * hw + 4 * j);
    const unsigned off_last = (unsigned)(min(kk, K - 1 - (Kq - 1) * 4) * hw + 4 * j); // last quad: rows >= K clamped
    auto load_tile = [&](int t, float4(&x)[KQ]) {
        t = min(t, ntiles - 1); // beyond the wave's last tile: a harmless re-read
        const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 64;
        const float *inb = in + (size_t)b * K * hw + p0;
#pragma unroll
        for (int q = 0; q < KQ; ++q)
            if (EXACT || q < Kq) {
                const float *rowq = inb + (size_t)(q * 4) * hw;
                x[q] = *reinterpret_cast<const float4 *>(rowq + ((EXACT ? q == KQ - 1 : q == Kq - 1) ? off_last : off_main));
            }
    };
    // PRO: the coefficients of the current image sit in a wave-private strip of LDS, refreshed when the image changes —
    // read from global memory at compute time they would be the newest loads in flight, and waiting for them means
    // waiting for the prefetched tile as well.
    float *cw = a_lds + (size_t)Mt * 64 * a_ld + wave * (2 * KQ * 4);
    int coef_b = -1;
    // STATS: fp64 partial sums per (64-row tile, 16-row block, row quad) in a wave-private strip of LDS (each slot has one
    // writer: lane 16 kk of the wave), flushed to the global accumulators when the wave moves on to another sample — a
    // wave walks a CONTIGUOUS range of position tiles, so that happens once or twice per launch.
    double *sacc = reinterpret_cast<double *>(a_lds + (size_t)Mt * 64 * a_ld + (PRO ? WG_WAVES * 2 * FW_KQ_MAX * 4 : 0)) +
                   wave * (4 * 16 * 2);
    int stat_b = -1;
    auto flush_stats = [&]() {
        if constexpr (STATS) {
            if (stat_b >= 0) {
                const int cpg = M / groups; // channels per group, a multiple of 4 on this path
                double *dst = stats + ((size_t)(blockIdx.x % GN_SLOTS) * nbatch + stat_b) * 2 * groups;
                if (lane < Mt * 16) { // slot = (mt * 4 + a) * 4 + kk  ->  rows mt * 64 + a * 16 + kk * 4 ..
                    const int m = (lane >> 4) * 64 + ((lane >> 2) & 3) * 16 + (lane & 3) * 4;
                    if (m < M) {
                        unsafeAtomicAdd(dst + 2 * (m / cpg), sacc[2 * lane]);
                        unsafeAtomicAdd(dst + 2 * (m / cpg) + 1, sacc[2 * lane + 1]);
                    }
                }
            }
            if (lane < 64) { sacc[2 * lane] = 0.0; sacc[2 * lane + 1] = 0.0; }
            __builtin_amdgcn_wave_barrier();
        }
    };
    auto compute_store = [&](int t, float4(&x)[KQ]) {
        const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 64;
        if constexpr (STATS) {
            if (b != stat_b) { flush_stats(); stat_b = b; }
        }
        if (PRO) { // the previous layer's GroupNorm (+ ReLU) applied to the tile in place
            if (b != coef_b) {
                coef_b = b;
                for (int r = lane; r < KQ * 4; r += OGC_WAVE) { // padding rows: act(0 * x + 0) = 0
                    cw[r] = r < K ? pa[(size_t)b * K + r] : 0.f;
                    cw[KQ * 4 + r] = r < K ? pb[(size_t)b * K + r] : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (EXACT || q < Kq) {
                    const float ca = cw[q * 4 + kk], cb = cw[KQ * 4 + q * 4 + kk];
                    float4 v = x[q];
                    v.x = fmaf(ca, v.x, cb); v.y = fmaf(ca, v.y, cb);
                    v.z = fmaf(ca, v.z, cb); v.w = fmaf(ca, v.w, cb);
                    if (pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    x[q] = v;
                }
            }
        }
        float *outb = out + (size_t)b * M * hw + p0;
        const unsigned off_out = (unsigned)(kk * 4 * hw + 4 * j);
        for (int mt = 0; mt < Mt; ++mt) {
            const float *at = a_lds + (size_t)mt * 64 * a_ld;
            const int nblk = EXACT ? 4 : min(4, (M - mt * 64 + 15) >> 4);
            v4f acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (EXACT || q < Kq) {
                    float av[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) av[a] = at[(a * 16 + j) * a_ld + q * 4 + kk];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (EXACT || a < nblk) {
                            acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], x[q].x, acc[a][0], 0, 0, 0);
                            acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], x[q].y, acc[a][1], 0, 0, 0);
                            acc[a][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], x[q].z, acc[a][2], 0, 0, 0);
                            acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], x[q].w, acc[a][3], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m0 = mt * 64 + a * 16 + r; // the row is m0 + 4 kk
                    if (EXACT || m0 + kk * 4 < M)
                        *reinterpret_cast<float4 *>(outb + (size_t)m0 * hw + off_out) =
                            make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
                }
            if constexpr (POOL) {
                const int m0p = mt * 64;
                const int centres = hw / pool.s;
                const int centre = (p0 + 4 * j) / pool.s;    // of this lane's four positions
                const size_t o0 = ((size_t)b * M + m0p + kk * 4) * centres + centre;
                const float *sg = sgn_all + m0p;
                if (pool.s == 64) ogc_pool_extremes_epilogue<16>(acc, nblk, sg, j, kk, m0p, M, centres, 64, pool.yext + o0, pool.aext + o0);
                else if (pool.s == 32) ogc_pool_extremes_epilogue<8>(acc, nblk, sg, j, kk, m0p, M, centres, 32, pool.yext + o0, pool.aext + o0);
                else ogc_pool_extremes_epilogue<4>(acc, nblk, sg, j, kk, m0p, M, centres, 16, pool.yext + o0, pool.aext + o0);
            }
            if constexpr (STATS) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (EXACT || a < nblk) {
                        float sm = 0.f, sq = 0.f; // the lane's 4 rows x 4 positions; rows >= M are exact zeros
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float v = acc[a][c][r];
                                sm += v;
                                sq += v * v;
                            }
                        sm += ogc_dpp_f32<0xB1>(sm); sq += ogc_dpp_f32<0xB1>(sq);
                        sm += ogc_dpp_f32<0x4E>(sm); sq += ogc_dpp_f32<0x4E>(sq);
                        sm += ogc_dpp_f32<0x141>(sm); sq += ogc_dpp_f32<0x141>(sq);
                        sm += ogc_dpp_f32<0x140>(sm); sq += ogc_dpp_f32<0x140>(sq);
                        if (j == 0) {
                            const int slot = (mt * 4 + a) * 4 + kk;
                            sacc[2 * slot] += (double)sm;
                            sacc[2 * slot + 1] += (double)sq;
                        }
                    }
                }
            }
        }
    };
    // a wave walks a contiguous range of position tiles (it stays inside one sample most of the time)
    const int per = (ntiles + nw - 1) / nw;
    int t = (blockIdx.x * WG_WAVES + wave) * per;
    const int t_end = min(ntiles, t + per);
    if (STATS) flush_stats(); // (clears the strip)
    if constexpr (DUAL) {
        float4 x1[KQ];
        for (; t < t_end; ++t) {
            load_tile(t, x1);
            compute_store(t, x1);
        }
        if (STATS) flush_stats();
        return;
    }
    float4 xa[KQ], xb[KQ];
    if (t < t_end) {
        load_tile(t, xa);
        for (; t + 1 < t_end; t += 2) { // two tiles per round, no branch in the body
            load_tile(t + 1, xb);
            compute_store(t, xa);
            load_tile(t + 2, xa);
            compute_store(t + 1, xb);
        }
        if (t < t_end) compute_store(t, xa);
    }
    if (STATS) flush_stats();
}

// the streaming kernel for this shape, or false when the tile kernel should run
// shapes the streaming kernel takes (fp32 operands only)
// (pool: the launch also carries the sign strip of the pooled epilogue — one float per staged row)
// (any_precision: the variants with GroupNorm statistics / neighbourhood extremes in their epilogue keep fp32 operands under
// `matmul_precision: bf16` too — the alternative there is the bf16 tile kernel followed by a statistics pass of its own, and for
// a pooled tail the dense backward path)
bool gemm_stream_eligible(int b, int M, int K, int hw, bool pro, bool pool = false, bool any_precision = false) {
    const int Kq = (K + 3) / 4, Mt = (M + 63) / 64;
    const long long ntiles = (long long)b * (hw / 64);
    const size_t lds = ((size_t)Mt * 64 * ogc_a_ld(Kq) + (pro ? WG_WAVES * 2 * FW_KQ_MAX * 4 : 0)) * sizeof(float) +
                       WG_WAVES * 4 * 16 * 2 * sizeof(double) + (pool ? (size_t)Mt * 64 * sizeof(float) : 0);
    static const bool off = getenv("OGC_GEMM_STREAM") && getenv("OGC_GEMM_STREAM")[0] == '0';
    return !(off || (g_matmul_bf16 && !any_precision) || (hw & 63) != 0 || Kq <= 25 || Kq > FW_KQ_MAX || lds > 156 * 1024 || ntiles < 2048 ||
             ntiles >= (1ll << 31));
}

template <bool PRO, bool STATS = false, bool POOL = false, bool TRANS = false>
bool gemm_stream_launch(int b, int M, int K, int hw, const float *w, const float *in, float *out, const float *pa,
                        const float *pb, int pro_relu, hipStream_t s, int groups = 1, double *stats = nullptr,
                        PoolOut pool = PoolOut()) {
    const int Kq = (K + 3) / 4, Mt = (M + 63) / 64;
    const long long ntiles = (long long)b * (hw / 64);
    const size_t lds = ((size_t)Mt * 64 * ogc_a_ld(Kq) + (PRO ? WG_WAVES * 2 * FW_KQ_MAX * 4 : 0)) * sizeof(float) +
                       WG_WAVES * 4 * 16 * 2 * sizeof(double) + (POOL ? (size_t)Mt * 64 * sizeof(float) : 0);
    if (!gemm_stream_eligible(b, M, K, hw, PRO, POOL, STATS) || lds > 156 * 1024) return false;
    const int wgs = (int)(ntiles / WG_WAVES < 256 ? ntiles / WG_WAVES : 256);
#define OGC_STREAM_D(KQV, EX, DU, WGS)                                                                                       \
    do {                                                                                                                     \
        static bool raised = false;                                                                                          \
        const void *fn = reinterpret_cast<const void *>(&conv1x1_gemm_stream_kernel<KQV, PRO, EX, STATS, POOL, DU, TRANS>);         \
        if (!raised) {                                                                                                       \
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return false;  \
            raised = true;                                                                                                   \
        }                                                                                                                    \
        hipLaunchKernelGGL((conv1x1_gemm_stream_kernel<KQV, PRO, EX, STATS, POOL, DU, TRANS>), dim3(WGS), dim3(WG_WAVES * OGC_WAVE), \
                           lds, s, M, K, hw, (int)ntiles, w, in, out, pa, pb, pro_relu, groups, b, stats, pool);             \
    } while (0)
    // two workgroups per CU where the weights of both fit the LDS (the 128-channel layers; OGC_GEMM_STREAM_DUAL=0: one)
    static const bool dual_on = !(getenv("OGC_GEMM_STREAM_DUAL") && getenv("OGC_GEMM_STREAM_DUAL")[0] == '0');
    const bool dual = dual_on && !POOL && 2 * (lds + 1024) <= 156 * 1024 && ntiles >= 4096;
    const int wgs2 = (int)(ntiles / WG_WAVES < 512 ? ntiles / WG_WAVES : 512);
#define OGC_STREAM(KQV, EX)                                                                                                  \
    do {                                                                                                                     \
        if (dual) OGC_STREAM_D(KQV, EX, true, wgs2);                                                                         \
        else OGC_STREAM_D(KQV, EX, false, wgs);                                                                              \
    } while (0)
    const bool full_rows = M % 64 == 0;
    if (Kq == 32 && full_rows) OGC_STREAM(32, true);         // K = 125 .. 128 (the 128-channel layers)
    else if (Kq == 33 && full_rows) OGC_STREAM(33, true);    // K = 129 .. 132 (131: 128 features + 3 coordinates)
    else if (Kq <= 33) OGC_STREAM(33, false);
    else OGC_STREAM(40, false);
#undef OGC_STREAM
#undef OGC_STREAM_D
    return true;
}

template <bool T, bool STATS, bool PRO, bool POOL = false>
int gemm_launch(int b, int M, int K, int hw, int groups, const float *w, const float *in, float *out, double *stats,
                const float *pa, const float *pb, int pro_relu, hipStream_t s, PoolOut pool = PoolOut()) {
    const int Kq = (K + 3) / 4;
    if constexpr (!T && (!POOL || (STATS && PRO))) {
        static const bool wide_fp32 = [] { const char *e = getenv("OGC_BF16_WIDE_STATS"); return e && e[0] == '1'; }();
        if ((!g_matmul_bf16 || (STATS && wide_fp32)) &&
            gemm_stream_launch<PRO, STATS, POOL>(b, M, K, hw, w, in, out, pa, pb, pro_relu, s, groups, stats, pool))
            return OGC_OK;
    }
    if constexpr (T && !STATS && !PRO && !POOL) { // the plain input gradient of a 101 .. 160-channel layer
        if (!g_matmul_bf16 && gemm_stream_launch<false, false, false, true>(b, M, K, hw, w, in, out, pa, pb, pro_relu, s))
            return OGC_OK;
    }
    const size_t lds = (size_t)64 * ogc_a_ld(Kq) * sizeof(float); // (the bf16 staging needs half of it)
    dim3 grid(ogc_divup(hw, 64 * WG_WAVES), b);
#define OGC_GEMM(KQV)                                                                                                  \
    do {                                                                                                               \
        if (g_matmul_bf16)                                                                                             \
            hipLaunchKernelGGL((conv1x1_gemm_kernel<T, KQV, STATS, PRO, true, POOL>), grid, dim3(WG_WAVES * OGC_WAVE), \
                               lds, s, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, pool);                   \
        else                                                                                                           \
            hipLaunchKernelGGL((conv1x1_gemm_kernel<T, KQV, STATS, PRO, false, POOL>), grid, dim3(WG_WAVES * OGC_WAVE), \
                               lds, s, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, pool);                   \
    } while (0)
    if (Kq <= 2) OGC_GEMM(2);
    else if (Kq <= 8) OGC_GEMM(8);
    else if (Kq <= 16) OGC_GEMM(16);
    else if (Kq <= 25) OGC_GEMM(25);
    else if (POOL && !g_matmul_bf16) return OGC_ERR_UNSUPPORTED; // fp32 operands: the pooled variant is the streaming kernel's beyond K = 100
    else {
        // (bf16 operands: the register tile has MFMA time to spare at these widths — the kernel is bound by its loads and stores —
        // so the statistics / extremes epilogues ride along here too instead of a pass of their own)
        if (Kq <= 33) OGC_GEMM(33);
        else OGC_GEMM(40);
    }
#undef OGC_GEMM
    return OGC_OK;
}

int gemm_check(const char *name, int b, int M, int K, int hw, const float *w, const float *in, const float *out) {
    OGC_REQUIRE(b >= 0 && M >= 1 && K >= 1 && hw >= 1, "%s: bad shape", name);
    OGC_REQUIRE(w && in && out, "%s: null pointer", name);
    if ((hw & 63) != 0 || K > 4 * FW_KQ_MAX || (((uintptr_t)in | (uintptr_t)out) & 15) != 0) {
        ogc_set_error("%s: needs hw %% 64 == 0, K <= %d and 16-byte aligned tensors (hw=%d, K=%d)", name,
                      4 * FW_KQ_MAX, hw, K);
        return OGC_ERR_UNSUPPORTED;
    }
    // batch offsets are 64-bit in the kernels; only one sample's activation must fit 32-bit offsets
    OGC_REQUIRE((long long)M * hw < (1ll << 31) && (long long)K * hw < (1ll << 31) && b <= 65535,
                "%s: one sample exceeds 32-bit indexing", name);
    return OGC_OK;
}

} // namespace

// OUT[b, m, p] = sum_k A[m, k] IN[b, k, p];  transpose_a == 0: A = w (M x K);  != 0: A = w^T with w stored (K x M).
extern "C" int ogc_conv1x1_gemm(int b, int M, int K, int hw, int transpose_a, const float *w, const float *in,
                                float *out, ogc_stream_t stream) {
    const int rc = gemm_check("ogc_conv1x1_gemm", b, M, K, hw, w, in, out);
    if (rc != OGC_OK) return rc;
    if (b == 0) return OGC_OK;
    if (transpose_a) gemm_launch<true, false, false>(b, M, K, hw, 1, w, in, out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
    else gemm_launch<false, false, false>(b, M, K, hw, 1, w, in, out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
    OGC_CHECK_LAUNCH("ogc_conv1x1_gemm");
    return OGC_OK;
}

extern "C" int ogc_conv1x1_gn_slots(void) { return GN_SLOTS; }

extern "C" int ogc_conv1x1_gemm_stats_supported(int b, int M, int K, int hw, int affine) {
    // can ogc_conv1x1_gemm_gnstats (affine = 0) / ogc_conv1x1_gemm_affine with groups > 0 (affine = 1) take this shape?
    if (b < 1 || M < 1 || K < 1 || hw < 1) return 0;
    return (K <= 100 || g_matmul_bf16 || gemm_stream_eligible(b, M, K, hw, affine != 0, false, true)) ? 1 : 0;
}

extern "C" int ogc_conv1x1_gemm_stream_supported(int b, int M, int K, int hw) {
    // does ogc_conv1x1_gemm (either orientation) run this shape on the streaming kernel — 101 .. 160 reduction channels, enough
    // position tiles — where it is level with or ahead of the vendor GEMM?  (fp32 operands only)
    if (b < 1 || M < 1 || K < 1 || hw < 1) return 0;
    return gemm_stream_eligible(b, M, K, hw, false) ? 1 : 0;
}

extern "C" int ogc_set_matmul_precision(int bf16) {
    const int previous = g_matmul_bf16;
    g_matmul_bf16 = bf16 ? 1 : 0;
    return previous;
}

// Forward convolution that also produces the statistics of the GroupNorm that follows it.
extern "C" int ogc_conv1x1_gemm_gnstats(int b, int M, int K, int hw, int groups, const float *w, const float *in,
                                        float *out, double *stats, ogc_stream_t stream) {
    const int rc = gemm_check("ogc_conv1x1_gemm_gnstats", b, M, K, hw, w, in, out);
    if (rc != OGC_OK) return rc;
    OGC_REQUIRE(stats, "ogc_conv1x1_gemm_gnstats: null pointer");
    if (groups < 1 || groups > 32 || M % groups != 0 || (M / groups) % 4 != 0) {
        ogc_set_error("ogc_conv1x1_gemm_gnstats: needs 1 <= groups <= 32 and (M / groups) %% 4 == 0 (M=%d, groups=%d)", M,
                      groups);
        return OGC_ERR_UNSUPPORTED;
    }
    if (K > 100 && !g_matmul_bf16 && !gemm_stream_eligible(b, M, K, hw, false, false, true)) {
        // with the 33- and 40-float4 input tiles the TILE kernel has no registers to spare: the statistics epilogue costs
        // 0.08-0.15 ms there against 0.05-0.10 ms for the separate statistics pass (tools/bench_ops.py --ops conv); the
        // streaming kernel (ogc_conv1x1_gemm_stats_supported) adds them for ~nothing
        ogc_set_error("ogc_conv1x1_gemm_gnstats: not profitable for K > 100 (K=%d) on this shape; use ogc_conv1x1_gemm", K);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    if (ogc_zero_async(stats, sizeof(double) * 2 * GN_SLOTS * (size_t)b * groups, (hipStream_t)stream) != hipSuccess) {
        ogc_set_error("ogc_conv1x1_gemm_gnstats: memset failed");
        return OGC_ERR_LAUNCH;
    }
    gemm_launch<false, true, false>(b, M, K, hw, groups, w, in, out, stats, nullptr, nullptr, 0, (hipStream_t)stream);
    OGC_CHECK_LAUNCH("ogc_conv1x1_gemm_gnstats");
    return OGC_OK;
}

namespace {
int wgrad_impl(const char *name, int b, int cin, int cout, int hw, const float *x, const float *dy, float *dw,
               const float *pa, const float *pb, int pro_relu, ogc_stream_t stream, const float2 *coef2 = nullptr,
               const float2 *inj = nullptr, int s_shift = 0) {
    OGC_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && hw >= 1, "%s: bad shape", name);
    OGC_REQUIRE(x && dy && dw, "%s: null pointer", name);
    if ((hw & 15) != 0 || (((uintptr_t)x | (uintptr_t)dy) & 15) != 0) {
        ogc_set_error("%s: hw=%d must be a multiple of 16 and x/dy 16-byte aligned", name, hw);
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
    // layers of 128 channels and more on both sides: a 128 x 128 tile per workgroup, operands shared through LDS
    if (wgrad_shared_launch(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s, coef2, inj, s_shift)) {
        OGC_CHECK_LAUNCH(name);
        return OGC_OK;
    }
    // register tile per wave: (16*COB) x (16*CIB) outputs.  Small channel counts use small tiles so that no MFMA
    // work is spent on padding; wide layers use 64x64 tiles (16 accumulators) and split the rest over the grid.
    if (inj) { // (the pooled form is offered for the wide tails only: 64-row tiles)
        if (cin <= 32) wgrad_launch<4, 2>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s, coef2, inj, s_shift);
        else wgrad_launch<4, 4>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s, coef2, inj, s_shift);
        OGC_CHECK_LAUNCH(name);
        return OGC_OK;
    }
    if (cout <= 16 && cin <= 16) wgrad_launch<1, 1>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cout <= 32 && cin <= 16) wgrad_launch<2, 1>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cout <= 32 && cin <= 32) wgrad_launch<2, 2>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cin <= 16) wgrad_launch<4, 1>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cin <= 32) wgrad_launch<4, 2>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else if (cout <= 32) wgrad_launch<2, 4>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    else wgrad_launch<4, 4>(b, cin, cout, hw, x, dy, dw, pa, pb, pro_relu, s);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

extern "C" int ogc_conv1x1_wgrad(int b, int cin, int cout, int hw, const float *x, const float *dy, float *dw,
                                 ogc_stream_t stream) {
    return wgrad_impl("ogc_conv1x1_wgrad", b, cin, cout, hw, x, dy, dw, nullptr, nullptr, 0, stream);
}

// The two kernels above with the previous layer's GroupNorm (+ ReLU) folded into the operand load:
// in' = act(pa[b, k] * in + pb[b, k]).  groups == 0: no statistics of the output.
extern "C" int ogc_conv1x1_gemm_affine(int b, int M, int K, int hw, int relu, int groups, const float *w, const float *in,
                                       const float *pa, const float *pb, float *out, double *stats,
                                       ogc_stream_t stream) {
    const int rc = gemm_check("ogc_conv1x1_gemm_affine", b, M, K, hw, w, in, out);
    if (rc != OGC_OK) return rc;
    OGC_REQUIRE(pa && pb, "ogc_conv1x1_gemm_affine: null pointer");
    if (b == 0) return OGC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (groups > 0) {
        OGC_REQUIRE(stats, "ogc_conv1x1_gemm_affine: null pointer");
        if (groups > 32 || M % groups != 0 || (M / groups) % 4 != 0 || (K > 100 && !g_matmul_bf16 && !gemm_stream_eligible(b, M, K, hw, true, false, true))) {
            ogc_set_error("ogc_conv1x1_gemm_affine: output statistics need groups <= 32, (M / groups) %% 4 == 0, K <= 100 "
                          "(or a shape of the streaming kernel: ogc_conv1x1_gemm_stats_supported)");
            return OGC_ERR_UNSUPPORTED;
        }
        if (ogc_zero_async(stats, sizeof(double) * 2 * GN_SLOTS * (size_t)b * groups, s) != hipSuccess) {
            ogc_set_error("ogc_conv1x1_gemm_affine: memset failed");
            return OGC_ERR_LAUNCH;
        }
        gemm_launch<false, true, true>(b, M, K, hw, groups, w, in, out, stats, pa, pb, relu, s);
    } else {
        gemm_launch<false, false, true>(b, M, K, hw, 1, w, in, out, nullptr, pa, pb, relu, s);
    }
    OGC_CHECK_LAUNCH("ogc_conv1x1_gemm_affine");
    return OGC_OK;
}

// ogc_conv1x1_gemm_affine with output statistics, for the LAST layer of a set-abstraction MLP: positions are
// (centre, neighbour) pairs, hw = centres * nsample, and the extreme of the raw output over each neighbourhood comes out
// as well (see PoolOut) — the max-pool over the normalised activation is then ogc_group_norm_pool_extremes.
extern "C" int ogc_conv1x1_gemm_affine_pool(int b, int M, int K, int hw, int relu, int groups, int nsample,
                                            const float *w, const float *in, const float *pa, const float *pb,
                                            const float *next_gamma, float *out, double *stats, float *yext, int *aext,
                                            ogc_stream_t stream) {
    const int rc = gemm_check("ogc_conv1x1_gemm_affine_pool", b, M, K, hw, w, in, out);
    if (rc != OGC_OK) return rc;
    OGC_REQUIRE(pa && pb && next_gamma && stats && yext && aext, "ogc_conv1x1_gemm_affine_pool: null pointer");
    if (groups < 1 || groups > 32 || M % groups != 0 || (M / groups) % 4 != 0 ||
        (K > 100 && !g_matmul_bf16 && !gemm_stream_eligible(b, M, K, hw, true, true, true)) || (nsample != 16 && nsample != 32 && nsample != 64) ||
        hw % nsample != 0) {
        ogc_set_error("ogc_conv1x1_gemm_affine_pool: needs 1 <= groups <= 32, (M / groups) %% 4 == 0, K <= 100 and "
                      "nsample in {16, 32, 64} dividing hw (M=%d, groups=%d, K=%d, nsample=%d)", M, groups, K, nsample);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (ogc_zero_async(stats, sizeof(double) * 2 * GN_SLOTS * (size_t)b * groups, s) != hipSuccess) {
        ogc_set_error("ogc_conv1x1_gemm_affine_pool: memset failed");
        return OGC_ERR_LAUNCH;
    }
    PoolOut pool;
    pool.yext = yext; pool.aext = aext; pool.sign = next_gamma; pool.s = nsample;
    const int lrc = gemm_launch<false, true, true, true>(b, M, K, hw, groups, w, in, out, stats, pa, pb, relu, s, pool);
    if (lrc != OGC_OK) return lrc;
    OGC_CHECK_LAUNCH("ogc_conv1x1_gemm_affine_pool");
    return OGC_OK;
}

extern "C" int ogc_conv1x1_wgrad_affine(int b, int cin, int cout, int hw, int relu, const float *x, const float *pa,
                                        const float *pb, const float *dy, float *dw, ogc_stream_t stream) {
    OGC_REQUIRE(pa && pb, "ogc_conv1x1_wgrad_affine: null pointer");
    return wgrad_impl("ogc_conv1x1_wgrad_affine", b, cin, cout, hw, x, dy, dw, pa, pb, relu, stream);
}

// ogc_conv1x1_wgrad_affine with dy in the sparse form of ogc_group_norm_maxpool_bwd_sparse: y is the convolution's raw output
// (b, cout, hw) and g_y is rebuilt from (y, coef2, inj) while y is loaded (see POOLED at conv1x1_wgrad_kernel) — the weight
// gradient of the LAST layer of a set-abstraction MLP without the dense gradient of its pooled GroupNorm.  fp32 operands.
extern "C" int ogc_conv1x1_wgrad_affine_pooled(int b, int cin, int cout, int hw, int relu, int nsample, const float *x,
                                               const float *pa, const float *pb, const float *y, const float *coef2,
                                               const float *inj, float *dw, ogc_stream_t stream) {
    OGC_REQUIRE(pa && pb && coef2 && inj, "ogc_conv1x1_wgrad_affine_pooled: null pointer");
    const int sh = nsample == 16 ? 4 : nsample == 32 ? 5 : nsample == 64 ? 6 : -1;
    if (sh < 0 || hw % nsample != 0 || (((uintptr_t)coef2 | (uintptr_t)inj) & 7) != 0) {
        ogc_set_error("ogc_conv1x1_wgrad_affine_pooled: nsample=%d must be 16, 32 or 64 and divide hw=%d", nsample, hw);
        return OGC_ERR_UNSUPPORTED;
    }
    return wgrad_impl("ogc_conv1x1_wgrad_affine_pooled", b, cin, cout, hw, x, y, dw, pa, pb, relu, stream,
                      reinterpret_cast<const float2 *>(coef2), reinterpret_cast<const float2 *>(inj), sh);
}
