// conv1x1_h.hip — the forward / input-gradient GEMM of the 1x1 convolutions for 16-BIT tensors (bf16 in, bf16 out, bf16 operands),
// as a PERSISTENT kernel.  Reference: the Conv2d of utils/nn_util.py:45-85 inside the set-abstraction MLPs of
// models/segnet_ogcdr.py:26-41 (BASELINE config 2).
//
//     OUT[b, m, p] = sum_k A[m, k] act(pa[b, k] IN[b, k, p] + pb[b, k]),    A = w (M x K)  or  w^T (w stored K x M)
//
// conv1x1_gemm_kernel (conv1x1.hip) gives a wavefront ONE tile of 64 positions and a workgroup four: per workgroup it stages a
// 64-row tile of the weights (one L2 round trip per 16 input rows), fetches the coefficients of the folded norm (two more), loads,
// computes, stores — a chain of ~7 dependent memory round trips, ~15 us, for four tiles.  With fp32 tensors that chain hides behind
// the tiles' bytes (the kernel runs at ~5 TB/s); with 16-bit tensors the bytes halve and the chain is what is left: 64 -> 64
// channels on 4.2 M positions took 0.33 ms for 0.54 GB (1.6 TB/s), the pooled 128 -> 256 tail 0.93 ms at one workgroup per CU.
// Here a workgroup stages ALL row tiles of the weights once, packed as bf16 MFMA operands, and its wavefronts walk contiguous
// ranges of position tiles: the raw tile of step t + 1 is requested as soon as step t's has been turned into operands (its
// registers are free from then on), so the loads fly during the MFMAs and the epilogue of step t; the norm's coefficients live in
// a wave-private LDS strip that is refreshed when the sample changes; statistics go to a wave-private fp64 strip that is flushed
// to the global accumulators when the sample changes (once or twice per launch).  Products on gfx950's v_mfma_f32_16x16x32_bf16.
// Epilogues — statistics of the STORED values, neighbourhood extremes (PoolOut) — are those of conv1x1_gemm_kernel, bit for bit:
// the two kernels give identical outputs (tests/test_act16_gpu.py::test_persistent_kernel_equals_tile_kernel).
#include "conv1x1_epilogue.h"
#include "act_io.h"

namespace {

constexpr int H_WAVES = 4;

// PF: request the raw tile of step t + 1 as soon as step t's has been turned into operands (its loads then fly during step t's
// MFMAs, epilogue and stores; costs its registers through that phase).  !PF: load at the top of the step and let the other
// wavefronts of the SIMD cover the latency.  OCC: wavefronts per SIMD the kernel is built for (its register budget).
// RAG: K does not fill the KQ quads (K % 4 != 0 or K < 4 KQ): rows beyond K — read from inside the tensor — are zeroed by selects.
// (As a run-time test per quad it was sixteen branches per step.)
// ADJ (with TRANS, nothing else): the input gradient of  y = conv(relu(GroupNorm(y_prev)))  with the GroupNorm adjoint in the epilogue
// (dgrad_adjoint_kernel of gn_fused_bwd.hip, the same expressions):  out[b, m, p] = alpha[b, m] mask (sum_k w[k, m] in[b, k, p]) +
// c2[b, m] y_prev[b, m, p] + c3[b, m],  mask = [pa[b, m] y_prev + pb[b, m] > 0] — pa, pb are indexed by OUTPUT row here, coef holds
// (alpha, c2, c3) per (b, m); the five coefficients of a sample's rows sit in a wave-private LDS strip.
struct AdjIn {
    const ogc_bf16 *yprev; // (B, M, hw)
    const float *coef;     // (B, M, 3)
};

template <bool TRANS, int KQ, bool STATS, bool PRO, bool POOL, bool PF, int OCC, bool RAG, bool ADJ = false>
__global__ __launch_bounds__(H_WAVES *OGC_WAVE, OCC) void conv1x1_gemm16_kernel(
    int M, int K, int hw, int ntiles, int nbatch, int groups, const float *__restrict__ w, const ogc_bf16 *__restrict__ in,
    ogc_bf16 *__restrict__ out, double *__restrict__ stats, const float *__restrict__ pa, const float *__restrict__ pb, int pro_relu,
    PoolOut pool, AdjIn adj = AdjIn()) {
    static_assert(!ADJ || (TRANS && !STATS && !PRO && !POOL), "the adjoint epilogue belongs to the plain input gradient");
    constexpr int GQ = (KQ + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float h_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kk = lane >> 4;
    const int Kq = (K + 3) >> 2, Gq = (Kq + 3) >> 2, Mt = (M + 63) >> 6;
    const int tiles_per_img = hw >> 6;
    // LDS: [Mt][Gq * 256] packed operand quads (8 bytes) | PRO: [wave][2][KQ * 4] floats | STATS: [wave][Mt * 16][2] doubles | POOL: [Mt * 64] signs
    v4s *a_all = reinterpret_cast<v4s *>(h_lds);
    float *after_a = h_lds + (size_t)Mt * Gq * 512;
    float *cw = after_a + wave * (2 * KQ * 4);
    double *sacc = reinterpret_cast<double *>(after_a + (PRO ? H_WAVES * 2 * KQ * 4 : 0)) + wave * (Mt * 32);
    float *sgn_all = after_a + (PRO ? H_WAVES * 2 * KQ * 4 : 0) + (STATS ? H_WAVES * Mt * 64 : 0);
    float *cf = after_a + wave * (Mt * 64 * 5); // ADJ: [Mt * 64][5] = pa, pb, alpha, c2, c3 of the current sample's rows
    for (int mt = 0; mt < Mt; ++mt) // (nothing else is live yet: all of a tile's loads in flight at once)
        ogc_stage_weight_tile_bf16<TRANS, H_WAVES, GQ>(a_all + (size_t)mt * Gq * 256, w, mt * 64, M, K, Kq);
    if constexpr (POOL) {
        for (int t = threadIdx.x; t < Mt * 64; t += H_WAVES * OGC_WAVE) sgn_all[t] = (t < M && pool.sign[t] < 0.f) ? -1.f : 1.f;
    }
    __syncthreads();

    // ---- a wave's walk over its contiguous range of position tiles
    const int nw = gridDim.x * H_WAVES;
    const int per = (ntiles + nw - 1) / nw;
    int t = (blockIdx.x * H_WAVES + wave) * per;
    const int t_end = min(ntiles, t + per);
    if (t >= t_end) return;
    // unconditional loads of one shape: rows beyond K re-read the last row (zeroed below when K is not a multiple of 4; rows of
    // whole missing quads meet zero weights only after that zeroing too)
    // addresses: a wave-uniform base per row quad + one of two 32-bit lane offsets (the last real quad clamps its rows)
    const unsigned off_main = (unsigned)(kk * hw + 4 * j);
    const unsigned off_last = (unsigned)(min(kk, K - 1 - (Kq - 1) * 4) * hw + 4 * j);
    auto load_tile = [&](int tt, uint2(&x)[KQ]) {
        tt = min(tt, t_end - 1); // beyond the wave's last tile: a harmless re-read
        const int b = tt / tiles_per_img, p0 = (tt - b * tiles_per_img) * 64;
        const ogc_bf16 *inb = in + (size_t)b * K * hw + p0;
        unsigned om = off_main, ol = off_last; // (opaque: the per-quad choice below is loop-invariant, and hoisted it is KQ 64-bit lane
        asm volatile("" : "+v"(om), "+v"(ol)); //  addresses held — and spilled — across the loop)
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const ogc_bf16 *rowq = inb + (size_t)(min(q, Kq - 1) * 4) * hw; // quads beyond K re-read the last one
            x[q] = *reinterpret_cast<const uint2 *>(rowq + (q >= Kq - 1 ? ol : om));
        }
    };
    int coef_b = -1, stat_b = -1;
    auto flush_stats = [&]() {
        if constexpr (STATS) {
            if (stat_b >= 0) {
                const int cpg = M / groups; // channels per group, a multiple of 4 on this path
                double *dst = stats + ((size_t)(blockIdx.x % GN_SLOTS) * nbatch + stat_b) * 2 * groups;
                for (int sl = lane; sl < Mt * 16; sl += OGC_WAVE) { // slot = (mt * 4 + a) * 4 + kk -> rows mt * 64 + a * 16 + kk * 4 ..
                    const int m = (sl >> 4) * 64 + ((sl >> 2) & 3) * 16 + (sl & 3) * 4;
                    if (m < M) {
                        unsafeAtomicAdd(dst + 2 * (m / cpg), sacc[2 * sl]);
                        unsafeAtomicAdd(dst + 2 * (m / cpg) + 1, sacc[2 * sl + 1]);
                    }
                }
            }
            for (int sl = lane; sl < Mt * 16; sl += OGC_WAVE) { sacc[2 * sl] = 0.0; sacc[2 * sl + 1] = 0.0; }
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (STATS) flush_stats(); // (clears the strip)

    uint2 raw[KQ];
    if (PF) load_tile(t, raw);
    for (; t < t_end; ++t) {
        const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 64;
        if (!PF) load_tile(t, raw);
        // (opaque per step: the weights' operand reads and the coefficient reads below do not depend on the step, and hoisted out
        // of this loop they would live in ~100 registers across it — the kernel then spills its tiles to scratch memory)
        int aoff = j * 4 + kk, coff = kk;
        unsigned soff = (unsigned)(kk * 4 * hw + 4 * j);
        asm volatile("" : "+v"(aoff), "+v"(coff), "+v"(soff));
        if constexpr (STATS) {
            if (b != stat_b) { flush_stats(); stat_b = b; }
        }
        if constexpr (PRO) {
            if (b != coef_b) {
                coef_b = b;
                for (int r = lane; r < KQ * 4; r += OGC_WAVE) { // padding rows: act(0 * x + 0) = 0
                    cw[r] = r < K ? pa[(size_t)b * K + r] : 0.f;
                    cw[KQ * 4 + r] = r < K ? pb[(size_t)b * K + r] : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if constexpr (ADJ) {
            if (b != coef_b) {
                coef_b = b;
                for (int r = lane; r < Mt * 64; r += OGC_WAVE) {
                    const bool in_m = r < M;
                    const size_t e = (size_t)b * M + r;
                    cf[r * 5 + 0] = in_m ? pa[e] : 0.f;
                    cf[r * 5 + 1] = in_m ? pb[e] : 0.f;
                    cf[r * 5 + 2] = in_m ? adj.coef[e * 3] : 0.f;
                    cf[r * 5 + 3] = in_m ? adj.coef[e * 3 + 1] : 0.f;
                    cf[r * 5 + 4] = in_m ? adj.coef[e * 3 + 2] : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- the raw tile -> packed operands (k-slot i of lane group kk of group g = input row 4 (4 g + i) + kk)
        v4s xb[GQ][4];
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            float4 r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 4 * g + i;
                if (q < KQ) {
                    const uint2 u = raw[q];
                    float4 v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                                           __uint_as_float(u.y & 0xFFFF0000u));
                    if constexpr (PRO) {
                        const float ca = cw[q * 4 + coff], cb = cw[KQ * 4 + q * 4 + coff];
                        v.x = fmaf(ca, v.x, cb); v.y = fmaf(ca, v.y, cb); v.z = fmaf(ca, v.z, cb); v.w = fmaf(ca, v.w, cb);
                        if (pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
                    if constexpr (RAG) {
                        if (q * 4 + kk >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    r[i] = v;
                } else {
                    r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            xb[g][0] = ogc_pack_bf16(r[0].x, r[1].x, r[2].x, r[3].x);
            xb[g][1] = ogc_pack_bf16(r[0].y, r[1].y, r[2].y, r[3].y);
            xb[g][2] = ogc_pack_bf16(r[0].z, r[1].z, r[2].z, r[3].z);
            xb[g][3] = ogc_pack_bf16(r[0].w, r[1].w, r[2].w, r[3].w);
        }
        if (PF) load_tile(t + 1, raw); // in flight during this step's MFMAs, epilogue and stores

        ogc_bf16 *outb = out + (size_t)b * M * hw + p0;
        for (int mt = 0; mt < Mt; ++mt) {
            const v4s *a_bf = a_all + (size_t)mt * Gq * 256;
            const int nblk = min(4, (M - mt * 64 + 15) >> 4);
            v4f acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
            uint2 yv[ADJ ? 16 : 1]; // ADJ: y_prev at this lane's output positions, requested in front of the tile's MFMAs
            if constexpr (ADJ) {
                const ogc_bf16 *yb = adj.yprev + (size_t)b * M * hw + p0;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mu = min(mt * 64 + a * 16 + r, M - 1 - min(kk * 4, M - 1)); // (rows beyond M: a harmless re-read)
                        yv[a * 4 + r] = *reinterpret_cast<const uint2 *>(yb + (size_t)mu * hw + soff);
                    }
            }
#pragma unroll
            for (int g = 0; g + 1 < GQ; g += 2) {
                if (4 * (g + 1) < Kq) { // both groups were staged
                    v4s av0[4], av1[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        av0[a] = a_bf[(g * 64 + a * 16) * 4 + aoff];
                        av1[a] = a_bf[((g + 1) * 64 + a * 16) * 4 + aoff];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[a][c] = ogc_mfma_bf16_k32(av0[a], av1[a], xb[g][c], xb[g + 1][c], acc[a][c]);
                } else if (4 * g < Kq) {
                    v4s av[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) av[a] = a_bf[(g * 64 + a * 16) * 4 + aoff];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[a], xb[g][c], acc[a][c], 0, 0, 0);
                }
            }
            if constexpr (GQ % 2 == 1) {
                constexpr int g = GQ - 1;
                if (4 * g < Kq) {
                    v4s av[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) av[a] = a_bf[(g * 64 + a * 16) * 4 + aoff];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[a], xb[g][c], acc[a][c], 0, 0, 0);
                }
            }
            if constexpr (STATS || POOL) { // what follows sees the values as they are stored
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[a][c][r] = ogc_as_stored<ogc_bf16>(acc[a][c][r]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mu = mt * 64 + a * 16 + r; // the row is mu + 4 kk (C/D layout: row (l >> 4) * 4 + r, column l & 15):
                    float4 o = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
                    if constexpr (ADJ) {
                        const float *c5 = cf + (mu + coff * 4) * 5;
                        const float fa = c5[0], fb = c5[1], al = c5[2], c2 = c5[3], c3 = c5[4];
                        const uint2 u = yv[a * 4 + r];
                        const float4 y = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                                                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
                        if (pro_relu) {
                            o.x = fmaf(fa, y.x, fb) > 0.f ? o.x : 0.f; o.y = fmaf(fa, y.y, fb) > 0.f ? o.y : 0.f;
                            o.z = fmaf(fa, y.z, fb) > 0.f ? o.z : 0.f; o.w = fmaf(fa, y.w, fb) > 0.f ? o.w : 0.f;
                        }
                        o.x = fmaf(al, o.x, fmaf(c2, y.x, c3)); o.y = fmaf(al, o.y, fmaf(c2, y.y, c3));
                        o.z = fmaf(al, o.z, fmaf(c2, y.z, c3)); o.w = fmaf(al, o.w, fmaf(c2, y.w, c3));
                    }
                    if (mu + kk * 4 < M)                 // a wave-uniform base + ONE 32-bit lane offset (see load_tile)
                        ogc_st4(outb + (size_t)mu * hw + soff, o);
                }
            if constexpr (POOL) {
                const int m0p = mt * 64;
                const int centres = hw / pool.s;
                const int centre = (p0 + 4 * j) / pool.s;
                const size_t o0 = ((size_t)b * M + m0p + kk * 4) * centres + centre;
                const float *sg = sgn_all + m0p;
                if (pool.s == 64) ogc_pool_extremes_epilogue<16>(acc, nblk, sg, j, kk, m0p, M, centres, 64, pool.yext + o0, pool.aext + o0);
                else if (pool.s == 32) ogc_pool_extremes_epilogue<8>(acc, nblk, sg, j, kk, m0p, M, centres, 32, pool.yext + o0, pool.aext + o0);
                else ogc_pool_extremes_epilogue<4>(acc, nblk, sg, j, kk, m0p, M, centres, 16, pool.yext + o0, pool.aext + o0);
            }
            if constexpr (STATS) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < nblk) {
                        float sm = 0.f, sq = 0.f; // the lane's 4 rows x 4 positions, in conv1x1_gemm_kernel's order; rows >= M are exact zeros
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
    }
    if (STATS) flush_stats();
}

// ---- the same persistent walk for fp32 tensors with at most 64 reduction channels (round 5, last session) ---------------------------
// conv1x1_gemm_kernel's narrow fp32 layers — C4's first two levels: 32 -> 32 / 64 and 64 -> 64 / 128 channels on 2.1 M / 1.05 M
// positions — run at 2.7-3.5 TB/s: the same chain of dependent round trips per four tiles that the 16-bit form exposed, half hidden by
// twice the bytes.  (Wider layers have conv1x1_gemm_stream_kernel.)  Measured: a gain at 64 reduction channels (3.1 -> 4.0 TB/s),
// a loss at 32 — the launcher takes K = 33 .. 64 only; the C4 step does not move either way (10.76-10.79 ms).  Operands stay fp32 on v_mfma_f32_16x16x4_f32 and every
// accumulator sees its products in conv1x1_gemm_kernel's order (row quads ascending): the outputs are bit-identical
// (tests/test_ops_gpu.py::test_persistent_fp32_kernel_equals_tile_kernel).  Two register sets: the operand tile (transformed in
// place) and the raw tile of the next step, requested as soon as the operands exist.
template <bool TRANS, int KQ, bool STATS, bool PRO, bool POOL, int OCC>
__global__ __launch_bounds__(H_WAVES *OGC_WAVE, OCC) void conv1x1_gemm32_kernel(
    int M, int K, int hw, int ntiles, int nbatch, int groups, const float *__restrict__ w, const float *__restrict__ in,
    float *__restrict__ out, double *__restrict__ stats, const float *__restrict__ pa, const float *__restrict__ pb, int pro_relu,
    PoolOut pool) {
    extern __shared__ __attribute__((aligned(16))) float h_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kk = lane >> 4;
    const int Kq = (K + 3) >> 2, Mt = (M + 63) >> 6, a_ld = ogc_a_ld(Kq);
    const int tiles_per_img = hw >> 6;
    // LDS: [Mt][64][a_ld] weights | PRO: [wave][2][KQ * 4] | STATS: [wave][Mt * 16][2] doubles | POOL: [Mt * 64] signs
    float *after_a = h_lds + (size_t)Mt * 64 * a_ld;
    float *cw = after_a + wave * (2 * KQ * 4);
    double *sacc = reinterpret_cast<double *>(after_a + (PRO ? H_WAVES * 2 * KQ * 4 : 0)) + wave * (Mt * 32);
    float *sgn_all = after_a + (PRO ? H_WAVES * 2 * KQ * 4 : 0) + (STATS ? H_WAVES * Mt * 64 : 0);
    for (int mt = 0; mt < Mt; ++mt) ogc_stage_weight_tile<TRANS, H_WAVES>(h_lds + (size_t)mt * 64 * a_ld, w, mt * 64, M, K, Kq);
    if constexpr (POOL) {
        for (int t = threadIdx.x; t < Mt * 64; t += H_WAVES * OGC_WAVE) sgn_all[t] = (t < M && pool.sign[t] < 0.f) ? -1.f : 1.f;
    }
    __syncthreads();
    const int nw = gridDim.x * H_WAVES;
    const int per = (ntiles + nw - 1) / nw;
    int t = (blockIdx.x * H_WAVES + wave) * per;
    const int t_end = min(ntiles, t + per);
    if (t >= t_end) return;
    const unsigned off_main = (unsigned)(kk * hw + 4 * j);
    const unsigned off_last = (unsigned)(min(kk, K - 1 - (Kq - 1) * 4) * hw + 4 * j);
    auto load_tile = [&](int tt, float4(&x)[KQ]) {
        tt = min(tt, t_end - 1);
        const int b = tt / tiles_per_img, p0 = (tt - b * tiles_per_img) * 64;
        const float *inb = in + (size_t)b * K * hw + p0;
        unsigned om = off_main, ol = off_last; // (opaque per step: see conv1x1_gemm16_kernel)
        asm volatile("" : "+v"(om), "+v"(ol));
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const float *rowq = inb + (size_t)(min(q, Kq - 1) * 4) * hw;
            x[q] = *reinterpret_cast<const float4 *>(rowq + (q >= Kq - 1 ? ol : om));
        }
    };
    int coef_b = -1, stat_b = -1;
    auto flush_stats = [&]() {
        if constexpr (STATS) {
            if (stat_b >= 0) {
                const int cpg = M / groups;
                double *dst = stats + ((size_t)(blockIdx.x % GN_SLOTS) * nbatch + stat_b) * 2 * groups;
                for (int sl = lane; sl < Mt * 16; sl += OGC_WAVE) {
                    const int m = (sl >> 4) * 64 + ((sl >> 2) & 3) * 16 + (sl & 3) * 4;
                    if (m < M) {
                        unsafeAtomicAdd(dst + 2 * (m / cpg), sacc[2 * sl]);
                        unsafeAtomicAdd(dst + 2 * (m / cpg) + 1, sacc[2 * sl + 1]);
                    }
                }
            }
            for (int sl = lane; sl < Mt * 16; sl += OGC_WAVE) { sacc[2 * sl] = 0.0; sacc[2 * sl + 1] = 0.0; }
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (STATS) flush_stats();

    float4 raw[KQ], x[KQ];
    load_tile(t, raw);
    for (; t < t_end; ++t) {
        const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 64;
        int aoff = j * a_ld + kk, coff = kk;
        unsigned soff = (unsigned)(kk * 4 * hw + 4 * j);
        asm volatile("" : "+v"(aoff), "+v"(coff), "+v"(soff));
        if constexpr (STATS) {
            if (b != stat_b) { flush_stats(); stat_b = b; }
        }
        if constexpr (PRO) {
            if (b != coef_b) {
                coef_b = b;
                for (int r = lane; r < KQ * 4; r += OGC_WAVE) {
                    cw[r] = r < K ? pa[(size_t)b * K + r] : 0.f;
                    cw[KQ * 4 + r] = r < K ? pb[(size_t)b * K + r] : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) { // the operand tile: the folded norm in place; rows beyond K are exact zeros
            float4 v = raw[q];
            if constexpr (PRO) {
                const float ca = cw[q * 4 + coff], cb = cw[KQ * 4 + q * 4 + coff];
                v.x = fmaf(ca, v.x, cb); v.y = fmaf(ca, v.y, cb); v.z = fmaf(ca, v.z, cb); v.w = fmaf(ca, v.w, cb);
                if (pro_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (q * 4 + kk >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
            x[q] = v;
        }
        load_tile(t + 1, raw);

        float *outb = out + (size_t)b * M * hw + p0;
        for (int mt = 0; mt < Mt; ++mt) {
            const float *at = h_lds + (size_t)mt * 64 * a_ld;
            const int nblk = min(4, (M - mt * 64 + 15) >> 4);
            v4f acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                if (q < Kq) {
                    float av[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) av[a] = at[a * 16 * a_ld + q * 4 + aoff]; // A[mt * 64 + 16 a + j][4 q + kk]
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (a < nblk) {
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
                    const int mu = mt * 64 + a * 16 + r;
                    if (mu + kk * 4 < M)
                        *reinterpret_cast<float4 *>(outb + (size_t)mu * hw + soff) =
                            make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
                }
            if constexpr (POOL) {
                const int m0p = mt * 64;
                const int centres = hw / pool.s;
                const int centre = (p0 + 4 * j) / pool.s;
                const size_t o0 = ((size_t)b * M + m0p + kk * 4) * centres + centre;
                const float *sg = sgn_all + m0p;
                if (pool.s == 64) ogc_pool_extremes_epilogue<16>(acc, nblk, sg, j, kk, m0p, M, centres, 64, pool.yext + o0, pool.aext + o0);
                else if (pool.s == 32) ogc_pool_extremes_epilogue<8>(acc, nblk, sg, j, kk, m0p, M, centres, 32, pool.yext + o0, pool.aext + o0);
                else ogc_pool_extremes_epilogue<4>(acc, nblk, sg, j, kk, m0p, M, centres, 16, pool.yext + o0, pool.aext + o0);
            }
            if constexpr (STATS) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < nblk) {
                        float sm = 0.f, sq = 0.f;
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
    }
    if (STATS) flush_stats();
}

template <bool TRANS, int KQ, bool STATS, bool PRO, bool POOL, int OCC>
bool gemm32_go(int b, int M, int K, int hw, int groups, const float *w, const float *in, float *out, double *stats, const float *pa,
               const float *pb, int pro_relu, hipStream_t s, PoolOut pool) {
    const int Kq = (K + 3) / 4, Mt = (M + 63) / 64;
    const size_t lds = ((size_t)Mt * 64 * ogc_a_ld(Kq) + (PRO ? H_WAVES * 2 * KQ * 4 : 0) + (STATS ? H_WAVES * Mt * 64 : 0) +
                        (POOL ? Mt * 64 : 0)) * sizeof(float);
    const long long ntiles = (long long)b * (hw / 64);
    int per_cu = OCC;
    while (per_cu > 1 && (lds + 512) * per_cu > 156 * 1024) --per_cu;
    if (lds > 78 * 1024 || ntiles >= (1ll << 31)) return false;
    static bool raised = false;
    const void *fn = reinterpret_cast<const void *>(&conv1x1_gemm32_kernel<TRANS, KQ, STATS, PRO, POOL, OCC>);
    if (!raised) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        raised = true;
    }
    long long wgs = 256ll * per_cu;
    if (wgs > ntiles / H_WAVES) wgs = ntiles / H_WAVES;
    hipLaunchKernelGGL((conv1x1_gemm32_kernel<TRANS, KQ, STATS, PRO, POOL, OCC>), dim3((unsigned)wgs), dim3(H_WAVES * OGC_WAVE), lds, s, M,
                       K, hw, (int)ntiles, b, groups, w, in, out, stats, pa, pb, pro_relu, pool);
    return true;
}

template <bool TRANS, bool STATS, bool PRO, bool POOL>
bool gemm32_kq(int b, int M, int K, int hw, int groups, const float *w, const float *in, float *out, double *stats, const float *pa,
               const float *pb, int pro_relu, hipStream_t s, PoolOut pool) {
    const int Kq = (K + 3) / 4;
    if (Kq <= 8) return gemm32_go<TRANS, 8, STATS, PRO, POOL, 2>(b, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, s, pool);
    if (Kq <= 16) return gemm32_go<TRANS, 16, STATS, PRO, POOL, 2>(b, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, s, pool);
    return false;
}

size_t gemm16_lds(int M, int K, int KQ, bool stats_on, bool pro, bool pool_on, bool adj = false) {
    const int Kq = (K + 3) / 4, Gq = (Kq + 3) / 4, Mt = (M + 63) / 64;
    return ((size_t)Mt * Gq * 512 + (pro ? H_WAVES * 2 * KQ * 4 : 0) + (stats_on ? H_WAVES * Mt * 64 : 0) + (pool_on ? Mt * 64 : 0) +
            (adj ? H_WAVES * Mt * 64 * 5 : 0)) * sizeof(float);
}

template <bool TRANS, int KQ, bool STATS, bool PRO, bool POOL, bool PF, int OCC, bool RAG, bool ADJ = false>
bool gemm16_go(int b, int M, int K, int hw, int groups, const float *w, const ogc_bf16 *in, ogc_bf16 *out, double *stats,
               const float *pa, const float *pb, int pro_relu, hipStream_t s, PoolOut pool, AdjIn adj = AdjIn()) {
    const size_t lds = gemm16_lds(M, K, KQ, STATS, PRO, POOL, ADJ);
    const long long ntiles = (long long)b * (hw / 64);
    int per_cu = OCC; // workgroups per CU: what the registers allow, and the LDS (160 KiB per CU)
    while (per_cu > 1 && (lds + 512) * per_cu > 156 * 1024) --per_cu;
    if (lds > 78 * 1024 || ntiles >= (1ll << 31)) return false;
    static bool raised = false;
    const void *fn = reinterpret_cast<const void *>(&conv1x1_gemm16_kernel<TRANS, KQ, STATS, PRO, POOL, PF, OCC, RAG, ADJ>);
    if (!raised) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        raised = true;
    }
    long long wgs = 256ll * per_cu;
    if (wgs > ntiles / H_WAVES) wgs = ntiles / H_WAVES;
    hipLaunchKernelGGL((conv1x1_gemm16_kernel<TRANS, KQ, STATS, PRO, POOL, PF, OCC, RAG, ADJ>), dim3((unsigned)wgs), dim3(H_WAVES * OGC_WAVE), lds,
                       s, M, K, hw, (int)ntiles, b, groups, w, in, out, stats, pa, pb, pro_relu, pool, adj);
    return true;
}

template <bool TRANS, bool STATS, bool PRO, bool POOL>
bool gemm16_kq(int b, int M, int K, int hw, int groups, const float *w, const ogc_bf16 *in, ogc_bf16 *out, double *stats,
               const float *pa, const float *pb, int pro_relu, hipStream_t s, PoolOut pool) {
    const int Kq = (K + 3) / 4;
#define H_ARGS b, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, s, pool
    // two wavefronts per SIMD either way (three, built without the prefetch, spill at K <= 64); the prefetched tile does not fit the
    // registers next to 33 quads of operands
    const bool exact = (K & 3) == 0;
    if (Kq <= 16) {
        if (exact && Kq == 16) return gemm16_go<TRANS, 16, STATS, PRO, POOL, true, 2, false>(H_ARGS);
        return gemm16_go<TRANS, 16, STATS, PRO, POOL, true, 2, true>(H_ARGS);
    }
    if (exact && Kq == 32) return gemm16_go<TRANS, 32, STATS, PRO, POOL, false, 2, false>(H_ARGS);
    if (Kq <= 33) return gemm16_go<TRANS, 33, STATS, PRO, POOL, false, 2, true>(H_ARGS);
#undef H_ARGS
    return false;
}

} // namespace

// The persistent kernel for this call, or false: fewer than 8 position tiles per wavefront of a full launch (the staging would not
// amortise), more than 132 reduction channels, weights beyond 78 KiB of LDS, OGC_GEMM16=0 in the environment (A/B runs, tests).
bool ogc_gemm16_launch(bool transpose_a, bool stats_on, bool pro, bool pool_on, int b, int M, int K, int hw, int groups,
                       const float *w, const unsigned short *in, unsigned short *out, double *stats, const float *pa,
                       const float *pb, int pro_relu, hipStream_t s, const float *pool_sign, float *pool_yext, int *pool_aext,
                       int pool_s) {
    const char *e = getenv("OGC_GEMM16"); // (read per call: tests flip it)
    if (e && e[0] == '0') return false;
    if ((hw & 63) != 0 || (long long)b * (hw / 64) < 8192) return false;
    PoolOut pool;
    pool.yext = pool_yext; pool.aext = pool_aext; pool.sign = pool_sign; pool.s = pool_s;
#define H_GO(T, ST, PR, PO) return gemm16_kq<T, ST, PR, PO>(b, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, s, pool)
    if (transpose_a) {
        if (!stats_on && !pro && !pool_on) H_GO(true, false, false, false);
        return false;
    }
    if (pool_on) {
        if (stats_on && pro) H_GO(false, true, true, true);
        return false;
    }
    if (stats_on && pro) H_GO(false, true, true, false);
    if (!stats_on && pro) H_GO(false, false, true, false);
    if (!stats_on && !pro) H_GO(false, false, false, false);
#undef H_GO
    return false;
}

// The adjoint input gradient (ogc_conv1x1_dgrad_adjoint_h, dense form) on the persistent kernel, or false (fewer than 8192 position
// tiles, more than 64 reduction channels — the 33-quad form has no registers for the y_prev rows —, OGC_GEMM16=0).
bool ogc_gemm16_adjoint_launch(int b, int M, int K, int hw, int relu, const float *w, const unsigned short *gy,
                               const unsigned short *yprev, const float *pa, const float *pb, const float *coef,
                               unsigned short *out, hipStream_t s) {
    const char *e = getenv("OGC_GEMM16");
    if (e && e[0] == '0') return false;
    const int Kq = (K + 3) / 4;
    if ((hw & 63) != 0 || (long long)b * (hw / 64) < 8192 || Kq > 16 || M > 128) return false;
    AdjIn adj;
    adj.yprev = yprev; adj.coef = coef;
    if ((K & 3) == 0 && Kq == 16)
        return gemm16_go<true, 16, false, false, false, true, 2, false, true>(b, M, K, hw, 1, w, gy, out, nullptr, pa, pb, relu, s,
                                                                              PoolOut(), adj);
    return gemm16_go<true, 16, false, false, false, true, 2, true, true>(b, M, K, hw, 1, w, gy, out, nullptr, pa, pb, relu, s, PoolOut(),
                                                                         adj);
}

// The persistent fp32 kernel for this call (fp32 operands, <= 64 reduction channels, >= 8192 position tiles), or false.
// OGC_GEMM32=0: the tile kernel (A/B runs, tests).
bool ogc_gemm32_launch(bool transpose_a, bool stats_on, bool pro, bool pool_on, int b, int M, int K, int hw, int groups,
                       const float *w, const float *in, float *out, double *stats, const float *pa, const float *pb, int pro_relu,
                       hipStream_t s, const float *pool_sign, float *pool_yext, int *pool_aext, int pool_s) {
    const char *e = getenv("OGC_GEMM32");
    if (e && e[0] == '0') return false;
    // 33 .. 64 reduction channels only: at 32 the tile kernel (32 registers of tile, five wavefronts per SIMD) is the faster one —
    // 16 x 131072 positions, 32 -> 32: 0.168 against 0.179 ms, pooled 32 -> 64: 0.234 against 0.284; 64 -> 64 on 16 x 65536: 0.173
    // against 0.134, pooled 64 -> 128: 0.292 against 0.270 (tools/gemm32_bench.py).  OGC_GEMM32=all: every K <= 64 (tests).
    const bool all = e && e[0] == 'a';
    if ((hw & 63) != 0 || (long long)b * (hw / 64) < 8192 || K > 64 || (K <= 32 && !all)) return false;
    PoolOut pool;
    pool.yext = pool_yext; pool.aext = pool_aext; pool.sign = pool_sign; pool.s = pool_s;
#define H_GO(T, ST, PR, PO) return gemm32_kq<T, ST, PR, PO>(b, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, s, pool)
    if (transpose_a) {
        if (!stats_on && !pro && !pool_on) H_GO(true, false, false, false);
        return false;
    }
    if (pool_on) {
        if (stats_on && pro) H_GO(false, true, true, true);
        return false;
    }
    if (stats_on && pro) H_GO(false, true, true, false);
    if (stats_on && !pro) H_GO(false, true, false, false);
    if (!stats_on && pro) H_GO(false, false, true, false);
    if (!stats_on && !pro) H_GO(false, false, false, false);
#undef H_GO
    return false;
}
