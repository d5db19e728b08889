// conv1x1.hip — forward and input-gradient GEMMs of the per-point (1x1) convolutions of the shared MLPs on the MFMA pipe
// (reference: utils/nn_util.py:45-85); the weight gradients live in conv1x1_wgrad.hip, what both share in conv1x1_shared.h.
#include "conv1x1_shared.h"
#include "act_io.h"
#include "conv1x1_epilogue.h"

int ogc_g_matmul_bf16 = 0;


namespace {

// ---- forward / input-gradient GEMM:  OUT[b, m, p] = sum_k A[m, k] * IN[b, k, p] -----------------------------------
// forward: A = W (M = Cout, K = Cin, IN = x);  input gradient: A = W^T (M = Cin, K = Cout, IN = dy).
// A wave owns 64 positions.  Lane (k' = l >> 4, j = l & 15) loads the float4 IN[k0 + k'][p0 + 4j .. 4j+3]; MFMA column
// block c (c = 0..3) takes component c of it, i.e. column j of block c is position p0 + 4j + c.  The same lane then
// holds the four consecutive positions of an output row in its four column-block accumulators, so results leave as
// float4 stores — no shuffles on either side.  The whole IN tile (K x 64) stays in registers (read from HBM exactly
// once); A is staged through LDS in [k/4][m][4] order (conflict-free ds_read_b32 of the A operand) one 64-row tile at
// a time and shared by the four waves of the workgroup.
// (FW_KQ_MAX, GN_SLOTS, PoolOut and the pooled epilogue: conv1x1_epilogue.h, shared with conv1x1_h.hip)

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
// IT / OT: element types of `in` and `out` (float, or ogc_bf16 for activations kept in 16 bits: act_io.h); the statistics and
// the extremes are then those of the ROUNDED outputs.
template <bool TRANSPOSE_A, int KQ, bool STATS, bool PRO, bool BF, bool POOL = false, typename IT = float, typename OT = float>
__global__ __launch_bounds__(WG_WAVES *OGC_WAVE) void conv1x1_gemm_kernel(int M, int K, int hw, int groups,
                                                                          const float *__restrict__ w, // (Cout, Cin)
                                                                          const IT *__restrict__ in,
                                                                          OT *__restrict__ out,
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
    const IT *inb = in + (size_t)b * K * hw;
    OT *outb = out + (size_t)b * M * hw;
    if (STATS && threadIdx.x < 2 * groups) s_stats[threadIdx.x] = 0.0;

    float4 xin[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const int row = q * 4 + kk;
        xin[q] = (live && q < Kq && row < K) ? ogc_ld4(inb + (size_t)row * hw + p0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
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
            // (one group per round.  All groups' loads in flight at once — ogc_stage_weight_tile_bf16 of conv_stage.h, which the
            // adjoint kernel of gn_fused_bwd.hip uses — measured SLOWER here: 128 -> 256 pooled 0.93 -> 1.66 ms, 64 -> 64 0.33 -> 0.41)
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
            // pairs of groups (32 input rows) on gfx950's v_mfma_f32_16x16x32_bf16, a single last group on the 16x16x16 form
#pragma unroll
            for (int g = 0; g + 1 < GQ; g += 2) {
                if (4 * (g + 1) < Kq) { // both groups hold rows (a group that was not staged is stale LDS, not zeros)
                    v4s av0[4], av1[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        av0[a] = a_bf[(g * 64 + a * 16 + j) * 4 + kk];
                        av1[a] = a_bf[((g + 1) * 64 + a * 16 + j) * 4 + kk];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (a < nblk) {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[a][c] = ogc_mfma_bf16_k32(av0[a], av1[a], xb[g][c], xb[g + 1][c], acc[a][c]);
                        }
                    }
                } else if (4 * g < Kq) {
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
            if constexpr (GQ % 2 == 1) {
                constexpr int g = GQ - 1;
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
            if constexpr (sizeof(OT) == 2 && (STATS || POOL)) { // what follows sees the values as they are stored
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[a][c][r] = ogc_as_stored<OT>(acc[a][c][r]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + a * 16 + kk * 4 + r; // C/D layout: row (l >> 4) * 4 + r, column l & 15
                    if (m < M) {
                        const float4 o = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
                        ogc_st4(outb + (size_t)m * hw + p0 + 4 * j, o);
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

template <bool T, bool STATS, bool PRO, bool POOL = false, typename AT = float>
int gemm_launch(int b, int M, int K, int hw, int groups, const float *w, const AT *in, AT *out, double *stats,
                const float *pa, const float *pb, int pro_relu, hipStream_t s, PoolOut pool = PoolOut()) {
    const int Kq = (K + 3) / 4;
    constexpr bool F32 = sizeof(AT) == 4;
    if constexpr (F32 && !T && (!POOL || (STATS && PRO))) {
        static const bool wide_fp32 = [] { const char *e = getenv("OGC_BF16_WIDE_STATS"); return e && e[0] == '1'; }();
        if ((!g_matmul_bf16 || (STATS && wide_fp32)) &&
            gemm_stream_launch<PRO, STATS, POOL>(b, M, K, hw, w, in, out, pa, pb, pro_relu, s, groups, stats, pool))
            return OGC_OK;
    }
    if constexpr (F32 && T && !STATS && !PRO && !POOL) { // the plain input gradient of a 101 .. 160-channel layer
        if (!g_matmul_bf16 && gemm_stream_launch<false, false, false, true>(b, M, K, hw, w, in, out, pa, pb, pro_relu, s))
            return OGC_OK;
    }
    if constexpr (F32) { // narrow fp32 layers with many position tiles: the persistent kernel of conv1x1_h.hip (same arithmetic)
        if (!g_matmul_bf16 && ogc_gemm32_launch(T, STATS, PRO, POOL, b, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, s,
                                                pool.sign, pool.yext, pool.aext, pool.s))
            return OGC_OK;
    }
    if constexpr (!F32) { // 16-bit tensors: the persistent kernel of conv1x1_h.hip where there are enough position tiles
        if (ogc_gemm16_launch(T, STATS, PRO, POOL, b, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, s, pool.sign, pool.yext,
                              pool.aext, pool.s))
            return OGC_OK;
    }
    const size_t lds = (size_t)64 * ogc_a_ld(Kq) * sizeof(float); // (the bf16 staging needs half of it)
    dim3 grid(ogc_divup(hw, 64 * WG_WAVES), b);
    // 16-bit activations come with bf16 operands only (the entry points check the precision switch)
#define OGC_GEMM(KQV)                                                                                                  \
    do {                                                                                                               \
        if constexpr (!F32)                                                                                            \
            hipLaunchKernelGGL((conv1x1_gemm_kernel<T, KQV, STATS, PRO, true, POOL, AT, AT>), grid,                     \
                               dim3(WG_WAVES * OGC_WAVE), lds, s, M, K, hw, groups, w, in, out, stats, pa, pb, pro_relu, pool); \
        else if (g_matmul_bf16)                                                                                        \
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

template <typename AT>
int gemm_check(const char *name, int b, int M, int K, int hw, const float *w, const AT *in, const AT *out) {
    OGC_REQUIRE(b >= 0 && M >= 1 && K >= 1 && hw >= 1, "%s: bad shape", name);
    OGC_REQUIRE(w && in && out, "%s: null pointer", name);
    if ((hw & 63) != 0 || K > 4 * FW_KQ_MAX || (((uintptr_t)in | (uintptr_t)out) & ogc_act_mask<AT>()) != 0) {
        ogc_set_error("%s: needs hw %% 64 == 0, K <= %d and 16-byte aligned tensors (hw=%d, K=%d)", name,
                      4 * FW_KQ_MAX, hw, K);
        return OGC_ERR_UNSUPPORTED;
    }
    if (sizeof(AT) == 2 && !g_matmul_bf16) {
        ogc_set_error("%s: 16-bit activations need ogc_set_matmul_precision(1)", name);
        return OGC_ERR_UNSUPPORTED;
    }
    // batch offsets are 64-bit in the kernels; only one sample's activation must fit 32-bit offsets
    OGC_REQUIRE((long long)M * hw < (1ll << 31) && (long long)K * hw < (1ll << 31) && b <= 65535,
                "%s: one sample exceeds 32-bit indexing", name);
    return OGC_OK;
}

} // namespace

// OUT[b, m, p] = sum_k A[m, k] IN[b, k, p];  transpose_a == 0: A = w (M x K);  != 0: A = w^T with w stored (K x M).
namespace {
template <typename AT>
int gemm_plain_impl(const char *name, int b, int M, int K, int hw, int transpose_a, const float *w, const AT *in, AT *out,
                    ogc_stream_t stream) {
    const int rc = gemm_check(name, b, M, K, hw, w, in, out);
    if (rc != OGC_OK) return rc;
    if (b == 0) return OGC_OK;
    if (transpose_a)
        gemm_launch<true, false, false, false, AT>(b, M, K, hw, 1, w, in, out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
    else
        gemm_launch<false, false, false, false, AT>(b, M, K, hw, 1, w, in, out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

extern "C" int ogc_conv1x1_gemm(int b, int M, int K, int hw, int transpose_a, const float *w, const float *in,
                                float *out, ogc_stream_t stream) {
    return gemm_plain_impl<float>("ogc_conv1x1_gemm", b, M, K, hw, transpose_a, w, in, out, stream);
}

// 16-bit activations (`in`, `out` bf16; bf16 operands: needs ogc_set_matmul_precision(1)) — see act_io.h
extern "C" int ogc_conv1x1_gemm_h(int b, int M, int K, int hw, int transpose_a, const float *w, const ogc_bf16_t *in,
                                  ogc_bf16_t *out, ogc_stream_t stream) {
    return gemm_plain_impl<ogc_bf16>("ogc_conv1x1_gemm_h", b, M, K, hw, transpose_a, w, in, out, stream);
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
    const int rc = gemm_check<float>("ogc_conv1x1_gemm_gnstats", b, M, K, hw, w, in, out);
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


// The two kernels above with the previous layer's GroupNorm (+ ReLU) folded into the operand load:
// in' = act(pa[b, k] * in + pb[b, k]).  groups == 0: no statistics of the output.
namespace {
template <typename AT>
int gemm_affine_impl(const char *name, int b, int M, int K, int hw, int relu, int groups, const float *w, const AT *in,
                     const float *pa, const float *pb, AT *out, double *stats, ogc_stream_t stream) {
    const int rc = gemm_check(name, b, M, K, hw, w, in, out);
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
        gemm_launch<false, true, true, false, AT>(b, M, K, hw, groups, w, in, out, stats, pa, pb, relu, s);
    } else {
        gemm_launch<false, false, true, false, AT>(b, M, K, hw, 1, w, in, out, nullptr, pa, pb, relu, s);
    }
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

extern "C" int ogc_conv1x1_gemm_affine(int b, int M, int K, int hw, int relu, int groups, const float *w, const float *in,
                                       const float *pa, const float *pb, float *out, double *stats,
                                       ogc_stream_t stream) {
    return gemm_affine_impl<float>("ogc_conv1x1_gemm_affine", b, M, K, hw, relu, groups, w, in, pa, pb, out, stats, stream);
}

extern "C" int ogc_conv1x1_gemm_affine_h(int b, int M, int K, int hw, int relu, int groups, const float *w, const ogc_bf16_t *in,
                                         const float *pa, const float *pb, ogc_bf16_t *out, double *stats,
                                         ogc_stream_t stream) {
    return gemm_affine_impl<ogc_bf16>("ogc_conv1x1_gemm_affine_h", b, M, K, hw, relu, groups, w, in, pa, pb, out, stats, stream);
}

// ogc_conv1x1_gemm_affine with output statistics, for the LAST layer of a set-abstraction MLP: positions are
// (centre, neighbour) pairs, hw = centres * nsample, and the extreme of the raw output over each neighbourhood comes out
// as well (see PoolOut) — the max-pool over the normalised activation is then ogc_group_norm_pool_extremes.
namespace {
template <typename AT>
int gemm_affine_pool_impl(const char *name, int b, int M, int K, int hw, int relu, int groups, int nsample, const float *w,
                          const AT *in, const float *pa, const float *pb, const float *next_gamma, AT *out, double *stats,
                          float *yext, int *aext, ogc_stream_t stream) {
    const int rc = gemm_check(name, b, M, K, hw, w, in, out);
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
    const int lrc = gemm_launch<false, true, true, true, AT>(b, M, K, hw, groups, w, in, out, stats, pa, pb, relu, s, pool);
    if (lrc != OGC_OK) return lrc;
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

extern "C" int ogc_conv1x1_gemm_affine_pool(int b, int M, int K, int hw, int relu, int groups, int nsample,
                                            const float *w, const float *in, const float *pa, const float *pb,
                                            const float *next_gamma, float *out, double *stats, float *yext, int *aext,
                                            ogc_stream_t stream) {
    return gemm_affine_pool_impl<float>("ogc_conv1x1_gemm_affine_pool", b, M, K, hw, relu, groups, nsample, w, in, pa, pb,
                                        next_gamma, out, stats, yext, aext, stream);
}

// (yext holds the ROUNDED extremes: the values `out` holds at aext)
extern "C" int ogc_conv1x1_gemm_affine_pool_h(int b, int M, int K, int hw, int relu, int groups, int nsample,
                                              const float *w, const ogc_bf16_t *in, const float *pa, const float *pb,
                                              const float *next_gamma, ogc_bf16_t *out, double *stats, float *yext, int *aext,
                                              ogc_stream_t stream) {
    return gemm_affine_pool_impl<ogc_bf16>("ogc_conv1x1_gemm_affine_pool_h", b, M, K, hw, relu, groups, nsample, w, in, pa, pb,
                                           next_gamma, out, stats, yext, aext, stream);
}
