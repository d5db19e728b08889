// gemm_chunk.hip — the 1x1 convolution product for the shapes the register-tile kernels of conv1x1.hip do not take.
//
//     OUT[b, m, p] = sum_k A[m, k] IN[b, k, p],     A = w (M x K)  or  A = w^T (w stored K x M: the input gradient)
//
// Reference: the Conv1d / Conv2d of SharedMLP (utils/nn_util.py:45-85) in the feature-propagation modules
// (utils/pointnet2_util.py:96-120: 384 -> 128 on 1024 points, 224 -> 64 on 2048, 67 -> 64 on 8192) and the input gradient of
// the widest set-abstraction layer (128 <- 256 channels on 16 x 32768 positions).  conv1x1_gemm_kernel keeps the whole K x 64
// input tile of a wavefront in registers (K <= 160) and gives a wavefront all output rows, so that (a) a reduction over 224,
// 256 or 384 channels does not fit and (b) a layer of 16384 positions is 64 workgroups on 256 compute units; those products
// went to the vendor library.  Here K is walked in CHUNKS of 4 KQ rows:
//   * the IN chunk (4 KQ rows x 64 positions) of a wavefront goes from memory straight into registers — lane (kk = l >> 4,
//     j = l & 15) holds IN[k0 + 4 q + kk][p0 + 4 j .. 4 j + 3], component c being column j of MFMA column block c, exactly as in
//     conv1x1_gemm_kernel — double-buffered: the loads of chunk i + 1 are issued before the MFMAs of chunk i;
//   * the A chunk (rows of the workgroup x 4 KQ) is staged through LDS (row stride 4 * odd: conflict-free ds_read_b32 of the
//     operand), also double-buffered: ONE barrier per chunk;
//   * SPLIT_M = false: the four wavefronts of a workgroup own four position tiles and the same 16 RB rows (IN is read once per
//     row tile; large position counts);  SPLIT_M = true: the four wavefronts own the same position tile and 16 RB rows each
//     (the IN chunk is loaded by all four — L1 / L2 hits — so that a layer of a few thousand positions still fills the chip).
// v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation in k order within a lane's chain.
#include <stdlib.h>

#include "ogc_common.h"
#include "act_io.h"
#include "conv1x1_shared.h"

namespace {

constexpr int GC_WAVES = 4;

// (ogc_pack_bf16_rr of act_io.h: the union / vector-conversion form leaves this kernel's input chunks in scratch memory)
__device__ __forceinline__ v4s gc_pack_bf16(float a, float b, float c, float d) { return ogc_pack_bf16_rr(a, b, c, d); }

// POOLED: IN is not stored — `in` holds y, the raw output of the LAST convolution of a set-abstraction MLP, and the operand is the
// gradient of the pooled GroupNorm that follows it, rebuilt while y is loaded (include/ogc_ops.h, ogc_group_norm_maxpool_bwd_sparse):
//     IN[b, k, p] = fmaf(c2, y, c3) + (p % S == arg ? ag : 0),   (c2, c3) = coef2[b, k],   (ag, arg) = inj[b, k, p / S]
// — the expression of gn_maxpool_bwd_dx_kernel, bit for bit; a lane's four positions lie inside one neighbourhood (S >= 16).
// KA = 2: the weights are staged 2 x 4 KQ rows at a time — ONE barrier per two input chunks.  (The 128-row tile with 16-row chunks
// reaches 111 TFLOP/s on 128 <- 256 channels, with 8-row chunks 86: the barrier is what it waits for; 32-row input chunks spill.)
// AT: element type of in / out (float / ogc_bf16: act_io.h).  BF: operands rounded to bf16 on v_mfma_f32_16x16x16_bf16 — four
// row quads (q .. q + 3) of the input chunk feed one MFMA: k-slot i of lane group kk is chunk row 4 (q + i) + kk for BOTH operands
// (a permutation of the sixteen rows, which the sum over k does not see; cf. conv1x1_gemm_kernel).
// TAB (POOLED): the (c2, c3, ag, arg) of the wavefront's tile — K rows x (64 >> s_shift) neighbourhoods — are fetched ONCE per tile
// into a wave-private LDS table (as dgrad_adjoint_kernel does) and read back one 16-byte entry per row and chunk; without it every
// chunk of sixteen rows issues eight 8-byte loads per lane for them next to its four data loads (K = 256: 128 against 64 + 8 load
// instructions per lane and tile).
template <int RB, int KQ, bool SPLIT_M, bool TRANS, bool POOLED = false, int KA = 1, typename AT = float, bool BF = false,
          bool TAB = false>
__global__ __launch_bounds__(GC_WAVES *OGC_WAVE, 2) void gemm_chunk_kernel(int M, int K, int hw, const float *__restrict__ w,
                                                                         const AT *__restrict__ in,
                                                                         AT *__restrict__ out,
                                                                         const float2 *__restrict__ coef2 = nullptr,
                                                                         const float2 *__restrict__ inj = nullptr,
                                                                         int s_shift = 0) {
    constexpr int KC = 4 * KQ, KCA = KC * KA, LD = 4 * ((KQ * KA) | 1);
    constexpr int MT = SPLIT_M ? 64 * RB : 16 * RB;         // rows of A staged per workgroup
    constexpr int PER = (MT * KCA) / (GC_WAVES * OGC_WAVE); // staged elements per thread and A chunk
    static_assert((MT * KCA) % (GC_WAVES * OGC_WAVE) == 0 && PER >= 1 && (KA == 1 || KA == 2), "staging split");
    extern __shared__ __attribute__((aligned(16))) float gc_lds[]; // [2][MT][LD]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int j = lane & 15, kk = lane >> 4;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * MT;
    const int wrow = SPLIT_M ? wave * 16 * RB : 0;                    // this wavefront's first row inside the staged tile
    const int p0 = SPLIT_M ? blockIdx.x * 64 : (blockIdx.x * GC_WAVES + wave) * 64;
    const bool active = p0 < hw;                                      // (wave-uniform; !SPLIT_M: the last workgroup may be ragged)
    const int pl = active ? p0 : 0;
    const int nblk = min(RB, (M - m0 - wrow + 15) >> 4);              // row blocks with at least one real row (may be <= 0)
    const AT *inb = in + (size_t)b * K * hw + pl + 4 * j;
    const int nchunks = (K + KC - 1) / KC;

    v4f acc[RB][4];
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};

    // ---- loads of a chunk: A elements (zeros beyond the matrix), IN rows (clamped onto the last row, zeroed in `rebuild`)
    auto load_a = [&](int k0, float(&av)[PER]) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = t + i * (GC_WAVES * OGC_WAVE);
            int mi, ki;
            if (TRANS) { mi = e % MT; ki = e / MT; } else { mi = e / KCA; ki = e % KCA; } // consecutive lanes: consecutive addresses
            const int m = m0 + mi, k = k0 + ki;
            const bool ok = m < M && k < K;
            const int mc = min(m, M - 1), kc = min(k, K - 1);
            const float v = TRANS ? w[(size_t)kc * M + mc] : w[(size_t)mc * K + kc];
            av[i] = ok ? v : 0.f;
        }
    };
    auto store_a = [&](float *dst, const float(&av)[PER]) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = t + i * (GC_WAVES * OGC_WAVE);
            int mi, ki;
            if (TRANS) { mi = e % MT; ki = e / MT; } else { mi = e / KCA; ki = e % KCA; }
            dst[mi * LD + ki] = av[i];
        }
    };
    constexpr int NP = (POOLED && !TAB) ? KQ : 1;
    const int centres = hw >> s_shift;
    const int my_centre = (pl + 4 * j) >> s_shift, jpos = (pl + 4 * j) & ((1 << s_shift) - 1);
    const int cpw = 64 >> s_shift; // neighbourhoods of a 64-position tile
    float4 *tab = reinterpret_cast<float4 *>(gc_lds + 2 * MT * LD) + (size_t)wave * K * cpw;
    if constexpr (POOLED && TAB) {
        for (int e = lane; e < K * cpw; e += OGC_WAVE) {
            const int row = e / cpw, centre = (pl >> s_shift) + (e - row * cpw);
            const float2 c2v = coef2[(size_t)b * K + row];
            const float2 jj = centre < centres ? inj[((size_t)b * K + row) * centres + centre] : make_float2(0.f, __int_as_float(-1));
            tab[e] = make_float4(c2v.x, c2v.y, jj.x, jj.y);
        }
        __builtin_amdgcn_wave_barrier();
    }
    const int mine_c = (4 * j) >> s_shift;
    auto load_in = [&](int k0, float4(&xv)[KQ], float2(&cc)[NP], float2(&jv)[NP]) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const int k = min(k0 + 4 * q + kk, K - 1);
            xv[q] = ogc_ld4(inb + (size_t)k * hw);
            if constexpr (POOLED && !TAB) {
                cc[q] = coef2[(size_t)b * K + k];
                jv[q] = inj[((size_t)b * K + k) * centres + my_centre];
            }
        }
    };
    auto rebuild = [&](int k0, float4(&xv)[KQ], const float2(&cc)[NP], const float2(&jv)[NP]) {
        if constexpr (POOLED && TAB) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const float4 te = tab[min(k0 + 4 * q + kk, K - 1) * cpw + mine_c];
                const int rel = __float_as_int(te.w) - jpos;
                xv[q].x = fmaf(te.x, xv[q].x, te.y) + (rel == 0 ? te.z : 0.f);
                xv[q].y = fmaf(te.x, xv[q].y, te.y) + (rel == 1 ? te.z : 0.f);
                xv[q].z = fmaf(te.x, xv[q].z, te.y) + (rel == 2 ? te.z : 0.f);
                xv[q].w = fmaf(te.x, xv[q].w, te.y) + (rel == 3 ? te.z : 0.f);
            }
        } else if constexpr (POOLED) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const int rel = __float_as_int(jv[q].y) - jpos;
                const float ag = jv[q].x;
                xv[q].x = fmaf(cc[q].x, xv[q].x, cc[q].y) + (rel == 0 ? ag : 0.f);
                xv[q].y = fmaf(cc[q].x, xv[q].y, cc[q].y) + (rel == 1 ? ag : 0.f);
                xv[q].z = fmaf(cc[q].x, xv[q].z, cc[q].y) + (rel == 2 ? ag : 0.f);
                xv[q].w = fmaf(cc[q].x, xv[q].w, cc[q].y) + (rel == 3 ? ag : 0.f);
            }
        }
        // the ragged last chunk (K not a multiple of the chunk; wave-uniform test): rows beyond K were read from row K - 1 —
        // they become zeros here, not "anything times a zero weight": an Inf / NaN in the last real channel must not turn every
        // output row of its position into NaN (0 * Inf) where the library's product gives what the real terms give
        if (k0 + KC > K) {
#pragma unroll
            for (int q = 0; q < KQ; ++q)
                if (k0 + 4 * q + kk >= K) xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // Every row block is computed: rows beyond M are zeros in LDS, and the entry point picks RB so that few are (16 RB >= the
    // rows of a tile, 67 rows -> RB = 5) — a wave-uniform `a < nblk` branch per row block between the MFMAs costs the matrix pipe
    // issue slots, and the loop written twice (with and without it) spills.
    auto compute = [&](const float *a_lds, const float4(&xv)[KQ], int koff = 0) {
        if constexpr (BF) {
            static_assert(!BF || KQ % 4 == 0, "bf16 operands: whole groups of four row quads per input chunk");
#pragma unroll
            for (int g = 0; g < KQ / 4; ++g) {
                const float4 v0 = xv[4 * g], v1 = xv[4 * g + 1], v2 = xv[4 * g + 2], v3 = xv[4 * g + 3];
                const v4s bx = gc_pack_bf16(v0.x, v1.x, v2.x, v3.x);
                const v4s by = gc_pack_bf16(v0.y, v1.y, v2.y, v3.y);
                const v4s bz = gc_pack_bf16(v0.z, v1.z, v2.z, v3.z);
                const v4s bw = gc_pack_bf16(v0.w, v1.w, v2.w, v3.w);
                const float *ar = a_lds + (wrow + j) * LD + koff + 16 * g + kk;
#pragma unroll
                for (int a = 0; a < RB; ++a) {
                    const v4s av = gc_pack_bf16(ar[a * 16 * LD], ar[a * 16 * LD + 4], ar[a * 16 * LD + 8], ar[a * 16 * LD + 12]);
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bx, acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, by, acc[a][1], 0, 0, 0);
                    acc[a][2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bz, acc[a][2], 0, 0, 0);
                    acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bw, acc[a][3], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const float bx = xv[q].x, by = xv[q].y, bz = xv[q].z, bw = xv[q].w;
#pragma unroll
                for (int a = 0; a < RB; ++a) {
                    const float av = a_lds[(wrow + a * 16 + j) * LD + koff + q * 4 + kk];
                    acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bx, acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, by, acc[a][1], 0, 0, 0);
                    acc[a][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bz, acc[a][2], 0, 0, 0);
                    acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw, acc[a][3], 0, 0, 0);
                }
            }
        }
    };

    float *buf0 = gc_lds, *buf1 = gc_lds + MT * LD;
    float4 x0[KQ], x1[KQ];
    float2 c0[NP], c1[NP], j0[NP], j1[NP];
    float a_next[PER];
    load_a(0, a_next);
    load_in(0, x0, c0, j0);
    store_a(buf0, a_next);
    __syncthreads();
    const bool work = nblk > 0; // (SPLIT_M: a wavefront whose rows all lie beyond M only helps staging)
    if constexpr (KA == 2) {
        // one A chunk (2 KC rows) per iteration: both input register sets are consumed against it, one barrier; input chunks
        // beyond K are zero rows against zero weights (no branch around them)
        const int npairs = (nchunks + 1) / 2;
        for (int p = 0; p < npairs; ++p) {
            const bool more = p + 1 < npairs;
            if (more) load_a((p + 1) * KCA, a_next);
            load_in((2 * p + 1) * KC, x1, c1, j1);
            rebuild(2 * p * KC, x0, c0, j0);
            if (work) compute(buf0, x0, 0);
            if (more) load_in((2 * p + 2) * KC, x0, c0, j0);
            rebuild((2 * p + 1) * KC, x1, c1, j1);
            if (work) compute(buf0, x1, KC);
            if (more) store_a(buf1, a_next);
            __syncthreads();
            float *tmp = buf0; buf0 = buf1; buf1 = tmp;
        }
    } else
    // chunks in pairs so that the two register sets and the two LDS buffers are named, not indexed
    for (int i = 0; i < nchunks; i += 2) {
        const bool more1 = i + 1 < nchunks, more2 = i + 2 < nchunks;
        if (more1) {
            load_a((i + 1) * KC, a_next);
            load_in((i + 1) * KC, x1, c1, j1);
        }
        rebuild(i * KC, x0, c0, j0);
        if (work) compute(buf0, x0);
        if (more1) store_a(buf1, a_next);
        __syncthreads();
        if (!more1) break;
        if (more2) {
            load_a((i + 2) * KC, a_next);
            load_in((i + 2) * KC, x0, c0, j0);
        }
        rebuild((i + 1) * KC, x1, c1, j1);
        if (work) compute(buf1, x1);
        if (more2) store_a(buf0, a_next);
        __syncthreads();
    }
    if (!active) return;
    // acc[a][c][r]: row a * 16 + kk * 4 + r, position p0 + 4 j + c  ->  one float4 per row
    AT *ob = out + (size_t)b * M * hw + p0 + 4 * j;
#pragma unroll
    for (int a = 0; a < RB; ++a) {
        if (a < nblk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wrow + a * 16 + kk * 4 + r;
                if (m < M) ogc_st4(ob + (size_t)m * hw, make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]));
            }
        }
    }
}

template <int RB, int KQ, bool SPLIT_M, int KA = 1>
void gemm_chunk_launch(int b, int M, int K, int hw, int transpose_a, const float *w, const float *in, float *out,
                       hipStream_t s) {
    constexpr int MT = SPLIT_M ? 64 * RB : 16 * RB;
    const size_t lds = (size_t)2 * MT * 4 * ((KQ * KA) | 1) * sizeof(float);
    dim3 grid(SPLIT_M ? hw / 64 : ogc_divup(hw, 64 * GC_WAVES), ogc_divup(M, MT), b);
    // operands: fp32, or rounded to bf16 under ogc_set_matmul_precision(1) (since the end of round 5: until then this entry point
    // kept fp32 operands whatever the switch said, and a bf16 configuration mixed the two)
#define GC_GO(TR, BFV)                                                                                                           \
    hipLaunchKernelGGL((gemm_chunk_kernel<RB, KQ, SPLIT_M, TR, false, KA, float, BFV>), grid, dim3(GC_WAVES * OGC_WAVE), lds, s, M, K, \
                       hw, w, in, out, nullptr, nullptr, 0)
    if (ogc_g_matmul_bf16) {
        if (transpose_a) GC_GO(true, true);
        else GC_GO(false, true);
    } else {
        if (transpose_a) GC_GO(true, false);
        else GC_GO(false, false);
    }
#undef GC_GO
}

} // namespace

// OUT[b, m, p] = sum_k A[m, k] IN[b, k, p] for ANY reduction length and row count (hw % 64 == 0): transpose_a == 0: A = w
// (M x K), != 0: A = w^T with w stored (K x M).  Operands follow ogc_set_matmul_precision (fp32, or rounded to bf16).
extern "C" int ogc_conv1x1_gemm_any(int b, int M, int K, int hw, int transpose_a, const float *w, const float *in, float *out,
                                    ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && M >= 1 && K >= 1 && hw >= 1, "ogc_conv1x1_gemm_any: bad shape");
    OGC_REQUIRE(w && in && out, "ogc_conv1x1_gemm_any: null pointer");
    if ((hw & 63) != 0 || (((uintptr_t)in | (uintptr_t)out) & 15) != 0) {
        ogc_set_error("ogc_conv1x1_gemm_any: needs hw %% 64 == 0 and 16-byte aligned tensors (hw=%d)", hw);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE((long long)M * hw < (1ll << 31) && (long long)K * hw < (1ll << 31) && b <= 65535,
                "ogc_conv1x1_gemm_any: one sample exceeds 32-bit indexing");
    if (b == 0) return OGC_OK;
    hipStream_t s = (hipStream_t)stream;
    const long long tiles = (long long)b * (hw / 64); // 64-position tiles
#define GC_CASE(RBV, KQV, SPLIT)                                                              \
    case RBV: gemm_chunk_launch<RBV, KQV, SPLIT>(b, M, K, hw, transpose_a, w, in, out, s); break
    if (tiles >= 4096) {
        // enough position tiles for one wavefront each with all rows of a (<= 128-row) tile: IN is read once per row tile.
        // Row blocks per wavefront: the smallest that covers M in ceil(M / 128) tiles (224 rows: two tiles of 7 blocks)
        const int ntile = ogc_divup(M, 128), rb = ogc_divup(M, 16 * ntile);
        switch (rb) {
            GC_CASE(1, 8, false); GC_CASE(2, 8, false); GC_CASE(3, 8, false); GC_CASE(4, 8, false);
            GC_CASE(5, 8, false); GC_CASE(6, 8, false); GC_CASE(7, 4, false);
            default: {
                static const char *kq = getenv("OGC_GEMM_CHUNK_KQ"); // development: 4 = one barrier per 16-row chunk
                if (kq && kq[0] == '4') gemm_chunk_launch<8, 4, false>(b, M, K, hw, transpose_a, w, in, out, s);
                else gemm_chunk_launch<8, 4, false, 2>(b, M, K, hw, transpose_a, w, in, out, s);
            }
        }
    } else {
        // few positions: the wavefronts of a workgroup split the rows, 32 rows per wavefront where that still gives ~2 workgroups
        // per compute unit and M fills them, 16 otherwise
        const bool same_rows = 128 * ogc_divup(M, 128) == 64 * ogc_divup(M, 64); // (96 rows: 2 x 64 either way; 192: 256 vs 192)
        const int rb = (M > 64 && same_rows && tiles * ogc_divup(M, 128) >= 512) ? 2 : 1;
        switch (rb) {
            GC_CASE(2, 8, true);
            default: gemm_chunk_launch<1, 8, true>(b, M, K, hw, transpose_a, w, in, out, s);
        }
    }
#undef GC_CASE
    OGC_CHECK_LAUNCH("ogc_conv1x1_gemm_any");
    return OGC_OK;
}

// The input gradient of the LAST convolution of a set-abstraction MLP with the gradient of its pooled GroupNorm in the sparse form
// of ogc_group_norm_maxpool_bwd_sparse:  grad_z[b] = w^T . g_y[b],  g_y rebuilt from (y, coef2, inj) while y is loaded (see
// POOLED above) — the dense g_y (the size of the layer's activation) is neither written nor read.  w (cout, cin), y (b, cout, hw),
// grad_z (b, cin, hw); any cin (row tiles of 64 up to 64 channels, of 128 beyond).  hw % 64 == 0,
// nsample in {16, 32, 64} dividing hw, at least 1024 tiles of 64 positions (fewer: OGC_ERR_UNSUPPORTED, the caller keeps the dense path).
namespace {
template <typename AT>
int dgrad_pooled_impl(int b, int cin, int cout, int hw, int nsample, const float *w, const AT *y, const float *coef2,
                      const float *inj, AT *grad_z, ogc_stream_t stream) {
    constexpr bool BF = sizeof(AT) == 2; // 16-bit tensors come with bf16 operands
    OGC_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && hw >= 1, "ogc_conv1x1_dgrad_pooled: bad shape");
    OGC_REQUIRE(w && y && coef2 && inj && grad_z, "ogc_conv1x1_dgrad_pooled: null pointer");
    const int sh = nsample == 16 ? 4 : nsample == 32 ? 5 : nsample == 64 ? 6 : -1;
    if (sh < 0 || hw % nsample != 0 || (hw & 63) != 0 || (long long)b * (hw / 64) < 1024 ||
        (((uintptr_t)y | (uintptr_t)grad_z) & ogc_act_mask<AT>()) != 0 || (((uintptr_t)coef2 | (uintptr_t)inj) & 7) != 0 ||
        (BF && !ogc_g_matmul_bf16)) {
        ogc_set_error("ogc_conv1x1_dgrad_pooled: needs nsample in {16, 32, 64} dividing hw, hw %% 64 == 0, >= 1024 tiles of 64 "
                      "positions and aligned tensors (hw=%d, nsample=%d, b=%d)", hw, nsample, b);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE((long long)cin * hw < (1ll << 31) && (long long)cout * hw < (1ll << 31) && b <= 65535,
                "ogc_conv1x1_dgrad_pooled: one sample exceeds 32-bit indexing");
    if (b == 0) return OGC_OK;
    hipStream_t s = (hipStream_t)stream;
    const float2 *c2 = reinterpret_cast<const float2 *>(coef2), *ij = reinterpret_cast<const float2 *>(inj);
    const int M = cin, K = cout;
    // the per-tile table of the sparse gradient's entries (TAB) where it fits next to the weights (K rows x 64 / nsample
    // neighbourhoods x 16 bytes per wavefront; OGC_DGRAD_POOLED_TAB=0: per-chunk loads, for A/B runs)
    static const bool tab_off = [] { const char *e = getenv("OGC_DGRAD_POOLED_TAB"); return e && e[0] == '0'; }();
    const size_t tab_bytes = (size_t)GC_WAVES * K * (64 >> sh) * sizeof(float4);
    const bool tab = !tab_off && tab_bytes <= 24 * 1024;
    if (M <= 64) {
        constexpr int RB = 4, KQ = 4, KA = 2; // (weights staged 32 rows at a time: registers to spare at 64 rows)
        dim3 grid(ogc_divup(hw, 64 * GC_WAVES), ogc_divup(M, 16 * RB), b);
        const size_t lds = (size_t)2 * 16 * RB * 4 * ((KQ * KA) | 1) * sizeof(float);
        if (tab)
            hipLaunchKernelGGL((gemm_chunk_kernel<RB, KQ, false, true, true, KA, AT, BF, true>), grid, dim3(GC_WAVES * OGC_WAVE),
                               lds + tab_bytes, s, M, K, hw, w, y, grad_z, c2, ij, sh);
        else
            hipLaunchKernelGGL((gemm_chunk_kernel<RB, KQ, false, true, true, KA, AT, BF>), grid, dim3(GC_WAVES * OGC_WAVE), lds, s, M,
                               K, hw, w, y, grad_z, c2, ij, sh);
    } else {
        constexpr int RB = 8, KQ = 4;
        dim3 grid(ogc_divup(hw, 64 * GC_WAVES), ogc_divup(M, 16 * RB), b);
        const size_t lds = (size_t)2 * 16 * RB * 4 * (KQ | 1) * sizeof(float);
        if (tab)
            hipLaunchKernelGGL((gemm_chunk_kernel<RB, KQ, false, true, true, 1, AT, BF, true>), grid, dim3(GC_WAVES * OGC_WAVE),
                               lds + tab_bytes, s, M, K, hw, w, y, grad_z, c2, ij, sh);
        else
            hipLaunchKernelGGL((gemm_chunk_kernel<RB, KQ, false, true, true, 1, AT, BF>), grid, dim3(GC_WAVES * OGC_WAVE), lds, s, M, K,
                               hw, w, y, grad_z, c2, ij, sh);
    }
    OGC_CHECK_LAUNCH("ogc_conv1x1_dgrad_pooled");
    return OGC_OK;
}
} // namespace

extern "C" int ogc_conv1x1_dgrad_pooled(int b, int cin, int cout, int hw, int nsample, const float *w, const float *y,
                                        const float *coef2, const float *inj, float *grad_z, ogc_stream_t stream) {
    return dgrad_pooled_impl<float>(b, cin, cout, hw, nsample, w, y, coef2, inj, grad_z, stream);
}

// 16-bit y / grad_z, bf16 operands (needs ogc_set_matmul_precision(1))
extern "C" int ogc_conv1x1_dgrad_pooled_h(int b, int cin, int cout, int hw, int nsample, const float *w, const ogc_bf16_t *y,
                                          const float *coef2, const float *inj, ogc_bf16_t *grad_z, ogc_stream_t stream) {
    return dgrad_pooled_impl<ogc_bf16>(b, cin, cout, hw, nsample, w, y, coef2, inj, grad_z, stream);
}

