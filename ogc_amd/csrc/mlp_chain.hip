// mlp_chain.hip — a whole per-neighbourhood MLP (1x1 conv -> folded BatchNorm -> ReLU, two or three layers) and the max over
// the neighbourhood in ONE kernel, for inference.
//
// Replaces, in evaluation mode, the tail of the FlowStep3D blocks (reference utils/flowstep3d_util.py:57-66 — the
// correlation layer `FlowEmbedding` — and :126-138, the set-abstraction layers): `for conv, bn: x = relu(bn(conv(x)))`
// followed by `torch.max(x, -1)`.  With BatchNorm in evaluation mode every layer is an affine map per output channel,
// y = a_c (W x)_c + b_c, which the host folds into the weights (W' = diag(a) W) and a bias; as separate launches the
// C3 correlation layer (B = 1, 2048 points x 16 neighbours, 131 -> 128 -> 128 -> 128) is ~10 kernels of a few
// microseconds of work each, 0.21 ms in all for 3.2 GFLOP (20 us at the fp32 MFMA rate).
//
// Layout of the computation (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains):
//   * a wavefront owns 16 consecutive positions (= columns) of the (C, P*S) activation: for S = 16 one point's whole
//     neighbourhood, for S = 8 / 4 two / four of them; S = 32 gives a point to two column blocks (NCB = 2);
//   * D(16 rows x 16 columns) += A(16 x 4) B(4 x 16): A = a 16-row block of W' (lane i + 16 kk holds W'[i][kk]),
//     B = four rows of the input (lane j + 16 kk holds X[kk][j]), D: lane j + 16 ib, register r = row 4 ib + r;
//   * the OUTPUT registers of one layer are the B operands of the next without any data movement: the k-step (rb, r) of
//     the next layer is made of the rows {16 rb + 4 kk + r : kk = 0..3} — exactly what register r of row block rb holds
//     in the lanes ib = kk.  The sum over k does not care about the order of its terms; the A operand just has to be
//     read in the matching order: lane (i, kk) takes W'[.][16 rb + 4 kk + r], one conflict-free ds_read_b32 from the
//     transposed, padded copy of W' in LDS;
//   * bias = initial value of the accumulators, ReLU in place, max over the S lanes of a DPP row at the end.
// A workgroup is four wavefronts sharing the LDS copy of the current layer's weights (<= 70 KB, so two workgroups per CU:
// one loads its next layer while the other computes).
#include "ogc_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int MC_WAVES = 4;
constexpr int MC_THREADS = MC_WAVES * OGC_WAVE;

template <int CTRL>
__device__ __forceinline__ float mc_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// max over the S consecutive lanes (S = 4, 8, 16; aligned groups inside a 16-lane DPP row) that hold one neighbourhood
template <int S>
__device__ __forceinline__ float mc_group_max(float v) {
    v = fmaxf(v, mc_dpp<0xB1>(v));              // quad_perm [1,0,3,2]
    v = fmaxf(v, mc_dpp<0x4E>(v));              // quad_perm [2,3,0,1]
    if (S >= 8) v = fmaxf(v, mc_dpp<0x141>(v)); // row_half_mirror
    if (S >= 16) v = fmaxf(v, mc_dpp<0x140>(v)); // row_mirror
    return v;
}

// width a neighbourhood of S positions takes in a wavefront's columns
constexpr int mc_padded(int S) { return S <= 16 ? S : (S + 15) / 16 * 16; }

// weights of one layer: global Wt (KP rows x COUT, row-major, KP = K rounded up to 4, zero rows beyond K) -> LDS with
// row stride COUT + 4 (4 * stride = 16 mod 32 banks: the two 16-float runs a half-wave reads never share a bank),
// followed by the COUT biases
template <int KP, int COUT>
__device__ __forceinline__ void mc_stage_weights(float *lds, const float *__restrict__ wt, const float *__restrict__ bias) {
    constexpr int LD = COUT + 4;
    // eight independent float4 loads in flight per thread and round (one load per round = one L2 round trip per round)
    constexpr int U = 8;
    for (int e0 = threadIdx.x * 4; e0 < KP * COUT; e0 += MC_THREADS * 4 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * MC_THREADS * 4, KP * COUT - 4);
            v[u] = *reinterpret_cast<const float4 *>(wt + e);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * MC_THREADS * 4;
            if (e < KP * COUT) {
                const int k = e / COUT, m = e - k * COUT;
                *reinterpret_cast<float4 *>(lds + k * LD + m) = v[u];
            }
        }
    }
    for (int m = threadIdx.x; m < COUT; m += MC_THREADS) lds[KP * LD + m] = bias[m];
}

// one layer whose input rows sit in the accumulators of the previous one: out[rb'] += W'[16 rb' .., k] * in[k]
template <int CIN, int COUT, int NCB>
__device__ __forceinline__ void mc_layer_from_regs(const float *lds, const v4f (&in)[CIN / 16][NCB], v4f (&out)[COUT / 16][NCB]) {
    constexpr int LD = COUT + 4;
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const float *bias = lds + CIN * LD;
#pragma unroll
    for (int ob = 0; ob < COUT / 16; ++ob) {
        const float4 b4 = *reinterpret_cast<const float4 *>(bias + ob * 16 + kk * 4); // rows 16 ob + 4 ib + r, ib = kk
#pragma unroll
        for (int c = 0; c < NCB; ++c) out[ob][c] = v4f{b4.x, b4.y, b4.z, b4.w};
    }
#pragma unroll
    for (int rb = 0; rb < CIN / 16; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float *wrow = lds + (rb * 16 + kk * 4 + r) * LD + i; // W'[16 ob + i][16 rb + 4 kk + r] for ob = 0..
#pragma unroll
            for (int ob = 0; ob < COUT / 16; ++ob) {
                const float a = wrow[ob * 16];
#pragma unroll
                for (int c = 0; c < NCB; ++c)
                    out[ob][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, in[rb][c][r], out[ob][c], 0, 0, 0);
            }
        }
#pragma unroll
    for (int ob = 0; ob < COUT / 16; ++ob)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[ob][c][r] = fmaxf(out[ob][c][r], 0.0f);
}

// x (B, C0, hw), hw = P * S; wt1 (C0P, C1), wt2 (C1, C2), wt3 (C2, C3) transposed folded weights; out (B, CL, P), CL = the
// last layer's width (C3, or C2 when C3 == 0)
// CORR: the input rows are not read from a materialised (B, C0, P, S) tensor but formed while loading, as FlowEmbedding
// builds them (utils/flowstep3d_util.py:53-60): rows 0-2 = pos2[:, idx] - pos1, rows 3 .. 3 + CF - 1 = feat2[:, idx],
// the last CF rows = feat1 of the point itself (C0 = 3 + 2 CF); idx (B, P, S) are the clamped neighbour indices.
struct McGather {
    const float *pos1, *pos2, *feat1, *feat2; // (B, 3, P), (B, 3, n2), (B, CF, P), (B, CF, n2)
    const int *idx;                           // (B, P, S)
    int n2;
};

template <int C0, int C1, int C2, int C3, int S, bool CORR>
__global__ __launch_bounds__(MC_THREADS) void mlp_chain_pool_kernel(int hw, McGather ga, const float *__restrict__ x,
                                                                   const float *__restrict__ wt1, const float *__restrict__ b1,
                                                                   const float *__restrict__ wt2, const float *__restrict__ b2,
                                                                   const float *__restrict__ wt3, const float *__restrict__ b3,
                                                                   float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float mc_lds[];
    // A neighbourhood that is not 4, 8 or a multiple of 16 wide (S = 24) is padded to SP columns by repeating its last
    // neighbour: a maximum does not change when one of its arguments is taken twice.
    constexpr int SP = mc_padded(S);
    constexpr int NCB = SP > 16 ? SP / 16 : 1;     // column blocks (of 16 positions) per wavefront
    constexpr int C0P = (C0 + 3) / 4 * 4;
    constexpr int CL = C3 > 0 ? C3 : C2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kk = lane >> 4;
    const int b = blockIdx.y;
    const long long col0 = ((long long)blockIdx.x * MC_WAVES + wave) * (16 * NCB); // first (padded) position of this wavefront
    const int p_total = hw / S;
    const long long hwp = (long long)p_total * SP;
    const float *xb = x + (size_t)b * C0 * hw;
    // padded position -> position in the (P, S) input
    auto source = [&](long long colp) -> long long {
        if constexpr (SP == S) return colp;
        const long long point = colp / SP;
        const int sp = (int)(colp - point * SP);
        return point * S + (sp < S ? sp : S - 1);
    };

    // ---- layer 1: B operands straight from global memory (k-step s = input rows 4 s .. 4 s + 3, lane (j, kk) row 4 s + kk)
    float xin[C0P / 4][NCB];
    if constexpr (CORR) {
        constexpr int CF = (C0 - 3) / 2;
        static_assert(SP == S, "the gathering variant has no padded neighbourhoods");
        const int P = p_total;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const long long col = col0 + c * 16 + j;
            const bool in = col < hw;
            const int point = in ? (int)(col / S) : 0;
            const int nbr = in ? ga.idx[(size_t)b * hw + col] : 0;
#pragma unroll
            for (int s = 0; s < C0P / 4; ++s) {
                const int row = 4 * s + kk;
                float v = 0.0f;
                if (in) {
                    if (row < 3)
                        v = ga.pos2[((size_t)b * 3 + row) * ga.n2 + nbr] - ga.pos1[((size_t)b * 3 + row) * P + point];
                    else if (row < 3 + CF)
                        v = ga.feat2[((size_t)b * CF + (row - 3)) * ga.n2 + nbr];
                    else if (row < C0)
                        v = ga.feat1[((size_t)b * CF + (row - 3 - CF)) * P + point];
                }
                xin[s][c] = v;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const long long colp = col0 + c * 16 + j;
            const bool in = colp < hwp;
            const long long col = in ? source(colp) : 0;
#pragma unroll
            for (int s = 0; s < C0P / 4; ++s) {
                const int row = 4 * s + kk;
                xin[s][c] = (row < C0 && in) ? xb[(size_t)row * hw + col] : 0.0f;
            }
        }
    }
    mc_stage_weights<C0P, C1>(mc_lds, wt1, b1);
    __syncthreads();
    v4f h1[C1 / 16][NCB];
    {
        constexpr int LD = C1 + 4;
        const float *bias = mc_lds + C0P * LD;
#pragma unroll
        for (int ob = 0; ob < C1 / 16; ++ob) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bias + ob * 16 + kk * 4);
#pragma unroll
            for (int c = 0; c < NCB; ++c) h1[ob][c] = v4f{b4.x, b4.y, b4.z, b4.w};
        }
#pragma unroll
        for (int s = 0; s < C0P / 4; ++s) {
            const float *wrow = mc_lds + (4 * s + kk) * LD + j; // W'[16 ob + i][4 s + kk], i = lane & 15
#pragma unroll
            for (int ob = 0; ob < C1 / 16; ++ob) {
                const float a = wrow[ob * 16];
#pragma unroll
                for (int c = 0; c < NCB; ++c) h1[ob][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xin[s][c], h1[ob][c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ob = 0; ob < C1 / 16; ++ob)
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[ob][c][r] = fmaxf(h1[ob][c][r], 0.0f);
    }
    __syncthreads();                       // everybody is done with layer 1's weights
    mc_stage_weights<C1, C2>(mc_lds, wt2, b2);
    __syncthreads();
    v4f h2[C2 / 16][NCB];
    mc_layer_from_regs<C1, C2, NCB>(mc_lds, h1, h2);

    // ---- max over the neighbourhood; lane j == first lane of its group stores (row 16 ob + 4 ib + r, point)
    constexpr int SG = SP > 16 ? 16 : SP;
    float *ob_out = out + (size_t)b * CL * p_total;
    auto pool_and_store = [&](const auto &last) {
#pragma unroll
        for (int ob = 0; ob < CL / 16; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = last[ob][0][r];
#pragma unroll
                for (int c = 1; c < NCB; ++c) v = fmaxf(v, last[ob][c][r]);   // S = 32: the point's second column block
                v = mc_group_max<SG>(v);
                const long long point = (col0 + (NCB > 1 ? 0 : j)) / SP;
                if ((j % SG) == 0 && point < p_total) ob_out[(size_t)(ob * 16 + kk * 4 + r) * p_total + point] = v;
            }
    };
    if constexpr (C3 > 0) {
        __syncthreads();
        mc_stage_weights<C2, C3>(mc_lds, wt3, b3);
        __syncthreads();
        v4f h3[C3 / 16][NCB];
        mc_layer_from_regs<C2, C3, NCB>(mc_lds, h2, h3);
        pool_and_store(h3);
    } else {
        pool_and_store(h2);
    }
}

template <int C0, int C1, int C2, int C3, int S, bool CORR>
int mc_launch(int b, int hw, McGather ga, const float *x, const float *wt1, const float *b1, const float *wt2, const float *b2,
              const float *wt3, const float *b3, float *out, hipStream_t s) {
    constexpr int SP = mc_padded(S);
    constexpr int NCB = SP > 16 ? SP / 16 : 1;
    static_assert(S == 4 || S == 8 || S >= 16, "neighbourhoods of 4, 8 or at least 16 positions");
    static_assert(NCB <= 2, "a point's neighbourhood must fit one wavefront's column blocks");
    constexpr int C0P = (C0 + 3) / 4 * 4;
    constexpr int w1 = C0P * (C1 + 4) + C1, w2 = C1 * (C2 + 4) + C2, w3 = C3 > 0 ? C2 * (C3 + 4) + C3 : 0;
    constexpr int lds_floats = w1 > w2 ? (w1 > w3 ? w1 : w3) : (w2 > w3 ? w2 : w3);
    static_assert(lds_floats * 4 <= 160 * 1024, "weights of one layer must fit the LDS");
    const void *fn = reinterpret_cast<const void *>(&mlp_chain_pool_kernel<C0, C1, C2, C3, S, CORR>);
    static bool raised = false;
    if (!raised && lds_floats * 4 > 64 * 1024) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4) != hipSuccess) {
            ogc_set_error("ogc_mlp_chain_pool: cannot raise the dynamic LDS limit to %d bytes", lds_floats * 4);
            return OGC_ERR_LAUNCH;
        }
        raised = true;
    }
    dim3 grid(ogc_divup((long long)(hw / S) * SP, 16 * NCB * MC_WAVES), b);
    hipLaunchKernelGGL((mlp_chain_pool_kernel<C0, C1, C2, C3, S, CORR>), grid, dim3(MC_THREADS), lds_floats * 4, s, hw, ga, x, wt1, b1,
                       wt2, b2, wt3, b3, out);
    OGC_CHECK_LAUNCH("ogc_mlp_chain_pool");
    return OGC_OK;
}

} // namespace

// The shapes with a kernel: the set-abstraction blocks of the FlowStep3D nets at their C3 sizes (38 blocks per forward pass,
// 7-9 launches each when run layer by layer) — (channels in, three widths, neighbours)
#define OGC_MC_SHAPES(X)                                                                                        \
    X(131, 128, 128, 128, 16) X(131, 128, 128, 128, 32) X(67, 128, 128, 128, 32) X(35, 64, 64, 64, 32)         \
    X(6, 32, 32, 32, 32) X(6, 32, 32, 32, 16) X(6, 32, 32, 64, 16) X(67, 64, 64, 128, 16)                      \
    X(35, 16, 16, 16, 8) X(67, 128, 128, 128, 8) X(131, 128, 128, 128, 24)

extern "C" int ogc_mlp_chain_pool_supported(int c0, int c1, int c2, int c3, int nsample) {
#define OGC_MC_TEST(A, B, C, D, S) if (c0 == A && c1 == B && c2 == C && c3 == D && nsample == S) return 1;
    OGC_MC_SHAPES(OGC_MC_TEST)
#undef OGC_MC_TEST
    return 0;
}

extern "C" int ogc_mlp_chain_pool(int b, int c0, int c1, int c2, int c3, int p, int nsample, const float *x,
                                  const float *wt1, const float *b1, const float *wt2, const float *b2, const float *wt3,
                                  const float *b3, float *out, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && p >= 0 && nsample >= 1, "ogc_mlp_chain_pool: bad shape");
    if (b == 0 || p == 0) return OGC_OK;
    OGC_REQUIRE(x && wt1 && b1 && wt2 && b2 && out && (c3 == 0 || (wt3 && b3)), "ogc_mlp_chain_pool: null pointer");
    OGC_REQUIRE((long long)p * (nsample + 15) < (1ll << 31) && b <= 65535, "ogc_mlp_chain_pool: sample exceeds 32-bit indexing");
    OGC_REQUIRE(((uintptr_t)wt1 & 15) == 0 && ((uintptr_t)wt2 & 15) == 0 && ((uintptr_t)wt3 & 15) == 0,
                "ogc_mlp_chain_pool: weights must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int hw = p * nsample;
#define OGC_MC_RUN(A, B, C, D, S)                                                   \
    if (c0 == A && c1 == B && c2 == C && c3 == D && nsample == S)                   \
        return mc_launch<A, B, C, D, S, false>(b, hw, McGather{}, x, wt1, b1, wt2, b2, wt3, b3, out, s);
    OGC_MC_SHAPES(OGC_MC_RUN)
#undef OGC_MC_RUN
    ogc_set_error("ogc_mlp_chain_pool: no kernel for %d -> %d -> %d -> %d with nsample %d", c0, c1, c2, c3, nsample);
    return OGC_ERR_UNSUPPORTED;
}

extern "C" int ogc_corr_layer_pool_supported(int cf, int c1, int c2, int c3, int nsample) {
    return (cf == 64 && c1 == 128 && c2 == 128 && c3 == 128 && nsample == 16) ? 1 : 0;
}

extern "C" int ogc_corr_layer_pool(int b, int cf, int c1, int c2, int c3, int n1, int n2, int nsample, const float *pos1,
                                   const float *pos2, const float *feat1, const float *feat2, const int *idx,
                                   const float *wt1, const float *b1, const float *wt2, const float *b2, const float *wt3,
                                   const float *b3, float *out, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n1 >= 0 && n2 >= 1 && nsample >= 1, "ogc_corr_layer_pool: bad shape");
    if (b == 0 || n1 == 0) return OGC_OK;
    OGC_REQUIRE(pos1 && pos2 && feat1 && feat2 && idx && wt1 && b1 && wt2 && b2 && wt3 && b3 && out,
                "ogc_corr_layer_pool: null pointer");
    OGC_REQUIRE((long long)n1 * nsample < (1ll << 31) && b <= 65535, "ogc_corr_layer_pool: sample exceeds 32-bit indexing");
    OGC_REQUIRE(((uintptr_t)wt1 & 15) == 0 && ((uintptr_t)wt2 & 15) == 0 && ((uintptr_t)wt3 & 15) == 0,
                "ogc_corr_layer_pool: weights must be 16-byte aligned");
    McGather ga{pos1, pos2, feat1, feat2, idx, n2};
    if (cf == 64 && c1 == 128 && c2 == 128 && c3 == 128 && nsample == 16)
        return mc_launch<131, 128, 128, 128, 16, true>(b, n1 * nsample, ga, nullptr, wt1, b1, wt2, b2, wt3, b3, out,
                                                       (hipStream_t)stream);
    ogc_set_error("ogc_corr_layer_pool: no kernel for %d features, %d -> %d -> %d, nsample %d", cf, c1, c2, c3, nsample);
    return OGC_ERR_UNSUPPORTED;
}
