// soft_nn.hip — the soft nearest-neighbour step of object-aware ICP, fused.
//
// Reference (oa_icp.py:62-72), per refinement iteration:
//     corr12 = (-torch.cdist(pc1 + flow, pc2) / temperature).softmax(-1)        # (B, N1, N2)
//     corr12 = corr12 * consistency12                                            # consistency12 = mask1 @ mask2^T
//     corr12 = corr12 / corr12.sum(-1, keepdim=True).clamp(1e-10)
//     target = corr12 @ pc2                                                      # flow = target - pc1
// At the C4 refinement shape (B = 4, N = 8192) that is two 1 GB (B, N, N) tensors written and re-read several times
// per iteration, twenty iterations per round.  Nothing of size N x N needs to exist: per query point m
//     target_m = (sum_n e_mn c_mn q_n / Z_m) / max(sum_n e_mn c_mn / Z_m, 1e-10),
//     e_mn = exp(x_mn - max_n x_mn),  x_mn = -d_mn / temperature,  Z_m = sum_n e_mn,  c_mn = <mask1_m, mask2_n>,
// which one pass over the candidates accumulates with the usual running-maximum rescaling.
// Distances follow torch.cdist's large-matrix path (|a|^2 + |b|^2 - 2 a.b accumulated as one 5-term dot product,
// clamp, sqrt) rather than coordinate differences, so the values agree with the reference's to fp32 rounding even
// where that formulation is ill-conditioned (metre-scale coordinates, centimetre distances).
// One lane per query; the eight wavefronts of a workgroup split the candidates of the same 64 queries and merge
// their partial (max, Z, A, V) through LDS; candidates are staged through LDS as 16-byte-aligned records
// (x, y, z, |q|^2, mask[0..K)) read back as broadcasts.
#include <stdlib.h>

#include "ogc_common.h"

namespace {

constexpr int SN_WAVES = 8;   // wavefronts per workgroup = splits of the candidate range
constexpr int SN_CHUNK = 8;   // candidates per rescaling step

template <int KMAX>
__global__ __launch_bounds__(SN_WAVES *OGC_WAVE) void soft_nn_kernel(int n1, int n2, int k, float tau,
                                                                     const float *__restrict__ p1,
                                                                     const float *__restrict__ p2,
                                                                     const float *__restrict__ m1,
                                                                     const float *__restrict__ m2,
                                                                     float *__restrict__ out) {
    constexpr int REC = 4 + KMAX;               // floats per staged candidate (multiple of 4)
    constexpr int SN_TILE = KMAX > 16 ? 32 : 64; // candidates staged per wavefront at a time (LDS budget)
    __shared__ __attribute__((aligned(16))) float tile[SN_WAVES][SN_TILE * REC];
    __shared__ float part[SN_WAVES][6][OGC_WAVE]; // per wave: max, Z, A, Vx, Vy, Vz of every query lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int q = blockIdx.x * OGC_WAVE + lane;
    const float *p1b = p1 + (size_t)b * n1 * 3, *p2b = p2 + (size_t)b * n2 * 3;
    const float *m1b = m1 + (size_t)b * n1 * k, *m2b = m2 + (size_t)b * n2 * k;

    // the query: -2a, |a|^2, mask row
    float ax = 0.f, ay = 0.f, az = 0.f, an = 0.f, mq[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) mq[c] = 0.f;
    if (q < n1) {
        const float x = p1b[q * 3], y = p1b[q * 3 + 1], z = p1b[q * 3 + 2];
        an = (x * x + y * y) + z * z;
        ax = -2.f * x; ay = -2.f * y; az = -2.f * z;
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if (c < k) mq[c] = m1b[(size_t)q * k + c];
    }
    float mx = -INFINITY, Z = 0.f, A = 0.f, Vx = 0.f, Vy = 0.f, Vz = 0.f;

    // this wavefront's share of the candidates
    const int per = (n2 + SN_WAVES - 1) / SN_WAVES;
    const int j_begin = wave * per, j_end = min(n2, j_begin + per);
    float *t = tile[wave];
    for (int j0 = j_begin; j0 < j_end; j0 += SN_TILE) {
        // stage up to SN_TILE candidates: lane l writes record l
        if (lane < SN_TILE) {
            const int j = j0 + lane;
            float x = 0.f, y = 0.f, z = 0.f, nn = INFINITY; // padding: infinitely far
            if (j < j_end) {
                x = p2b[j * 3]; y = p2b[j * 3 + 1]; z = p2b[j * 3 + 2];
                nn = (x * x + y * y) + z * z;
            }
            float *r = t + lane * REC;
            r[0] = x; r[1] = y; r[2] = z; r[3] = nn;
#pragma unroll
            for (int c = 0; c < KMAX; ++c) r[4 + c] = (j < j_end && c < k) ? m2b[(size_t)j * k + c] : 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int cnt = min(SN_TILE, j_end - j0);
        for (int c0 = 0; c0 < cnt; c0 += SN_CHUNK) {
            float x[SN_CHUNK], cw[SN_CHUNK], qx[SN_CHUNK], qy[SN_CHUNK], qz[SN_CHUNK];
            float cmax = -INFINITY;
#pragma unroll
            for (int u = 0; u < SN_CHUNK; ++u) {
                const float4 *rec = reinterpret_cast<const float4 *>(t + (c0 + u) * REC);
                const float4 g = rec[0]; // broadcast read
                // torch.cdist (mm path): one dot product over (-2a, |a|^2, 1) . (b, 1, |b|^2), then clamp + sqrt
                float d2 = ax * g.x;
                d2 = fmaf(ay, g.y, d2);
                d2 = fmaf(az, g.z, d2);
                d2 = d2 + an;
                d2 = d2 + g.w;
                const float d = sqrtf(fmaxf(d2, 1e-30f));
                x[u] = -d / tau;
                qx[u] = g.x; qy[u] = g.y; qz[u] = g.z;
                float cc = 0.f;
#pragma unroll
                for (int v = 0; v < KMAX / 4; ++v) {
                    const float4 mm = rec[1 + v];
                    cc = fmaf(mq[4 * v], mm.x, cc);
                    cc = fmaf(mq[4 * v + 1], mm.y, cc);
                    cc = fmaf(mq[4 * v + 2], mm.z, cc);
                    cc = fmaf(mq[4 * v + 3], mm.w, cc);
                }
                cw[u] = cc;
                cmax = fmaxf(cmax, x[u]);
            }
            if (cmax > mx) { // rescale the running sums to the new maximum (first chunk: mx = -inf -> factor 0)
                const float s = expf(mx - cmax);
                Z *= s; A *= s; Vx *= s; Vy *= s; Vz *= s;
                mx = cmax;
            }
#pragma unroll
            for (int u = 0; u < SN_CHUNK; ++u) {
                const float e = expf(x[u] - mx); // padding: x = -inf -> 0
                const float w = e * cw[u];
                Z += e;
                A += w;
                Vx = fmaf(w, qx[u], Vx); Vy = fmaf(w, qy[u], Vy); Vz = fmaf(w, qz[u], Vz);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    part[wave][0][lane] = mx; part[wave][1][lane] = Z; part[wave][2][lane] = A;
    part[wave][3][lane] = Vx; part[wave][4][lane] = Vy; part[wave][5][lane] = Vz;
    __syncthreads();
    if (wave == 0 && q < n1) {
        float gm = -INFINITY;
#pragma unroll
        for (int w = 0; w < SN_WAVES; ++w) gm = fmaxf(gm, part[w][0][lane]);
        float z = 0.f, a = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
        for (int w = 0; w < SN_WAVES; ++w) {
            const float pm = part[w][0][lane];
            const float s = pm > -INFINITY ? expf(pm - gm) : 0.f;
            z = fmaf(part[w][1][lane], s, z);
            a = fmaf(part[w][2][lane], s, a);
            vx = fmaf(part[w][3][lane], s, vx);
            vy = fmaf(part[w][4][lane], s, vy);
            vz = fmaf(part[w][5][lane], s, vz);
        }
        // corr = softmax * consistency; corr /= corr.sum().clamp(1e-10); target = corr @ pc2
        const float inv_z = 1.0f / z;
        const float denom = fmaxf(a * inv_z, 1e-10f);
        float *o = out + ((size_t)b * n1 + q) * 3;
        o[0] = (vx * inv_z) / denom;
        o[1] = (vy * inv_z) / denom;
        o[2] = (vz * inv_z) / denom;
    }
}


// ---- the same step on the matrix cores (round 6) ---------------------------------------------------------------------------------
// The lane-per-query kernel above spends ~50 vector instructions per (query, candidate) pair — a 5-term distance dot product, a
// K-term consistency dot product fed by five broadcast LDS reads, a division, two library exponentials — and runs at a tenth of
// the vector issue peak: 1.0 ms per iteration at the refinement shape (B = 4, N = 8192, K = 10; oa_icp.py:175: 20 iterations).
// Both dot products are GEMM-shaped: rows = queries, columns = candidates, reduction length 5 and K.  Here a wavefront owns
// SM_QT tiles of 16 queries and walks its share of the candidates in tiles of 16:
//     D  = v_mfma_f32_16x16x4_f32( (-2a, |a|^2 | 1, 0, 0, 0),  (b, 1 | |b|^2, 0, 0, 0) )        two MFMAs: torch.cdist's mm path
//     C  = v_mfma_f32_16x16x4_f32( mask1 rows, mask2 rows )                                    ceil(K / 4) MFMAs
// so the vector unit only does max / sqrt / scale / exp2 / five multiply-adds per pair (~17 instructions).  In the MFMA's result
// layout a lane holds four queries (rows 4 (l >> 4) + r) of ONE candidate column (l & 15): it keeps the running (max, Z, A, V) of
// its four queries over the candidates of its column, rescaled once per SM_U tiles; the sixteen lanes of a row group are merged
// by a butterfly at the end, the SM_SPLITS wavefronts of a workgroup (same queries, different candidates) through LDS.
// The candidates are constant over a call: ogc_soft_nn_target packs them once into the operand layout (64 consecutive floats per
// MFMA operand: one coalesced 4-byte load per lane, no LDS staging) in the stream's workspace.  Exponentials in base 2 on
// v_exp_f32 (x log2(e) folded into the -1 / tau scale), v_sqrt_f32 for the root: 1 ulp each, against the sums' 1e-5 bar.
typedef float sn_v4f __attribute__((ext_vector_type(4)));
constexpr int SM_SPLITS = 8;   // wavefronts per workgroup = splits of the candidate range
constexpr int SM_QT = 2;       // query tiles of 16 per wavefront
constexpr int SM_U = 2;        // candidate tiles per rescaling step
constexpr float SM_FAR = 1e30f; // |b|^2 of a padding candidate: finite (no inf - inf in the running maximum), weight exp2(-huge) = 0

// table (b, tiles_pad, 2 + NM, 4, 16): operand blocks of a candidate tile; coords (b, tiles_pad * 16) float4 = (x, y, z, 0)
template <int NM>
__global__ __launch_bounds__(256) void soft_nn_pack_kernel(int n2, int k, int tiles_pad, const float *__restrict__ p2,
                                                           const float *__restrict__ m2, float *__restrict__ table,
                                                           float4 *__restrict__ coords) {
    constexpr int NB = 2 + NM;
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= tiles_pad * 16) return;
    const bool real = j < n2;
    float x = 0.f, y = 0.f, z = 0.f, nn = SM_FAR;
    if (real) {
        const float *p = p2 + ((size_t)b * n2 + j) * 3;
        x = p[0]; y = p[1]; z = p[2];
        nn = (x * x + y * y) + z * z;
    }
    float *t = table + (((size_t)b * tiles_pad + (j >> 4)) * NB) * 64 + (j & 15);
    t[0] = x; t[16] = y; t[32] = z; t[48] = 1.0f;
    t[64] = nn; t[64 + 16] = 0.f; t[64 + 32] = 0.f; t[64 + 48] = 0.f;
    const float *m = m2 + ((size_t)b * n2 + (real ? j : 0)) * k;
#pragma unroll
    for (int v = 0; v < NM; ++v)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) t[(2 + v) * 64 + kk * 16] = (real && 4 * v + kk < k) ? m[4 * v + kk] : 0.f;
    coords[(size_t)b * tiles_pad * 16 + j] = make_float4(x, y, z, 0.f);
}

template <int NM>
__global__ __launch_bounds__(SM_SPLITS *OGC_WAVE) void soft_nn_mfma_kernel(int n1, int k, int tiles_pad, float scale2,
                                                                           const float *__restrict__ p1, const float *__restrict__ m1,
                                                                           const float *__restrict__ table,
                                                                           const float4 *__restrict__ coords, float *__restrict__ out) {
    constexpr int NB = 2 + NM;
    __shared__ float part[SM_SPLITS][SM_QT * 16][6];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), b = blockIdx.y;
    const int col = lane & 15, kk = lane >> 4;
    const int qbase = blockIdx.x * (SM_QT * 16);
    // A operands: lane (col, kk) supplies element kk of query row `col`
    float a_d1[SM_QT], a_d2[SM_QT], a_m[SM_QT][NM];
#pragma unroll
    for (int qt = 0; qt < SM_QT; ++qt) {
        const int q = qbase + qt * 16 + col;
        float x = 0.f, y = 0.f, z = 0.f;
        if (q < n1) {
            const float *p = p1 + ((size_t)b * n1 + q) * 3;
            x = p[0]; y = p[1]; z = p[2];
        }
        const float an = (x * x + y * y) + z * z;
        a_d1[qt] = kk == 0 ? -2.f * x : kk == 1 ? -2.f * y : kk == 2 ? -2.f * z : an;
        a_d2[qt] = kk == 0 ? 1.f : 0.f;
#pragma unroll
        for (int v = 0; v < NM; ++v) a_m[qt][v] = (q < n1 && 4 * v + kk < k) ? m1[((size_t)b * n1 + q) * k + 4 * v + kk] : 0.f;
    }
    float mx[SM_QT][4], Z[SM_QT][4], A[SM_QT][4], Vx[SM_QT][4], Vy[SM_QT][4], Vz[SM_QT][4];
#pragma unroll
    for (int qt = 0; qt < SM_QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mx[qt][r] = -3.0e38f;
            Z[qt][r] = A[qt][r] = Vx[qt][r] = Vy[qt][r] = Vz[qt][r] = 0.f;
        }
    const int per = tiles_pad / SM_SPLITS; // (a multiple of SM_U)
    const float *tb = table + (((size_t)b * tiles_pad + (size_t)wave * per) * NB) * 64 + lane;
    const float4 *cb = coords + ((size_t)b * tiles_pad + (size_t)wave * per) * 16 + col;
    const sn_v4f zero = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < per; t0 += SM_U, tb += SM_U * NB * 64, cb += SM_U * 16) {
        float bd1[SM_U], bd2[SM_U], bm[SM_U][NM];
        float4 cq[SM_U];
#pragma unroll
        for (int u = 0; u < SM_U; ++u) {
            bd1[u] = tb[u * NB * 64];
            bd2[u] = tb[u * NB * 64 + 64];
#pragma unroll
            for (int v = 0; v < NM; ++v) bm[u][v] = tb[u * NB * 64 + (2 + v) * 64];
            cq[u] = cb[u * 16];
        }
#pragma unroll
        for (int qt = 0; qt < SM_QT; ++qt) {
            float y[SM_U][4], c[SM_U][4];
#pragma unroll
            for (int u = 0; u < SM_U; ++u) {
                sn_v4f dd = __builtin_amdgcn_mfma_f32_16x16x4f32(a_d1[qt], bd1[u], zero, 0, 0, 0);
                dd = __builtin_amdgcn_mfma_f32_16x16x4f32(a_d2[qt], bd2[u], dd, 0, 0, 0);
                sn_v4f cc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_m[qt][0], bm[u][0], zero, 0, 0, 0);
#pragma unroll
                for (int v = 1; v < NM; ++v) cc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_m[qt][v], bm[u][v], cc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[u][r] = __builtin_amdgcn_sqrtf(fmaxf(dd[r], 1e-30f)) * scale2; // log2(e) x, x = -d / tau
                    c[u][r] = cc[r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float cmax = y[0][r];
#pragma unroll
                for (int u = 1; u < SM_U; ++u) cmax = fmaxf(cmax, y[u][r]);
                const float mn = fmaxf(mx[qt][r], cmax);
                const float sc = __builtin_amdgcn_exp2f(mx[qt][r] - mn); // 1 when the maximum stands
                mx[qt][r] = mn;
                float z_ = Z[qt][r] * sc, a_ = A[qt][r] * sc, vx = Vx[qt][r] * sc, vy = Vy[qt][r] * sc, vz = Vz[qt][r] * sc;
#pragma unroll
                for (int u = 0; u < SM_U; ++u) {
                    const float e = __builtin_amdgcn_exp2f(y[u][r] - mn);
                    const float w = e * c[u][r];
                    z_ += e;
                    a_ += w;
                    vx = fmaf(w, cq[u].x, vx); vy = fmaf(w, cq[u].y, vy); vz = fmaf(w, cq[u].z, vz);
                }
                Z[qt][r] = z_; A[qt][r] = a_; Vx[qt][r] = vx; Vy[qt][r] = vy; Vz[qt][r] = vz;
            }
        }
    }
    // merge the sixteen candidate columns of a row group (lanes with equal kk): butterfly over col
#pragma unroll
    for (int qt = 0; qt < SM_QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m_ = mx[qt][r], z_ = Z[qt][r], a_ = A[qt][r], vx = Vx[qt][r], vy = Vy[qt][r], vz = Vz[qt][r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const float mo = __shfl_xor(m_, off, 64), zo = __shfl_xor(z_, off, 64), ao = __shfl_xor(a_, off, 64);
                const float xo = __shfl_xor(vx, off, 64), yo = __shfl_xor(vy, off, 64), wo = __shfl_xor(vz, off, 64);
                const float mn = fmaxf(m_, mo);
                const float s1 = __builtin_amdgcn_exp2f(m_ - mn), s2 = __builtin_amdgcn_exp2f(mo - mn);
                z_ = fmaf(z_, s1, zo * s2); a_ = fmaf(a_, s1, ao * s2);
                vx = fmaf(vx, s1, xo * s2); vy = fmaf(vy, s1, yo * s2); vz = fmaf(vz, s1, wo * s2);
                m_ = mn;
            }
            if (col == 0) {
                float *o = part[wave][qt * 16 + kk * 4 + r];
                o[0] = m_; o[1] = z_; o[2] = a_; o[3] = vx; o[4] = vy; o[5] = vz;
            }
        }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < SM_QT * 16 && qbase + t < n1) {
        float gm = part[0][t][0];
#pragma unroll
        for (int w = 1; w < SM_SPLITS; ++w) gm = fmaxf(gm, part[w][t][0]);
        float z = 0.f, a = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
        for (int w = 0; w < SM_SPLITS; ++w) {
            const float s = __builtin_amdgcn_exp2f(part[w][t][0] - gm);
            z = fmaf(part[w][t][1], s, z);
            a = fmaf(part[w][t][2], s, a);
            vx = fmaf(part[w][t][3], s, vx);
            vy = fmaf(part[w][t][4], s, vy);
            vz = fmaf(part[w][t][5], s, vz);
        }
        // corr = softmax * consistency; corr /= corr.sum().clamp(1e-10); target = corr @ pc2
        const float inv_z = 1.0f / z;
        const float denom = fmaxf(a * inv_z, 1e-10f);
        float *o = out + ((size_t)b * n1 + qbase + t) * 3;
        o[0] = (vx * inv_z) / denom;
        o[1] = (vy * inv_z) / denom;
        o[2] = (vz * inv_z) / denom;
    }
}

template <int NM>
int soft_nn_mfma_launch(int b, int n1, int n2, int k, float temperature, const float *p1, const float *p2, const float *m1,
                        const float *m2, float *target, hipStream_t s) {
    constexpr int NB = 2 + NM;
    const int group = SM_SPLITS * SM_U;
    const int tiles_pad = ogc_divup(ogc_divup(n2, 16), group) * group;
    const size_t table_floats = (size_t)b * tiles_pad * NB * 64, coord_floats = (size_t)b * tiles_pad * 16 * 4;
    float *ws = static_cast<float *>(ogc_workspace(s, (table_floats + coord_floats) * sizeof(float) + 64));
    if (!ws) {
        ogc_set_error("ogc_soft_nn_target: no workspace for the packed candidates");
        return OGC_ERR_LAUNCH;
    }
    float *table = ws;
    float4 *coords = reinterpret_cast<float4 *>(ws + (table_floats + 3) / 4 * 4);
    hipLaunchKernelGGL(soft_nn_pack_kernel<NM>, dim3(ogc_divup(tiles_pad * 16, 256), b), dim3(256), 0, s, n2, k, tiles_pad, p2, m2, table,
                       coords);
    const float scale2 = (float)(-1.4426950408889634 / (double)temperature);
    hipLaunchKernelGGL(soft_nn_mfma_kernel<NM>, dim3(ogc_divup(n1, SM_QT * 16), b), dim3(SM_SPLITS * OGC_WAVE), 0, s, n1, k, tiles_pad,
                       scale2, p1, m1, table, coords, target);
    OGC_CHECK_LAUNCH("ogc_soft_nn_target");
    return OGC_OK;
}

} // namespace

extern "C" int ogc_soft_nn_target(int b, int n1, int n2, int k, float temperature, const float *p1, const float *p2,
                                  const float *mask1, const float *mask2, float *target, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n1 >= 0 && n2 >= 1 && k >= 1, "ogc_soft_nn_target: bad shape");
    OGC_REQUIRE(temperature > 0.0f, "ogc_soft_nn_target: temperature must be positive");
    if (b == 0 || n1 == 0) return OGC_OK;
    OGC_REQUIRE(p1 && p2 && mask1 && mask2 && target, "ogc_soft_nn_target: null pointer");
    if (k > 32) {
        ogc_set_error("ogc_soft_nn_target: more than 32 slots (k=%d)", k);
        return OGC_ERR_UNSUPPORTED;
    }
    const dim3 grid(ogc_divup(n1, OGC_WAVE), b), block(SN_WAVES * OGC_WAVE);
    hipStream_t s = (hipStream_t)stream;
    // the matrix-core form from a few hundred pairs per query on (below that the packing and the butterfly are the cost);
    // OGC_SOFT_NN_MFMA=0: the lane-per-query kernel for every shape (A/B runs, tests of both)
    static const bool mfma_off = [] { const char *e = getenv("OGC_SOFT_NN_MFMA"); return e && e[0] == '0'; }();
    if (!mfma_off && n2 >= 256 && b <= 65535) {
        if (k <= 8) return soft_nn_mfma_launch<2>(b, n1, n2, k, temperature, p1, p2, mask1, mask2, target, s);
        if (k <= 12) return soft_nn_mfma_launch<3>(b, n1, n2, k, temperature, p1, p2, mask1, mask2, target, s);
        if (k <= 16) return soft_nn_mfma_launch<4>(b, n1, n2, k, temperature, p1, p2, mask1, mask2, target, s);
        return soft_nn_mfma_launch<8>(b, n1, n2, k, temperature, p1, p2, mask1, mask2, target, s);
    }
    if (k <= 8)
        hipLaunchKernelGGL(soft_nn_kernel<8>, grid, block, 0, s, n1, n2, k, temperature, p1, p2, mask1, mask2, target);
    else if (k <= 16)
        hipLaunchKernelGGL(soft_nn_kernel<16>, grid, block, 0, s, n1, n2, k, temperature, p1, p2, mask1, mask2, target);
    else
        hipLaunchKernelGGL(soft_nn_kernel<32>, grid, block, 0, s, n1, n2, k, temperature, p1, p2, mask1, mask2, target);
    OGC_CHECK_LAUNCH("ogc_soft_nn_target");
    return OGC_OK;
}
