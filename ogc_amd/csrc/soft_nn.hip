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
    if (k <= 8)
        hipLaunchKernelGGL(soft_nn_kernel<8>, grid, block, 0, s, n1, n2, k, temperature, p1, p2, mask1, mask2, target);
    else if (k <= 16)
        hipLaunchKernelGGL(soft_nn_kernel<16>, grid, block, 0, s, n1, n2, k, temperature, p1, p2, mask1, mask2, target);
    else
        hipLaunchKernelGGL(soft_nn_kernel<32>, grid, block, 0, s, n1, n2, k, temperature, p1, p2, mask1, mask2, target);
    OGC_CHECK_LAUNCH("ogc_soft_nn_target");
    return OGC_OK;
}
