// attention.hip — the attention core of the MaskFormer head's nn.MultiheadAttention layers, forward and backward.
//
// Reference: utils/transformer_util.py:5-62 (TransformerDecoderLayer: cross-attention of K slots over the N points of
// the coarsest level, self-attention among the slots), models/segnet_kitti.py:49-51 (8 heads, embed 128).
// At the C4 shapes that is 16 samples x 8 heads x 10 queries x {512, 10} keys of 16 floats: a few MFLOP, which
// torch runs as ~14 launches forward and ~30 backward per attention (head split / merge copies, scaling, two batched
// GEMMs, softmax, and their adjoints) — the head's 286 launches cost 1.7 ms of GPU time per training step for
// ~0.1 ms of arithmetic.  Here the core  O = softmax(scale * Q K^T) V  per (sample, head) is one kernel each way,
// reading Q / K / V in place from the projection outputs (row stride = width of the packed projection, head = column
// block) and writing the merged-head layout the output projection takes: no copies on either side.
//
// forward : one wavefront per (sample, head, query): lanes over the keys, two passes (maximum; exponentials, their
//           sum and the weighted sum of V), probabilities kept for the backward pass.
// backward: one workgroup per (sample, head).  With D_q = dO_q . O_q  (= sum_n P_qn dP_qn):
//             dS_qn = P_qn (dO_q . V_n - D_q),  dV_n = sum_q P_qn dO_q,  dK_n = scale sum_q dS_qn Q_q   (thread per key)
//             dQ_q  = scale sum_n dS_qn K_n                                               (thread per (query, column))
//           dS lives in LDS between the two phases (lq * lk floats).
#include "ogc_common.h"

namespace {

constexpr int AT_THREADS = 256;

template <int D>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(int items, int lq, int lk, int h, float scale,
                                                             const float *__restrict__ q, int ldq,
                                                             const float *__restrict__ k, int ldk,
                                                             const float *__restrict__ v, int ldv,
                                                             float *__restrict__ out, float *__restrict__ prob) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * (AT_THREADS / 64) + (threadIdx.x >> 6); // (sample, head, query)
    if (item >= items) return; // whole wavefront
    const int qi = item % lq, hi = (item / lq) % h, bi = item / (lq * h);
    const float *qp = q + ((size_t)bi * lq + qi) * ldq + hi * D;
    const float *kp = k + (size_t)bi * lk * ldk + hi * D;
    const float *vp = v + (size_t)bi * lk * ldv + hi * D;
    float *pp = prob + (((size_t)bi * h + hi) * lq + qi) * lk;
    float qv[D];
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(qp + c);
        qv[c] = t.x * scale; qv[c + 1] = t.y * scale; qv[c + 2] = t.z * scale; qv[c + 3] = t.w * scale;
    }
    auto score = [&](int n) {
        const float *kr = kp + (size_t)n * ldk;
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(kr + c);
            s = fmaf(qv[c], t.x, s); s = fmaf(qv[c + 1], t.y, s); s = fmaf(qv[c + 2], t.z, s); s = fmaf(qv[c + 3], t.w, s);
        }
        return s;
    };
    float m = -INFINITY;
    for (int n = lane; n < lk; n += 64) m = fmaxf(m, score(n));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float z = 0.0f, o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] = 0.0f;
    for (int n = lane; n < lk; n += 64) {
        const float e = expf(score(n) - m);
        pp[n] = e;
        z += e;
        const float *vr = vp + (size_t)n * ldv;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(vr + c);
            o[c] = fmaf(e, t.x, o[c]); o[c + 1] = fmaf(e, t.y, o[c + 1]);
            o[c + 2] = fmaf(e, t.z, o[c + 2]); o[c + 3] = fmaf(e, t.w, o[c + 3]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        z += __shfl_xor(z, off, 64);
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] += __shfl_xor(o[c], off, 64);
    }
    const float inv = 1.0f / z;
    for (int n = lane; n < lk; n += 64) pp[n] *= inv; // each lane rescales what it wrote itself
    if (lane == 0) {
        float *op = out + ((size_t)bi * lq + qi) * (h * D) + hi * D;
#pragma unroll
        for (int c = 0; c < D; c += 4)
            *reinterpret_cast<float4 *>(op + c) = make_float4(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv);
    }
}

template <int D>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_kernel(int lq, int lk, int h, float scale,
                                                             const float *__restrict__ q, int ldq,
                                                             const float *__restrict__ k, int ldk,
                                                             const float *__restrict__ v, int ldv,
                                                             const float *__restrict__ out,
                                                             const float *__restrict__ prob,
                                                             const float *__restrict__ dout, float *__restrict__ dq,
                                                             int lddq, float *__restrict__ dk, int lddk,
                                                             float *__restrict__ dv, int lddv) {
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    float *qs = at_smem;             // [lq][D]
    float *dos = qs + lq * D;        // [lq][D]
    float *dsum = dos + lq * D;      // [lq]   D_q
    float *dsl = dsum + ((lq + 3) & ~3); // [lq][lk]
    const int hi = blockIdx.x % h, bi = blockIdx.x / h;
    const int e = h * D;
    for (int t = threadIdx.x; t < lq * D; t += AT_THREADS) {
        const int qi = t / D, c = t % D;
        qs[t] = q[((size_t)bi * lq + qi) * ldq + hi * D + c];
        dos[t] = dout[((size_t)bi * lq + qi) * e + hi * D + c];
    }
    __syncthreads();
    for (int qi = threadIdx.x; qi < lq; qi += AT_THREADS) {
        const float *orow = out + ((size_t)bi * lq + qi) * e + hi * D;
        float s = 0.0f;
        for (int c = 0; c < D; ++c) s = fmaf(dos[qi * D + c], orow[c], s);
        dsum[qi] = s;
    }
    __syncthreads();
    const float *pb = prob + ((size_t)bi * h + hi) * lq * lk;
    for (int n = threadIdx.x; n < lk; n += AT_THREADS) {
        const float *kr = k + ((size_t)bi * lk + n) * ldk + hi * D;
        const float *vr = v + ((size_t)bi * lk + n) * ldv + hi * D;
        float vn[D], gk[D], gv[D];
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(vr + c);
            vn[c] = t.x; vn[c + 1] = t.y; vn[c + 2] = t.z; vn[c + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < D; ++c) gk[c] = gv[c] = 0.0f;
        for (int qi = 0; qi < lq; ++qi) {
            const float p = pb[(size_t)qi * lk + n];
            const float *dor = dos + qi * D, *qr = qs + qi * D;
            float dp = 0.0f;
#pragma unroll
            for (int c = 0; c < D; ++c) dp = fmaf(dor[c], vn[c], dp);
            const float ds = p * (dp - dsum[qi]);
            dsl[qi * lk + n] = ds;
            const float dss = ds * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                gv[c] = fmaf(p, dor[c], gv[c]);
                gk[c] = fmaf(dss, qr[c], gk[c]);
            }
        }
        float *gkr = dk + ((size_t)bi * lk + n) * lddk + hi * D;
        float *gvr = dv + ((size_t)bi * lk + n) * lddv + hi * D;
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            *reinterpret_cast<float4 *>(gkr + c) = make_float4(gk[c], gk[c + 1], gk[c + 2], gk[c + 3]);
            *reinterpret_cast<float4 *>(gvr + c) = make_float4(gv[c], gv[c + 1], gv[c + 2], gv[c + 3]);
        }
        (void)kr;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < lq * D; t += AT_THREADS) {
        const int qi = t / D, c = t % D;
        const float *kc = k + (size_t)bi * lk * ldk + hi * D + c;
        const float *dsr = dsl + qi * lk;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int n = 0;
        for (; n + 3 < lk; n += 4) {
            a0 = fmaf(dsr[n], kc[(size_t)n * ldk], a0);
            a1 = fmaf(dsr[n + 1], kc[(size_t)(n + 1) * ldk], a1);
            a2 = fmaf(dsr[n + 2], kc[(size_t)(n + 2) * ldk], a2);
            a3 = fmaf(dsr[n + 3], kc[(size_t)(n + 3) * ldk], a3);
        }
        for (; n < lk; ++n) a0 = fmaf(dsr[n], kc[(size_t)n * ldk], a0);
        dq[((size_t)bi * lq + qi) * lddq + hi * D + c] = ((a0 + a1) + (a2 + a3)) * scale;
    }
}

size_t attn_bwd_lds(int lq, int lk, int d) {
    return sizeof(float) * ((size_t)2 * lq * d + ((lq + 3) & ~3) + (size_t)lq * lk);
}

int attn_check(const char *name, int b, int lq, int lk, int h, int d, int ldq, int ldk, int ldv) {
    OGC_REQUIRE(b >= 0 && lq >= 1 && lk >= 1 && h >= 1, "%s: bad shape", name);
    if (d != 16 && d != 32) {
        ogc_set_error("%s: head width %d (supported: 16, 32)", name, d);
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE(ldq >= h * d && ldk >= h * d && ldv >= h * d && ((ldq | ldk | ldv) & 3) == 0,
                "%s: row strides must cover all heads and be multiples of 4 floats", name);
    OGC_REQUIRE((long long)b * h * lq * lk < (1ll << 31) && (long long)b * lk * ldk < (1ll << 31) &&
                    (long long)b * lk * ldv < (1ll << 31) && (long long)b * lq * ldq < (1ll << 31),
                "%s: tensor exceeds 32-bit indexing", name);
    return OGC_OK;
}

} // namespace

extern "C" int ogc_attention_fwd(int b, int lq, int lk, int h, int d, float scale, const float *q, int ldq,
                                 const float *k, int ldk, const float *v, int ldv, float *out, float *prob,
                                 ogc_stream_t stream) {
    const int rc = attn_check("ogc_attention_fwd", b, lq, lk, h, d, ldq, ldk, ldv);
    if (rc != OGC_OK) return rc;
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(q && k && v && out && prob, "ogc_attention_fwd: null pointer");
    OGC_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0,
                "ogc_attention_fwd: q, k, v, out must be 16-byte aligned");
    const int items = b * h * lq;
    const dim3 grid(ogc_divup(items, AT_THREADS / 64)), block(AT_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (d == 16)
        hipLaunchKernelGGL(attn_fwd_kernel<16>, grid, block, 0, s, items, lq, lk, h, scale, q, ldq, k, ldk, v, ldv, out,
                           prob);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, 0, s, items, lq, lk, h, scale, q, ldq, k, ldk, v, ldv, out,
                           prob);
    OGC_CHECK_LAUNCH("ogc_attention_fwd");
    return OGC_OK;
}

extern "C" int ogc_attention_bwd(int b, int lq, int lk, int h, int d, float scale, const float *q, int ldq,
                                 const float *k, int ldk, const float *v, int ldv, const float *out,
                                 const float *prob, const float *dout, float *dq, int lddq, float *dk, int lddk,
                                 float *dv, int lddv, ogc_stream_t stream) {
    const int rc = attn_check("ogc_attention_bwd", b, lq, lk, h, d, ldq, ldk, ldv);
    if (rc != OGC_OK) return rc;
    OGC_REQUIRE(lddq >= h * d && lddk >= h * d && lddv >= h * d && ((lddk | lddv) & 3) == 0,
                "ogc_attention_bwd: gradient row strides must cover all heads (dk, dv: multiples of 4 floats)");
    const size_t lds = attn_bwd_lds(lq, lk, d);
    if (lds > 64 * 1024) {
        ogc_set_error("ogc_attention_bwd: lq * lk = %d x %d does not fit the workgroup's 64 KiB of LDS", lq, lk);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(q && k && v && out && prob && dout && dq && dk && dv, "ogc_attention_bwd: null pointer");
    OGC_REQUIRE((((uintptr_t)k | (uintptr_t)v | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                "ogc_attention_bwd: k, v, dk, dv must be 16-byte aligned");
    const dim3 grid(b * h), block(AT_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (d == 16)
        hipLaunchKernelGGL(attn_bwd_kernel<16>, grid, block, lds, s, lq, lk, h, scale, q, ldq, k, ldk, v, ldv, out, prob,
                           dout, dq, lddq, dk, lddk, dv, lddv);
    else
        hipLaunchKernelGGL(attn_bwd_kernel<32>, grid, block, lds, s, lq, lk, h, scale, q, ldq, k, ldk, v, ldv, out, prob,
                           dout, dq, lddq, dk, lddk, dv, lddv);
    OGC_CHECK_LAUNCH("ogc_attention_bwd");
    return OGC_OK;
}
