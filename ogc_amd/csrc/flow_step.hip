// flow_step.hip — the element-wise glue of FlowStep3D's refinement loop (inference), one launch per group of framework operators.
//
// Reference: models/flownet_kitti.py:135-151 (GRU), :22-38 (FlowRegressor's fc between two transposes), :229-250 (the loop:
// pc1_new_lr - pc1_l_loc[2], / (k_decay_fact * it + 1), pc1_new_lr + delta, pc1_new + delta_flow, pc1_new - pc1) and
// utils/flowstep3d_util.py:110-118 (gather of the cached centres, then their transpose).  One forward at C3 (one pair of 8192
// points, iters = 5) is ~450 launches of a few microseconds each and its time is their number: every kernel here stands for
// three to six framework launches on tensors of a few thousand elements.  Arithmetic: the same fp32 operations in the same
// order as the operator sequence it replaces, one rounding per operation (no contraction).
//     gather_xyz_pair   out = xyz[:, :, idx] as (b, 3, m) AND as (b, m, 3)
//     flow_advance      d = delta * scale;  new = cur + d;  flow = new - ref   (+ new as (b, n, 3))
//     linear_cn         y (b, cout, n) = W x + bias on channel-major x (b, cin, n), cout <= 4
//     gru_reset         out = cat([sigmoid(max_s rc) * h, x])          with hx = cat([h, x]) as the input
//     gru_blend         z = sigmoid(max_s zc), q = tanh(max_s qc):  out = (1 - z) * h + z * q
#include "ogc_common.h"

namespace {

// torch.amax's maximum: a NaN stays
__device__ __forceinline__ float max_nan(float a, float b) { return (a > b || a != a) ? a : b; }
// torch's sigmoid (1 / (1 + exp(-x)), fp32) and tanh
__device__ __forceinline__ float sigmoid_f32(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }

__device__ __forceinline__ float pooled(const float *__restrict__ p, int s) {
    float m;
    if ((s & 3) == 0) {
        const float4 *q = reinterpret_cast<const float4 *>(p);
        float4 v = q[0];
        m = max_nan(max_nan(v.x, v.y), max_nan(v.z, v.w));
        for (int j = 1; j < (s >> 2); ++j) {
            v = q[j];
            m = max_nan(m, max_nan(max_nan(v.x, v.y), max_nan(v.z, v.w)));
        }
    } else {
        m = p[0];
        for (int j = 1; j < s; ++j) m = max_nan(m, p[j]);
    }
    return m;
}

__global__ __launch_bounds__(256) void gather_xyz_pair_kernel(int n, int m, const float *__restrict__ xyz,
                                                              const int *__restrict__ idx, float *__restrict__ out,
                                                              float *__restrict__ out_t) {
    const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int i = idx[(size_t)b * m + j];
    const float *src = xyz + (size_t)b * 3 * n;
    const float x = src[i], y = src[n + i], z = src[2 * (size_t)n + i];
    float *o = out + (size_t)b * 3 * m;
    o[j] = x; o[m + j] = y; o[2 * (size_t)m + j] = z;
    float *t = out_t + ((size_t)b * m + j) * 3;
    t[0] = x; t[1] = y; t[2] = z;
}

__global__ __launch_bounds__(256) void flow_advance_kernel(int n, float scale, const float *__restrict__ cur,
                                                           const float *__restrict__ delta, const float *__restrict__ ref,
                                                           float *__restrict__ out_delta, float *__restrict__ out_new,
                                                           float *__restrict__ out_new_t, float *__restrict__ out_flow) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const size_t e = ((size_t)b * 3 + a) * n + i;
        const float d = scale == 1.0f ? delta[e] : __fmul_rn(delta[e], scale);
        const float v = __fadd_rn(cur[e], d);
        if (out_delta) out_delta[e] = d;
        out_new[e] = v;
        if (out_new_t) out_new_t[((size_t)b * n + i) * 3 + a] = v;
        if (out_flow) out_flow[e] = __fsub_rn(v, ref[e]);
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void linear_cn_kernel(int cin, int n, const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ y) {
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *src = x + (size_t)b * cin * n + i;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
    for (int c = 0; c < cin; ++c) {
        const float v = src[(size_t)c * n];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(w[o * cin + c], v, acc[o]); // (w: wave-uniform addresses, scalar loads)
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) y[((size_t)b * COUT + o) * n + i] = bias ? acc[o] + bias[o] : acc[o];
}

__global__ __launch_bounds__(256) void gru_reset_kernel(int c, int cx, int n, int s, const float *__restrict__ rc, long long rc_bs,
                                                        const float *__restrict__ hx, float *__restrict__ out) {
    const int b = blockIdx.z, ch = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t e = ((size_t)b * (c + cx) + ch) * n + i;
    float v = hx[e];
    if (ch < c) {
        const float r = sigmoid_f32(pooled(rc + (size_t)b * rc_bs + ((size_t)ch * n + i) * s, s));
        v = __fmul_rn(r, v);
    }
    out[e] = v;
}

__global__ __launch_bounds__(256) void gru_blend_kernel(int c, int n, int s, const float *__restrict__ zc, long long zc_bs,
                                                        const float *__restrict__ qc, long long qc_bs, const float *__restrict__ h,
                                                        long long h_bs, float *__restrict__ out) {
    const int b = blockIdx.z, ch = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t p = ((size_t)ch * n + i);
    const float z = sigmoid_f32(pooled(zc + (size_t)b * zc_bs + p * s, s));
    const float q = tanhf(pooled(qc + (size_t)b * qc_bs + p * s, s));
    const float hv = h[(size_t)b * h_bs + p];
    out[((size_t)b * c + ch) * n + i] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, z), hv), __fmul_rn(z, q));
}

} // namespace

extern "C" int ogc_gather_xyz_pair(int b, int n, int m, const float *xyz, const int *idx, float *out, float *out_t,
                                   ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ogc_gather_xyz_pair: negative dimension");
    if (b == 0 || m == 0) return OGC_OK;
    OGC_REQUIRE(n >= 1, "ogc_gather_xyz_pair: indices into an empty cloud");
    OGC_REQUIRE(xyz && idx && out && out_t, "ogc_gather_xyz_pair: null pointer");
    OGC_REQUIRE(b <= 65535 && (long long)b * (n > m ? n : m) * 3 < (1ll << 31), "ogc_gather_xyz_pair: exceeds 32-bit indexing");
    hipLaunchKernelGGL(gather_xyz_pair_kernel, dim3(ogc_divup(m, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, xyz, idx, out,
                       out_t);
    OGC_CHECK_LAUNCH("ogc_gather_xyz_pair");
    return OGC_OK;
}

extern "C" int ogc_flow_advance(int b, int n, float scale, const float *cur, const float *delta, const float *ref,
                                float *out_delta, float *out_new, float *out_new_t, float *out_flow, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0, "ogc_flow_advance: negative dimension");
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(cur && delta && out_new && (ref || !out_flow), "ogc_flow_advance: null pointer");
    OGC_REQUIRE(b <= 65535 && (long long)b * n * 3 < (1ll << 31), "ogc_flow_advance: exceeds 32-bit indexing");
    hipLaunchKernelGGL(flow_advance_kernel, dim3(ogc_divup(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, scale, cur, delta, ref,
                       out_delta, out_new, out_new_t, out_flow);
    OGC_CHECK_LAUNCH("ogc_flow_advance");
    return OGC_OK;
}

extern "C" int ogc_linear_cn(int b, int cin, int cout, int n, const float *x, const float *weight, const float *bias, float *y,
                             ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && cin >= 0 && n >= 0, "ogc_linear_cn: negative dimension");
    OGC_REQUIRE(cout >= 1 && cout <= 4, "ogc_linear_cn: 1 .. 4 output channels (got %d)", cout);
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE((x || cin == 0) && (weight || cin == 0) && y, "ogc_linear_cn: null pointer");
    OGC_REQUIRE(b <= 65535 && (long long)b * (cin > cout ? cin : cout) * n < (1ll << 31), "ogc_linear_cn: exceeds 32-bit indexing");
    const dim3 grid(ogc_divup(n, 256), b), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (cout) {
    case 1: hipLaunchKernelGGL(linear_cn_kernel<1>, grid, block, 0, s, cin, n, x, weight, bias, y); break;
    case 2: hipLaunchKernelGGL(linear_cn_kernel<2>, grid, block, 0, s, cin, n, x, weight, bias, y); break;
    case 3: hipLaunchKernelGGL(linear_cn_kernel<3>, grid, block, 0, s, cin, n, x, weight, bias, y); break;
    default: hipLaunchKernelGGL(linear_cn_kernel<4>, grid, block, 0, s, cin, n, x, weight, bias, y); break;
    }
    OGC_CHECK_LAUNCH("ogc_linear_cn");
    return OGC_OK;
}

extern "C" int ogc_gru_reset(int b, int c, int cx, int n, int s, const float *rc, long long rc_batch_stride, const float *hx,
                             float *out, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 0 && cx >= 0 && n >= 0 && s >= 1, "ogc_gru_reset: bad dimension");
    if (b == 0 || n == 0 || c + cx == 0) return OGC_OK;
    OGC_REQUIRE((rc || c == 0) && hx && out, "ogc_gru_reset: null pointer");
    OGC_REQUIRE(rc_batch_stride >= (long long)c * n * s && (rc_batch_stride & 3) == 0 && ((uintptr_t)rc & 15) == 0,
                "ogc_gru_reset: batch stride / alignment of the gate");
    OGC_REQUIRE(b <= 65535 && c + cx <= 65535 && (long long)b * (c + cx) * n < (1ll << 31) &&
                    (long long)b * rc_batch_stride < (1ll << 31), "ogc_gru_reset: exceeds 32-bit indexing");
    hipLaunchKernelGGL(gru_reset_kernel, dim3(ogc_divup(n, 256), c + cx, b), dim3(256), 0, (hipStream_t)stream, c, cx, n, s, rc,
                       rc_batch_stride, hx, out);
    OGC_CHECK_LAUNCH("ogc_gru_reset");
    return OGC_OK;
}

extern "C" int ogc_gru_blend(int b, int c, int n, int s, const float *zc, long long zc_batch_stride, const float *qc,
                             long long qc_batch_stride, const float *h, long long h_batch_stride, float *out, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 0 && n >= 0 && s >= 1, "ogc_gru_blend: bad dimension");
    if (b == 0 || n == 0 || c == 0) return OGC_OK;
    OGC_REQUIRE(zc && qc && h && out, "ogc_gru_blend: null pointer");
    const long long need = (long long)c * n * s;
    OGC_REQUIRE(zc_batch_stride >= need && qc_batch_stride >= need && h_batch_stride >= (long long)c * n &&
                    ((zc_batch_stride | qc_batch_stride) & 3) == 0 && (((uintptr_t)zc | (uintptr_t)qc) & 15) == 0,
                "ogc_gru_blend: batch strides / alignment of the gates");
    OGC_REQUIRE(b <= 65535 && c <= 65535 && (long long)b * zc_batch_stride < (1ll << 31) && (long long)b * qc_batch_stride < (1ll << 31) &&
                    (long long)b * h_batch_stride < (1ll << 31), "ogc_gru_blend: exceeds 32-bit indexing");
    hipLaunchKernelGGL(gru_blend_kernel, dim3(ogc_divup(n, 256), c, b), dim3(256), 0, (hipStream_t)stream, c, n, s, zc,
                       zc_batch_stride, qc, qc_batch_stride, h, h_batch_stride, out);
    OGC_CHECK_LAUNCH("ogc_gru_blend");
    return OGC_OK;
}
