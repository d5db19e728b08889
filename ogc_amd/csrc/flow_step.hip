// flow_step.hip — the element-wise glue of FlowStep3D's refinement loop (inference), one launch per group of framework operators.
//
// Reference: models/flownet_kitti.py:135-151 (GRU), :22-38 (FlowRegressor's fc between two transposes), :229-250 (the loop:
// pc1_new_lr - pc1_l_loc[2], / (k_decay_fact * it + 1), pc1_new_lr + delta, pc1_new + delta_flow, pc1_new - pc1) and
// utils/flowstep3d_util.py:110-118 (gather of the cached centres, then their transpose).  One forward at C3 (one pair of 8192
// points, iters = 5) is ~450 launches of a few microseconds each and its time is their number: every kernel here stands for
// three to six framework launches on tensors of a few thousand elements.  Arithmetic: the same fp32 operations in the same
// order as the operator sequence it replaces, one rounding per operation (no contraction) — except soft_corr_flow and
// linear_cn, whose dot products are fused multiply-add chains (the reference normalises the features and calls bmm / addmm,
// a different summation order): those two are held to a tolerance, not to the bit (tests/test_flow_glue_gpu.py).
//     gather_xyz_pair   out = xyz[:, :, idx] as (b, 3, m) AND as (b, m, 3)
//     flow_advance      d = delta * scale;  new = cur + d;  flow = new - ref   (+ new as (b, n, 3))
//     linear_cn         y (b, cout, n) = W x + bias on channel-major x (b, cin, n), cout <= 4
//     gru_reset         out = cat([sigmoid(max_s rc) * h, x])          with hx = cat([h, x]) as the input
//     gru_blend         z = sigmoid(max_s zc), q = tanh(max_s qc):  out = (1 - z) * h + z * q
//     three_nn_weights  normalised inverse distances of the three neighbours from ogc_three_nn's squared distances
//     soft_corr_flow    the dense soft correlation of the coarsest levels and the flow it implies (see the kernel)
#include <type_traits>

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

// 64 points per workgroup, one wavefront per quarter of the input channels (a thread's loads are independent of its sums, eight in
// flight at a time; consecutive lanes read consecutive points): a lane per point walking all cin channels alone is one dependent
// chain of cin loads — 32 us for 128 channels on 2048 points, against 4 us here.  The four partial sums are added in slice order.
template <int COUT>
__global__ __launch_bounds__(256) void linear_cn_kernel(int cin, int n, const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ y) {
    __shared__ float part[3][COUT][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, slice = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
    const int per = (cin + 3) >> 2, c0 = slice * per, c1 = min(cin, c0 + per);
    const float *src = x + (size_t)b * cin * n + min(i, n - 1);
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(c + u) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[o] = fmaf(w[o * cin + c + u], v[u], acc[o]); // (w: wave-uniform, scalar loads)
    }
    for (; c < c1; ++c) {
        const float v = src[(size_t)c * n];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(w[o * cin + c], v, acc[o]);
    }
    if (slice > 0) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) part[slice - 1][o][lane] = acc[o];
    }
    __syncthreads();
    if (slice == 0 && i < n) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float sum = ((acc[o] + part[0][o][lane]) + part[1][o][lane]) + part[2][o][lane];
            y[((size_t)b * COUT + o) * n + i] = bias ? sum + bias[o] : sum;
        }
    }
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

// inverse-distance weights of the three nearest neighbours from their SQUARED distances (what ogc_three_nn returns):
// mode 0 (utils/flowstep3d_util.py:169-170)  r_k = 1 / max(sqrt(d2_k), 1e-10);  mode 1 (utils/pointnet2_util.py:99-101)  r_k = 1 /
// (sqrt(d2_k) + 1e-8);  weight_k = r_k / ((r_0 + r_1) + r_2).  sqrt, clamp, reciprocal, sum and division as one launch.
__global__ __launch_bounds__(256) void three_nn_weights_kernel(long long rows, int mode, const float *__restrict__ dist2,
                                                               float *__restrict__ weight) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = sqrtf(dist2[i * 3 + k]);
        r[k] = __fdiv_rn(1.0f, mode == 0 ? fmaxf(d, 1e-10f) : __fadd_rn(d, 1e-8f));
    }
    const float sum = __fadd_rn(__fadd_rn(r[0], r[1]), r[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) weight[i * 3 + k] = __fdiv_rn(r[k], sum);
}

// The dense soft correlation of the coarsest levels and the flow it implies (GlobalCorrLayer.calc_corr_mat + the three lines
// after it, flownet_kitti.py:53-70): w_ij = exp(-(1 - cos(f1_i, f2_j)) / (e^epsilon + 0.03)) for pairs with
// (|p_i|^2 + |q_j|^2) - 2 p_i.q_j < support, else 0;  flow_i = sum_j w_ij q_j / (sum_j w_ij + 1e-8) - p_i.  ~25 framework launches
// on 256 x 256 matrices there.  A workgroup owns SC_ROWS points of cloud 1 (their features in LDS), a thread one point j of cloud 2
// at a time: the dot products of its feature column with the SC_ROWS rows, the weights, and their sums over j reduced over the
// workgroup at the end.  The cosine is (f1.f2) / (|f1| |f2|) with the reference's 1e-8 under both roots.
constexpr int SC_ROWS = 4, SC_MAXC = 256;
__global__ __launch_bounds__(256) void soft_corr_flow_kernel(int n1, int n2, int c, float support, const float *__restrict__ epsilon,
                                                             const float *__restrict__ pc1, const float *__restrict__ pc2,
                                                             const float *__restrict__ f1, const float *__restrict__ f2,
                                                             float *__restrict__ flow) {
    __shared__ float f1s[SC_MAXC][SC_ROWS];
    __shared__ float rows[SC_ROWS][8];    // x, y, z, |p|^2, 1 / |f1| of the workgroup's rows
    __shared__ float part[4][SC_ROWS][4]; // per wavefront: sum w, sum w q
    const int b = blockIdx.y, i0 = blockIdx.x * SC_ROWS, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *p1 = pc1 + (size_t)b * 3 * n1, *p2 = pc2 + (size_t)b * 3 * n2;
    const float *g1 = f1 + (size_t)b * c * n1, *g2 = f2 + (size_t)b * c * n2;
    for (int e = t; e < c * SC_ROWS; e += 256) {
        const int ch = e / SC_ROWS, ii = e % SC_ROWS;
        f1s[ch][ii] = i0 + ii < n1 ? g1[(size_t)ch * n1 + i0 + ii] : 0.0f;
    }
    __syncthreads();
    if (t < SC_ROWS) {
        const int i = min(i0 + t, n1 - 1);
        const float x = p1[i], y = p1[n1 + i], z = p1[2 * (size_t)n1 + i];
        float nn = 0.0f;
        for (int ch = 0; ch < c; ++ch) nn = fmaf(f1s[ch][t], f1s[ch][t], nn);
        rows[t][0] = x; rows[t][1] = y; rows[t][2] = z;
        rows[t][3] = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
        rows[t][4] = __fdiv_rn(1.0f, sqrtf(nn + 1e-8f));
    }
    __syncthreads();
    const float temperature = expf(epsilon[0]) + 0.03f;
    float sw[SC_ROWS], sx[SC_ROWS], sy[SC_ROWS], sz[SC_ROWS];
#pragma unroll
    for (int ii = 0; ii < SC_ROWS; ++ii) sw[ii] = sx[ii] = sy[ii] = sz[ii] = 0.0f;
    for (int j = t; j < n2; j += 256) {
        float dot[SC_ROWS], nn = 0.0f;
#pragma unroll
        for (int ii = 0; ii < SC_ROWS; ++ii) dot[ii] = 0.0f;
        auto channels = [&](int ch, auto width) { // `width` independent loads in flight (the loop is their latency otherwise)
            constexpr int U = decltype(width)::value;
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = g2[(size_t)(ch + u) * n2 + j];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                nn = fmaf(v[u], v[u], nn);
#pragma unroll
                for (int ii = 0; ii < SC_ROWS; ++ii) dot[ii] = fmaf(f1s[ch + u][ii], v[u], dot[ii]);
            }
        };
        int ch = 0;
        for (; ch + 16 <= c; ch += 16) channels(ch, std::integral_constant<int, 16>());
        for (; ch < c; ch += 4) channels(ch, std::integral_constant<int, 4>()); // (c is a multiple of 4)
        const float inv2 = __fdiv_rn(1.0f, sqrtf(nn + 1e-8f));
        const float qx = p2[j], qy = p2[n2 + j], qz = p2[2 * (size_t)n2 + j];
        const float qq = __fadd_rn(__fadd_rn(__fmul_rn(qx, qx), __fmul_rn(qy, qy)), __fmul_rn(qz, qz));
#pragma unroll
        for (int ii = 0; ii < SC_ROWS; ++ii) {
            const float gram = fmaf(rows[ii][2], qz, fmaf(rows[ii][1], qy, __fmul_rn(rows[ii][0], qx)));
            const bool near = __fsub_rn(__fadd_rn(rows[ii][3], qq), __fmul_rn(2.0f, gram)) < support;
            const float cosine = __fmul_rn(__fmul_rn(dot[ii], rows[ii][4]), inv2);
            const float w = near ? expf(__fdiv_rn(-__fsub_rn(1.0f, cosine), temperature)) : 0.0f;
            sw[ii] += w;
            sx[ii] = fmaf(w, qx, sx[ii]); sy[ii] = fmaf(w, qy, sy[ii]); sz[ii] = fmaf(w, qz, sz[ii]);
        }
    }
#pragma unroll
    for (int ii = 0; ii < SC_ROWS; ++ii) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sw[ii] += __shfl_xor(sw[ii], off, 64); sx[ii] += __shfl_xor(sx[ii], off, 64);
            sy[ii] += __shfl_xor(sy[ii], off, 64); sz[ii] += __shfl_xor(sz[ii], off, 64);
        }
        if (lane == 0) { part[wave][ii][0] = sw[ii]; part[wave][ii][1] = sx[ii]; part[wave][ii][2] = sy[ii]; part[wave][ii][3] = sz[ii]; }
    }
    __syncthreads();
    if (t < SC_ROWS * 3 && i0 + t / 3 < n1) {
        const int ii = t / 3, a = t % 3;
        const float w = (part[0][ii][0] + part[1][ii][0]) + (part[2][ii][0] + part[3][ii][0]);
        const float v = (part[0][ii][1 + a] + part[1][ii][1 + a]) + (part[2][ii][1 + a] + part[3][ii][1 + a]);
        flow[((size_t)b * 3 + a) * n1 + i0 + ii] = __fsub_rn(__fdiv_rn(v, __fadd_rn(w, 1e-8f)), rows[ii][a]);
    }
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
    const dim3 grid(ogc_divup(n, 64), b), block(256);
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
    // (16-byte loads only when a neighbourhood is a whole number of float4: other widths take pooled()'s scalar walk)
    OGC_REQUIRE(rc_batch_stride >= (long long)c * n * s && ((s & 3) != 0 || ((rc_batch_stride & 3) == 0 && ((uintptr_t)rc & 15) == 0)),
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
                    ((s & 3) != 0 || (((zc_batch_stride | qc_batch_stride) & 3) == 0 && (((uintptr_t)zc | (uintptr_t)qc) & 15) == 0)),
                "ogc_gru_blend: batch strides / alignment of the gates");
    OGC_REQUIRE(b <= 65535 && c <= 65535 && (long long)b * zc_batch_stride < (1ll << 31) && (long long)b * qc_batch_stride < (1ll << 31) &&
                    (long long)b * h_batch_stride < (1ll << 31), "ogc_gru_blend: exceeds 32-bit indexing");
    hipLaunchKernelGGL(gru_blend_kernel, dim3(ogc_divup(n, 256), c, b), dim3(256), 0, (hipStream_t)stream, c, n, s, zc,
                       zc_batch_stride, qc, qc_batch_stride, h, h_batch_stride, out);
    OGC_CHECK_LAUNCH("ogc_gru_blend");
    return OGC_OK;
}

extern "C" int ogc_soft_corr_flow(int b, int n1, int n2, int c, float support, const float *epsilon, const float *pc1,
                                  const float *pc2, const float *f1, const float *f2, float *flow, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n1 >= 0 && n2 >= 0 && c >= 0, "ogc_soft_corr_flow: negative dimension");
    OGC_REQUIRE(c <= SC_MAXC && (c & 3) == 0, "ogc_soft_corr_flow: feature width must be a multiple of 4 up to %d (got %d)", SC_MAXC, c);
    if (b == 0 || n1 == 0) return OGC_OK;
    OGC_REQUIRE(epsilon && pc1 && flow && (n2 == 0 || pc2) && (c == 0 || (f1 && (n2 == 0 || f2))), "ogc_soft_corr_flow: null pointer");
    OGC_REQUIRE(b <= 65535 && (long long)b * (c > 3 ? c : 3) * (n1 > n2 ? n1 : n2) < (1ll << 31), "ogc_soft_corr_flow: exceeds 32-bit indexing");
    hipLaunchKernelGGL(soft_corr_flow_kernel, dim3(ogc_divup(n1, SC_ROWS), b), dim3(256), 0, (hipStream_t)stream, n1, n2, c, support,
                       epsilon, pc1, pc2, f1, f2, flow);
    OGC_CHECK_LAUNCH("ogc_soft_corr_flow");
    return OGC_OK;
}

extern "C" int ogc_three_nn_weights(int b, int n, int mode, const float *dist2, float *weight, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0, "ogc_three_nn_weights: negative dimension");
    OGC_REQUIRE(mode == 0 || mode == 1, "ogc_three_nn_weights: mode 0 (clamp at 1e-10) or 1 (+ 1e-8), got %d", mode);
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(dist2 && weight, "ogc_three_nn_weights: null pointer");
    const long long rows = (long long)b * n;
    OGC_REQUIRE(rows * 3 < (1ll << 31), "ogc_three_nn_weights: exceeds 32-bit indexing");
    hipLaunchKernelGGL(three_nn_weights_kernel, dim3(ogc_divup(rows, 256)), dim3(256), 0, (hipStream_t)stream, rows, mode, dist2, weight);
    OGC_CHECK_LAUNCH("ogc_three_nn_weights");
    return OGC_OK;
}
