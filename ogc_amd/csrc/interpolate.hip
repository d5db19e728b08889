// interpolate.hip — three-point weighted interpolation and its gradient.
//
// Replaces three_interpolate_kernel_fast / three_interpolate_grad_kernel_fast
//   (reference: pointnet2/src/interpolate_gpu.cu:149-169, :192-214).
// A thread owns one target point: it loads the point's three indices and weights once and walks a
// slab of channels (the reference re-reads idx/weight for every channel through blockIdx.y).
// Forward evaluation order is the reference's: (w0*p0 + w1*p1) + w2*p2, no FMA contraction.
#include "ogc_common.h"

namespace {

constexpr int TI_THREADS = 256;

__global__ __launch_bounds__(TI_THREADS) void three_interp_fwd_kernel(int c, int m, int n, int ch_per_block,
                                                                      const float *__restrict__ points,
                                                                      const int *__restrict__ idx,
                                                                      const float *__restrict__ weight,
                                                                      float *__restrict__ out) {
    const int b = blockIdx.z;
    const int i = blockIdx.x * TI_THREADS + threadIdx.x;
    if (i >= n) return;
    const int c0 = blockIdx.y * ch_per_block;
    const int c1 = min(c, c0 + ch_per_block);
    const int *id = idx + ((size_t)b * n + i) * 3;
    const float *w = weight + ((size_t)b * n + i) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *p = points + ((size_t)b * c + c0) * m;
    float *o = out + ((size_t)b * c + c0) * n + i;
    for (int ch = c0; ch < c1; ++ch, p += m, o += n) {
        const float t0 = __fmul_rn(w0, p[i0]);
        const float t1 = __fmul_rn(w1, p[i1]);
        const float t2 = __fmul_rn(w2, p[i2]);
        *o = __fadd_rn(__fadd_rn(t0, t1), t2);
    }
}

__global__ __launch_bounds__(TI_THREADS) void three_interp_bwd_kernel(int c, int n, int m, int ch_per_block,
                                                                      const float *__restrict__ grad_out,
                                                                      const int *__restrict__ idx,
                                                                      const float *__restrict__ weight,
                                                                      float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int i = blockIdx.x * TI_THREADS + threadIdx.x;
    if (i >= n) return;
    const int c0 = blockIdx.y * ch_per_block;
    const int c1 = min(c, c0 + ch_per_block);
    const int *id = idx + ((size_t)b * n + i) * 3;
    const float *w = weight + ((size_t)b * n + i) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *g = grad_out + ((size_t)b * c + c0) * n + i;
    float *gp = grad_points + ((size_t)b * c + c0) * m;
    for (int ch = c0; ch < c1; ++ch, g += n, gp += m) {
        const float go = *g;
        unsafeAtomicAdd(gp + i0, __fmul_rn(go, w0));
        unsafeAtomicAdd(gp + i1, __fmul_rn(go, w1));
        unsafeAtomicAdd(gp + i2, __fmul_rn(go, w2));
    }
}

// LDS-privatised variant of the backward: a workgroup owns CC channels x a slice of the n target points,
// accumulates the three weighted contributions per point into an LDS image acc[CC][m] (ds_add_f32) and merges the
// non-zero entries into grad_points with one global atomic each (cf. group_bwd_lds_kernel).
template <int CC>
__global__ __launch_bounds__(TI_THREADS) void three_interp_bwd_lds_kernel(int c, int n, int m, int n_per_block,
                                                                          const float *__restrict__ grad_out,
                                                                          const int *__restrict__ idx,
                                                                          const float *__restrict__ weight,
                                                                          float *__restrict__ grad_points) {
    extern __shared__ __attribute__((aligned(16))) float ti_acc[]; // [CC][m]
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CC;
    const int ncc = min(CC, c - c0);
    const int i_begin = blockIdx.x * n_per_block;
    const int i_end = min(n, i_begin + n_per_block);
    for (int t = threadIdx.x; t < CC * m; t += TI_THREADS) ti_acc[t] = 0.0f;
    __syncthreads();
    const float *g = grad_out + ((size_t)b * c + c0) * n;
    for (int i = i_begin + threadIdx.x; i < i_end; i += TI_THREADS) {
        const int *id = idx + ((size_t)b * n + i) * 3;
        const float *w = weight + ((size_t)b * n + i) * 3;
        const int i0 = id[0], i1 = id[1], i2 = id[2];
        const float w0 = w[0], w1 = w[1], w2 = w[2];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            if (cc < ncc) {
                const float go = g[(size_t)cc * n + i];
                float *a = ti_acc + cc * m;
                atomicAdd(a + i0, __fmul_rn(go, w0));
                atomicAdd(a + i1, __fmul_rn(go, w1));
                atomicAdd(a + i2, __fmul_rn(go, w2));
            }
        }
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)b * c + c0) * m;
    for (int t = threadIdx.x; t < ncc * m; t += TI_THREADS) {
        const float v = ti_acc[t];
        if (v != 0.0f) unsafeAtomicAdd(gp + t, v);
    }
}

int ti_ch_per_block(int b, int c, int blocks_x) {
    int cpb = 1;
    while (cpb < c && (long long)b * blocks_x * ((c + 2 * cpb - 1) / (2 * cpb)) >= 2048) cpb *= 2;
    return cpb;
}

} // namespace

extern "C" int ogc_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                     const float *weight, float *out, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "ogc_three_interpolate: negative dimension");
    if (b == 0 || c == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(points && idx && weight && out, "ogc_three_interpolate: null pointer");
    OGC_REQUIRE((long long)b * c * n < (1ll << 31) && (long long)b * c * m < (1ll << 31),
                "ogc_three_interpolate: tensor exceeds 32-bit indexing");
    const int bx = ogc_divup(n, TI_THREADS);
    const int cpb = ti_ch_per_block(b, c, bx);
    dim3 grid(bx, ogc_divup(c, cpb), b);
    hipLaunchKernelGGL(three_interp_fwd_kernel, grid, dim3(TI_THREADS), 0, (hipStream_t)stream, c, m, n, cpb,
                       points, idx, weight, out);
    OGC_CHECK_LAUNCH("ogc_three_interpolate");
    return OGC_OK;
}

extern "C" int ogc_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                          const float *weight, float *grad_points, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 0 && m >= 0 && n >= 0, "ogc_three_interpolate_grad: negative dimension");
    if (b == 0 || c == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(grad_out && idx && weight && grad_points, "ogc_three_interpolate_grad: null pointer");
    OGC_REQUIRE((long long)b * c * n < (1ll << 31) && (long long)b * c * m < (1ll << 31),
                "ogc_three_interpolate_grad: tensor exceeds 32-bit indexing");
    if (ogc_deterministic()) // the three products of a target in position order, every sum ascending (det.hip)
        return ogc_det_scatter_add("ogc_three_interpolate_grad", b, c, m, 3ll * n, idx, grad_out, (long long)c * n, weight, 1,
                                   grad_points, 1, (hipStream_t)stream);
    if (m <= 16384 && n >= 1024) { // LDS-privatised path: CC channel images of m floats within 64 KiB
        int cc = 4096 / m;   // 16 KiB images (one channel when m > 4096): more workgroups per CU, see group_bwd
        if (cc < 1) cc = 1;
        cc = cc >= 8 ? 8 : (cc >= 4 ? 4 : (cc >= 2 ? 2 : 1));
        while (cc > 1 && cc / 2 >= c) cc /= 2;
        const int chunks = ogc_divup(c, cc);
        int splits = 1;
        while ((long long)b * chunks * splits < 512 && n / (splits * 2) >= 2048) splits *= 2;
        const int npb = ogc_divup(n, splits);
        dim3 grid(ogc_divup(n, npb), chunks, b);
        const size_t lds = (size_t)cc * m * sizeof(float);
#define TI_LAUNCH(CCV)                                                                                          \
    hipLaunchKernelGGL(three_interp_bwd_lds_kernel<CCV>, grid, dim3(TI_THREADS), lds, (hipStream_t)stream, c, n, m, \
                       npb, grad_out, idx, weight, grad_points)
        if (cc == 8) TI_LAUNCH(8);
        else if (cc == 4) TI_LAUNCH(4);
        else if (cc == 2) TI_LAUNCH(2);
        else TI_LAUNCH(1);
#undef TI_LAUNCH
        OGC_CHECK_LAUNCH("ogc_three_interpolate_grad");
        return OGC_OK;
    }
    const int bx = ogc_divup(n, TI_THREADS);
    const int cpb = ti_ch_per_block(b, c, bx);
    dim3 grid(bx, ogc_divup(c, cpb), b);
    hipLaunchKernelGGL(three_interp_bwd_kernel, grid, dim3(TI_THREADS), 0, (hipStream_t)stream, c, n, m, cpb,
                       grad_out, idx, weight, grad_points);
    OGC_CHECK_LAUNCH("ogc_three_interpolate_grad");
    return OGC_OK;
}
