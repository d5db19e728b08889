// loss_glue.hip — the scalar algebra either side of the fused loss kernels, one launch per direction.
//
// Reference: losses/seg_loss_unsup.py:353-392 (UnsupervisedOGCLoss.forward): every term is a mean over the points of a view, the
// means are summed over views and weighted.  As framework operators on the stacked views that is a mean per term, a concatenation
// and a dot product forward, and per term a broadcast division backward; the segmentation masks feed five consumers (the
// dynamic term, the two neighbour terms, and the two halves of the invariance term), whose gradients autograd adds pairwise —
// four additions and two zero-filled embeddings of the half-batch slices.  Twenty launches of a few microseconds each on the
// step's main queue, where every launch also costs its dispatch gap.  Here:
//     ogc_view_means       v[k] = mean of row k, over up to eight dense fp32 tensors viewed as (rows_i, len_i)
//     ogc_view_means_grad  the adjoint of the means folded with the weighted sum: grad_i[r, :] = (weight[k(i, r)] * g) * (1 / len_i)
//     ogc_sum_ranges       out[e] = sum of the parts that cover element e (parts are dense runs [first_i, first_i + count_i))
// Deterministic: a row is summed by one workgroup in a fixed tree (double accumulators); ranges are added in part order.
#include "ogc_common.h"

namespace {

constexpr int LG_PARTS = 8;
constexpr int LG_THREADS = 1024;

struct MeanParts {
    const float *src[LG_PARTS];
    long long len[LG_PARTS];
    int first_row[LG_PARTS + 1]; // prefix sums of rows
};

__global__ __launch_bounds__(LG_THREADS) void view_means_kernel(MeanParts p, int parts, float *__restrict__ out) {
    __shared__ double red[LG_THREADS / 64];
    const int k = blockIdx.x;
    int i = 0;
    while (i + 1 < parts && k >= p.first_row[i + 1]) ++i;
    const long long len = p.len[i];
    const float *row = p.src[i] + (size_t)(k - p.first_row[i]) * len;
    double s = 0.0;
    if ((len & 3) == 0 && ((uintptr_t)row & 15) == 0) {
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
        for (long long j = threadIdx.x; j < (len >> 2); j += LG_THREADS) {
            const float4 v = r4[j];
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
    } else {
        for (long long j = threadIdx.x; j < len; j += LG_THREADS) s += (double)row[j];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < LG_THREADS / 64; ++w) t += red[w];
        out[k] = (float)(t / (double)len);
    }
}

struct GradParts {
    float *dst[LG_PARTS];
    long long len[LG_PARTS];
    int first_row[LG_PARTS + 1];
    int first_block[LG_PARTS + 1]; // prefix sums of ceil(rows_i len_i / (4 LG_THREADS))
};

// MeanBackward's arithmetic — (weight * g) * (1 / len), the framework's division by a scalar, one rounding each — so that the
// gradients are the bits the framework formed
__global__ __launch_bounds__(LG_THREADS) void view_means_grad_kernel(GradParts p, int parts, const float *__restrict__ weight,
                                                                     const float *__restrict__ g_loss) {
    int i = 0;
    while (i + 1 < parts && (int)blockIdx.x >= p.first_block[i + 1]) ++i;
    const long long len = p.len[i];
    const long long total = (long long)(p.first_row[i + 1] - p.first_row[i]) * len;
    const long long e = ((long long)(blockIdx.x - p.first_block[i]) * LG_THREADS + threadIdx.x) * 4;
    if (e >= total) return;
    const float g = g_loss[0];
    float *dst = p.dst[i];
    const float inv = __fdiv_rn(1.0f, (float)len);
    if ((len & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
        const float v = __fmul_rn(__fmul_rn(weight[p.first_row[i] + (int)(e / len)], g), inv);
        *reinterpret_cast<float4 *>(dst + e) = make_float4(v, v, v, v);
    } else {
        for (long long j = e; j < e + 4 && j < total; ++j)
            dst[j] = __fmul_rn(__fmul_rn(weight[p.first_row[i] + (int)(j / len)], g), inv);
    }
}

struct SumParts {
    const float *src[LG_PARTS];
    long long first[LG_PARTS];
    long long count[LG_PARTS];
};

template <bool VEC>
__global__ __launch_bounds__(256) void sum_ranges_kernel(SumParts p, int parts, long long total, float *__restrict__ out) {
    const long long stride = (long long)gridDim.x * 256 * (VEC ? 4 : 1);
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1); e < total; e += stride) {
        if constexpr (VEC) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            bool any = false;
#pragma unroll
            for (int i = 0; i < LG_PARTS; ++i) {
                if (i < parts && e >= p.first[i] && e < p.first[i] + p.count[i]) {
                    const float4 v = *reinterpret_cast<const float4 *>(p.src[i] + (e - p.first[i]));
                    a = any ? make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w) : v;
                    any = true;
                }
            }
            *reinterpret_cast<float4 *>(out + e) = a;
        } else {
            float a = 0.f;
            bool any = false;
            for (int i = 0; i < parts; ++i) {
                if (e >= p.first[i] && e < p.first[i] + p.count[i]) {
                    const float v = p.src[i][e - p.first[i]];
                    a = any ? a + v : v;
                    any = true;
                }
            }
            out[e] = a;
        }
    }
}

} // namespace

extern "C" int ogc_view_means(int parts, const float *const *src, const int *rows, const long long *len, float *out,
                              ogc_stream_t stream) {
    OGC_REQUIRE(parts >= 0 && parts <= LG_PARTS, "ogc_view_means: between 0 and 8 tensors");
    if (parts == 0) return OGC_OK;
    OGC_REQUIRE(src && rows && len && out, "ogc_view_means: null pointer");
    MeanParts p;
    p.first_row[0] = 0;
    for (int i = 0; i < parts; ++i) {
        OGC_REQUIRE(src[i] && rows[i] >= 1 && len[i] >= 1 && rows[i] <= 65536, "ogc_view_means: tensor %d is missing or empty", i);
        p.src[i] = src[i];
        p.len[i] = len[i];
        p.first_row[i + 1] = p.first_row[i] + rows[i];
    }
    hipLaunchKernelGGL(view_means_kernel, dim3(p.first_row[parts]), dim3(LG_THREADS), 0, (hipStream_t)stream, p, parts, out);
    OGC_CHECK_LAUNCH("ogc_view_means");
    return OGC_OK;
}

extern "C" int ogc_view_means_grad(int parts, float *const *grad, const int *rows, const long long *len, const float *weight,
                                   const float *g_loss, ogc_stream_t stream) {
    OGC_REQUIRE(parts >= 0 && parts <= LG_PARTS, "ogc_view_means_grad: between 0 and 8 tensors");
    if (parts == 0) return OGC_OK;
    OGC_REQUIRE(grad && rows && len && weight && g_loss, "ogc_view_means_grad: null pointer");
    GradParts p;
    p.first_row[0] = p.first_block[0] = 0;
    for (int i = 0; i < parts; ++i) {
        OGC_REQUIRE(grad[i] && rows[i] >= 1 && len[i] >= 1 && rows[i] <= 65536, "ogc_view_means_grad: tensor %d is missing or empty", i);
        const long long blocks = ((long long)rows[i] * len[i] + 4 * LG_THREADS - 1) / (4 * LG_THREADS);
        OGC_REQUIRE(p.first_block[i] + blocks < (1ll << 30), "ogc_view_means_grad: too many elements");
        p.dst[i] = grad[i];
        p.len[i] = len[i];
        p.first_row[i + 1] = p.first_row[i] + rows[i];
        p.first_block[i + 1] = p.first_block[i] + (int)blocks;
    }
    hipLaunchKernelGGL(view_means_grad_kernel, dim3(p.first_block[parts]), dim3(LG_THREADS), 0, (hipStream_t)stream, p, parts,
                       weight, g_loss);
    OGC_CHECK_LAUNCH("ogc_view_means_grad");
    return OGC_OK;
}

extern "C" int ogc_sum_ranges(int parts, const float *const *src, const long long *first, const long long *count, long long total,
                              float *out, ogc_stream_t stream) {
    OGC_REQUIRE(parts >= 0 && parts <= LG_PARTS && total >= 0, "ogc_sum_ranges: between 0 and 8 parts, total >= 0");
    if (total == 0) return OGC_OK;
    OGC_REQUIRE(out && (parts == 0 || (src && first && count)), "ogc_sum_ranges: null pointer");
    SumParts p;
    bool vec = (total & 3) == 0 && ((uintptr_t)out & 15) == 0;
    for (int i = 0; i < parts; ++i) {
        OGC_REQUIRE(src[i] && first[i] >= 0 && count[i] >= 0 && first[i] + count[i] <= total, "ogc_sum_ranges: part %d out of range", i);
        p.src[i] = src[i];
        p.first[i] = first[i];
        p.count[i] = count[i];
        vec = vec && (first[i] & 3) == 0 && (count[i] & 3) == 0 && ((uintptr_t)src[i] & 15) == 0;
    }
    const long long items = vec ? total / 4 : total;
    long long blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (vec) hipLaunchKernelGGL(sum_ranges_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, parts, total, out);
    else hipLaunchKernelGGL(sum_ranges_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, parts, total, out);
    OGC_CHECK_LAUNCH("ogc_sum_ranges");
    return OGC_OK;
}
