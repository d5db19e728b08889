// group_norm.hip — fused GroupNorm (+ ReLU) forward and backward for the per-point MLPs.
//
// Replaces the Conv2d -> nn.GroupNorm(4) -> ReLU(inplace) tail of every SharedMLP layer of the segmentation nets
// (reference: utils/nn_util.py:6-11, :45-85; models/segnet_kitti.py:8 BN_CONFIG).  PyTorch's GroupNorm forward
// reduces each (sample, group) row — here 64 rows of ~1M elements — with ONE workgroup per row
// (RowwiseMomentsCUDAKernel: 0.74 ms per layer, 22 % of the training step on MI355X); ReLU and its backward are
// separate full passes.  Here:
//   forward : stats  (many workgroups per row, fp64 partials)            1 read
//             apply  y = relu(a_c * x + b_c)                             1 read + 1 write
//   backward: sums   ds = sum dy'*x, db = sum dy' per (sample, channel)  2 reads       (dy' = dy * [y > 0])
//             params c2, c3 per (sample, group); dgamma, dbeta           tiny
//             dx     = dy' * gamma_c * rstd + c2 * x + c3                2 reads + 1 write
// x is (B, C, HW) fp32 contiguous; the channels of a group are contiguous, so a (sample, group) row is one
// contiguous run of (C/G)*HW floats.  Statistics: biased variance, rstd = 1/sqrt(var + eps), as nn.GroupNorm.
#include "ogc_common.h"
#include "act_io.h"

namespace {

constexpr int GN_THREADS = 256;

// block-wide sum of two doubles; result valid in thread 0
__device__ __forceinline__ void gn_block_sum2(double &a, double &b, double *smem /* [2*GN_THREADS/64] */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        smem[wave * 2] = a;
        smem[wave * 2 + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.0;
        b = 0.0;
        for (int w = 0; w < GN_THREADS / 64; ++w) {
            a += smem[w * 2];
            b += smem[w * 2 + 1];
        }
    }
}

// ---- forward ------------------------------------------------------------------------------------------------
// grid (chunks, B*G): sum / sum of squares of one slice of a row -> ws[chunk][row][0..1] (fp64).  Every slot is written
// exactly once, so the buffer needs no clearing and no atomics; the consumers add the `slots` = chunks partial sums.
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(long long row_len, long long chunk_len,
                                                              const float *__restrict__ x, double *__restrict__ ws) {
    __shared__ double smem[2 * GN_THREADS / 64];
    const long long row = blockIdx.y;
    const long long begin = (long long)blockIdx.x * chunk_len;
    const long long end = min(row_len, begin + chunk_len);
    const float *p = x + row * row_len;
    double s = 0.0, ss = 0.0;
    if (((row_len | chunk_len) & 3) == 0 && ((uintptr_t)p & 15) == 0) {
        for (long long i = begin + threadIdx.x * 4; i < end; i += GN_THREADS * 4 * 4) {
            float fs = 0.0f, fss = 0.0f; // fp32 over <= 16 elements, then fp64
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long j = i + (long long)u * GN_THREADS * 4;
                if (j < end) {
                    const float4 v = *reinterpret_cast<const float4 *>(p + j);
                    fs += (v.x + v.y) + (v.z + v.w);
                    fss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
            s += fs;
            ss += fss;
        }
    } else {
        for (long long i = begin + threadIdx.x; i < end; i += GN_THREADS) {
            const float v = p[i];
            s += v;
            ss += (double)v * v;
        }
    }
    gn_block_sum2(s, ss, smem);
    if (threadIdx.x == 0) {
        double *dst = ws + ((size_t)blockIdx.x * gridDim.y + row) * 2;
        dst[0] = s;
        dst[1] = ss;
    }
}

// grid (chunks over HW, C, B): y = act(a_c * x + b_c).  The first chunk of the first channel of a group also
// publishes mean / rstd for the backward pass.
// Sum of the `slots` copies of a (sum, sum of squares) accumulator, in slot order (the order every consumer must share: the
// results are compared bit for bit), with the loads of eight slots in flight at a time instead of a load -> add chain per slot.
__device__ __forceinline__ void gn_sum_slots(const double *__restrict__ first, size_t stride, int slots, double &sum,
                                             double &sumsq) {
    for (int s0 = 0; s0 < slots; s0 += 8) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = *reinterpret_cast<const double2 *>(first + (size_t)min(s0 + u, slots - 1) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < slots) {
                sum += v[u].x;
                sumsq += v[u].y;
            }
    }
}

template <bool RELU>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(int c, int hw, int groups, float eps,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const double *__restrict__ ws, int slots,
                                                              float *__restrict__ y,
                                                              float *__restrict__ mean_out,
                                                              float *__restrict__ rstd_out) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int cg = c / groups, g = ch / cg;
    const int row = b * groups + g;
    const double n = (double)cg * hw;
    double sum = 0.0, sumsq = 0.0; // `slots` copies of the accumulator (1 from gn_stats_kernel, more from the fused conv)
    gn_sum_slots(ws + (size_t)row * 2, (size_t)gridDim.z * groups * 2, slots, sum, sumsq);
    const double m = sum / n;
    const double var = fmax(sumsq / n - m * m, 0.0);
    const float mean = (float)m;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && ch == g * cg && threadIdx.x == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
    const float a = rstd * gamma[ch];
    const float bb = beta[ch] - mean * a;
    const size_t base = ((size_t)b * c + ch) * hw;
    const float *px = x + base;
    float *py = y + base;
    if ((hw & 3) == 0 && (((uintptr_t)px | (uintptr_t)py) & 15) == 0) {
        for (int i = (blockIdx.x * GN_THREADS + threadIdx.x) * 4; i < hw; i += gridDim.x * GN_THREADS * 4) {
            float4 v = *reinterpret_cast<const float4 *>(px + i);
            v.x = fmaf(a, v.x, bb); v.y = fmaf(a, v.y, bb); v.z = fmaf(a, v.z, bb); v.w = fmaf(a, v.w, bb);
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(py + i) = v;
        }
    } else {
        for (int i = blockIdx.x * GN_THREADS + threadIdx.x; i < hw; i += gridDim.x * GN_THREADS) {
            float v = fmaf(a, px[i], bb);
            py[i] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
}

// ---- backward -----------------------------------------------------------------------------------------------
// grid (chunks, C, B): ds = sum dy'*x, db = sum dy' of one slice of a channel -> dsdb[chunk][b*c][2] (fp64, every slot
// written once: no clearing, no atomics)
// AT: element type of x, dy (and dx below): float / ogc_bf16 (act_io.h)
template <bool RELU, typename AT = float>
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_sums_kernel(int c, int hw, int groups,
                                                                 const AT *__restrict__ x,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ rstd,
                                                                 const AT *__restrict__ dy,
                                                                 double *__restrict__ dsdb) {
    __shared__ double smem[2 * GN_THREADS / 64];
    const int b = blockIdx.z, ch = blockIdx.y;
    const int cg = c / groups, row = b * groups + ch / cg;
    const float a = rstd[row] * gamma[ch];
    const float bb = beta[ch] - mean[row] * a;
    const size_t base = ((size_t)b * c + ch) * hw;
    const AT *px = x + base, *pd = dy + base;
    double s = 0.0, sb = 0.0;
    if ((hw & 3) == 0 && (((uintptr_t)px | (uintptr_t)pd) & ogc_act_mask<AT>()) == 0) {
        for (int i = (blockIdx.x * GN_THREADS + threadIdx.x) * 4; i < hw; i += gridDim.x * GN_THREADS * 4) {
            const float4 v = ogc_ld4(px + i);
            float4 d = ogc_ld4(pd + i);
            if (RELU) {
                d.x = fmaf(a, v.x, bb) > 0.f ? d.x : 0.f; d.y = fmaf(a, v.y, bb) > 0.f ? d.y : 0.f;
                d.z = fmaf(a, v.z, bb) > 0.f ? d.z : 0.f; d.w = fmaf(a, v.w, bb) > 0.f ? d.w : 0.f;
            }
            s += (double)((d.x * v.x + d.y * v.y) + (d.z * v.z + d.w * v.w));
            sb += (double)((d.x + d.y) + (d.z + d.w));
        }
    } else {
        for (int i = blockIdx.x * GN_THREADS + threadIdx.x; i < hw; i += gridDim.x * GN_THREADS) {
            const float v = ogc_ld1(px + i);
            float d = ogc_ld1(pd + i);
            if (RELU) d = fmaf(a, v, bb) > 0.f ? d : 0.f;
            s += (double)d * v;
            sb += d;
        }
    }
    gn_block_sum2(s, sb, smem);
    if (threadIdx.x == 0) {
        double *dst = dsdb + (((size_t)blockIdx.x * gridDim.z + b) * c + ch) * 2;
        dst[0] = s;
        dst[1] = sb;
    }
}

// Prologue of the dx kernels (block (chunk, ch, b)): from the partial sums of gn_*_bwd_sums_kernel
//   c2, c3 of the block's (b, group):  dx = a * dy' + c2 * x + c3   (returned to every thread), and
//   dgamma[ch], dbeta[ch] — summed over the batch by the block (0, ch, 0).
// This used to be a launch of its own between the two passes; the few hundred fp64 values it reads per block are
// nothing next to the block's share of the activation.
__device__ __forceinline__ void gn_bwd_prologue(int c, int groups, double n_group, int slots,
                                                const float *__restrict__ gamma, const float *__restrict__ mean,
                                                const float *__restrict__ rstd, const double *__restrict__ dsdb,
                                                float *__restrict__ dgamma, float *__restrict__ dbeta, float &c2,
                                                float &c3) {
    __shared__ double red[2 * GN_THREADS / 64];
    __shared__ float coef[2];
    const int b = blockIdx.z, ch = blockIdx.y, nb = gridDim.z;
    const int cg = c / groups, g = ch / cg, row = b * groups + g;
    double dsg = 0.0, dbg = 0.0;
    for (int e = threadIdx.x; e < cg * slots; e += GN_THREADS) {
        const int k = g * cg + e % cg, sl = e / cg;
        const double *src = dsdb + (((size_t)sl * nb + b) * c + k) * 2;
        const double gk = (double)gamma[k];
        dsg += src[0] * gk;
        dbg += src[1] * gk;
    }
    gn_block_sum2(dsg, dbg, red);
    if (threadIdx.x == 0) {
        const double m = mean[row], r = rstd[row];
        const double v2 = (dbg * m - dsg) * r * r * r / n_group;
        coef[0] = (float)v2;
        coef[1] = (float)(-v2 * m - dbg * r / n_group);
    }
    __syncthreads();
    c2 = coef[0];
    c3 = coef[1];
    if (blockIdx.x == 0 && b == 0) { // uniform per block
        __syncthreads();
        double dg = 0.0, dbt = 0.0;
        for (int e = threadIdx.x; e < nb * slots; e += GN_THREADS) {
            const int bi = e % nb, sl = e / nb;
            const double *src = dsdb + (((size_t)sl * nb + bi) * c + ch) * 2;
            const int r2 = bi * groups + g;
            dg += (src[0] - (double)mean[r2] * src[1]) * (double)rstd[r2];
            dbt += src[1];
        }
        gn_block_sum2(dg, dbt, red);
        if (threadIdx.x == 0) {
            dgamma[ch] = (float)dg;
            dbeta[ch] = (float)dbt;
        }
    }
}

template <bool RELU, typename AT = float>
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_dx_kernel(int c, int hw, int groups,
                                                               const AT *__restrict__ x,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ rstd,
                                                               const double *__restrict__ dsdb, int slots,
                                                               const AT *__restrict__ dy, AT *__restrict__ dx,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int cg = c / groups, row = b * groups + ch / cg;
    const float r = rstd[row];
    const float a = r * gamma[ch];
    const float bb = beta[ch] - mean[row] * a;
    float c2, c3;
    const size_t base = ((size_t)b * c + ch) * hw;
    const AT *px = x + base, *pd = dy + base;
    AT *po = dx + base;
    const bool vec = (hw & 3) == 0 && (((uintptr_t)px | (uintptr_t)pd | (uintptr_t)po) & ogc_act_mask<AT>()) == 0;
    // the first pieces of x and dy are requested BEFORE the prologue (a few hundred fp64 partial sums and two block reductions:
    // several microseconds of a block that lives ~20): their latency runs underneath it
    const int i0 = (blockIdx.x * GN_THREADS + threadIdx.x) * 4;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), d0 = v0;
    if (vec && i0 < hw) {
        v0 = ogc_ld4(px + i0);
        d0 = ogc_ld4(pd + i0);
    }
    gn_bwd_prologue(c, groups, (double)cg * hw, slots, gamma, mean, rstd, dsdb, dgamma, dbeta, c2, c3);
    if (vec) {
        for (int i = i0; i < hw; i += gridDim.x * GN_THREADS * 4) {
            const float4 v = i == i0 ? v0 : ogc_ld4(px + i);
            float4 d = i == i0 ? d0 : ogc_ld4(pd + i);
            if (RELU) {
                d.x = fmaf(a, v.x, bb) > 0.f ? d.x : 0.f; d.y = fmaf(a, v.y, bb) > 0.f ? d.y : 0.f;
                d.z = fmaf(a, v.z, bb) > 0.f ? d.z : 0.f; d.w = fmaf(a, v.w, bb) > 0.f ? d.w : 0.f;
            }
            float4 o;
            o.x = fmaf(a, d.x, fmaf(c2, v.x, c3)); o.y = fmaf(a, d.y, fmaf(c2, v.y, c3));
            o.z = fmaf(a, d.z, fmaf(c2, v.z, c3)); o.w = fmaf(a, d.w, fmaf(c2, v.w, c3));
            ogc_st4(po + i, o);
        }
    } else {
        if constexpr (sizeof(AT) == 4) { // (16-bit tensors: the entry point insists on the vector path)
            for (int i = blockIdx.x * GN_THREADS + threadIdx.x; i < hw; i += gridDim.x * GN_THREADS) {
                const float v = px[i];
                float d = pd[i];
                if (RELU) d = fmaf(a, v, bb) > 0.f ? d : 0.f;
                po[i] = fmaf(a, d, fmaf(c2, v, c3));
            }
        }
    }
}

// ---- GroupNorm + ReLU + max over the neighbourhood (last layer of a set-abstraction MLP) ----------------------
// x (B, C, P, S) -> out (B, C, P) = max_s act(a_c x + b_c), argmax (B, C, P).  The normalised activation is never
// written: this removes one full write, the separate max-reduction pass (utils/pointnet2_util.py:39-42 max_pool2d)
// and, in the backward pass, the dense gradient of the max.  L = S/4 lanes share a row (one float4 each, fully
// coalesced); the row maximum is reduced with xor-shuffles inside the L-lane group (first index wins ties).
template <bool RELU>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_maxpool_kernel(int c, int p, int s, int groups, float eps,
                                                                      const float *__restrict__ x,
                                                                      const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta,
                                                                      const double *__restrict__ ws, int slots,
                                                                      float *__restrict__ out, int *__restrict__ arg,
                                                                      float *__restrict__ mean_out,
                                                                      float *__restrict__ rstd_out) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int cg = c / groups, g = ch / cg, row = b * groups + g;
    const double n = (double)cg * p * s;
    double sum = 0.0, sumsq = 0.0;
    gn_sum_slots(ws + (size_t)row * 2, (size_t)gridDim.z * groups * 2, slots, sum, sumsq);
    const double m = sum / n;
    const double var = fmax(sumsq / n - m * m, 0.0);
    const float mean = (float)m;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && ch == g * cg && threadIdx.x == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
    const float a = rstd * gamma[ch];
    const float bb = beta[ch] - mean * a;
    const int L = s >> 2;                         // lanes per row (power of two, <= 64)
    const int rows_per_block = GN_THREADS / L;
    const int sub = threadIdx.x % L;
    const size_t base = ((size_t)b * c + ch) * p;
    for (int pr = blockIdx.x * rows_per_block + threadIdx.x / L; pr < p + (rows_per_block - 1);
         pr += gridDim.x * rows_per_block) { // uniform trip count for the shuffles; tail rows are clamped
        const int prc = min(pr, p - 1);
        const float4 v = *reinterpret_cast<const float4 *>(x + (base + prc) * s + sub * 4);
        float y0 = fmaf(a, v.x, bb), y1 = fmaf(a, v.y, bb), y2 = fmaf(a, v.z, bb), y3 = fmaf(a, v.w, bb);
        if (RELU) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); y2 = fmaxf(y2, 0.f); y3 = fmaxf(y3, 0.f); }
        float best = y0;
        int bi = sub * 4;
        if (y1 > best) { best = y1; bi = sub * 4 + 1; }
        if (y2 > best) { best = y2; bi = sub * 4 + 2; }
        if (y3 > best) { best = y3; bi = sub * 4 + 3; }
        for (int off = 1; off < L; off <<= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (sub == 0 && pr < p) {
            out[base + pr] = best;
            arg[base + pr] = bi;
        }
    }
}

// The same result from the extreme of x over each neighbourhood (written by the convolution that produced x,
// ogc_conv1x1_gemm_affine_pool: the largest x where gamma >= 0, the smallest where gamma < 0): y -> act(a * y + bb) is
// non-decreasing for a >= 0 and non-increasing for a < 0 — rounding included — so the maximum of the activation is the
// activation of that extreme, and the first neighbour attaining it is the one the convolution recorded.  One workgroup
// per (batch, channel); x is not read.
template <bool RELU>
__global__ __launch_bounds__(GN_THREADS) void gn_pool_extremes_kernel(int c, int p, int s, int groups, float eps,
                                                                      const float *__restrict__ yext,
                                                                      const int *__restrict__ aext,
                                                                      const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta,
                                                                      const double *__restrict__ ws, int slots,
                                                                      float *__restrict__ out, int *__restrict__ arg,
                                                                      float *__restrict__ mean_out,
                                                                      float *__restrict__ rstd_out) {
    const int b = blockIdx.y, ch = blockIdx.x;
    const int cg = c / groups, g = ch / cg, row = b * groups + g;
    const double n = (double)cg * p * s;
    double sum = 0.0, sumsq = 0.0;
    gn_sum_slots(ws + (size_t)row * 2, (size_t)gridDim.y * groups * 2, slots, sum, sumsq);
    const double m = sum / n;
    const double var = fmax(sumsq / n - m * m, 0.0);
    const float mean = (float)m;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (ch == g * cg && threadIdx.x == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
    const float a = rstd * gamma[ch];
    const float bb = beta[ch] - mean * a;
    const size_t base = ((size_t)b * c + ch) * p;
    for (int pr = threadIdx.x; pr < p; pr += GN_THREADS) {
        float y = fmaf(a, yext[base + pr], bb);
        if (RELU) y = fmaxf(y, 0.f);
        out[base + pr] = y;
        arg[base + pr] = a != 0.f ? aext[base + pr] : 0;   // a == 0: every neighbour gives bb, the first wins the tie
    }
}

// grid (chunks over P, C, B): ds, db from the sparse gradient (non-zero only at the arg-max element)
template <bool RELU, typename XT = float>
__global__ __launch_bounds__(GN_THREADS) void gn_maxpool_bwd_sums_kernel(int c, int p, int s,
                                                                         const XT *__restrict__ x,
                                                                         const float *__restrict__ out,
                                                                         const int *__restrict__ arg,
                                                                         const float *__restrict__ gout,
                                                                         const float *__restrict__ xext, // x at arg, or null
                                                                         const float *__restrict__ gamma,
                                                                         const float *__restrict__ rstd, int groups,
                                                                         double *__restrict__ dsdb) {
    __shared__ double smem[2 * GN_THREADS / 64];
    const int b = blockIdx.z, ch = blockIdx.y;
    // xext holds x at the neighbourhood's EXTREME; arg is that position unless the channel's scale is zero
    // (gn_pool_extremes_kernel: arg = 0 there, as the pass over the full tensor would have it): gather for such a channel
    if (xext && !(rstd[b * groups + ch / (c / groups)] * gamma[ch] != 0.f)) xext = nullptr;
    const size_t base = ((size_t)b * c + ch) * p;
    double ds = 0.0, db = 0.0;
    for (int pr = blockIdx.x * GN_THREADS + threadIdx.x; pr < p; pr += gridDim.x * GN_THREADS) {
        float g = gout[base + pr];
        if (RELU && !(out[base + pr] > 0.f)) g = 0.f;
        // (the gather costs a 32-byte sector per element: 50 us at C4's widths against 5 when the forward pass kept the values)
        ds += (double)g * (double)(xext ? xext[base + pr] : ogc_ld1(x + (base + pr) * s + arg[base + pr]));
        db += g;
    }
    gn_block_sum2(ds, db, smem);
    if (threadIdx.x == 0) {
        double *dst = dsdb + (((size_t)blockIdx.x * gridDim.z + b) * c + ch) * 2;
        dst[0] = ds;
        dst[1] = db;
    }
}

// dx = c2 * x + c3 everywhere, + a * g' at the arg-max element of each row
template <bool RELU>
__global__ __launch_bounds__(GN_THREADS) void gn_maxpool_bwd_dx_kernel(int c, int p, int s, int groups,
                                                                       const float *__restrict__ x,
                                                                       const float *__restrict__ gamma,
                                                                       const float *__restrict__ mean,
                                                                       const float *__restrict__ rstd,
                                                                       const double *__restrict__ dsdb, int slots,
                                                                       const float *__restrict__ out,
                                                                       const int *__restrict__ arg,
                                                                       const float *__restrict__ gout,
                                                                       float *__restrict__ dx,
                                                                       float *__restrict__ dgamma,
                                                                       float *__restrict__ dbeta) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int cg = c / groups, row = b * groups + ch / cg;
    const float a = rstd[row] * gamma[ch];
    float c2, c3;
    gn_bwd_prologue(c, groups, (double)cg * p * s, slots, gamma, mean, rstd, dsdb, dgamma, dbeta, c2, c3);
    const int L = s >> 2;
    const int rows_per_block = GN_THREADS / L;
    const int sub = threadIdx.x % L;
    const size_t base = ((size_t)b * c + ch) * p;
    for (int pr = blockIdx.x * rows_per_block + threadIdx.x / L; pr < p; pr += gridDim.x * rows_per_block) {
        const size_t off = (base + pr) * s + sub * 4;
        const float4 v = *reinterpret_cast<const float4 *>(x + off);
        float g = gout[base + pr];
        if (RELU && !(out[base + pr] > 0.f)) g = 0.f;
        const int rel = arg[base + pr] - sub * 4;
        const float ag = a * g;
        float4 o;
        o.x = fmaf(c2, v.x, c3) + (rel == 0 ? ag : 0.f);
        o.y = fmaf(c2, v.y, c3) + (rel == 1 ? ag : 0.f);
        o.z = fmaf(c2, v.z, c3) + (rel == 2 ? ag : 0.f);
        o.w = fmaf(c2, v.w, c3) + (rel == 3 ? ag : 0.f);
        *reinterpret_cast<float4 *>(dx + off) = o;
    }
}

// The same gradient in SPARSE form, for consumers that rebuild it while loading x (gn_fused_bwd.hip: the weight- and input-
// gradient kernels of the convolution that wrote x): dx[b, ch, pr, j] = fmaf(c2, x, c3) + (j == arg ? ag : 0) with
//   coef2[b, ch] = (c2, c3)   and   inj[b, ch, pr] = (ag, arg as bits)
// — 1/32 of dx's bytes instead of a pass that reads x and writes dx.
template <bool RELU>
__global__ __launch_bounds__(GN_THREADS) void gn_maxpool_bwd_sparse_kernel(int c, int p, int s, int groups,
                                                                           const float *__restrict__ gamma,
                                                                           const float *__restrict__ mean,
                                                                           const float *__restrict__ rstd,
                                                                           const double *__restrict__ dsdb, int slots,
                                                                           const float *__restrict__ out,
                                                                           const int *__restrict__ arg,
                                                                           const float *__restrict__ gout,
                                                                           float2 *__restrict__ coef2,
                                                                           float2 *__restrict__ inj,
                                                                           float *__restrict__ dgamma,
                                                                           float *__restrict__ dbeta) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int cg = c / groups, row = b * groups + ch / cg;
    const float a = rstd[row] * gamma[ch];
    float c2, c3;
    gn_bwd_prologue(c, groups, (double)cg * p * s, slots, gamma, mean, rstd, dsdb, dgamma, dbeta, c2, c3);
    const size_t base = ((size_t)b * c + ch) * p;
    for (int pr = blockIdx.x * GN_THREADS + threadIdx.x; pr < p; pr += gridDim.x * GN_THREADS) {
        float g = gout[base + pr];
        if (RELU && !(out[base + pr] > 0.f)) g = 0.f;
        inj[base + pr] = make_float2(a * g, __int_as_float(arg[base + pr]));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) coef2[(size_t)b * c + ch] = make_float2(c2, c3);
}

constexpr int GN_STATS_SLOTS = 32; // most partial sums per (b, group) the first pass writes  (ogc_group_norm_stats_slots)
constexpr int GN_BWD_SLOTS = 8;    // most partial sums per (b, channel) of the backward sums  (ogc_group_norm_bwd_slots)

int hw_chunks(int b, int c, int hw) {
    // enough workgroups to fill the chip (>= ~2048) without making them tiny (>= 4096 elements each)
    int chunks = 1;
    while ((long long)b * c * chunks < 2048 && hw / (chunks * 2) >= 4096 && chunks < GN_BWD_SLOTS) chunks *= 2;
    return chunks;
}

// First pass of a GroupNorm: partial (sum, sum of squares) of every (b, group) row into ws[slot][rows][2]; returns the
// number of slots written (<= GN_STATS_SLOTS).
int launch_gn_stats(int rows, long long row_len, const float *x, double *ws, hipStream_t s) {
    int chunks = 1;
    while ((long long)rows * chunks < 2048 && row_len / (chunks * 2) >= 16384 && chunks < GN_STATS_SLOTS) chunks *= 2;
    long long chunk_len = (row_len + chunks - 1) / chunks;
    chunk_len = (chunk_len + 3) / 4 * 4;
    const int slots = (int)ogc_divup(row_len, chunk_len);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(slots, rows), dim3(GN_THREADS), 0, s, row_len, chunk_len, x, ws);
    return slots;
}

} // namespace

namespace {
// stats == nullptr: first pass (sum, sum of squares per (batch, group), fp64) into ws, then apply.
// stats != nullptr: `slots` copies of that accumulator already exist (ogc_conv1x1_gemm_gnstats), apply only.
int gn_fwd_impl(const char *name, int b, int c, int hw, int groups, float eps, int relu, const float *x,
                const float *gamma, const float *beta, float *y, float *mean, float *rstd, double *ws,
                const double *stats, int slots, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && hw >= 1 && groups >= 1 && c % groups == 0, "%s: bad shape", name);
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(x && gamma && beta && y && mean && rstd && (ws || stats), "%s: null pointer", name);
    OGC_REQUIRE((long long)c * hw < (1ll << 31) && b <= 65535, "%s: one sample exceeds 32-bit indexing", name);
    hipStream_t s = (hipStream_t)stream;
    if (!stats) {
        const int rows = b * groups;
        const long long row_len = (long long)(c / groups) * hw;
        slots = launch_gn_stats(rows, row_len, x, ws, s);
        stats = ws;
    }
    dim3 grid(hw_chunks(b, c, hw), c, b);
    if (relu)
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(GN_THREADS), 0, s, c, hw, groups, eps, x, gamma, beta, stats,
                           slots, y, mean, rstd);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(GN_THREADS), 0, s, c, hw, groups, eps, x, gamma, beta,
                           stats, slots, y, mean, rstd);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

namespace {
// one thread per (b, channel): the affine map y -> a * y + bb that GroupNorm applies to channel ch of sample b
__global__ void gn_coeffs_kernel(int b, int c, int hw, int groups, float eps, const float *__restrict__ gamma,
                                 const float *__restrict__ beta, const double *__restrict__ stats, int slots,
                                 float *__restrict__ mean_out, float *__restrict__ rstd_out, float *__restrict__ a_out,
                                 float *__restrict__ bb_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * c) return;
    const int bi = t / c, ch = t % c;
    const int cg = c / groups, g = ch / cg, row = bi * groups + g;
    const double n = (double)cg * hw;
    double sum = 0.0, sumsq = 0.0;
    gn_sum_slots(stats + (size_t)row * 2, (size_t)b * groups * 2, slots, sum, sumsq);
    const double m = sum / n;
    const double var = fmax(sumsq / n - m * m, 0.0);
    const float mean = (float)m;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (ch == g * cg) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
    const float a = rstd * gamma[ch];
    a_out[t] = a;
    bb_out[t] = beta[ch] - mean * a;
}
} // namespace

// Statistics (unless supplied) -> mean, rstd per (b, group) and the per-(b, channel) affine coefficients, WITHOUT
// applying them: the consumer (ogc_conv1x1_gemm_affine / ogc_conv1x1_wgrad_affine) applies them while loading.
extern "C" int ogc_group_norm_coeffs(int b, int c, int hw, int groups, float eps, const float *x, const float *gamma,
                                     const float *beta, const double *stats, int slots, double *ws, float *mean,
                                     float *rstd, float *a, float *bb, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && hw >= 1 && groups >= 1 && c % groups == 0, "ogc_group_norm_coeffs: bad shape");
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(gamma && beta && mean && rstd && a && bb && (stats || (ws && x)), "ogc_group_norm_coeffs: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!stats) {
        const int rows = b * groups;
        const long long row_len = (long long)(c / groups) * hw;
        slots = launch_gn_stats(rows, row_len, x, ws, s);
        stats = ws;
    }
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3(ogc_divup(b * c, 256)), dim3(256), 0, s, b, c, hw, groups, eps, gamma, beta,
                       stats, slots, mean, rstd, a, bb);
    OGC_CHECK_LAUNCH("ogc_group_norm_coeffs");
    return OGC_OK;
}

extern "C" int ogc_group_norm_stats_slots(void) { return GN_STATS_SLOTS; }
extern "C" int ogc_group_norm_bwd_slots(void) { return GN_BWD_SLOTS; }

extern "C" int ogc_group_norm_fwd(int b, int c, int hw, int groups, float eps, int relu, const float *x,
                                  const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                                  double *ws, ogc_stream_t stream) {
    return gn_fwd_impl("ogc_group_norm_fwd", b, c, hw, groups, eps, relu, x, gamma, beta, y, mean, rstd, ws, nullptr, 0,
                       stream);
}

extern "C" int ogc_group_norm_fwd_stats(int b, int c, int hw, int groups, float eps, int relu, const float *x,
                                        const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                                        const double *stats, int slots, ogc_stream_t stream) {
    OGC_REQUIRE(stats && slots >= 1, "ogc_group_norm_fwd_stats: no statistics");
    return gn_fwd_impl("ogc_group_norm_fwd_stats", b, c, hw, groups, eps, relu, x, gamma, beta, y, mean, rstd, nullptr,
                       stats, slots, stream);
}

namespace {
template <typename AT>
int gn_bwd_impl(int b, int c, int hw, int groups, int relu, const AT *x, const float *gamma, const float *beta, const float *mean,
                const float *rstd, const AT *grad_y, AT *grad_x, float *grad_gamma, float *grad_beta, double *ws,
                ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && hw >= 1 && groups >= 1 && c % groups == 0, "ogc_group_norm_bwd: bad shape");
    if (b == 0) return OGC_OK;
    if (sizeof(AT) == 2 && ((hw & 3) != 0 || (((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x) & 7) != 0)) {
        ogc_set_error("ogc_group_norm_bwd_h: needs hw %% 4 == 0 and 8-byte aligned tensors");
        return OGC_ERR_UNSUPPORTED;
    }
    OGC_REQUIRE(x && gamma && beta && mean && rstd && grad_y && grad_x && grad_gamma && grad_beta && ws,
                "ogc_group_norm_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // ws: partial sums dsdb[slot][b*c][2] (fp64), slot < ogc_group_norm_bwd_slots(); written by the first pass, read by
    // the prologue of the second (which also produces grad_gamma / grad_beta)
    double *dsdb = ws;
    dim3 grid(hw_chunks(b, c, hw), c, b);
    const int slots = (int)grid.x;
    if (relu) {
        hipLaunchKernelGGL((gn_bwd_sums_kernel<true, AT>), grid, dim3(GN_THREADS), 0, s, c, hw, groups, x, gamma, beta, mean,
                           rstd, grad_y, dsdb);
        hipLaunchKernelGGL((gn_bwd_dx_kernel<true, AT>), grid, dim3(GN_THREADS), 0, s, c, hw, groups, x, gamma, beta, mean,
                           rstd, dsdb, slots, grad_y, grad_x, grad_gamma, grad_beta);
    } else {
        hipLaunchKernelGGL((gn_bwd_sums_kernel<false, AT>), grid, dim3(GN_THREADS), 0, s, c, hw, groups, x, gamma, beta, mean,
                           rstd, grad_y, dsdb);
        hipLaunchKernelGGL((gn_bwd_dx_kernel<false, AT>), grid, dim3(GN_THREADS), 0, s, c, hw, groups, x, gamma, beta, mean,
                           rstd, dsdb, slots, grad_y, grad_x, grad_gamma, grad_beta);
    }
    OGC_CHECK_LAUNCH("ogc_group_norm_bwd");
    return OGC_OK;
}
} // namespace

extern "C" int ogc_group_norm_bwd(int b, int c, int hw, int groups, int relu, const float *x, const float *gamma,
                                  const float *beta, const float *mean, const float *rstd, const float *grad_y,
                                  float *grad_x, float *grad_gamma, float *grad_beta, double *ws,
                                  ogc_stream_t stream) {
    return gn_bwd_impl<float>(b, c, hw, groups, relu, x, gamma, beta, mean, rstd, grad_y, grad_x, grad_gamma, grad_beta, ws, stream);
}

extern "C" int ogc_group_norm_bwd_h(int b, int c, int hw, int groups, int relu, const ogc_bf16_t *x, const float *gamma,
                                    const float *beta, const float *mean, const float *rstd, const ogc_bf16_t *grad_y,
                                    ogc_bf16_t *grad_x, float *grad_gamma, float *grad_beta, double *ws,
                                    ogc_stream_t stream) {
    return gn_bwd_impl<ogc_bf16>(b, c, hw, groups, relu, x, gamma, beta, mean, rstd, grad_y, grad_x, grad_gamma, grad_beta, ws, stream);
}

static bool gn_pool_shape_ok(int s) { return s >= 4 && s <= 256 && (s & (s - 1)) == 0; }

namespace {
int gn_pool_fwd_impl(const char *name, int b, int c, int p, int s, int groups, float eps, int relu, const float *x,
                     const float *gamma, const float *beta, float *out, int *argmax, float *mean, float *rstd,
                     double *ws, const double *stats, int slots, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && p >= 1 && groups >= 1 && c % groups == 0, "%s: bad shape", name);
    if (!gn_pool_shape_ok(s) || ((uintptr_t)x & 15) != 0) {
        ogc_set_error("%s: nsample=%d must be a power of two in [4,256] and x 16-byte aligned", name, s);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(x && gamma && beta && out && argmax && mean && rstd && (ws || stats), "%s: null pointer", name);
    OGC_REQUIRE((long long)c * p * s < (1ll << 31) && b <= 65535, "%s: one sample exceeds 32-bit indexing", name);
    hipStream_t st = (hipStream_t)stream;
    if (!stats) {
        const int rows = b * groups;
        const long long row_len = (long long)(c / groups) * p * s;
        slots = launch_gn_stats(rows, row_len, x, ws, st);
        stats = ws;
    }
    const int rows_per_block = GN_THREADS / (s / 4);
    int bx = ogc_divup(p, rows_per_block);
    while (bx > 1 && (long long)bx * c * b > 8192) bx = (bx + 1) / 2;
    dim3 grid(bx, c, b);
    if (relu)
        hipLaunchKernelGGL(gn_apply_maxpool_kernel<true>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, eps, x, gamma,
                           beta, stats, slots, out, argmax, mean, rstd);
    else
        hipLaunchKernelGGL(gn_apply_maxpool_kernel<false>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, eps, x,
                           gamma, beta, stats, slots, out, argmax, mean, rstd);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
} // namespace

extern "C" int ogc_group_norm_maxpool_fwd(int b, int c, int p, int s, int groups, float eps, int relu, const float *x,
                                          const float *gamma, const float *beta, float *out, int *argmax,
                                          float *mean, float *rstd, double *ws, ogc_stream_t stream) {
    return gn_pool_fwd_impl("ogc_group_norm_maxpool_fwd", b, c, p, s, groups, eps, relu, x, gamma, beta, out, argmax,
                            mean, rstd, ws, nullptr, 0, stream);
}

extern "C" int ogc_group_norm_maxpool_fwd_stats(int b, int c, int p, int s, int groups, float eps, int relu,
                                                const float *x, const float *gamma, const float *beta, float *out,
                                                int *argmax, float *mean, float *rstd, const double *stats, int slots,
                                                ogc_stream_t stream) {
    OGC_REQUIRE(stats && slots >= 1, "ogc_group_norm_maxpool_fwd_stats: no statistics");
    return gn_pool_fwd_impl("ogc_group_norm_maxpool_fwd_stats", b, c, p, s, groups, eps, relu, x, gamma, beta, out,
                            argmax, mean, rstd, nullptr, stats, slots, stream);
}

extern "C" int ogc_group_norm_pool_extremes(int b, int c, int p, int s, int groups, float eps, int relu,
                                            const float *yext, const int *aext, const float *gamma, const float *beta,
                                            float *out, int *argmax, float *mean, float *rstd, const double *stats,
                                            int slots, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && p >= 1 && s >= 1 && groups >= 1 && c % groups == 0 && slots >= 1,
                "ogc_group_norm_pool_extremes: bad shape");
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(yext && aext && gamma && beta && out && argmax && mean && rstd && stats,
                "ogc_group_norm_pool_extremes: null pointer");
    OGC_REQUIRE(b <= 65535, "ogc_group_norm_pool_extremes: too many samples for one grid");
    const dim3 grid(c, b);
    hipStream_t st = (hipStream_t)stream;
    if (relu)
        hipLaunchKernelGGL(gn_pool_extremes_kernel<true>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, eps, yext, aext,
                           gamma, beta, stats, slots, out, argmax, mean, rstd);
    else
        hipLaunchKernelGGL(gn_pool_extremes_kernel<false>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, eps, yext, aext,
                           gamma, beta, stats, slots, out, argmax, mean, rstd);
    OGC_CHECK_LAUNCH("ogc_group_norm_pool_extremes");
    return OGC_OK;
}

static int gn_maxpool_bwd_impl(int b, int c, int p, int s, int groups, int relu, const float *x, const float *x_at_argmax,
                               const float *gamma, const float *mean, const float *rstd, const float *out,
                               const int *argmax, const float *grad_out, float *grad_x, float *grad_gamma,
                               float *grad_beta, double *ws, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && p >= 1 && groups >= 1 && c % groups == 0, "ogc_group_norm_maxpool_bwd: bad shape");
    if (!gn_pool_shape_ok(s) || (((uintptr_t)x | (uintptr_t)grad_x) & 15) != 0) {
        ogc_set_error("ogc_group_norm_maxpool_bwd: unsupported nsample=%d or misaligned tensors", s);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(x && gamma && mean && rstd && out && argmax && grad_out && grad_x && grad_gamma && grad_beta && ws,
                "ogc_group_norm_maxpool_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    double *dsdb = ws; // partial sums dsdb[slot][b*c][2], as in ogc_group_norm_bwd
    int slots = ogc_divup(p, GN_THREADS * 4) > 0 ? ogc_divup(p, GN_THREADS * 4) : 1;
    if (slots > GN_BWD_SLOTS) slots = GN_BWD_SLOTS; // the kernel strides over the rest
    dim3 gsum(slots, c, b);
    const int rows_per_block = GN_THREADS / (s / 4);
    int bx = ogc_divup(p, rows_per_block);
    while (bx > 1 && (long long)bx * c * b > 8192) bx = (bx + 1) / 2;
    dim3 grid(bx, c, b);
    if (relu) {
        hipLaunchKernelGGL(gn_maxpool_bwd_sums_kernel<true>, gsum, dim3(GN_THREADS), 0, st, c, p, s, x, out, argmax,
                           grad_out, x_at_argmax, gamma, rstd, groups, dsdb);
        hipLaunchKernelGGL(gn_maxpool_bwd_dx_kernel<true>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, x, gamma, mean,
                           rstd, dsdb, slots, out, argmax, grad_out, grad_x, grad_gamma, grad_beta);
    } else {
        hipLaunchKernelGGL(gn_maxpool_bwd_sums_kernel<false>, gsum, dim3(GN_THREADS), 0, st, c, p, s, x, out, argmax,
                           grad_out, x_at_argmax, gamma, rstd, groups, dsdb);
        hipLaunchKernelGGL(gn_maxpool_bwd_dx_kernel<false>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, x, gamma,
                           mean, rstd, dsdb, slots, out, argmax, grad_out, grad_x, grad_gamma, grad_beta);
    }
    OGC_CHECK_LAUNCH("ogc_group_norm_maxpool_bwd");
    return OGC_OK;
}

extern "C" int ogc_group_norm_maxpool_bwd(int b, int c, int p, int s, int groups, int relu, const float *x,
                                          const float *gamma, const float *mean, const float *rstd, const float *out,
                                          const int *argmax, const float *grad_out, float *grad_x, float *grad_gamma,
                                          float *grad_beta, double *ws, ogc_stream_t stream) {
    return gn_maxpool_bwd_impl(b, c, p, s, groups, relu, x, nullptr, gamma, mean, rstd, out, argmax, grad_out, grad_x, grad_gamma,
                               grad_beta, ws, stream);
}

// The same with x_at_argmax (b, c, p) = x[b, c, pr, argmax[b, c, pr]] handed in (ogc_conv1x1_gemm_affine_pool's yext): the
// sums pass then reads no element of x.
extern "C" int ogc_group_norm_maxpool_bwd_ext(int b, int c, int p, int s, int groups, int relu, const float *x,
                                              const float *x_at_argmax, const float *gamma, const float *mean,
                                              const float *rstd, const float *out, const int *argmax, const float *grad_out,
                                              float *grad_x, float *grad_gamma, float *grad_beta, double *ws,
                                              ogc_stream_t stream) {
    return gn_maxpool_bwd_impl(b, c, p, s, groups, relu, x, x_at_argmax, gamma, mean, rstd, out, argmax, grad_out, grad_x,
                               grad_gamma, grad_beta, ws, stream);
}

// ogc_group_norm_maxpool_bwd without the dense result: coef2 (B, C, 2) and inj (B, C, P, 2) from which
// ogc_conv1x1_wgrad_moments_pooled / ogc_conv1x1_dgrad_adjoint_pooled rebuild grad_x element by element (same expression,
// same bits) while they load x.
namespace {
template <typename XT>
int gn_maxpool_bwd_sparse_impl(int b, int c, int p, int s, int groups, int relu, const XT *x, const float *x_at_argmax,
                               const float *gamma, const float *mean, const float *rstd, const float *out, const int *argmax,
                               const float *grad_out, float *coef2, float *inj, float *grad_gamma, float *grad_beta, double *ws,
                               ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && p >= 1 && groups >= 1 && c % groups == 0, "ogc_group_norm_maxpool_bwd_sparse: bad shape");
    if (!gn_pool_shape_ok(s) || (((uintptr_t)coef2 | (uintptr_t)inj) & 7) != 0) {
        ogc_set_error("ogc_group_norm_maxpool_bwd_sparse: unsupported nsample=%d or misaligned tensors", s);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(x && gamma && mean && rstd && out && argmax && grad_out && coef2 && inj && grad_gamma &&
                    grad_beta && ws,
                "ogc_group_norm_maxpool_bwd_sparse: null pointer");
    OGC_REQUIRE(b <= 65535, "ogc_group_norm_maxpool_bwd_sparse: batch exceeds the grid limit");
    hipStream_t st = (hipStream_t)stream;
    double *dsdb = ws;
    int slots = ogc_divup(p, GN_THREADS * 4) > 0 ? ogc_divup(p, GN_THREADS * 4) : 1;
    if (slots > GN_BWD_SLOTS) slots = GN_BWD_SLOTS;
    dim3 gsum(slots, c, b);
    int bx = ogc_divup(p, GN_THREADS * 4);
    while (bx > 1 && (long long)bx * c * b > 4096) bx = (bx + 1) / 2;
    dim3 grid(bx, c, b);
    float2 *c2 = reinterpret_cast<float2 *>(coef2), *ij = reinterpret_cast<float2 *>(inj);
    if (relu) {
        hipLaunchKernelGGL((gn_maxpool_bwd_sums_kernel<true, XT>), gsum, dim3(GN_THREADS), 0, st, c, p, s, x, out, argmax,
                           grad_out, x_at_argmax, gamma, rstd, groups, dsdb);
        hipLaunchKernelGGL(gn_maxpool_bwd_sparse_kernel<true>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, gamma, mean,
                           rstd, dsdb, slots, out, argmax, grad_out, c2, ij, grad_gamma, grad_beta);
    } else {
        hipLaunchKernelGGL((gn_maxpool_bwd_sums_kernel<false, XT>), gsum, dim3(GN_THREADS), 0, st, c, p, s, x, out, argmax,
                           grad_out, x_at_argmax, gamma, rstd, groups, dsdb);
        hipLaunchKernelGGL(gn_maxpool_bwd_sparse_kernel<false>, grid, dim3(GN_THREADS), 0, st, c, p, s, groups, gamma, mean,
                           rstd, dsdb, slots, out, argmax, grad_out, c2, ij, grad_gamma, grad_beta);
    }
    OGC_CHECK_LAUNCH("ogc_group_norm_maxpool_bwd_sparse");
    return OGC_OK;
}
} // namespace

extern "C" int ogc_group_norm_maxpool_bwd_sparse(int b, int c, int p, int s, int groups, int relu, const float *x,
                                                 const float *x_at_argmax, const float *gamma, const float *mean, const float *rstd,
                                                 const float *out, const int *argmax, const float *grad_out, float *coef2,
                                                 float *inj, float *grad_gamma, float *grad_beta, double *ws,
                                                 ogc_stream_t stream) {
    return gn_maxpool_bwd_sparse_impl<float>(b, c, p, s, groups, relu, x, x_at_argmax, gamma, mean, rstd, out, argmax, grad_out,
                                             coef2, inj, grad_gamma, grad_beta, ws, stream);
}

// (x in 16 bits; x_at_argmax stays fp32: the rounded extremes ogc_conv1x1_gemm_affine_pool_h wrote)
extern "C" int ogc_group_norm_maxpool_bwd_sparse_h(int b, int c, int p, int s, int groups, int relu, const ogc_bf16_t *x,
                                                   const float *x_at_argmax, const float *gamma, const float *mean,
                                                   const float *rstd, const float *out, const int *argmax, const float *grad_out,
                                                   float *coef2, float *inj, float *grad_gamma, float *grad_beta, double *ws,
                                                   ogc_stream_t stream) {
    return gn_maxpool_bwd_sparse_impl<ogc_bf16>(b, c, p, s, groups, relu, x, x_at_argmax, gamma, mean, rstd, out, argmax, grad_out,
                                                coef2, inj, grad_gamma, grad_beta, ws, stream);
}
