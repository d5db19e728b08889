// batch_norm.hip — fused BatchNorm2d (+ ReLU) (+ max over the neighbourhood), forward and backward, for the per-point
// MLPs of the FlowStep3D nets.
//
// Replaces the `F.relu(bn(conv(x)))` chains and the trailing `.max(dim=-1)` of FlowEmbedding / PointNetSetAbstraction
// (reference: utils/flowstep3d_util.py:57-66, :126-138; nn.BatchNorm2d in its default configuration).  MIOpen runs
// BatchNorm on these (B, C, N, nsample) tensors as separate statistics / normalise / backward kernels around NCHW
// transposes, ReLU and the max as further full passes (BatchNorm + max-reduce ≈ 25 % of a FlowStep3D training step).
// Same plan as group_norm.hip with per-channel rows over the whole batch:
//   forward : stats   per channel over (B, HW), fp64 partials          1 read   (skipped when the producing
//                                                                                convolution delivers them)
//             finalize mean, rstd per channel; running statistics        tiny
//             apply   y = relu(a_c x + b_c)      [or max over nsample]   1 read + 1 write  [1 read]
//   backward: sums    ds = sum dy'*x, db = sum dy' per channel           2 reads      (dy' = dy * [y > 0])
//             params  dgamma, dbeta, c2, c3 per channel                  tiny
//             dx      = dy' * gamma_c * rstd_c + c2 * x + c3             2 reads + 1 write
// Semantics of nn.BatchNorm2d: training = batch statistics with biased variance for the normalisation, running
// statistics updated with `momentum` and the UNBIASED variance; evaluation = running statistics, whose gradient is
// the plain affine map (c2 = c3 = 0).
#include <stdlib.h>

#include "ogc_common.h"

namespace {

constexpr int BN_THREADS = 256;

__device__ __forceinline__ void bn_block_sum2(double &a, double &b, double *smem /* [2*BN_THREADS/64] */) {
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
        for (int w = 0; w < BN_THREADS / 64; ++w) {
            a += smem[w * 2];
            b += smem[w * 2 + 1];
        }
    }
}

// ---- finalize folded into the consumers (round 6) ---------------------------------------------------------------------------------
// A training-mode BatchNorm used to be stats -> finalize -> apply forwards and sums -> params -> dx backwards: the two middle
// launches are one thread per channel, ~4.5 us each plus a dependent kernel boundary, 210 of them per FlowStep3D training step
// whose launch thread is as busy as its GPU.  Every workgroup of the apply / dx kernel now derives ITS channel's coefficients from
// the channel's sums itself (the same double-precision expressions, so the same values bit for bit), and the workgroup
// (chunk 0, sample 0) of a channel also writes what the middle launch wrote: mean / rstd / running statistics, dgamma / dbeta.
struct BnFin {   // stats == nullptr: not folded (the coefficients come from mean / rstd / var as before)
    const double *stats;
    int slots;
    double count;
    float momentum;
    float *running_mean, *running_var, *mean_out, *rstd_out;
};

__device__ __forceinline__ void bn_fin(const BnFin &f, int c, int ch, float eps, bool writer, float &mean, float &rstd) {
    double sum = 0.0, sumsq = 0.0;
    for (int sl = 0; sl < f.slots; ++sl) {
        sum += f.stats[((size_t)sl * c + ch) * 2];
        sumsq += f.stats[((size_t)sl * c + ch) * 2 + 1];
    }
    const double m = sum / f.count;
    const double var = fmax(sumsq / f.count - m * m, 0.0);
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (writer) {
        f.mean_out[ch] = mean;
        f.rstd_out[ch] = rstd;
        if (f.running_mean && f.running_var) {
            const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
            f.running_mean[ch] = (float)((1.0 - (double)f.momentum) * (double)f.running_mean[ch] + (double)f.momentum * m);
            f.running_var[ch] = (float)((1.0 - (double)f.momentum) * (double)f.running_var[ch] + (double)f.momentum * unbiased);
        }
    }
}

struct BnBwdFin {   // dsdb == nullptr: not folded (c2, c3 come from the c2c3 buffer)
    const double *dsdb;
    double count;
    int training;
    float *dgamma, *dbeta;
};

__device__ __forceinline__ void bn_bwd_fin(const BnBwdFin &f, int ch, float gamma, float mean, float rstd, bool writer, float &c2f,
                                           float &c3f) {
    const double ds = f.dsdb[(size_t)ch * 2], db = f.dsdb[(size_t)ch * 2 + 1];
    const double m = mean, r = rstd, g = gamma;
    double c2 = 0.0, c3 = 0.0;
    if (f.training) {
        c2 = g * (db * m - ds) * r * r * r / f.count;
        c3 = -c2 * m - g * db * r / f.count;
    }
    c2f = (float)c2;
    c3f = (float)c3;
    if (writer) {
        f.dgamma[ch] = (float)((ds - m * db) * r);
        f.dbeta[ch] = (float)db;
    }
}

// grid (chunks, C, B): partial sum / sum of squares of one (sample, channel) segment -> ws[c][0..1] (fp64 atomics)
// part (deterministic mode, else null): the workgroup's pair goes to slab (sample, chunk) of part instead — summed in slab order
// by ogc_det_reduce_f64
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(int c, int hw, const float *__restrict__ x,
                                                              double *__restrict__ ws, double *__restrict__ part) {
    __shared__ double smem[2 * BN_THREADS / 64];
    const int b = blockIdx.z, ch = blockIdx.y;
    const float *p = x + ((size_t)b * c + ch) * hw;
    double s = 0.0, ss = 0.0;
    if ((hw & 3) == 0 && ((uintptr_t)p & 15) == 0) {
        for (int i = (blockIdx.x * BN_THREADS + threadIdx.x) * 4; i < hw; i += gridDim.x * BN_THREADS * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p + i);
            s += (double)((v.x + v.y) + (v.z + v.w));
            ss += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
        }
    } else {
        for (int i = blockIdx.x * BN_THREADS + threadIdx.x; i < hw; i += gridDim.x * BN_THREADS) {
            const float v = p[i];
            s += v;
            ss += (double)v * v;
        }
    }
    bn_block_sum2(s, ss, smem);
    if (threadIdx.x == 0) {
        if (part) {
            double *slab = part + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 * c;
            slab[(size_t)ch * 2] = s;
            slab[(size_t)ch * 2 + 1] = ss;
        } else {
            atomicAdd(ws + (size_t)ch * 2, s);
            atomicAdd(ws + (size_t)ch * 2 + 1, ss);
        }
    }
}

// one thread per channel
__global__ void bn_finalize_kernel(int c, double count, float eps, int training, float momentum,
                                   const double *__restrict__ stats, int slots, float *__restrict__ running_mean,
                                   float *__restrict__ running_var, float *__restrict__ mean_out,
                                   float *__restrict__ rstd_out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    if (!training) {
        mean_out[ch] = running_mean[ch];
        rstd_out[ch] = (float)(1.0 / sqrt((double)running_var[ch] + (double)eps));
        return;
    }
    double sum = 0.0, sumsq = 0.0;
    for (int sl = 0; sl < slots; ++sl) {
        sum += stats[((size_t)sl * c + ch) * 2];
        sumsq += stats[((size_t)sl * c + ch) * 2 + 1];
    }
    const double m = sum / count;
    const double var = fmax(sumsq / count - m * m, 0.0);
    mean_out[ch] = (float)m;
    rstd_out[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[ch] = (float)((1.0 - (double)momentum) * (double)running_mean[ch] + (double)momentum * m);
        running_var[ch] = (float)((1.0 - (double)momentum) * (double)running_var[ch] + (double)momentum * unbiased);
    }
}

template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(int c, int hw, const float *__restrict__ x,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ rstd,
                                                              const float *__restrict__ var, float eps,
                                                              float *__restrict__ y, BnFin fin) {
    const int b = blockIdx.z, ch = blockIdx.y;
    // var != nullptr: evaluation straight from the running statistics (mean = running_mean), no finalize launch
    float mu, rs;
    if (fin.stats) bn_fin(fin, c, ch, eps, blockIdx.x == 0 && b == 0 && threadIdx.x == 0, mu, rs);
    else { mu = mean[ch]; rs = var ? (float)(1.0 / sqrt((double)var[ch] + (double)eps)) : rstd[ch]; }
    const float a = rs * gamma[ch];
    const float bb = beta[ch] - mu * a;
    const size_t base = ((size_t)b * c + ch) * hw;
    const float *px = x + base;
    float *py = y + base;
    if ((hw & 3) == 0 && (((uintptr_t)px | (uintptr_t)py) & 15) == 0) {
        for (int i = (blockIdx.x * BN_THREADS + threadIdx.x) * 4; i < hw; i += gridDim.x * BN_THREADS * 4) {
            float4 v = *reinterpret_cast<const float4 *>(px + i);
            v.x = fmaf(a, v.x, bb); v.y = fmaf(a, v.y, bb); v.z = fmaf(a, v.z, bb); v.w = fmaf(a, v.w, bb);
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(py + i) = v;
        }
    } else {
        for (int i = blockIdx.x * BN_THREADS + threadIdx.x; i < hw; i += gridDim.x * BN_THREADS) {
            const float v = fmaf(a, px[i], bb);
            py[i] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
}

// x (B, C, P, S) -> out (B, C, P) = max_s act(a_c x + b_c), argmax (first index on ties); L = S/4 lanes per row
template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_maxpool_kernel(int c, int p, int s,
                                                                      const float *__restrict__ x,
                                                                      const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta,
                                                                      const float *__restrict__ mean,
                                                                      const float *__restrict__ rstd,
                                                                      const float *__restrict__ var, float eps,
                                                                      float *__restrict__ out, int *__restrict__ arg, BnFin fin) {
    const int b = blockIdx.z, ch = blockIdx.y;
    float mu, rs;
    if (fin.stats) bn_fin(fin, c, ch, eps, blockIdx.x == 0 && b == 0 && threadIdx.x == 0, mu, rs);
    else { mu = mean[ch]; rs = var ? (float)(1.0 / sqrt((double)var[ch] + (double)eps)) : rstd[ch]; }
    const float a = rs * gamma[ch];
    const float bb = beta[ch] - mu * a;
    const int L = s >> 2;
    const int rows_per_block = BN_THREADS / L;
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

// ---- backward ------------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_sums_kernel(int c, int hw, const float *__restrict__ x,
                                                                 const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta,
                                                                 const float *__restrict__ mean,
                                                                 const float *__restrict__ rstd,
                                                                 const float *__restrict__ dy,
                                                                 double *__restrict__ dsdb, double *__restrict__ part) {
    __shared__ double smem[2 * BN_THREADS / 64];
    const int b = blockIdx.z, ch = blockIdx.y;
    const float a = rstd[ch] * gamma[ch];
    const float bb = beta[ch] - mean[ch] * a;
    const size_t base = ((size_t)b * c + ch) * hw;
    const float *px = x + base, *pd = dy + base;
    double s = 0.0, sb = 0.0;
    if ((hw & 3) == 0 && (((uintptr_t)px | (uintptr_t)pd) & 15) == 0) {
        for (int i = (blockIdx.x * BN_THREADS + threadIdx.x) * 4; i < hw; i += gridDim.x * BN_THREADS * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(px + i);
            float4 d = *reinterpret_cast<const float4 *>(pd + i);
            if (RELU) {
                d.x = fmaf(a, v.x, bb) > 0.f ? d.x : 0.f; d.y = fmaf(a, v.y, bb) > 0.f ? d.y : 0.f;
                d.z = fmaf(a, v.z, bb) > 0.f ? d.z : 0.f; d.w = fmaf(a, v.w, bb) > 0.f ? d.w : 0.f;
            }
            s += (double)((d.x * v.x + d.y * v.y) + (d.z * v.z + d.w * v.w));
            sb += (double)((d.x + d.y) + (d.z + d.w));
        }
    } else {
        for (int i = blockIdx.x * BN_THREADS + threadIdx.x; i < hw; i += gridDim.x * BN_THREADS) {
            const float v = px[i];
            float d = pd[i];
            if (RELU) d = fmaf(a, v, bb) > 0.f ? d : 0.f;
            s += (double)d * v;
            sb += d;
        }
    }
    bn_block_sum2(s, sb, smem);
    if (threadIdx.x == 0) {
        if (part) {
            double *slab = part + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 * c;
            slab[(size_t)ch * 2] = s;
            slab[(size_t)ch * 2 + 1] = sb;
        } else {
            atomicAdd(dsdb + (size_t)ch * 2, s);
            atomicAdd(dsdb + (size_t)ch * 2 + 1, sb);
        }
    }
}

// one thread per channel: dgamma, dbeta, and the coefficients of dx = a dy' + c2 x + c3
__global__ void bn_bwd_params_kernel(int c, double count, int training, const float *__restrict__ gamma,
                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                     const double *__restrict__ dsdb, float *__restrict__ dgamma,
                                     float *__restrict__ dbeta, float *__restrict__ c2c3) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double ds = dsdb[(size_t)ch * 2], db = dsdb[(size_t)ch * 2 + 1];
    const double m = mean[ch], r = rstd[ch], g = gamma[ch];
    dgamma[ch] = (float)((ds - m * db) * r);
    dbeta[ch] = (float)db;
    double c2 = 0.0, c3 = 0.0;
    if (training) {
        c2 = g * (db * m - ds) * r * r * r / count;
        c3 = -c2 * m - g * db * r / count;
    }
    c2c3[ch * 2] = (float)c2;
    c2c3[ch * 2 + 1] = (float)c3;
}

template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_dx_kernel(int c, int hw, const float *__restrict__ x,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ rstd,
                                                               const float *__restrict__ c2c3,
                                                               const float *__restrict__ dy, float *__restrict__ dx, BnBwdFin fin) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const float a = rstd[ch] * gamma[ch];
    const float bb = beta[ch] - mean[ch] * a;
    float c2, c3;
    if (fin.dsdb) bn_bwd_fin(fin, ch, gamma[ch], mean[ch], rstd[ch], blockIdx.x == 0 && b == 0 && threadIdx.x == 0, c2, c3);
    else { c2 = c2c3[ch * 2]; c3 = c2c3[ch * 2 + 1]; }
    const size_t base = ((size_t)b * c + ch) * hw;
    const float *px = x + base, *pd = dy + base;
    float *po = dx + base;
    if ((hw & 3) == 0 && (((uintptr_t)px | (uintptr_t)pd | (uintptr_t)po) & 15) == 0) {
        for (int i = (blockIdx.x * BN_THREADS + threadIdx.x) * 4; i < hw; i += gridDim.x * BN_THREADS * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(px + i);
            float4 d = *reinterpret_cast<const float4 *>(pd + i);
            if (RELU) {
                d.x = fmaf(a, v.x, bb) > 0.f ? d.x : 0.f; d.y = fmaf(a, v.y, bb) > 0.f ? d.y : 0.f;
                d.z = fmaf(a, v.z, bb) > 0.f ? d.z : 0.f; d.w = fmaf(a, v.w, bb) > 0.f ? d.w : 0.f;
            }
            float4 o;
            o.x = fmaf(a, d.x, fmaf(c2, v.x, c3)); o.y = fmaf(a, d.y, fmaf(c2, v.y, c3));
            o.z = fmaf(a, d.z, fmaf(c2, v.z, c3)); o.w = fmaf(a, d.w, fmaf(c2, v.w, c3));
            *reinterpret_cast<float4 *>(po + i) = o;
        }
    } else {
        for (int i = blockIdx.x * BN_THREADS + threadIdx.x; i < hw; i += gridDim.x * BN_THREADS) {
            const float v = px[i];
            float d = pd[i];
            if (RELU) d = fmaf(a, v, bb) > 0.f ? d : 0.f;
            po[i] = fmaf(a, d, fmaf(c2, v, c3));
        }
    }
}

// sparse gradient of the max: non-zero only at the arg-max element of each row
template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_maxpool_bwd_sums_kernel(int c, int p, int s,
                                                                         const float *__restrict__ x,
                                                                         const float *__restrict__ out,
                                                                         const int *__restrict__ arg,
                                                                         const float *__restrict__ gout,
                                                                         double *__restrict__ dsdb, double *__restrict__ part) {
    __shared__ double smem[2 * BN_THREADS / 64];
    const int b = blockIdx.z, ch = blockIdx.y;
    const size_t base = ((size_t)b * c + ch) * p;
    double ds = 0.0, db = 0.0;
    for (int pr = blockIdx.x * BN_THREADS + threadIdx.x; pr < p; pr += gridDim.x * BN_THREADS) {
        float g = gout[base + pr];
        if (RELU && !(out[base + pr] > 0.f)) g = 0.f;
        ds += (double)g * (double)x[(base + pr) * s + arg[base + pr]];
        db += g;
    }
    bn_block_sum2(ds, db, smem);
    if (threadIdx.x == 0) {
        if (part) {
            double *slab = part + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 * c;
            slab[(size_t)ch * 2] = ds;
            slab[(size_t)ch * 2 + 1] = db;
        } else {
            atomicAdd(dsdb + (size_t)ch * 2, ds);
            atomicAdd(dsdb + (size_t)ch * 2 + 1, db);
        }
    }
}

template <bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_maxpool_bwd_dx_kernel(int c, int p, int s,
                                                                       const float *__restrict__ x,
                                                                       const float *__restrict__ gamma,
                                                                       const float *__restrict__ rstd,
                                                                       const float *__restrict__ c2c3,
                                                                       const float *__restrict__ out,
                                                                       const int *__restrict__ arg,
                                                                       const float *__restrict__ gout,
                                                                       float *__restrict__ dx, const float *__restrict__ mean,
                                                                       BnBwdFin fin) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const float a = rstd[ch] * gamma[ch];
    float c2, c3;
    if (fin.dsdb) bn_bwd_fin(fin, ch, gamma[ch], mean[ch], rstd[ch], blockIdx.x == 0 && b == 0 && threadIdx.x == 0, c2, c3);
    else { c2 = c2c3[ch * 2]; c3 = c2c3[ch * 2 + 1]; }
    const int L = s >> 2;
    const int rows_per_block = BN_THREADS / L;
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

int bn_chunks(int b, int c, int hw) {
    int chunks = 1;
    while ((long long)b * c * chunks < 2048 && hw / (chunks * 2) >= 4096) chunks *= 2;
    return chunks;
}

bool bn_pool_shape_ok(int s) { return s >= 4 && s <= 256 && (s & (s - 1)) == 0; }

// OGC_BN_FOLD=0 in the environment: the finalize / params launches of their own, as until round 5 (A/B runs, tests of both)
bool bn_fold_finalize() {
    static const bool on = [] { const char *e = getenv("OGC_BN_FOLD"); return !(e && e[0] == '0'); }();
    return on;
}

// deterministic mode: one slab of 2 c doubles per workgroup of the split (grid.x chunks x grid.z samples), every entry written
double *bn_partials(const char *name, dim3 grid, int c, hipStream_t s) {
    if (!ogc_deterministic()) return nullptr;
    double *part = static_cast<double *>(ogc_det_scratch(s, sizeof(double) * 2 * (size_t)c * grid.x * grid.z));
    if (!part) ogc_set_error("%s (deterministic): no scratch memory", name);
    return part;
}

// statistics (unless supplied) + finalize; leaves mean / rstd per channel
// fin: training mode — the finalize step is left to the apply kernel (see BnFin); evaluation mode keeps the finalize launch
int bn_prepare(const char *name, int b, int c, int hw, float eps, int training, float momentum, const float *x,
               float *running_mean, float *running_var, float *mean, float *rstd, double *ws, const double *stats,
               int slots, hipStream_t s, BnFin *fin) {
    *fin = BnFin{nullptr, 0, 0.0, 0.f, nullptr, nullptr, nullptr, nullptr};
    if (training && !stats) {
        OGC_REQUIRE(ws, "%s: null workspace", name);
        if (ogc_zero_async(ws, sizeof(double) * 2 * c, s) != hipSuccess) {
            ogc_set_error("%s: memset failed", name);
            return OGC_ERR_LAUNCH;
        }
        const dim3 grid(bn_chunks(b, c, hw), c, b);
        double *part = bn_partials(name, grid, c, s);
        if (ogc_deterministic() && !part) return OGC_ERR_LAUNCH;
        hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(BN_THREADS), 0, s, c, hw, x, ws, part);
        if (part && ogc_det_reduce_f64(ws, part, (int)(grid.x * grid.z), 2ll * c, 0, s) != hipSuccess) return OGC_ERR_LAUNCH;
        stats = ws;
        slots = 1;
    }
    if (!training) OGC_REQUIRE(running_mean && running_var, "%s: evaluation mode needs running statistics", name);
    if (training && bn_fold_finalize()) {
        *fin = BnFin{stats, slots, (double)b * (double)hw, momentum, running_mean, running_var, mean, rstd};
        return OGC_OK;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ogc_divup(c, 256)), dim3(256), 0, s, c, (double)b * (double)hw, eps,
                       training, momentum, stats, slots, running_mean, running_var, mean, rstd);
    return OGC_OK;
}

} // namespace

extern "C" int ogc_batch_norm_fwd(int b, int c, int hw, float eps, int relu, int training, float momentum,
                                  const float *x, const float *gamma, const float *beta, float *running_mean,
                                  float *running_var, float *y, float *mean, float *rstd, double *ws,
                                  const double *stats, int slots, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && hw >= 1, "ogc_batch_norm_fwd: bad shape");
    if (b == 0) return OGC_OK;
    // evaluation without mean / rstd outputs (nothing will be back-propagated): the affine map is formed from the running
    // statistics inside the apply kernel — one launch
    const bool inline_eval = !training && !mean && !rstd;
    OGC_REQUIRE(x && gamma && beta && y && (inline_eval || (mean && rstd)), "ogc_batch_norm_fwd: null pointer");
    OGC_REQUIRE((long long)c * hw < (1ll << 31) && b <= 65535, "ogc_batch_norm_fwd: one sample exceeds 32-bit indexing");
    hipStream_t s = (hipStream_t)stream;
    const float *var = nullptr;
    BnFin fin{nullptr, 0, 0.0, 0.f, nullptr, nullptr, nullptr, nullptr};
    if (inline_eval) {
        OGC_REQUIRE(running_mean && running_var, "ogc_batch_norm_fwd: evaluation mode needs running statistics");
        mean = running_mean;
        var = running_var;
    } else {
        const int rc = bn_prepare("ogc_batch_norm_fwd", b, c, hw, eps, training, momentum, x, running_mean, running_var,
                                  mean, rstd, ws, stats, slots, s, &fin);
        if (rc != OGC_OK) return rc;
    }
    dim3 grid(bn_chunks(b, c, hw), c, b);
    if (relu)
        hipLaunchKernelGGL(bn_apply_kernel<true>, grid, dim3(BN_THREADS), 0, s, c, hw, x, gamma, beta, mean, rstd, var, eps,
                           y, fin);
    else
        hipLaunchKernelGGL(bn_apply_kernel<false>, grid, dim3(BN_THREADS), 0, s, c, hw, x, gamma, beta, mean, rstd, var,
                           eps, y, fin);
    OGC_CHECK_LAUNCH("ogc_batch_norm_fwd");
    return OGC_OK;
}

extern "C" int ogc_batch_norm_bwd(int b, int c, int hw, int relu, int training, const float *x, const float *gamma,
                                  const float *beta, const float *mean, const float *rstd, const float *grad_y,
                                  float *grad_x, float *grad_gamma, float *grad_beta, double *ws,
                                  ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && hw >= 1, "ogc_batch_norm_bwd: bad shape");
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(x && gamma && beta && mean && rstd && grad_y && grad_x && grad_gamma && grad_beta && ws,
                "ogc_batch_norm_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    double *dsdb = ws; // [c][2] fp64, then c2c3 [c][2] fp32
    float *c2c3 = reinterpret_cast<float *>(ws + (size_t)2 * c);
    if (ogc_zero_async(dsdb, sizeof(double) * 2 * c, s) != hipSuccess) {
        ogc_set_error("ogc_batch_norm_bwd: memset failed");
        return OGC_ERR_LAUNCH;
    }
    dim3 grid(bn_chunks(b, c, hw), c, b);
    double *part = bn_partials("ogc_batch_norm_bwd", grid, c, s);
    if (ogc_deterministic() && !part) return OGC_ERR_LAUNCH;
    if (relu)
        hipLaunchKernelGGL(bn_bwd_sums_kernel<true>, grid, dim3(BN_THREADS), 0, s, c, hw, x, gamma, beta, mean, rstd,
                           grad_y, dsdb, part);
    else
        hipLaunchKernelGGL(bn_bwd_sums_kernel<false>, grid, dim3(BN_THREADS), 0, s, c, hw, x, gamma, beta, mean, rstd,
                           grad_y, dsdb, part);
    if (part && ogc_det_reduce_f64(dsdb, part, (int)(grid.x * grid.z), 2ll * c, 0, s) != hipSuccess) return OGC_ERR_LAUNCH;
    BnBwdFin bfin{nullptr, 0.0, 0, nullptr, nullptr};
    if (bn_fold_finalize()) bfin = BnBwdFin{dsdb, (double)b * (double)hw, training, grad_gamma, grad_beta};
    else
        hipLaunchKernelGGL(bn_bwd_params_kernel, dim3(ogc_divup(c, 256)), dim3(256), 0, s, c, (double)b * (double)hw,
                           training, gamma, mean, rstd, dsdb, grad_gamma, grad_beta, c2c3);
    if (relu)
        hipLaunchKernelGGL(bn_bwd_dx_kernel<true>, grid, dim3(BN_THREADS), 0, s, c, hw, x, gamma, beta, mean, rstd, c2c3,
                           grad_y, grad_x, bfin);
    else
        hipLaunchKernelGGL(bn_bwd_dx_kernel<false>, grid, dim3(BN_THREADS), 0, s, c, hw, x, gamma, beta, mean, rstd,
                           c2c3, grad_y, grad_x, bfin);
    OGC_CHECK_LAUNCH("ogc_batch_norm_bwd");
    return OGC_OK;
}

extern "C" int ogc_batch_norm_maxpool_fwd(int b, int c, int p, int s, float eps, int relu, int training, float momentum,
                                          const float *x, const float *gamma, const float *beta, float *running_mean,
                                          float *running_var, float *out, int *argmax, float *mean, float *rstd,
                                          double *ws, const double *stats, int slots, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && p >= 1, "ogc_batch_norm_maxpool_fwd: bad shape");
    if (!bn_pool_shape_ok(s) || ((uintptr_t)x & 15) != 0) {
        ogc_set_error("ogc_batch_norm_maxpool_fwd: nsample=%d must be a power of two in [4,256] and x 16-byte aligned", s);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    const bool inline_eval = !training && !mean && !rstd; // as in ogc_batch_norm_fwd
    OGC_REQUIRE(x && gamma && beta && out && argmax && (inline_eval || (mean && rstd)),
                "ogc_batch_norm_maxpool_fwd: null pointer");
    OGC_REQUIRE((long long)c * p * s < (1ll << 31) && b <= 65535,
                "ogc_batch_norm_maxpool_fwd: one sample exceeds 32-bit indexing");
    hipStream_t st = (hipStream_t)stream;
    const float *var = nullptr;
    BnFin fin{nullptr, 0, 0.0, 0.f, nullptr, nullptr, nullptr, nullptr};
    if (inline_eval) {
        OGC_REQUIRE(running_mean && running_var, "ogc_batch_norm_maxpool_fwd: evaluation mode needs running statistics");
        mean = running_mean;
        var = running_var;
    } else {
        const int rc = bn_prepare("ogc_batch_norm_maxpool_fwd", b, c, p * s, eps, training, momentum, x, running_mean,
                                  running_var, mean, rstd, ws, stats, slots, st, &fin);
        if (rc != OGC_OK) return rc;
    }
    const int rows_per_block = BN_THREADS / (s / 4);
    int bx = ogc_divup(p, rows_per_block);
    while (bx > 1 && (long long)bx * c * b > 8192) bx = (bx + 1) / 2;
    dim3 grid(bx, c, b);
    if (relu)
        hipLaunchKernelGGL(bn_apply_maxpool_kernel<true>, grid, dim3(BN_THREADS), 0, st, c, p, s, x, gamma, beta, mean,
                           rstd, var, eps, out, argmax, fin);
    else
        hipLaunchKernelGGL(bn_apply_maxpool_kernel<false>, grid, dim3(BN_THREADS), 0, st, c, p, s, x, gamma, beta, mean,
                           rstd, var, eps, out, argmax, fin);
    OGC_CHECK_LAUNCH("ogc_batch_norm_maxpool_fwd");
    return OGC_OK;
}

extern "C" int ogc_batch_norm_maxpool_bwd(int b, int c, int p, int s, int relu, int training, const float *x,
                                          const float *gamma, const float *mean, const float *rstd, const float *out,
                                          const int *argmax, const float *grad_out, float *grad_x, float *grad_gamma,
                                          float *grad_beta, double *ws, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 1 && p >= 1, "ogc_batch_norm_maxpool_bwd: bad shape");
    if (!bn_pool_shape_ok(s) || (((uintptr_t)x | (uintptr_t)grad_x) & 15) != 0) {
        ogc_set_error("ogc_batch_norm_maxpool_bwd: nsample=%d must be a power of two in [4,256], tensors 16-byte aligned",
                      s);
        return OGC_ERR_UNSUPPORTED;
    }
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(x && gamma && mean && rstd && out && argmax && grad_out && grad_x && grad_gamma && grad_beta && ws,
                "ogc_batch_norm_maxpool_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    double *dsdb = ws;
    float *c2c3 = reinterpret_cast<float *>(ws + (size_t)2 * c);
    if (ogc_zero_async(dsdb, sizeof(double) * 2 * c, st) != hipSuccess) {
        ogc_set_error("ogc_batch_norm_maxpool_bwd: memset failed");
        return OGC_ERR_LAUNCH;
    }
    int bs = ogc_divup(p, BN_THREADS);
    while (bs > 1 && (long long)bs * c * b > 4096) bs = (bs + 1) / 2;
    dim3 gsum(bs, c, b);
    double *part = bn_partials("ogc_batch_norm_maxpool_bwd", gsum, c, st);
    if (ogc_deterministic() && !part) return OGC_ERR_LAUNCH;
    if (relu)
        hipLaunchKernelGGL(bn_maxpool_bwd_sums_kernel<true>, gsum, dim3(BN_THREADS), 0, st, c, p, s, x, out, argmax,
                           grad_out, dsdb, part);
    else
        hipLaunchKernelGGL(bn_maxpool_bwd_sums_kernel<false>, gsum, dim3(BN_THREADS), 0, st, c, p, s, x, out, argmax,
                           grad_out, dsdb, part);
    if (part && ogc_det_reduce_f64(dsdb, part, (int)(gsum.x * gsum.z), 2ll * c, 0, st) != hipSuccess) return OGC_ERR_LAUNCH;
    BnBwdFin bfin{nullptr, 0.0, 0, nullptr, nullptr};
    if (bn_fold_finalize()) bfin = BnBwdFin{dsdb, (double)b * (double)p * (double)s, training, grad_gamma, grad_beta};
    else
        hipLaunchKernelGGL(bn_bwd_params_kernel, dim3(ogc_divup(c, 256)), dim3(256), 0, st, c,
                           (double)b * (double)p * (double)s, training, gamma, mean, rstd, dsdb, grad_gamma, grad_beta, c2c3);
    const int rows_per_block = BN_THREADS / (s / 4);
    int bx = ogc_divup(p, rows_per_block);
    while (bx > 1 && (long long)bx * c * b > 8192) bx = (bx + 1) / 2;
    dim3 grid(bx, c, b);
    if (relu)
        hipLaunchKernelGGL(bn_maxpool_bwd_dx_kernel<true>, grid, dim3(BN_THREADS), 0, st, c, p, s, x, gamma, rstd, c2c3,
                           out, argmax, grad_out, grad_x, mean, bfin);
    else
        hipLaunchKernelGGL(bn_maxpool_bwd_dx_kernel<false>, grid, dim3(BN_THREADS), 0, st, c, p, s, x, gamma, rstd, c2c3,
                           out, argmax, grad_out, grad_x, mean, bfin);
    OGC_CHECK_LAUNCH("ogc_batch_norm_maxpool_bwd");
    return OGC_OK;
}
