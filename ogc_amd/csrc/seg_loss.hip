// seg_loss.hip — the dynamic (rigid-motion) and invariance terms of the unsupervised OGC loss as a handful of launches.
//
// Reference: losses/seg_loss_unsup.py — DynamicLoss (:64-98) with fit_motion_svd_batch (:10-61), InvarianceLoss
// (:243-280) with match_mask_by_iou (:212-240).  Written with tensor ops they are ~65 and ~95 small launches per step
// (K-fold expanded clouds, one-hot tensors, einsums over (B, N, K) ...); each is a short reduction or a per-point map:
//
//   ogc_rigid_moments      per (cloud, slot): W = sum w, sum w p, sum w q, sum w p q^T  (fp64 accumulation) — one pass;
//                          then S = sum w (p - pbar)(q - qbar)^T = M - (sum w p)(sum w q)^T / W, means, in fp64
//   (ogc_kabsch_rotation)  R per slot (kabsch.hip)
//   ogc_rigid_blend_fwd    per point: || sum_k m_k (R_k p + t_k) - q ||_p              (t_k = qbar_k - R_k pbar_k)
//   ogc_rigid_blend_bwd    d/dm_k = g * <d||.||/d blended, R_k p + t_k>                (the fit is detached, :91)
//   ogc_mask_confusion     per sample: K x K counts of (argmax mask1, argmax mask2) -> IoU matrix (:220-233)
//   ogc_matched_distance_fwd / _bwd   per point: || m1 - m2[:, col] ||_p and the reverse direction, targets detached
#include "ogc_common.h"

namespace {

constexpr int SL_THREADS = 256;
constexpr int SL_KMAX = 32;

// store: a plain store instead of the atomic (deterministic mode: global_out is then this workgroup's own slab)
__device__ __forceinline__ void block_sum_doubles(double *vals, int nval, double *smem /* [nval][4] */,
                                                  double *global_out, bool store = false) {
    // wave reduce, then one LDS slot per wave, then thread 0..nval-1 adds the 4 waves and issues the global atomic
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v = 0; v < nval; ++v) {
        double x = vals[v];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) smem[v * (SL_THREADS / 64) + wave] = x;
    }
    __syncthreads();
    if ((int)threadIdx.x < nval) {
        double s = 0.0;
        for (int w = 0; w < SL_THREADS / 64; ++w) s += smem[threadIdx.x * (SL_THREADS / 64) + w];
        if (store) global_out[threadIdx.x] = s;
        else atomicAdd(global_out + threadIdx.x, s);
    }
}

// grid (chunks, K, VB).  mom[vb][k][16] = {W, wp(3), wq(3), wpq(9)}
__global__ __launch_bounds__(SL_THREADS) void rigid_moments_kernel(int n, int k, const float *__restrict__ pc,
                                                                   const float *__restrict__ pc2,
                                                                   const float *__restrict__ mask,
                                                                   double *__restrict__ mom, double *__restrict__ part) {
    __shared__ double smem[16 * (SL_THREADS / 64)];
    const int slot = blockIdx.y, vb = blockIdx.z;
    const float *p = pc + (size_t)vb * n * 3, *q = pc2 + (size_t)vb * n * 3;
    const float *m = mask + (size_t)vb * n * k + slot;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 0.0;
    for (int i = blockIdx.x * SL_THREADS + threadIdx.x; i < n; i += gridDim.x * SL_THREADS) {
        const double w = m[(size_t)i * k];
        const double px = p[i * 3], py = p[i * 3 + 1], pz = p[i * 3 + 2];
        const double qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
        const double wpx = w * px, wpy = w * py, wpz = w * pz;
        a[0] += w;
        a[1] += wpx; a[2] += wpy; a[3] += wpz;
        a[4] += w * qx; a[5] += w * qy; a[6] += w * qz;
        a[7] += wpx * qx; a[8] += wpx * qy; a[9] += wpx * qz;
        a[10] += wpy * qx; a[11] += wpy * qy; a[12] += wpy * qz;
        a[13] += wpz * qx; a[14] += wpz * qy; a[15] += wpz * qz;
    }
    // part (deterministic mode): chunk blockIdx.x's sums go to slab blockIdx.x; ogc_det_reduce_f64 adds the slabs in order
    if (part) block_sum_doubles(a, 16, smem, part + (((size_t)blockIdx.x * gridDim.z + vb) * k + slot) * 16, true);
    else block_sum_doubles(a, 16, smem, mom + ((size_t)vb * k + slot) * 16);
}

// one thread per (vb, slot): centred cross-covariance (float, for ogc_kabsch_rotation) and the weighted means
__global__ void rigid_finalize_kernel(int total, const double *__restrict__ mom, float *__restrict__ S,
                                      float *__restrict__ means /* [total][6] */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double *a = mom + (size_t)i * 16;
    const double W = a[0];
    double pm[3], qm[3];
    for (int d = 0; d < 3; ++d) { pm[d] = a[1 + d] / W; qm[d] = a[4 + d] / W; } // W == 0 -> NaN, as in the reference
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S[i * 9 + r * 3 + c] = (float)(a[7 + r * 3 + c] - a[1 + r] * qm[c]);
    for (int d = 0; d < 3; ++d) { means[i * 6 + d] = (float)pm[d]; means[i * 6 + 3 + d] = (float)qm[d]; }
}

// one thread per (vb, slot): t = qbar - R pbar; invalid fits (NaN covariance) -> identity, zero translation (:40-42)
__global__ void rigid_translation_kernel(int total, const float *__restrict__ means, const int *__restrict__ valid,
                                         float *__restrict__ R, float *__restrict__ t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float *Ri = R + i * 9;
    if (!valid[i]) {
        for (int e = 0; e < 9; ++e) Ri[e] = (e % 4 == 0) ? 1.0f : 0.0f;
        t[i * 3] = t[i * 3 + 1] = t[i * 3 + 2] = 0.0f;
        return;
    }
    const float *pm = means + i * 6, *qm = pm + 3;
    for (int r = 0; r < 3; ++r)
        t[i * 3 + r] = qm[r] - ((Ri[r * 3] * pm[0] + Ri[r * 3 + 1] * pm[1]) + Ri[r * 3 + 2] * pm[2]);
}

// grid (chunks, VB): per point ||sum_k m_k (R_k p + t_k) - q||_P.  BWD: grad_mask instead.
template <int P, bool BWD>
__global__ __launch_bounds__(SL_THREADS) void rigid_blend_kernel(int n, int k, const float *__restrict__ pc,
                                                                 const float *__restrict__ pc2,
                                                                 const float *__restrict__ mask,
                                                                 const float *__restrict__ R, const float *__restrict__ t,
                                                                 const float *__restrict__ grad_out,
                                                                 float *__restrict__ out /* fwd: (VB,N); bwd: (VB,N,K) */) {
    __shared__ float s_rt[SL_KMAX * 12];
    const int vb = blockIdx.y;
    for (int e = threadIdx.x; e < k * 12; e += SL_THREADS) {
        const int slot = e / 12, c = e % 12;
        s_rt[e] = c < 9 ? R[((size_t)vb * k + slot) * 9 + c] : t[((size_t)vb * k + slot) * 3 + (c - 9)];
    }
    __syncthreads();
    const int i = blockIdx.x * SL_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t pi = (size_t)vb * n + i;
    const float px = pc[pi * 3], py = pc[pi * 3 + 1], pz = pc[pi * 3 + 2];
    const float *m = mask + pi * k;
    float bx = 0.f, by = 0.f, bz = 0.f;
    for (int s = 0; s < k; ++s) {
        const float *rt = s_rt + s * 12;
        const float tx = (rt[0] * px + rt[1] * py) + rt[2] * pz + rt[9];
        const float ty = (rt[3] * px + rt[4] * py) + rt[5] * pz + rt[10];
        const float tz = (rt[6] * px + rt[7] * py) + rt[8] * pz + rt[11];
        const float w = m[s];
        bx = fmaf(w, tx, bx); by = fmaf(w, ty, by); bz = fmaf(w, tz, bz);
    }
    const float dx = bx - pc2[pi * 3], dy = by - pc2[pi * 3 + 1], dz = bz - pc2[pi * 3 + 2];
    if (P == 0) { // the difference itself (object-aware ICP: the mask-blended rigid flow, oa_icp.py:31-38 with pc2 = pc)
        out[pi * 3] = dx; out[pi * 3 + 1] = dy; out[pi * 3 + 2] = dz;
        return;
    }
    if (!BWD) {
        out[pi] = P == 1 ? (fabsf(dx) + fabsf(dy)) + fabsf(dz) : sqrtf((dx * dx + dy * dy) + dz * dz);
        return;
    }
    float gx, gy, gz; // d norm / d blended  (torch: sign(0) = 0; zero gradient at zero norm)
    if (P == 1) {
        gx = (float)((dx > 0.f) - (dx < 0.f)); gy = (float)((dy > 0.f) - (dy < 0.f)); gz = (float)((dz > 0.f) - (dz < 0.f));
    } else {
        const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float inv = nrm > 0.f ? 1.0f / nrm : 0.f;
        gx = dx * inv; gy = dy * inv; gz = dz * inv;
    }
    const float g = grad_out[pi];
    gx *= g; gy *= g; gz *= g;
    float *o = out + pi * k;
    for (int s = 0; s < k; ++s) {
        const float *rt = s_rt + s * 12;
        const float tx = (rt[0] * px + rt[1] * py) + rt[2] * pz + rt[9];
        const float ty = (rt[3] * px + rt[4] * py) + rt[5] * pz + rt[10];
        const float tz = (rt[6] * px + rt[7] * py) + rt[8] * pz + rt[11];
        o[s] = (gx * tx + gy * ty) + gz * tz;
    }
}

// grid (chunks, PB): counts[pb][g][p] += 1 for (g, p) = (argmax m1, argmax m2) of every point (first maximum wins,
// like torch.argmax), privatised in LDS; mask_iou_kernel turns the counts into IoUs
__global__ __launch_bounds__(SL_THREADS) void mask_confusion_kernel(int n, int k, const float *__restrict__ m1,
                                                                    const float *__restrict__ m2,
                                                                    int *__restrict__ counts) {
    __shared__ int s_cnt[SL_KMAX * SL_KMAX];
    const int pb = blockIdx.y;
    for (int e = threadIdx.x; e < k * k; e += SL_THREADS) s_cnt[e] = 0;
    __syncthreads();
    for (int i = blockIdx.x * SL_THREADS + threadIdx.x; i < n; i += gridDim.x * SL_THREADS) {
        const float *a = m1 + ((size_t)pb * n + i) * k, *b = m2 + ((size_t)pb * n + i) * k;
        int ga = 0, gb = 0;
        float va = a[0], vb = b[0];
        for (int s = 1; s < k; ++s) {
            if (a[s] > va) { va = a[s]; ga = s; }
            if (b[s] > vb) { vb = b[s]; gb = s; }
        }
        atomicAdd(&s_cnt[ga * k + gb], 1);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < k * k; e += SL_THREADS)
        if (s_cnt[e]) atomicAdd(&counts[(size_t)pb * k * k + e], s_cnt[e]);
}

// one thread per (pb, g, p): iou = inter / clamp(rows[g] + cols[p] - inter, 1e-10), in fp32 like the reference
__global__ void mask_iou_kernel(int total, int k, const int *__restrict__ counts, float *__restrict__ iou) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int pb = e / (k * k), g = (e / k) % k, p = e % k;
    const int *c = counts + (size_t)pb * k * k;
    int row = 0, col = 0;
    for (int s = 0; s < k; ++s) { row += c[g * k + s]; col += c[s * k + p]; }
    const float inter = (float)c[g * k + p];
    const float uni = ((float)row + (float)col) - inter;
    iou[e] = inter / fmaxf(uni, 1e-10f);
}

// grid (chunks, PB).  d12[n] = ||m1[n] - m2[n, col12]||_P, d21[n] = ||m2[n] - m1[n, col21]||_P   (targets detached)
template <int P, bool BWD>
__global__ __launch_bounds__(SL_THREADS) void matched_distance_kernel(int n, int k, const float *__restrict__ m1,
                                                                      const float *__restrict__ m2,
                                                                      const int *__restrict__ col12,
                                                                      const int *__restrict__ col21,
                                                                      const float *__restrict__ g12,
                                                                      const float *__restrict__ g21,
                                                                      float *__restrict__ o1, float *__restrict__ o2) {
    __shared__ int s_col[2 * SL_KMAX];
    const int pb = blockIdx.y;
    if ((int)threadIdx.x < k) {
        s_col[threadIdx.x] = col12[(size_t)pb * k + threadIdx.x];
        s_col[SL_KMAX + threadIdx.x] = col21[(size_t)pb * k + threadIdx.x];
    }
    __syncthreads();
    const int i = blockIdx.x * SL_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t pi = (size_t)pb * n + i;
    const float *a = m1 + pi * k, *b = m2 + pi * k;
    float s12 = 0.f, s21 = 0.f;
    for (int s = 0; s < k; ++s) {
        const float d12 = a[s] - b[s_col[s]], d21 = b[s] - a[s_col[SL_KMAX + s]];
        s12 += P == 1 ? fabsf(d12) : d12 * d12;
        s21 += P == 1 ? fabsf(d21) : d21 * d21;
    }
    if (P == 2) { s12 = sqrtf(s12); s21 = sqrtf(s21); }
    if (!BWD) {
        o1[pi] = s12;
        o2[pi] = s21;
        return;
    }
    const float w12 = P == 1 ? g12[pi] : (s12 > 0.f ? g12[pi] / s12 : 0.f);
    const float w21 = P == 1 ? g21[pi] : (s21 > 0.f ? g21[pi] / s21 : 0.f);
    float *ga = o1 + pi * k, *gb = o2 + pi * k;
    for (int s = 0; s < k; ++s) {
        const float d12 = a[s] - b[s_col[s]], d21 = b[s] - a[s_col[SL_KMAX + s]];
        ga[s] = P == 1 ? w12 * (float)((d12 > 0.f) - (d12 < 0.f)) : w12 * d12;
        gb[s] = P == 1 ? w21 * (float)((d21 > 0.f) - (d21 < 0.f)) : w21 * d21;
    }
}

int sl_chunks(int n, int rows) {
    int chunks = ogc_divup(n, SL_THREADS);
    while (chunks > 1 && (long long)chunks * rows > 4096) chunks = (chunks + 1) / 2;
    return chunks;
}

} // namespace

extern "C" int ogc_rigid_moments(int vb, int n, int k, const float *pc, const float *pc2, const float *mask, double *mom,
                                 float *S, float *means, ogc_stream_t stream) {
    OGC_REQUIRE(vb >= 0 && n >= 0 && k >= 1, "ogc_rigid_moments: bad shape");
    if (vb == 0) return OGC_OK;
    OGC_REQUIRE(pc && pc2 && mask && mom && S && means, "ogc_rigid_moments: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (ogc_zero_async(mom, sizeof(double) * 16 * (size_t)vb * k, s) != hipSuccess) {
        ogc_set_error("ogc_rigid_moments: memset failed");
        return OGC_ERR_LAUNCH;
    }
    if (n > 0) {
        const int chunks = sl_chunks(n, vb * k);
        double *part = nullptr;
        if (ogc_deterministic()) {
            part = static_cast<double *>(ogc_det_scratch(s, sizeof(double) * 16 * (size_t)vb * k * chunks));
            if (!part) {
                ogc_set_error("ogc_rigid_moments (deterministic): no scratch memory");
                return OGC_ERR_LAUNCH;
            }
        }
        hipLaunchKernelGGL(rigid_moments_kernel, dim3(chunks, k, vb), dim3(SL_THREADS), 0, s, n, k, pc, pc2, mask, mom, part);
        if (part && ogc_det_reduce_f64(mom, part, chunks, 16ll * vb * k, 0, s) != hipSuccess) return OGC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(rigid_finalize_kernel, dim3(ogc_divup(vb * k, 64)), dim3(64), 0, s, vb * k, mom, S, means);
    OGC_CHECK_LAUNCH("ogc_rigid_moments");
    return OGC_OK;
}

extern "C" int ogc_rigid_translation(int total, const float *means, const int *valid, float *R, float *t,
                                     ogc_stream_t stream) {
    OGC_REQUIRE(total >= 0, "ogc_rigid_translation: bad shape");
    if (total == 0) return OGC_OK;
    OGC_REQUIRE(means && valid && R && t, "ogc_rigid_translation: null pointer");
    hipLaunchKernelGGL(rigid_translation_kernel, dim3(ogc_divup(total, 64)), dim3(64), 0, (hipStream_t)stream, total, means,
                       valid, R, t);
    OGC_CHECK_LAUNCH("ogc_rigid_translation");
    return OGC_OK;
}

extern "C" int ogc_rigid_blend(int vb, int n, int k, int p, int backward, const float *pc, const float *pc2,
                               const float *mask, const float *R, const float *t, const float *grad_out, float *out,
                               ogc_stream_t stream) {
    OGC_REQUIRE(vb >= 0 && n >= 0 && k >= 1, "ogc_rigid_blend: bad shape");
    OGC_REQUIRE(p == 1 || p == 2 || (p == 0 && !backward), "ogc_rigid_blend: norm must be 1 or 2 (0: the difference vector, forward only)");
    if (k > SL_KMAX) {
        ogc_set_error("ogc_rigid_blend: more than %d slots", SL_KMAX);
        return OGC_ERR_UNSUPPORTED;
    }
    if (vb == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(pc && pc2 && mask && R && t && out && (!backward || grad_out), "ogc_rigid_blend: null pointer");
    const dim3 grid(ogc_divup(n, SL_THREADS), vb), block(SL_THREADS);
    hipStream_t s = (hipStream_t)stream;
#define OGC_BLEND(PV, BV) \
    hipLaunchKernelGGL((rigid_blend_kernel<PV, BV>), grid, block, 0, s, n, k, pc, pc2, mask, R, t, grad_out, out)
    if (p == 0) OGC_BLEND(0, false);
    else if (p == 1 && !backward) OGC_BLEND(1, false);
    else if (p == 1) OGC_BLEND(1, true);
    else if (!backward) OGC_BLEND(2, false);
    else OGC_BLEND(2, true);
#undef OGC_BLEND
    OGC_CHECK_LAUNCH("ogc_rigid_blend");
    return OGC_OK;
}

extern "C" int ogc_mask_iou(int pb, int n, int k, const float *mask1, const float *mask2, int *counts, float *iou,
                            ogc_stream_t stream) {
    OGC_REQUIRE(pb >= 0 && n >= 0 && k >= 1, "ogc_mask_iou: bad shape");
    if (k > SL_KMAX) {
        ogc_set_error("ogc_mask_iou: more than %d slots", SL_KMAX);
        return OGC_ERR_UNSUPPORTED;
    }
    if (pb == 0) return OGC_OK;
    OGC_REQUIRE(mask1 && mask2 && counts && iou, "ogc_mask_iou: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (ogc_zero_async(counts, sizeof(int) * (size_t)pb * k * k, s) != hipSuccess) {
        ogc_set_error("ogc_mask_iou: memset failed");
        return OGC_ERR_LAUNCH;
    }
    if (n > 0)
        hipLaunchKernelGGL(mask_confusion_kernel, dim3(sl_chunks(n, pb), pb), dim3(SL_THREADS), 0, s, n, k, mask1, mask2,
                           counts);
    hipLaunchKernelGGL(mask_iou_kernel, dim3(ogc_divup(pb * k * k, 256)), dim3(256), 0, s, pb * k * k, k, counts, iou);
    OGC_CHECK_LAUNCH("ogc_mask_iou");
    return OGC_OK;
}

extern "C" int ogc_matched_distance(int pb, int n, int k, int p, int backward, const float *mask1, const float *mask2,
                                    const int *col12, const int *col21, const float *grad12, const float *grad21,
                                    float *out1, float *out2, ogc_stream_t stream) {
    OGC_REQUIRE(pb >= 0 && n >= 0 && k >= 1, "ogc_matched_distance: bad shape");
    OGC_REQUIRE(p == 1 || p == 2, "ogc_matched_distance: norm must be 1 or 2");
    if (k > SL_KMAX) {
        ogc_set_error("ogc_matched_distance: more than %d slots", SL_KMAX);
        return OGC_ERR_UNSUPPORTED;
    }
    if (pb == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(mask1 && mask2 && col12 && col21 && out1 && out2 && (!backward || (grad12 && grad21)),
                "ogc_matched_distance: null pointer");
    const dim3 grid(ogc_divup(n, SL_THREADS), pb), block(SL_THREADS);
    hipStream_t s = (hipStream_t)stream;
#define OGC_MD(PV, BV)                                                                                              \
    hipLaunchKernelGGL((matched_distance_kernel<PV, BV>), grid, block, 0, s, n, k, mask1, mask2, col12, col21, grad12, \
                       grad21, out1, out2)
    if (p == 1 && !backward) OGC_MD(1, false);
    else if (p == 1) OGC_MD(1, true);
    else if (!backward) OGC_MD(2, false);
    else OGC_MD(2, true);
#undef OGC_MD
    OGC_CHECK_LAUNCH("ogc_matched_distance");
    return OGC_OK;
}
