// slot_masks.hip — the mask read-out of the segmentation nets, forward and backward.
//
// Reference: models/segnet_kitti.py:85-88 (same lines in segnet_sapien.py / segnet_ogcdr.py :77-80)
//     mask = einsum('bdn,bdk->bnk', F.normalize(feats, dim=1), F.normalize(slot, dim=1)) / 0.05;  mask.softmax(-1)
// feats (B, D, N) are the per-point features of the finest level (D = 64, N = 8192), slot (B, D, K) the K object
// embeddings.  As framework ops that is 9 launches forward and ~25 backward, most of them element-wise passes over the
// (B, D, N) tensor or its gradient (the two normalisations' adjoints alone read / write it nine times, and the
// gradient arrives transposed, so a contiguous copy follows): ~0.1 ms forward and ~0.35 ms backward of a 15.8 ms C4
// step for 33.5 MB of input.  Here each direction reads feats once:
//
// forward : thread per point.  The workgroup first normalises the K slot columns into LDS (D x KT floats), then each
//           thread streams its point's D features (coalesced over the points), accumulating the squared norm and the K
//           dot products, and finishes with the temperature and the softmax over K in registers.
// backward: thread per point, two sweeps over D.  Sweep 1 rebuilds the norm and the cosines c_k; with p the saved
//           probabilities and g their gradient,  gl_k = p_k (g_k - sum_j p_j g_j) / T  is the gradient of c_k, and
//               d feats_d = (sum_k gl_k s^_dk  -  f^_d sum_k gl_k c_k) / |f|            (sweep 2, L2-resident re-read)
//           The slot gradient  d s^_dk = sum_n gl_nk f^_nd  is reduced per workgroup through LDS (16 feature rows at a
//           time: points x rows and points x K tiles, thread per (row, k) pair) into per-workgroup partial sums; a
//           second small kernel adds the partials in a fixed order (no atomics: results repeat bit for bit) and applies
//           the adjoint of the slot normalisation.
// F.normalize clamps the norm at eps = 1e-12; below the clamp its adjoint treats the denominator as a constant, as
// autograd's clamp_min does.
#include "ogc_common.h"

namespace {

constexpr int SM_THREADS = 128;   // points per workgroup
constexpr int SM_ROWS = 16;       // feature rows per reduction chunk of the backward pass
constexpr int SM_PAD = SM_THREADS + 1;
constexpr float SM_EPS = 1e-12f;

// s^ = slot / max(|slot|, eps) per column, into LDS as sh[dd * KT + kk].  nrm: KT + SM_THREADS floats of scratch
// (column norms, then the partial sums of squares of SM_THREADS / KT interleaved row subsets).
template <int KT>
__device__ __forceinline__ void sm_normalise_slots(int d, int k, const float *__restrict__ slot, float *sh, float *nrm) {
    constexpr int PARTS = SM_THREADS / KT;
    const int t = threadIdx.x, kk = t % KT, part = t / KT;
    float *tmp = nrm + KT;
    for (int o = t; o < d * KT; o += SM_THREADS) sh[o] = (o % KT) < k ? slot[(size_t)(o / KT) * k + (o % KT)] : 0.f;
    __syncthreads();
    float ss = 0.f;
    for (int dd = part; dd < d; dd += PARTS) ss = fmaf(sh[dd * KT + kk], sh[dd * KT + kk], ss);
    tmp[t] = ss;
    __syncthreads();
    if (t < KT) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < PARTS; ++j) tot += tmp[j * KT + t];
        nrm[t] = fmaxf(sqrtf(tot), SM_EPS);
    }
    __syncthreads();
    for (int o = t; o < d * KT; o += SM_THREADS) sh[o] = sh[o] / nrm[o % KT];
    __syncthreads();
}

template <int KT>
__global__ __launch_bounds__(SM_THREADS) void slot_masks_fwd_kernel(int d, int n, int k, float temp,
                                                                   const float *__restrict__ feats,
                                                                   const float *__restrict__ slots,
                                                                   float *__restrict__ mask) {
    extern __shared__ float sm_lds[];
    float *sh = sm_lds, *nrm = sm_lds + (size_t)d * KT;
    const int bi = blockIdx.y;
    sm_normalise_slots<KT>(d, k, slots + (size_t)bi * d * k, sh, nrm);
    const int p = blockIdx.x * SM_THREADS + threadIdx.x;
    if (p >= n) return;
    const float *fp = feats + (size_t)bi * d * n + p;
    float ss = 0.f, dot[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) dot[kk] = 0.f;
#pragma unroll 16
    for (int dd = 0; dd < d; ++dd) {
        const float f = fp[(size_t)dd * n];
        ss = fmaf(f, f, ss);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) dot[kk] = fmaf(f, sh[dd * KT + kk], dot[kk]);
    }
    const float fn = fmaxf(sqrtf(ss), SM_EPS);
    float mx = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        dot[kk] = dot[kk] / fn / temp;
        if (kk < k) mx = fmaxf(mx, dot[kk]);
    }
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        dot[kk] = kk < k ? __expf(dot[kk] - mx) : 0.f;
        sum += dot[kk];
    }
    float *mp = mask + ((size_t)bi * n + p) * k;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
        if (kk < k) mp[kk] = dot[kk] / sum;
}

template <int KT>
__global__ __launch_bounds__(SM_THREADS) void slot_masks_bwd_kernel(int d, int n, int k, float temp,
                                                                   const float *__restrict__ feats,
                                                                   const float *__restrict__ slots,
                                                                   const float *__restrict__ mask,
                                                                   const float *__restrict__ gmask,
                                                                   float *__restrict__ gfeats,
                                                                   float *__restrict__ partial) {
    extern __shared__ float sm_lds[];
    float *sh = sm_lds, *nrm = sh + (size_t)d * KT;   // d*KT, KT
    float *fh = nrm + KT + SM_THREADS;                 // SM_ROWS x SM_PAD   f^ of this chunk's rows
    float *gls = fh + SM_ROWS * SM_PAD;                // KT x SM_PAD        gl of every point of the workgroup
    const int bi = blockIdx.y, t = threadIdx.x;
    sm_normalise_slots<KT>(d, k, slots + (size_t)bi * d * k, sh, nrm);
    const int p = blockIdx.x * SM_THREADS + t;
    const bool live = p < n;
    const float *fp = feats + (size_t)bi * d * n + (live ? p : 0);
    float ss = 0.f, gl[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) gl[kk] = 0.f;
    if (live) {
#pragma unroll 16
        for (int dd = 0; dd < d; ++dd) {
            const float f = fp[(size_t)dd * n];
            ss = fmaf(f, f, ss);
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) gl[kk] = fmaf(f, sh[dd * KT + kk], gl[kk]);
        }
    }
    const float raw = sqrtf(ss), fn = fmaxf(raw, SM_EPS);
    float tdot = 0.f;   // sum_k gl_k c_k
    if (live) {
        const float *mp = mask + ((size_t)bi * n + p) * k, *gp = gmask + ((size_t)bi * n + p) * k;
        float pg = 0.f, pr[KT], gr[KT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            pr[kk] = kk < k ? mp[kk] : 0.f;
            gr[kk] = kk < k ? gp[kk] : 0.f;
            pg = fmaf(pr[kk], gr[kk], pg);
        }
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const float c = gl[kk] / fn;   // cosine
            gl[kk] = pr[kk] * (gr[kk] - pg) / temp;
            tdot = fmaf(gl[kk], c, tdot);
        }
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) gls[kk * SM_PAD + t] = gl[kk];
    // below the clamp the normalisation is x / eps with a constant denominator
    const float proj = raw > SM_EPS ? tdot : 0.f;
    float *gfp = gfeats + (size_t)bi * d * n + (live ? p : 0);
    float *part = partial + ((size_t)bi * gridDim.x + blockIdx.x) * d * KT;
    for (int d0 = 0; d0 < d; d0 += SM_ROWS) {
        const int rows = min(SM_ROWS, d - d0);
        float fv[SM_ROWS];
#pragma unroll
        for (int r = 0; r < SM_ROWS; ++r) fv[r] = (live && r < rows) ? fp[(size_t)(d0 + r) * n] : 0.f;
#pragma unroll
        for (int r = 0; r < SM_ROWS; ++r) {
            const float fhat = fv[r] / fn;
            if (live && r < rows) {
                const int dd = d0 + r;
                float acc = 0.f;
#pragma unroll
                for (int kk = 0; kk < KT; ++kk) acc = fmaf(gl[kk], sh[dd * KT + kk], acc);
                gfp[(size_t)dd * n] = (acc - fhat * proj) / fn;
            }
            fh[r * SM_PAD + t] = fhat;
        }
        __syncthreads();
        for (int o = t; o < rows * KT; o += SM_THREADS) {
            const int r = o / KT, kk = o % KT;
            const float *a = fh + r * SM_PAD, *g = gls + kk * SM_PAD;
            float acc = 0.f;
#pragma unroll 8
            for (int q = 0; q < SM_THREADS; ++q) acc = fmaf(a[q], g[q], acc);
            part[(size_t)(d0 + r) * KT + kk] = acc;
        }
        __syncthreads();
    }
}

// d slot = adjoint of the column normalisation applied to the sum of the workgroups' partial d s^
constexpr int SF_THREADS = 1024;

template <int KT>
__global__ __launch_bounds__(SF_THREADS) void slot_masks_finish_kernel(int d, int k, int nblk,
                                                                      const float *__restrict__ slots,
                                                                      const float *__restrict__ partial,
                                                                      float *__restrict__ gslots) {
    constexpr int PARTS = SF_THREADS / KT;
    extern __shared__ float sm_lds[];
    float *gs = sm_lds;                  // d*KT  summed d s^
    float *raw = gs + (size_t)d * KT;    // KT    |s_k|
    float *dotk = raw + KT;              // KT    s^_k . d s^_k
    float *tss = dotk + KT;              // SF_THREADS partial sums of squares
    float *tdt = tss + SF_THREADS;       // SF_THREADS partial s . d s^
    const int bi = blockIdx.x, t = threadIdx.x, kk = t % KT, part = t / KT;
    const float *sp = slots + (size_t)bi * d * k;
    const float *pp = partial + (size_t)bi * nblk * d * KT;
    for (int o = t; o < d * KT; o += SF_THREADS) {
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < nblk; ++j) acc += pp[(size_t)j * d * KT + o];
        gs[o] = acc;
    }
    __syncthreads();
    float ss = 0.f, dt = 0.f;
    if (kk < k)
        for (int dd = part; dd < d; dd += PARTS) {
            const float v = sp[(size_t)dd * k + kk];
            ss = fmaf(v, v, ss);
            dt = fmaf(v, gs[dd * KT + kk], dt);
        }
    tss[t] = ss;
    tdt[t] = dt;
    __syncthreads();
    if (t < KT) {
        float a = 0.f, c = 0.f;
        for (int j = 0; j < PARTS; ++j) {
            a += tss[j * KT + t];
            c += tdt[j * KT + t];
        }
        raw[t] = sqrtf(a);
        dotk[t] = c / fmaxf(raw[t], SM_EPS);
    }
    __syncthreads();
    for (int o = t; o < d * k; o += SF_THREADS) {
        const int dd = o / k, kq = o % k;
        const float nr = fmaxf(raw[kq], SM_EPS);
        const float shat = sp[o] / nr;
        const float proj = raw[kq] > SM_EPS ? dotk[kq] : 0.f;
        gslots[(size_t)bi * d * k + o] = (gs[dd * KT + kq] - shat * proj) / nr;
    }
}

int sm_check(const char *who, int b, int d, int n, int k, float temp) {
    if (b < 0 || d <= 0 || n <= 0 || k <= 0 || !(temp > 0.f)) {
        ogc_set_error("%s: bad sizes b=%d d=%d n=%d k=%d temperature=%g", who, b, d, n, k, (double)temp);
        return OGC_ERR_INVALID_ARG;
    }
    if (k > 32 || d > 256) {
        ogc_set_error("%s: k=%d slots (max 32) / d=%d features (max 256) not supported", who, k, d);
        return OGC_ERR_UNSUPPORTED;
    }
    return OGC_OK;
}

inline int sm_kt(int k) { return k <= 16 ? 16 : 32; }

} // namespace

extern "C" long long ogc_slot_masks_ws_floats(int b, int d, int n, int k) {
    if (b <= 0 || d <= 0 || n <= 0 || k <= 0) return 0;
    return (long long)b * ogc_divup(n, SM_THREADS) * d * sm_kt(k);
}

extern "C" int ogc_slot_masks_fwd(int b, int d, int n, int k, float temperature, const float *feats,
                                  const float *slots, float *mask, ogc_stream_t stream) {
    const int rc = sm_check("ogc_slot_masks_fwd", b, d, n, k, temperature);
    if (rc != OGC_OK) return rc;
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(feats && slots && mask, "ogc_slot_masks_fwd: null pointer");
    const dim3 grid(ogc_divup(n, SM_THREADS), b), block(SM_THREADS);
    hipStream_t s = (hipStream_t)stream;
    const int kt = sm_kt(k);
    const size_t lds = ((size_t)d * kt + kt + SM_THREADS) * sizeof(float);
    if (kt == 16)
        hipLaunchKernelGGL(slot_masks_fwd_kernel<16>, grid, block, lds, s, d, n, k, temperature, feats, slots, mask);
    else
        hipLaunchKernelGGL(slot_masks_fwd_kernel<32>, grid, block, lds, s, d, n, k, temperature, feats, slots, mask);
    OGC_CHECK_LAUNCH("ogc_slot_masks_fwd");
    return OGC_OK;
}

extern "C" int ogc_slot_masks_bwd(int b, int d, int n, int k, float temperature, const float *feats,
                                  const float *slots, const float *mask, const float *grad_mask, float *grad_feats,
                                  float *grad_slots, float *ws, ogc_stream_t stream) {
    const int rc = sm_check("ogc_slot_masks_bwd", b, d, n, k, temperature);
    if (rc != OGC_OK) return rc;
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(feats && slots && mask && grad_mask && grad_feats && grad_slots && ws,
                "ogc_slot_masks_bwd: null pointer");
    const int nblk = ogc_divup(n, SM_THREADS);
    const dim3 grid(nblk, b), block(SM_THREADS);
    hipStream_t s = (hipStream_t)stream;
    const int kt = sm_kt(k);
    const size_t lds = ((size_t)d * kt + kt + SM_THREADS + (size_t)SM_ROWS * SM_PAD + (size_t)kt * SM_PAD) * sizeof(float);
    const size_t lds2 = ((size_t)d * kt + 2 * kt + 2 * SF_THREADS) * sizeof(float);
    if (kt == 16) {
        hipLaunchKernelGGL(slot_masks_bwd_kernel<16>, grid, block, lds, s, d, n, k, temperature, feats, slots, mask,
                           grad_mask, grad_feats, ws);
        hipLaunchKernelGGL(slot_masks_finish_kernel<16>, dim3(b), dim3(SF_THREADS), lds2, s, d, k, nblk, slots, ws, grad_slots);
    } else {
        hipLaunchKernelGGL(slot_masks_bwd_kernel<32>, grid, block, lds, s, d, n, k, temperature, feats, slots, mask,
                           grad_mask, grad_feats, ws);
        hipLaunchKernelGGL(slot_masks_finish_kernel<32>, dim3(b), dim3(SF_THREADS), lds2, s, d, k, nblk, slots, ws, grad_slots);
    }
    OGC_CHECK_LAUNCH("ogc_slot_masks_bwd");
    return OGC_OK;
}
