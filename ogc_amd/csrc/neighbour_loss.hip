// neighbour_loss.hip — the smoothness term of the OGC loss as one forward and one backward launch.
//
// Reference (losses/seg_loss_unsup.py:123-129 kNN, :152-158 ball query):
//     nn_mask = grouping_operation(mask, idx)                      # (B, C, N, k) gather — 335 MB at C4 for the ball term
//     loss    = (mask.unsqueeze(3) - nn_mask).norm(p, dim=1).mean(dim=-1)   # (B, N)
// i.e. per point i:  s_i = (1/k) sum_j || m_i - m_idx[i,j] ||_p  over C channels, p in {1, 2}.
// The reference's autograd materialises the gathered tensor, its difference, sign/abs and the scatter-add of the
// gather's gradient: ~60 launches and ~1.4 GB of traffic per step for values that need 5 MB of masks and 50 MB of
// indices.  Here:
//   ogc_neighbour_consistency_fwd   eight lanes per point stride over the k neighbours; masks are (B, N, C)
//                                   point-major, so one neighbour is one contiguous C-float read (L2 resident);
//   ogc_reverse_neighbours          coordinate-only: the transposed neighbour lists (CSR of incoming edges), so that
//   ogc_neighbour_consistency_bwd   the gradient is a GATHER over out- and in-edges — no atomics on the masks:
//       dL/dm_i = (1/k) [ g_i sum_j d(m_i, m_idx[i,j])  -  sum_{i' : i in idx[i']} g_i' d(m_i', m_i) ],
//       d(a, b) = sign(a - b) for p = 1 (sign(0) = 0), (a - b) / ||a - b||_2 for p = 2 (0 where the norm is 0),
//     which are the subgradients torch's norm backward uses.
#include "ogc_common.h"

namespace {

constexpr int SUBL = 8; // lanes per point

__device__ __forceinline__ float sgn(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }

template <int P>
__global__ __launch_bounds__(256) void nc_fwd_kernel(int n, int c, int k, const float *__restrict__ mask,
                                                     const int *__restrict__ idx, float *__restrict__ out) {
    const int b = blockIdx.y;
    const int sub = threadIdx.x & (SUBL - 1);
    const int i = blockIdx.x * (256 / SUBL) + (threadIdx.x >> 3);
    const float *mb = mask + (size_t)b * n * c;
    float acc = 0.0f;
    if (i < n) {
        const float *mi = mb + (size_t)i * c;
        const int *row = idx + ((size_t)b * n + i) * k;
        for (int j = sub; j < k; j += SUBL) {
            const float *mj = mb + (size_t)row[j] * c;
            float s = 0.0f;
            for (int ch = 0; ch < c; ++ch) {
                const float d = mi[ch] - mj[ch];
                s += P == 1 ? fabsf(d) : d * d;
            }
            acc += P == 1 ? s : sqrtf(s);
        }
    }
#pragma unroll
    for (int off = 1; off < SUBL; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (i < n && sub == 0) out[(size_t)b * n + i] = acc / (float)k;
}

// Transposed lists.  Rows produced by the ball query / radius-clamped kNN are "hits..., then copies of the first
// entry" and a row usually contains the point itself: a self edge has d(m_i, m_i) = 0 and is dropped, and the copies
// of a row's first entry are merged into ONE edge (flag bit 31 of rev_src) whose weight is the row's multiplicity
// mult[i] — otherwise thousands of edges pile up on a few low-index points and the atomics below serialise on them.
constexpr unsigned REV_FIRST = 0x80000000u;

__device__ __forceinline__ bool rev_keep(const int *__restrict__ idx, long long e, int n, int k, int &dst, int &src,
                                         bool &first_dup) {
    const int j = (int)(e % k);
    src = (int)((e / k) % n);
    dst = idx[e];
    const int first = idx[e - j];
    first_dup = j > 0 && dst == first;
    return dst != src && !first_dup;
}

// ROWWISE: k is a power of two <= 64, so a row is a group of k consecutive lanes of one wavefront: the row's
// multiplicity (1 + its padding copies) is one ballot + popcount instead of up to k-1 atomics on the same address —
// a ball-query row at the C4 loss shape holds ~14 hits and ~50 copies, and those same-address atomics were the whole
// cost of this kernel.  mult is written for every row, so it needs no initialisation on this path.
template <bool ROWWISE>
__global__ __launch_bounds__(256) void rev_count_kernel(long long total, int n, int k, const int *__restrict__ idx,
                                                        int *__restrict__ deg, int *__restrict__ mult) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in = e < total;
    int dst = 0, src = 0;
    bool dup = false, keep = false;
    long long b = 0;
    if (in) {
        b = e / ((long long)n * k);
        keep = rev_keep(idx, e, n, k, dst, src, dup);
    }
    if (keep) atomicAdd(&deg[b * (n + 1) + dst], 1);
    const bool extra = in && !keep && dup && dst != src;
    if (ROWWISE) {
        const unsigned long long copies = __builtin_amdgcn_ballot_w64(extra);
        const int lane = threadIdx.x & 63, row0 = lane & ~(k - 1);
        if (in && lane == row0) {
            const unsigned long long row_mask = k == 64 ? ~0ull : ((1ull << k) - 1ull) << row0;
            mult[b * n + src] = 1 + __popcll(copies & row_mask);
        }
    } else if (extra) {
        atomicAdd(&mult[b * n + src], 1);
    }
}

__global__ __launch_bounds__(1024) void rev_scan_kernel(int n, int *__restrict__ deg_to_start, int *__restrict__ cursor) {
    // one workgroup per cloud: exclusive scan of n counts (in place, n+1 entries), and a copy as the fill cursor
    __shared__ int part[1024];
    const int t = threadIdx.x, b = blockIdx.x;
    int *d = deg_to_start + (size_t)b * (n + 1);
    int *cur = cursor + (size_t)b * n;
    const int per = (n + 1023) / 1024;
    const int c0 = min(t * per, n), c1 = min(c0 + per, n);
    int sum = 0;
    for (int j = c0; j < c1; ++j) sum += d[j];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t > 0 ? part[t - 1] : 0;
    for (int j = c0; j < c1; ++j) {
        const int cnt = d[j];
        d[j] = run;
        cur[j] = run;
        run += cnt;
    }
    if (t == 1023) d[n] = part[1023];
}

__global__ __launch_bounds__(256) void rev_fill_kernel(long long total, int n, int k, const int *__restrict__ idx,
                                                       int *__restrict__ cursor, int *__restrict__ rev_src) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const long long b = e / ((long long)n * k);
    int dst, src;
    bool dup;
    if (!rev_keep(idx, e, n, k, dst, src, dup)) return;
    const int pos = atomicAdd(&cursor[b * n + dst], 1);
    rev_src[b * (long long)n * k + pos] = (int)((unsigned)src | (e % k == 0 ? REV_FIRST : 0u));
}

// deterministic mode: the slot an entry got inside its list was a race between the lanes of rev_fill_kernel — every list
// ascending by source point (the flag bit rides along), one thread per destination, in place
__global__ __launch_bounds__(256) void rev_sort_kernel(int n, int k, const int *__restrict__ rev_start, int *__restrict__ rev_src) {
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= n) return;
    const int *rs = rev_start + (size_t)b * (n + 1);
    int *a = rev_src + (size_t)b * n * k;
    const int lo = rs[j], hi = rs[j + 1];
    for (int i = lo + 1; i < hi; ++i) {
        const int v = a[i];
        const unsigned key = (unsigned)v & ~REV_FIRST;
        int q = i - 1;
        while (q >= lo && ((unsigned)a[q] & ~REV_FIRST) > key) {
            a[q + 1] = a[q];
            --q;
        }
        a[q + 1] = v;
    }
}

template <int P>
__device__ __forceinline__ void nc_edge(const float *a, const float *b_, int c, float w, float *g /*[c] in LDS*/,
                                        int stride) {
    // g += w * d(a, b)
    if (P == 1) {
        for (int ch = 0; ch < c; ++ch) g[ch * stride] += w * sgn(a[ch] - b_[ch]);
    } else {
        float s = 0.0f;
        for (int ch = 0; ch < c; ++ch) { const float d = a[ch] - b_[ch]; s += d * d; }
        const float nrm = sqrtf(s);
        if (nrm > 0.0f) {
            const float inv = w / nrm;
            for (int ch = 0; ch < c; ++ch) g[ch * stride] += inv * (a[ch] - b_[ch]);
        }
    }
}

template <int P>
__global__ __launch_bounds__(256) void nc_bwd_kernel(int n, int c, int k, const float *__restrict__ mask,
                                                     const int *__restrict__ idx, const int *__restrict__ rev_start,
                                                     const int *__restrict__ rev_src,
                                                     const int *__restrict__ rev_mult,
                                                     const float *__restrict__ grad_out,
                                                     float *__restrict__ grad_mask) {
    extern __shared__ float nc_smem[]; // [c][256] per-lane partial gradients
    const int b = blockIdx.y;
    const int sub = threadIdx.x & (SUBL - 1);
    const int i = blockIdx.x * (256 / SUBL) + (threadIdx.x >> 3);
    float *g = nc_smem + threadIdx.x;
    for (int ch = 0; ch < c; ++ch) g[ch * 256] = 0.0f;
    const float *mb = mask + (size_t)b * n * c;
    const float *go = grad_out + (size_t)b * n;
    if (i < n) {
        const float *mi = mb + (size_t)i * c;
        const float inv_k = 1.0f / (float)k;
        // out-edges: s_i depends on m_i through every neighbour
        const int *row = idx + ((size_t)b * n + i) * k;
        const float gi = go[i] * inv_k;
        for (int j = sub; j < k; j += SUBL) nc_edge<P>(mi, mb + (size_t)row[j] * c, c, gi, g, 256);
        // in-edges: s_i' depends on m_i when i is a neighbour of i'
        const int *rs = rev_start + (size_t)b * (n + 1);
        const int *src = rev_src + (size_t)b * n * k;
        for (int e = rs[i] + sub; e < rs[i + 1]; e += SUBL) {
            const unsigned raw = (unsigned)src[e];
            const int ip = (int)(raw & ~REV_FIRST);
            const float times = (raw & REV_FIRST) ? (float)rev_mult[(size_t)b * n + ip] : 1.0f;
            nc_edge<P>(mb + (size_t)ip * c, mi, c, -go[ip] * inv_k * times, g, 256);
        }
    }
    __syncthreads();
    // reduce the 8 lanes of a point and write C contiguous floats per point
    const int points = 256 / SUBL;
    for (int o = threadIdx.x; o < points * c; o += 256) {
        const int pl = o / c, ch = o % c;
        const int pi = blockIdx.x * points + pl;
        if (pi < n) {
            float s = 0.0f;
#pragma unroll
            for (int l = 0; l < SUBL; ++l) s += nc_smem[ch * 256 + pl * SUBL + l];
            grad_mask[((size_t)b * n + pi) * c + ch] = s;
        }
    }
}

} // namespace

extern "C" int ogc_neighbour_consistency_fwd(int b, int n, int c, int k, int p, const float *mask, const int *idx,
                                             float *out, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && c >= 0 && k >= 0, "ogc_neighbour_consistency_fwd: negative size");
    OGC_REQUIRE(p == 1 || p == 2, "ogc_neighbour_consistency_fwd: norm must be 1 or 2");
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(k > 0, "ogc_neighbour_consistency_fwd: k must be positive");
    OGC_REQUIRE(mask && idx && out, "ogc_neighbour_consistency_fwd: null pointer");
    const dim3 grid(ogc_divup(n, 256 / SUBL), b);
    if (p == 1) hipLaunchKernelGGL(nc_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, n, c, k, mask, idx, out);
    else hipLaunchKernelGGL(nc_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, n, c, k, mask, idx, out);
    OGC_CHECK_LAUNCH("ogc_neighbour_consistency_fwd");
    return OGC_OK;
}

__global__ __launch_bounds__(256) void fill_ones_kernel(long long total, int *__restrict__ p) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < total) p[e] = 1;
}

extern "C" int ogc_reverse_neighbours(int b, int n, int k, const int *idx, int *rev_start, int *rev_src, int *rev_mult,
                                      int *ws, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && k >= 0, "ogc_reverse_neighbours: negative size");
    if (b == 0) return OGC_OK;
    OGC_REQUIRE(rev_start, "ogc_reverse_neighbours: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (ogc_zero_async(rev_start, (size_t)b * (n + 1) * sizeof(int), s) != hipSuccess) {
        ogc_set_error("ogc_reverse_neighbours: zero fill failed");
        return OGC_ERR_LAUNCH;
    }
    const long long total = (long long)b * n * k;
    if (total == 0) return OGC_OK;
    OGC_REQUIRE(idx && rev_src && rev_mult && ws, "ogc_reverse_neighbours: null pointer");
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (k <= 64 && (k & (k - 1)) == 0) {
        hipLaunchKernelGGL(rev_count_kernel<true>, dim3(blocks), dim3(256), 0, s, total, n, k, idx, rev_start, rev_mult);
    } else {
        hipLaunchKernelGGL(fill_ones_kernel, dim3(ogc_divup(b * n, 256)), dim3(256), 0, s, (long long)b * n, rev_mult);
        hipLaunchKernelGGL(rev_count_kernel<false>, dim3(blocks), dim3(256), 0, s, total, n, k, idx, rev_start,
                           rev_mult);
    }
    hipLaunchKernelGGL(rev_scan_kernel, dim3(b), dim3(1024), 0, s, n, rev_start, ws);
    hipLaunchKernelGGL(rev_fill_kernel, dim3(blocks), dim3(256), 0, s, total, n, k, idx, ws, rev_src);
    if (ogc_deterministic()) hipLaunchKernelGGL(rev_sort_kernel, dim3(ogc_divup(n, 256), b), dim3(256), 0, s, n, k, rev_start, rev_src);
    OGC_CHECK_LAUNCH("ogc_reverse_neighbours");
    return OGC_OK;
}

extern "C" int ogc_neighbour_consistency_bwd(int b, int n, int c, int k, int p, const float *mask, const int *idx,
                                             const int *rev_start, const int *rev_src, const int *rev_mult,
                                             const float *grad_out, float *grad_mask, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && c >= 0 && k >= 0, "ogc_neighbour_consistency_bwd: negative size");
    OGC_REQUIRE(p == 1 || p == 2, "ogc_neighbour_consistency_bwd: norm must be 1 or 2");
    if (b == 0 || n == 0 || c == 0) return OGC_OK;
    OGC_REQUIRE(k > 0, "ogc_neighbour_consistency_bwd: k must be positive");
    OGC_REQUIRE(c <= 40, "ogc_neighbour_consistency_bwd: more than 40 channels");
    OGC_REQUIRE(mask && idx && rev_start && rev_src && rev_mult && grad_out && grad_mask,
                "ogc_neighbour_consistency_bwd: null pointer");
    const dim3 grid(ogc_divup(n, 256 / SUBL), b);
    const size_t smem = (size_t)c * 256 * sizeof(float);
    if (p == 1)
        hipLaunchKernelGGL(nc_bwd_kernel<1>, grid, dim3(256), smem, (hipStream_t)stream, n, c, k, mask, idx, rev_start,
                           rev_src, rev_mult, grad_out, grad_mask);
    else
        hipLaunchKernelGGL(nc_bwd_kernel<2>, grid, dim3(256), smem, (hipStream_t)stream, n, c, k, mask, idx, rev_start,
                           rev_src, rev_mult, grad_out, grad_mask);
    OGC_CHECK_LAUNCH("ogc_neighbour_consistency_bwd");
    return OGC_OK;
}
