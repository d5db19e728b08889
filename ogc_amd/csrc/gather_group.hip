// gather_group.hip — index gathers and their scatter-add gradients (bandwidth kernels).
//
// Replaces gather_points_kernel_fast / gather_points_grad_kernel_fast
//   (reference: pointnet2/src/sampling_gpu.cu:8-24, :46-63) and
// group_points_kernel_fast / group_points_grad_kernel_fast
//   (reference: pointnet2/src/group_points_gpu.cu:47-66, :8-25).
//
// The reference launches one thread per (channel, output element), so the index tensor is re-read
// once per channel (grid.y = C).  Here a thread owns up to four consecutive output positions, loads
// their indices once (one 16-byte load) and walks the channels, issuing 16-byte stores; a group of
// CH_PER_BLOCK channels per workgroup keeps enough workgroups in flight for small tensors.
//
// group_points is ONE kernel for both ops: gather_points is group_points with nsample = 1.
#include "ogc_common.h"
#include "act_io.h"

namespace {

constexpr int GG_THREADS = 256;

// out[b,c,t] = points[b,c,idx[b,t]],  t in [0, T)   (T = npoints*nsample)
template <bool VEC4>
__global__ __launch_bounds__(GG_THREADS) void group_fwd_kernel(int c, int n, int T, int ch_per_block,
                                                               long long out_bstride,
                                                               const float *__restrict__ points,
                                                               const int *__restrict__ idx,
                                                               float *__restrict__ out) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * ch_per_block;
    const int c1 = min(c, c0 + ch_per_block);
    const int *id = idx + (size_t)b * T;
    const float *p = points + ((size_t)b * c + c0) * n;
    float *o = out + (size_t)b * out_bstride + (size_t)c0 * T; // out_bstride = c*T for a dense (b,c,T) output
    if (VEC4) {
        const int t4 = (blockIdx.x * GG_THREADS + threadIdx.x) * 4;
        if (t4 >= T) return;
        const int4 i4 = *reinterpret_cast<const int4 *>(id + t4);
        for (int ch = c0; ch < c1; ++ch, p += n, o += T) {
            float4 v;
            v.x = p[i4.x]; v.y = p[i4.y]; v.z = p[i4.z]; v.w = p[i4.w];
            *reinterpret_cast<float4 *>(o + t4) = v;
        }
    } else {
        const int t = blockIdx.x * GG_THREADS + threadIdx.x;
        if (t >= T) return;
        const int i = id[t];
        for (int ch = c0; ch < c1; ++ch, p += n, o += T) o[t] = p[i];
    }
}

// grad_points[b,c,idx[b,t]] += grad_out[b,c,t]
template <bool VEC4>
__global__ __launch_bounds__(GG_THREADS) void group_bwd_kernel(int c, int n, int T, int ch_per_block,
                                                               long long go_bstride,
                                                               const float *__restrict__ grad_out,
                                                               const int *__restrict__ idx,
                                                               float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * ch_per_block;
    const int c1 = min(c, c0 + ch_per_block);
    const int *id = idx + (size_t)b * T;
    const float *g = grad_out + (size_t)b * go_bstride + (size_t)c0 * T;
    float *gp = grad_points + ((size_t)b * c + c0) * n;
    if (VEC4) {
        const int t4 = (blockIdx.x * GG_THREADS + threadIdx.x) * 4;
        if (t4 >= T) return;
        const int4 i4 = *reinterpret_cast<const int4 *>(id + t4);
        for (int ch = c0; ch < c1; ++ch, g += T, gp += n) {
            const float4 v = *reinterpret_cast<const float4 *>(g + t4);
            unsafeAtomicAdd(gp + i4.x, v.x);
            unsafeAtomicAdd(gp + i4.y, v.y);
            unsafeAtomicAdd(gp + i4.z, v.z);
            unsafeAtomicAdd(gp + i4.w, v.w);
        }
    } else {
        const int t = blockIdx.x * GG_THREADS + threadIdx.x;
        if (t >= T) return;
        const int i = id[t];
        for (int ch = c0; ch < c1; ++ch, g += T, gp += n) unsafeAtomicAdd(gp + i, g[t]);
    }
}

// LDS-privatised scatter-add: a workgroup owns CC channels x one slice of the T positions of batch b,
// accumulates into an LDS image acc[CC][n] with ds_add_f32 (no global atomic contention: kNN neighbour lists
// overlap heavily, group_points_gpu.cu:24 serialises on popular points), then merges the non-zero entries
// into grad_points with one global atomic each.
// WX: the gradient tensor is also multiplied with the relative coordinates rel (b, 3, T) on the way —
// dwx[ch, k] += sum_t grad_out[b, ch, t] * rel[b, k, t], the xyz columns of the weight gradient of ogc_group_linear_fwd's
// layer — so that the tensor is read once for both results.
template <int CC, bool WX>
__global__ __launch_bounds__(GG_THREADS) void group_bwd_lds_kernel(int c, int n, int T, int t_per_block,
                                                                   long long go_bstride,
                                                                   const float *__restrict__ grad_out,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ grad_points,
                                                                   const float *__restrict__ rel,
                                                                   float *__restrict__ dwx) {
    extern __shared__ __attribute__((aligned(16))) float gb_acc[]; // [CC][n]
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CC;
    const int ncc = min(CC, c - c0);
    const int t_begin = blockIdx.x * t_per_block;
    const int t_end = min(T, t_begin + t_per_block);
    for (int i = threadIdx.x; i < CC * n; i += GG_THREADS) gb_acc[i] = 0.0f;
    __syncthreads();
    const int *id = idx + (size_t)b * T;
    const float *g = grad_out + (size_t)b * go_bstride + (size_t)c0 * T;
    // A thread owns 16 consecutive positions (64 contiguous bytes per channel) and merges runs of equal indices
    // before touching LDS: neighbour rows end in long runs of the SAME index (kNN rows clamped to the nearest
    // neighbour beyond the radius, ball-query rows padded with the first hit), which would otherwise serialise as
    // same-address atomics.  t_begin, t_per_block and T are multiples of 16 on this path.
    float wacc[WX ? CC : 1][3];
#pragma unroll
    for (int cc = 0; cc < (WX ? CC : 1); ++cc) wacc[cc][0] = wacc[cc][1] = wacc[cc][2] = 0.0f;
    for (int t16 = t_begin + threadIdx.x * 16; t16 < t_end; t16 += GG_THREADS * 16) {
        int ids[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int4 i4 = *reinterpret_cast<const int4 *>(id + t16 + 4 * u);
            ids[4 * u] = i4.x; ids[4 * u + 1] = i4.y; ids[4 * u + 2] = i4.z; ids[4 * u + 3] = i4.w;
        }
        float rl[WX ? 3 : 1][16];
        if constexpr (WX) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 f = *reinterpret_cast<const float4 *>(rel + ((size_t)b * 3 + k) * T + t16 + 4 * u);
                    rl[k][4 * u] = f.x; rl[k][4 * u + 1] = f.y; rl[k][4 * u + 2] = f.z; rl[k][4 * u + 3] = f.w;
                }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            if (cc < ncc) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 f = *reinterpret_cast<const float4 *>(g + (size_t)cc * T + t16 + 4 * u);
                    v[4 * u] = f.x; v[4 * u + 1] = f.y; v[4 * u + 2] = f.z; v[4 * u + 3] = f.w;
                }
                if constexpr (WX) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int u = 0; u < 16; ++u) wacc[cc][k] = fmaf(v[u], rl[k][u], wacc[cc][k]);
                }
                float *a = gb_acc + cc * n;
                float run = v[0];
#pragma unroll
                for (int u = 1; u < 16; ++u) {
                    if (ids[u] == ids[u - 1]) {
                        run += v[u];
                    } else {
                        atomicAdd(a + ids[u - 1], run);
                        run = v[u];
                    }
                }
                atomicAdd(a + ids[15], run);
            }
        }
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)b * c + c0) * n;
    for (int i = threadIdx.x; i < ncc * n; i += GG_THREADS) {
        const float v = gb_acc[i];
        if (v != 0.0f) unsafeAtomicAdd(gp + i, v);
    }
    if constexpr (WX) { // wavefront sums, then one atomic per (channel, axis) and wavefront
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float w = wacc[cc][k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
                if ((threadIdx.x & 63) == 0 && cc < ncc && w != 0.0f) unsafeAtomicAdd(dwx + (size_t)(c0 + cc) * 3 + k, w);
            }
    }
}

// out[b, ch, p*nsample + s] = xyz[b, idx[b,p,s], ch] - new_xyz[b, p, ch],  ch in 0..2 — the relative coordinates that
// QueryAndGroup puts in front of the grouped features (pointnet2.py:286-288: group, then subtract the centre)
__global__ __launch_bounds__(GG_THREADS) void group_xyz_rel_kernel(int n, int npoints, int nsample,
                                                                   long long out_bstride,
                                                                   const float *__restrict__ xyz,
                                                                   const float *__restrict__ new_xyz,
                                                                   const int *__restrict__ idx,
                                                                   float *__restrict__ out) {
    const int b = blockIdx.y;
    const int T = npoints * nsample;
    const int t = blockIdx.x * GG_THREADS + threadIdx.x;
    if (t >= T) return;
    const int i = idx[(size_t)b * T + t];
    const float *p = xyz + ((size_t)b * n + i) * 3;
    const float *q = new_xyz + ((size_t)b * npoints + t / nsample) * 3;
    float *o = out + (size_t)b * out_bstride + t;
    o[0] = __fsub_rn(p[0], q[0]);
    o[T] = __fsub_rn(p[1], q[1]);
    o[2 * (size_t)T] = __fsub_rn(p[2], q[2]);
}

// ogc_group_concat in ONE launch: the workgroups of the last y-row write the three relative-coordinate rows (the expression of
// group_xyz_rel_kernel), the others the gathered feature rows behind them (group_fwd_kernel).  At B = 1 (FlowStep3D inference: ~46
// groupers per forward) the two launches of the pair are ~4 us each on a serial timeline.
template <bool VEC4>
__global__ __launch_bounds__(GG_THREADS) void group_concat_kernel(int c, int n, int T, int nsample, int ch_per_block,
                                                                  long long out_bstride, const float *__restrict__ points,
                                                                  const int *__restrict__ idx, const float *__restrict__ xyz,
                                                                  const float *__restrict__ new_xyz, float *__restrict__ out) {
    const int b = blockIdx.z;
    const int *id = idx + (size_t)b * T;
    float *ob = out + (size_t)b * out_bstride;
    if (blockIdx.y == gridDim.y - 1) { // rows 0..2: xyz[idx] - centre
        const int npoints = T / nsample;
        const float *xb = xyz + (size_t)b * n * 3, *qb = new_xyz + (size_t)b * npoints * 3;
        if (VEC4) {
            const int t4 = (blockIdx.x * GG_THREADS + threadIdx.x) * 4;
            if (t4 >= T) return;
            const int4 i4 = *reinterpret_cast<const int4 *>(id + t4);
            const int ii[4] = {i4.x, i4.y, i4.z, i4.w};
            float r[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float *p = xb + (size_t)ii[j] * 3, *q = qb + (size_t)((t4 + j) / nsample) * 3;
                r[0][j] = __fsub_rn(p[0], q[0]);
                r[1][j] = __fsub_rn(p[1], q[1]);
                r[2][j] = __fsub_rn(p[2], q[2]);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
                *reinterpret_cast<float4 *>(ob + (size_t)a * T + t4) = make_float4(r[a][0], r[a][1], r[a][2], r[a][3]);
        } else {
            const int t = blockIdx.x * GG_THREADS + threadIdx.x;
            if (t >= T) return;
            const float *p = xb + (size_t)id[t] * 3, *q = qb + (size_t)(t / nsample) * 3;
            ob[t] = __fsub_rn(p[0], q[0]);
            ob[T + t] = __fsub_rn(p[1], q[1]);
            ob[2 * (size_t)T + t] = __fsub_rn(p[2], q[2]);
        }
        return;
    }
    const int c0 = blockIdx.y * ch_per_block;
    const int c1 = min(c, c0 + ch_per_block);
    const float *p = points + ((size_t)b * c + c0) * n;
    float *o = ob + (size_t)(3 + c0) * T;
    if (VEC4) {
        const int t4 = (blockIdx.x * GG_THREADS + threadIdx.x) * 4;
        if (t4 >= T) return;
        const int4 i4 = *reinterpret_cast<const int4 *>(id + t4);
        for (int ch = c0; ch < c1; ++ch, p += n, o += T) {
            float4 v;
            v.x = p[i4.x]; v.y = p[i4.y]; v.z = p[i4.z]; v.w = p[i4.w];
            *reinterpret_cast<float4 *>(o + t4) = v;
        }
    } else {
        const int t = blockIdx.x * GG_THREADS + threadIdx.x;
        if (t >= T) return;
        const int i = id[t];
        for (int ch = c0; ch < c1; ++ch, p += n, o += T) o[t] = p[i];
    }
}

int pick_ch_per_block(int b, int c, int blocks_x) {
    // aim for >= ~2048 workgroups (8 per CU) before giving each workgroup more channels
    int cpb = 1;
    while (cpb < c && (long long)b * blocks_x * ((c + 2 * cpb - 1) / (2 * cpb)) >= 2048) cpb *= 2;
    return cpb;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int group_fwd(const char *name, int b, int c, int n, int T, const float *points, const int *idx,
              float *out, ogc_stream_t stream, long long out_bstride = -1) {
    if (out_bstride < 0) out_bstride = (long long)c * T;
    OGC_REQUIRE(b >= 0 && c >= 0 && n >= 0 && T >= 0, "%s: negative dimension", name);
    if (b == 0 || c == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(points && idx && out, "%s: null pointer", name);
    // the reference indexes the whole tensor with 32-bit ints (group_points_gpu.cu:63); here batch offsets are 64-bit
    OGC_REQUIRE(out_bstride < (1ll << 31) && (long long)c * n < (1ll << 31) && b <= 65535,
                "%s: one sample exceeds 32-bit indexing", name);
    const bool vec = (T % 4 == 0) && aligned16(idx) && aligned16(out) && out_bstride % 4 == 0;
    const int bx = ogc_divup(T, vec ? GG_THREADS * 4 : GG_THREADS);
    const int cpb = pick_ch_per_block(b, c, bx);
    dim3 grid(bx, ogc_divup(c, cpb), b);
    if (vec)
        hipLaunchKernelGGL(group_fwd_kernel<true>, grid, dim3(GG_THREADS), 0, (hipStream_t)stream, c, n, T,
                           cpb, out_bstride, points, idx, out);
    else
        hipLaunchKernelGGL(group_fwd_kernel<false>, grid, dim3(GG_THREADS), 0, (hipStream_t)stream, c, n, T,
                           cpb, out_bstride, points, idx, out);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}

// deterministic mode: dwx[ch][k] += sum over samples and positions of grad_out[b, ch, t] * rel[b, k, t] — one workgroup per channel,
// thread i takes positions i, i + 256, ... of sample 0, 1, ... in that order, then a fixed tree over the 256 partial sums
__global__ __launch_bounds__(256) void det_dwx_kernel(int b, int c, int T, long long go_bstride, const float *__restrict__ grad_out,
                                                      const float *__restrict__ rel, float *__restrict__ dwx) {
    __shared__ float red[3][256];
    const int ch = blockIdx.x, t0 = threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int bb = 0; bb < b; ++bb) {
        const float *g = grad_out + (size_t)bb * go_bstride + (size_t)ch * T;
        const float *r = rel + (size_t)bb * 3 * T;
        for (int t = t0; t < T; t += 256) {
            const float v = g[t];
            a0 = fmaf(v, r[t], a0);
            a1 = fmaf(v, r[(size_t)T + t], a1);
            a2 = fmaf(v, r[2 * (size_t)T + t], a2);
        }
    }
    red[0][t0] = a0; red[1][t0] = a1; red[2][t0] = a2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t0 < off) {
            red[0][t0] += red[0][t0 + off]; red[1][t0] += red[1][t0 + off]; red[2][t0] += red[2][t0 + off];
        }
        __syncthreads();
    }
    if (t0 < 3) dwx[ch * 3 + t0] += red[t0][0];
}

int group_bwd(const char *name, int b, int c, int n, int T, const float *grad_out, const int *idx,
              float *grad_points, ogc_stream_t stream, long long go_bstride = -1, const float *rel = nullptr,
              float *dwx = nullptr) {
    // rel / dwx: also accumulate dwx[ch, k] += sum grad_out * rel (LDS path only; the caller checks group_bwd_fuses_wx)
    if (go_bstride < 0) go_bstride = (long long)c * T;
    OGC_REQUIRE(b >= 0 && c >= 0 && n >= 0 && T >= 0, "%s: negative dimension", name);
    if (b == 0 || c == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(grad_out && idx && grad_points, "%s: null pointer", name);
    OGC_REQUIRE(go_bstride < (1ll << 31) && (long long)c * n < (1ll << 31) && b <= 65535,
                "%s: one sample exceeds 32-bit indexing", name);
    if (ogc_deterministic()) {
        // every sum in ascending position order, one thread per output (det.hip); the coordinate columns of a grouped first
        // layer's weight gradient by one workgroup per channel in a fixed order
        const int rc = ogc_det_scatter_add(name, b, c, n, T, idx, grad_out, go_bstride, nullptr, 0, grad_points, 1, (hipStream_t)stream);
        if (rc != OGC_OK || !dwx) return rc;
        hipLaunchKernelGGL(det_dwx_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, b, c, T, go_bstride, grad_out, rel, dwx);
        OGC_CHECK_LAUNCH(name);
        return OGC_OK;
    }
    const bool vec = (T % 4 == 0) && aligned16(idx) && aligned16(grad_out) && go_bstride % 4 == 0;
    // LDS-privatised path: the per-channel image (n floats) must fit a 64 KiB budget at least once
    if (vec && n <= 16384 && T >= 4096 && T % 16 == 0) {
        // channels per workgroup: one channel image of n floats (several only below 1024 points).  Small images mean
        // more workgroups per CU: at C4, 8 channels x 2048 points (64 KiB) ran at 235 us per call, 2 channels at 179 us,
        // 1 at 175 us; 2 x 8192 points at 151 us, 1 x 8192 at 135 us; 4 x 1024 at 231 us, 1 x 1024 at 175 us (the
        // index and rel rows a workgroup re-reads per channel group are small next to its share of the gradient tensor).
        int cc = 1024 / n;
        if (cc < 1) cc = 1;
        cc = cc >= 8 ? 8 : (cc >= 4 ? 4 : (cc >= 2 ? 2 : 1));
        while (cc > 1 && cc / 2 >= c) cc /= 2;
        const int chunks = ogc_divup(c, cc);
        // split T so that >= ~512 workgroups exist, but keep >= 8192 positions per workgroup
        int splits = 1;
        while ((long long)b * chunks * splits < 512 && T / (splits * 2) >= 8192) splits *= 2;
        int tpb = ogc_divup(T, splits);
        tpb = (tpb + 4095) / 4096 * 4096; // multiple of 256 threads x 16 positions
        dim3 grid(ogc_divup(T, tpb), chunks, b);
        const size_t lds = (size_t)cc * n * sizeof(float);
#define GB_LAUNCH(CCV)                                                                                             \
    do {                                                                                                           \
        if (dwx)                                                                                                   \
            hipLaunchKernelGGL((group_bwd_lds_kernel<CCV, true>), grid, dim3(GG_THREADS), lds, (hipStream_t)stream, c, n, \
                               T, tpb, go_bstride, grad_out, idx, grad_points, rel, dwx);                          \
        else                                                                                                       \
            hipLaunchKernelGGL((group_bwd_lds_kernel<CCV, false>), grid, dim3(GG_THREADS), lds, (hipStream_t)stream, c,  \
                               n, T, tpb, go_bstride, grad_out, idx, grad_points, rel, dwx);                       \
    } while (0)
        if (cc == 8) GB_LAUNCH(8);
        else if (cc == 4) GB_LAUNCH(4);
        else if (cc == 2) GB_LAUNCH(2);
        else GB_LAUNCH(1);
#undef GB_LAUNCH
        OGC_CHECK_LAUNCH(name);
        return OGC_OK;
    }
    if (dwx) {
        ogc_set_error("%s: the fused xyz weight gradient needs the LDS path (n <= 16384, T >= 4096, T %% 16 == 0)", name);
        return OGC_ERR_UNSUPPORTED;
    }
    const int bx = ogc_divup(T, vec ? GG_THREADS * 4 : GG_THREADS);
    const int cpb = pick_ch_per_block(b, c, bx);
    dim3 grid(bx, ogc_divup(c, cpb), b);
    if (vec)
        hipLaunchKernelGGL(group_bwd_kernel<true>, grid, dim3(GG_THREADS), 0, (hipStream_t)stream, c, n, T,
                           cpb, go_bstride, grad_out, idx, grad_points);
    else
        hipLaunchKernelGGL(group_bwd_kernel<false>, grid, dim3(GG_THREADS), 0, (hipStream_t)stream, c, n, T,
                           cpb, go_bstride, grad_out, idx, grad_points);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}

} // namespace

extern "C" int ogc_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                                 float *out, ogc_stream_t stream) {
    return group_fwd("ogc_gather_points", b, c, n, npoints, points, idx, out, stream);
}

extern "C" int ogc_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                      const int *idx, float *grad_points, ogc_stream_t stream) {
    return group_bwd("ogc_gather_points_grad", b, c, n, npoints, grad_out, idx, grad_points, stream);
}

extern "C" int ogc_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                const int *idx, float *out, ogc_stream_t stream) {
    OGC_REQUIRE(npoints >= 0 && nsample >= 0 && (long long)npoints * nsample < (1ll << 31),
                "ogc_group_points: bad npoints/nsample");
    return group_fwd("ogc_group_points", b, c, n, npoints * nsample, points, idx, out, stream);
}

extern "C" int ogc_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                     const int *idx, float *grad_points, ogc_stream_t stream) {
    OGC_REQUIRE(npoints >= 0 && nsample >= 0 && (long long)npoints * nsample < (1ll << 31),
                "ogc_group_points_grad: bad npoints/nsample");
    return group_bwd("ogc_group_points_grad", b, c, n, npoints * nsample, grad_out, idx, grad_points, stream);
}

extern "C" int ogc_group_concat(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                                const float *points, const int *idx, float *out, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0 &&
                    (long long)npoints * nsample < (1ll << 31),
                "ogc_group_concat: bad dimensions");
    const int T = npoints * nsample;
    if (b == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(xyz && new_xyz && idx && out && (points || c == 0), "ogc_group_concat: null pointer");
    const long long bstride = (long long)(3 + c) * T;
    OGC_REQUIRE(bstride < (1ll << 31) && b <= 65535, "ogc_group_concat: one sample exceeds 32-bit indexing");
    if (c > 0 && nsample > 0 && (long long)c * n < (1ll << 31)) { // both halves in one launch (group_concat_kernel)
        const bool vec = (T % 4 == 0) && aligned16(idx) && aligned16(out) && bstride % 4 == 0;
        const int bx = ogc_divup(T, vec ? GG_THREADS * 4 : GG_THREADS);
        const int cpb = pick_ch_per_block(b, c, bx);
        dim3 grid(bx, ogc_divup(c, cpb) + 1, b);
        if (vec)
            hipLaunchKernelGGL(group_concat_kernel<true>, grid, dim3(GG_THREADS), 0, (hipStream_t)stream, c, n, T, nsample, cpb,
                               bstride, points, idx, xyz, new_xyz, out);
        else
            hipLaunchKernelGGL(group_concat_kernel<false>, grid, dim3(GG_THREADS), 0, (hipStream_t)stream, c, n, T, nsample, cpb,
                               bstride, points, idx, xyz, new_xyz, out);
        OGC_CHECK_LAUNCH("ogc_group_concat");
        return OGC_OK;
    }
    hipLaunchKernelGGL(group_xyz_rel_kernel, dim3(ogc_divup(T, GG_THREADS), b), dim3(GG_THREADS), 0,
                       (hipStream_t)stream, n, npoints, nsample, bstride, xyz, new_xyz, idx, out);
    OGC_CHECK_LAUNCH("ogc_group_concat");
    if (c == 0) return OGC_OK;
    return group_fwd("ogc_group_concat", b, c, n, T, points, idx, out + 3 * (size_t)T, stream, bstride);
}

namespace {
// First layer of a set-abstraction MLP without the grouped tensor.  The layer is linear in [x_j - c_i ; f_j], so the
// feature part commutes with the gather:  y[b, m, (i, j)] = P[b, m, idx[b, i, j]] + sum_k wx[m, k] * rel[b, k, (i, j)]
// with P = W_f . f computed per POINT (a small GEMM) and rel = x_j - c_i formed first, exactly as the reference
// does (pointnet2.py:286-288), so nothing cancels.  A thread owns four consecutive positions: one 16-byte index load,
// three 16-byte loads of rel, then per output channel four gathered reads of P (L2-resident: b*m*n floats) and a
// 16-byte store.  A workgroup stays inside one GroupNorm group and adds its (sum, sum of squares) to one of
// GL_SLOTS copies of the (b, groups, 2) fp64 accumulator — the layout ogc_conv1x1_gemm_gnstats fills.
constexpr int GL_SLOTS = 16;

// OT: element type of y (float / ogc_bf16: act_io.h); the statistics are those of the stored values.
template <typename OT>
__global__ __launch_bounds__(GG_THREADS) void group_linear_fwd_kernel(int m, int n, int T, int cpb, int cg, int groups,
                                                                      const float *__restrict__ P,
                                                                      const int *__restrict__ idx,
                                                                      const float *__restrict__ rel,
                                                                      const float *__restrict__ wx,
                                                                      OT *__restrict__ y,
                                                                      double *__restrict__ stats) {
    __shared__ double red[2 * GG_THREADS / 64];
    const int b = blockIdx.z, c0 = blockIdx.y * cpb;
    const int t4 = (blockIdx.x * GG_THREADS + threadIdx.x) * 4;
    float s = 0.f, ss = 0.f;
    if (t4 < T) {
        const int4 i4 = *reinterpret_cast<const int4 *>(idx + (size_t)b * T + t4);
        const float *rb = rel + (size_t)b * 3 * T + t4;
        const float4 rx = *reinterpret_cast<const float4 *>(rb), ry = *reinterpret_cast<const float4 *>(rb + T),
                     rz = *reinterpret_cast<const float4 *>(rb + 2 * (size_t)T);
        const float *p = P + ((size_t)b * m + c0) * n;
        OT *o = y + ((size_t)b * m + c0) * T + t4;
        for (int ch = c0; ch < c0 + cpb; ++ch, p += n, o += T) {
            const float w0 = wx[ch * 3], w1 = wx[ch * 3 + 1], w2 = wx[ch * 3 + 2];
            float4 v;
            v.x = fmaf(w2, rz.x, fmaf(w1, ry.x, fmaf(w0, rx.x, p[i4.x])));
            v.y = fmaf(w2, rz.y, fmaf(w1, ry.y, fmaf(w0, rx.y, p[i4.y])));
            v.z = fmaf(w2, rz.z, fmaf(w1, ry.z, fmaf(w0, rx.z, p[i4.z])));
            v.w = fmaf(w2, rz.w, fmaf(w1, ry.w, fmaf(w0, rx.w, p[i4.w])));
            if constexpr (sizeof(OT) == 2) {
                v.x = ogc_as_stored<OT>(v.x); v.y = ogc_as_stored<OT>(v.y); v.z = ogc_as_stored<OT>(v.z); v.w = ogc_as_stored<OT>(v.w);
            }
            ogc_st4(o, v);
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    }
    if (stats) { // uniform
        double ds = s, dss = ss;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ds += __shfl_down(ds, off, 64);
            dss += __shfl_down(dss, off, 64);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
            red[wave * 2] = ds;
            red[wave * 2 + 1] = dss;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0, q = 0.0;
            for (int w = 0; w < GG_THREADS / 64; ++w) {
                a += red[w * 2];
                q += red[w * 2 + 1];
            }
            double *dst = stats + (((size_t)(blockIdx.x % GL_SLOTS) * gridDim.z + b) * groups + c0 / cg) * 2;
            atomicAdd(dst, a);
            atomicAdd(dst + 1, q);
        }
    }
}
} // namespace

namespace {
// The same layer for FEW feature channels (the first level of an encoder: the features are the coordinates, CF = 3): the reduction
// has 3 + CF terms, so the feature part is applied per POSITION as well — CF gathered feature values instead of one gathered value of
// P per OUTPUT channel (m >= 32 of them), and no point-wise product in front — and every output is ONE fused-multiply-add chain over
// the layer's input channels in their order, [rel x, y, z, f_0 ..] — the order in which the reference's convolution and this library's
// matrix kernels (v_mfma_f32_16x16x4_f32, k ascending) walk cat([grouped_xyz, grouped_features]).  P[idx] + W_xyz rel rounds the two
// halves separately; on the segnet_ogcdr fixture that one bit put an activation of the first level on the other side of its
// ReLU / max-pool gate and moved 68 gradient tensors by 1e-3 (tests/golden_cases.py, gate_flip_tensors.json until round 6).
template <typename OT, int CF>
__global__ __launch_bounds__(GG_THREADS) void group_linear_direct_kernel(int m, int n, int T, int cpb, int cg, int groups,
                                                                         const float *__restrict__ feats, // (b, CF, n)
                                                                         const int *__restrict__ idx,
                                                                         const float *__restrict__ rel,
                                                                         const float *__restrict__ w,     // (m, 3 + CF)
                                                                         OT *__restrict__ y, double *__restrict__ stats) {
    __shared__ double red[2 * GG_THREADS / 64];
    const int b = blockIdx.z, c0 = blockIdx.y * cpb;
    const int t4 = (blockIdx.x * GG_THREADS + threadIdx.x) * 4;
    float s = 0.f, ss = 0.f;
    if (t4 < T) {
        const int4 i4 = *reinterpret_cast<const int4 *>(idx + (size_t)b * T + t4);
        const float *rb = rel + (size_t)b * 3 * T + t4;
        const float4 rx = *reinterpret_cast<const float4 *>(rb), ry = *reinterpret_cast<const float4 *>(rb + T),
                     rz = *reinterpret_cast<const float4 *>(rb + 2 * (size_t)T);
        float4 f[CF];
#pragma unroll
        for (int c = 0; c < CF; ++c) {
            const float *fp = feats + ((size_t)b * CF + c) * n;
            f[c] = make_float4(fp[i4.x], fp[i4.y], fp[i4.z], fp[i4.w]);
        }
        OT *o = y + ((size_t)b * m + c0) * T + t4;
        for (int ch = c0; ch < c0 + cpb; ++ch, o += T) {
            const float *wr = w + (size_t)ch * (3 + CF);
            float4 v;
            v.x = fmaf(wr[2], rz.x, fmaf(wr[1], ry.x, wr[0] * rx.x));
            v.y = fmaf(wr[2], rz.y, fmaf(wr[1], ry.y, wr[0] * rx.y));
            v.z = fmaf(wr[2], rz.z, fmaf(wr[1], ry.z, wr[0] * rx.z));
            v.w = fmaf(wr[2], rz.w, fmaf(wr[1], ry.w, wr[0] * rx.w));
#pragma unroll
            for (int c = 0; c < CF; ++c) {
                const float wc = wr[3 + c];
                v.x = fmaf(wc, f[c].x, v.x); v.y = fmaf(wc, f[c].y, v.y); v.z = fmaf(wc, f[c].z, v.z); v.w = fmaf(wc, f[c].w, v.w);
            }
            if constexpr (sizeof(OT) == 2) {
                v.x = ogc_as_stored<OT>(v.x); v.y = ogc_as_stored<OT>(v.y); v.z = ogc_as_stored<OT>(v.z); v.w = ogc_as_stored<OT>(v.w);
            }
            ogc_st4(o, v);
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    }
    if (stats) { // uniform — as group_linear_fwd_kernel
        double ds = s, dss = ss;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ds += __shfl_down(ds, off, 64);
            dss += __shfl_down(dss, off, 64);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
            red[wave * 2] = ds;
            red[wave * 2 + 1] = dss;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0, q = 0.0;
            for (int wv = 0; wv < GG_THREADS / 64; ++wv) {
                a += red[wv * 2];
                q += red[wv * 2 + 1];
            }
            double *dst = stats + (((size_t)(blockIdx.x % GL_SLOTS) * gridDim.z + b) * groups + c0 / cg) * 2;
            atomicAdd(dst, a);
            atomicAdd(dst + 1, q);
        }
    }
}
} // namespace

// ogc_group_linear_fwd for 1 .. 4 feature channels without P: y[b, ch, (i, j)] = the fused-multiply-add chain over
// [rel (3), features[:, idx] (cf)] with the layer's weight rows w (m, 3 + cf), input channels ascending.  feats (b, cf, n); the other
// arguments, the statistics layout and the preconditions are ogc_group_linear_fwd's.
namespace {
template <typename OT>
int group_linear_fwd_direct_impl(int b, int m, int cf, int n, int npoints, int nsample, int groups, const float *feats,
                                 const int *idx, const float *rel, const float *w, OT *y, double *stats, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && m >= 1 && n >= 1 && npoints >= 0 && nsample >= 0 && groups >= 0 && (long long)npoints * nsample < (1ll << 31),
                "ogc_group_linear_fwd_direct: bad dimensions");
    OGC_REQUIRE(cf >= 1 && cf <= 4, "ogc_group_linear_fwd_direct: 1 .. 4 feature channels (got %d)", cf);
    const int T = npoints * nsample;
    if (b == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(feats && idx && rel && w && y && (stats || groups == 0), "ogc_group_linear_fwd_direct: null pointer");
    OGC_REQUIRE((long long)m * T < (1ll << 31) && (long long)cf * n < (1ll << 31) && b <= 65535,
                "ogc_group_linear_fwd_direct: one sample exceeds 32-bit indexing");
    if ((T & 3) != 0 || !aligned16(idx) || !aligned16(rel) || ((uintptr_t)y & ogc_act_mask<OT>()) != 0 || (groups > 0 && m % groups != 0)) {
        ogc_set_error("ogc_group_linear_fwd_direct: needs npoints * nsample %% 4 == 0, 16-byte aligned tensors, m %% groups == 0");
        return OGC_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    const int cg = groups > 0 ? m / groups : m;
    int cpb = 1;
    while (cpb < 16 && cg % (cpb * 2) == 0) cpb *= 2;
    if (groups > 0 && ogc_zero_async(stats, sizeof(double) * 2 * GL_SLOTS * (size_t)b * groups, s) != hipSuccess) {
        ogc_set_error("ogc_group_linear_fwd_direct: memset failed");
        return OGC_ERR_LAUNCH;
    }
    dim3 grid(ogc_divup(T, GG_THREADS * 4), m / cpb, b);
    double *st = groups > 0 ? stats : nullptr;
#define OGC_GLD(CFV) \
    hipLaunchKernelGGL((group_linear_direct_kernel<OT, CFV>), grid, dim3(GG_THREADS), 0, s, m, n, T, cpb, cg, groups, feats, idx, rel, w, y, st)
    if (cf == 1) OGC_GLD(1);
    else if (cf == 2) OGC_GLD(2);
    else if (cf == 3) OGC_GLD(3);
    else OGC_GLD(4);
#undef OGC_GLD
    OGC_CHECK_LAUNCH("ogc_group_linear_fwd_direct");
    return OGC_OK;
}
} // namespace

extern "C" int ogc_group_linear_fwd_direct(int b, int m, int cf, int n, int npoints, int nsample, int groups, const float *feats,
                                           const int *idx, const float *rel, const float *w, float *y, double *stats,
                                           ogc_stream_t stream) {
    return group_linear_fwd_direct_impl<float>(b, m, cf, n, npoints, nsample, groups, feats, idx, rel, w, y, stats, stream);
}

// 16-bit activations (act_io.h): y stored as bf16, the statistics those of the stored values — 12 gathered bytes per position
// instead of the 4 m bytes of a P row (ogc_group_linear_fwd_pt_h: 256 bytes at m = 64)
extern "C" int ogc_group_linear_fwd_direct_h(int b, int m, int cf, int n, int npoints, int nsample, int groups, const float *feats,
                                             const int *idx, const float *rel, const float *w, ogc_bf16_t *y, double *stats,
                                             ogc_stream_t stream) {
    return group_linear_fwd_direct_impl<ogc_bf16>(b, m, cf, n, npoints, nsample, groups, feats, idx, rel, w,
                                                  reinterpret_cast<ogc_bf16 *>(y), stats, stream);
}

namespace {
template <typename OT>
int group_linear_fwd_impl(int b, int m, int n, int npoints, int nsample, int groups, const float *P, const int *idx,
                          const float *rel, const float *wx, OT *y, double *stats, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && m >= 1 && n >= 1 && npoints >= 0 && nsample >= 0 && groups >= 0 &&
                    (long long)npoints * nsample < (1ll << 31),
                "ogc_group_linear_fwd: bad dimensions");
    const int T = npoints * nsample;
    if (b == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(P && idx && rel && wx && y && (stats || groups == 0), "ogc_group_linear_fwd: null pointer");
    OGC_REQUIRE((long long)m * T < (1ll << 31) && (long long)m * n < (1ll << 31) && b <= 65535,
                "ogc_group_linear_fwd: one sample exceeds 32-bit indexing");
    if ((T & 3) != 0 || !aligned16(idx) || !aligned16(rel) || ((uintptr_t)y & ogc_act_mask<OT>()) != 0 || (groups > 0 && m % groups != 0)) {
        ogc_set_error("ogc_group_linear_fwd: needs npoints * nsample %% 4 == 0, 16-byte aligned tensors, m %% groups == 0");
        return OGC_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    const int cg = groups > 0 ? m / groups : m;
    int cpb = 1; // channels per workgroup: the largest power of two <= 16 dividing the group width
    while (cpb < 16 && cg % (cpb * 2) == 0) cpb *= 2;
    if (groups > 0 &&
        ogc_zero_async(stats, sizeof(double) * 2 * GL_SLOTS * (size_t)b * groups, s) != hipSuccess) {
        ogc_set_error("ogc_group_linear_fwd: memset failed");
        return OGC_ERR_LAUNCH;
    }
    dim3 grid(ogc_divup(T, GG_THREADS * 4), m / cpb, b);
    hipLaunchKernelGGL(group_linear_fwd_kernel<OT>, grid, dim3(GG_THREADS), 0, s, m, n, T, cpb, cg, groups, P, idx, rel, wx,
                       y, groups > 0 ? stats : nullptr);
    OGC_CHECK_LAUNCH("ogc_group_linear_fwd");
    return OGC_OK;
}
} // namespace

extern "C" int ogc_group_linear_fwd(int b, int m, int n, int npoints, int nsample, int groups, const float *P,
                                    const int *idx, const float *rel, const float *wx, float *y, double *stats,
                                    ogc_stream_t stream) {
    return group_linear_fwd_impl<float>(b, m, n, npoints, nsample, groups, P, idx, rel, wx, y, stats, stream);
}

extern "C" int ogc_group_linear_fwd_h(int b, int m, int n, int npoints, int nsample, int groups, const float *P,
                                      const int *idx, const float *rel, const float *wx, ogc_bf16_t *y, double *stats,
                                      ogc_stream_t stream) {
    return group_linear_fwd_impl<ogc_bf16>(b, m, n, npoints, nsample, groups, P, idx, rel, wx, y, stats, stream);
}

namespace {
// ---- the same layer for 16-bit outputs with P stored POINT-MAJOR ------------------------------------------------------------------
// group_linear_fwd_kernel reads P (b, m, n) with one four-byte gather per output element — 268 M of them for C2's first level, each
// its own cache-line lookup, which is what its 0.32 ms are once the output is written as bf16 (0.54 GB: 1.7 TB/s).  Here P is
// (b, n, m): a position's 64 channels are 256 contiguous bytes, fetched as sixteen 16-byte loads by the ONE lane that owns the
// position (a quarter of the lookups, four channels each).  A lane owns two consecutive positions, so that every channel leaves as
// a 4-byte store and a wavefront writes 256 contiguous bytes per channel; a workgroup handles 64 channels (blockIdx.y picks the
// slice) of GLP_ITERS x 512 positions.  Same expression per element as group_linear_fwd_kernel (fmaf chain over x, y, z on top of
// P), outputs rounded before the statistics: y is bit-identical, the statistics differ in the order of their additions.
// CG: channels per GroupNorm group (16, 32 or 64: compile time, so that the per-group sums index registers).
constexpr int GLP_ITERS = 4;

template <int CG>
__global__ __launch_bounds__(GG_THREADS) void group_linear_fwd_pt_kernel(int m, int n, int T, int groups, int with_stats,
                                                                         const float *__restrict__ Pt,  // (b, n, m)
                                                                         const int *__restrict__ idx,   // (b, T)
                                                                         const float *__restrict__ rel, // (b, 3, T)
                                                                         const float *__restrict__ wx,  // (m, 3)
                                                                         ogc_bf16 *__restrict__ y,      // (b, m, T)
                                                                         double *__restrict__ stats) {
    constexpr int NG = 64 / CG; // groups inside the workgroup's 64 channels
    __shared__ float s_wx[64 * 3];
    __shared__ double red[GG_THREADS / 64][NG][2];
    const int b = blockIdx.z, c0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 3; e += GG_THREADS) s_wx[e] = c0 + e / 3 < m ? wx[(c0 + e / 3) * 3 + e % 3] : 0.f;
    __syncthreads();
    double dgs[NG], dgss[NG]; // per iteration the fp32 partial sums of 2 x CG values move into fp64
#pragma unroll
    for (int g = 0; g < NG; ++g) dgs[g] = dgss[g] = 0.0;
    const int *ib = idx + (size_t)b * T;
    const float *rb = rel + (size_t)b * 3 * T;
    const float *pb_ = Pt + (size_t)b * n * m + c0;
    ogc_bf16 *yb = y + ((size_t)b * m + c0) * T;
    for (int it = 0; it < GLP_ITERS; ++it) {
        const int t2 = ((blockIdx.x * GLP_ITERS + it) * GG_THREADS + threadIdx.x) * 2;
        if (t2 >= T) break;
        const int2 i2 = *reinterpret_cast<const int2 *>(ib + t2);
        const float2 rx = *reinterpret_cast<const float2 *>(rb + t2), ry = *reinterpret_cast<const float2 *>(rb + T + t2),
                     rz = *reinterpret_cast<const float2 *>(rb + 2 * (size_t)T + t2);
        const float4 *p0 = reinterpret_cast<const float4 *>(pb_ + (size_t)i2.x * m);
        const float4 *p1 = reinterpret_cast<const float4 *>(pb_ + (size_t)i2.y * m);
        float gs[NG], gss[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) gs[g] = gss[g] = 0.f;
        float4 a[16], c[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = p0[q];
#pragma unroll
        for (int q = 0; q < 16; ++q) c[q] = p1[q];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float pa4[4] = {a[q].x, a[q].y, a[q].z, a[q].w}, pc4[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = q * 4 + e;
                const float w0 = s_wx[ch * 3], w1 = s_wx[ch * 3 + 1], w2 = s_wx[ch * 3 + 2];
                float v0 = fmaf(w2, rz.x, fmaf(w1, ry.x, fmaf(w0, rx.x, pa4[e])));
                float v1 = fmaf(w2, rz.y, fmaf(w1, ry.y, fmaf(w0, rx.y, pc4[e])));
                v0 = ogc_as_stored<ogc_bf16>(v0);
                v1 = ogc_as_stored<ogc_bf16>(v1);
                if (c0 + ch < m)
                    *reinterpret_cast<unsigned *>(yb + (size_t)ch * T + t2) = (__float_as_uint(v0) >> 16) | (__float_as_uint(v1) & 0xFFFF0000u);
                gs[ch / CG] += v0 + v1;
                gss[ch / CG] += v0 * v0 + v1 * v1;
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) { dgs[g] += (double)gs[g]; dgss[g] += (double)gss[g]; }
    }
    if (with_stats) { // uniform
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double ds = dgs[g], dss = dgss[g];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                ds += __shfl_down(ds, off, 64);
                dss += __shfl_down(dss, off, 64);
            }
            if (lane == 0) { red[wave][g][0] = ds; red[wave][g][1] = dss; }
        }
        __syncthreads();
        if (threadIdx.x < NG * 2) {
            const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
            double v = 0.0;
            for (int w = 0; w < GG_THREADS / 64; ++w) v += red[w][g][which];
            const int gg = c0 / CG + g;
            if (gg < groups)
                atomicAdd(stats + (((size_t)(blockIdx.x % GL_SLOTS) * gridDim.z + b) * groups + gg) * 2 + which, v);
        }
    }
}
} // namespace

// ogc_group_linear_fwd_h with P stored point-major, Pt (b, n, m) — see group_linear_fwd_pt_kernel.  Needs m % 64 == 0, groups in
// {0} or m / groups in {16, 32, 64}, npoints * nsample % 2 == 0, n * m and m * T below 2^31 (OGC_ERR_UNSUPPORTED otherwise: the
// caller keeps the channel-major form).
extern "C" int ogc_group_linear_fwd_pt_h(int b, int m, int n, int npoints, int nsample, int groups, const float *Pt,
                                         const int *idx, const float *rel, const float *wx, ogc_bf16_t *y, double *stats,
                                         ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && m >= 1 && n >= 1 && npoints >= 0 && nsample >= 0 && groups >= 0 &&
                    (long long)npoints * nsample < (1ll << 31),
                "ogc_group_linear_fwd_pt_h: bad dimensions");
    const int T = npoints * nsample;
    if (b == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(Pt && idx && rel && wx && y && (stats || groups == 0), "ogc_group_linear_fwd_pt_h: null pointer");
    const int cg = groups > 0 ? m / groups : 64;
    if ((m & 63) != 0 || (T & 1) != 0 || (groups > 0 && (m % groups != 0 || (cg != 16 && cg != 32 && cg != 64))) ||
        (long long)m * T >= (1ll << 31) || (long long)m * n >= (1ll << 31) || b > 65535 ||
        (((uintptr_t)Pt) & 15) != 0 || (((uintptr_t)idx | (uintptr_t)rel) & 7) != 0 || (((uintptr_t)y) & 3) != 0) {
        ogc_set_error("ogc_group_linear_fwd_pt_h: needs m %% 64 == 0, m / groups in {16, 32, 64}, an even position count and aligned tensors");
        return OGC_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (groups > 0 && ogc_zero_async(stats, sizeof(double) * 2 * GL_SLOTS * (size_t)b * groups, s) != hipSuccess) {
        ogc_set_error("ogc_group_linear_fwd_pt_h: memset failed");
        return OGC_ERR_LAUNCH;
    }
    dim3 grid(ogc_divup(T, GG_THREADS * 2 * GLP_ITERS), m / 64, b);
#define GLP(CGV) hipLaunchKernelGGL(group_linear_fwd_pt_kernel<CGV>, grid, dim3(GG_THREADS), 0, s, m, n, T, groups, groups > 0 ? 1 : 0, \
                                    Pt, idx, rel, wx, y, stats)
    if (cg == 16) GLP(16);
    else if (cg == 32) GLP(32);
    else GLP(64);
#undef GLP
    OGC_CHECK_LAUNCH("ogc_group_linear_fwd_pt_h");
    return OGC_OK;
}

namespace {
} // namespace

extern "C" int ogc_group_linear_bwd(int b, int m, int n, int npoints, int nsample, const float *grad_y, const int *idx,
                                    const float *rel, float *grad_p, float *dwx, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && m >= 1 && n >= 1 && npoints >= 0 && nsample >= 0 &&
                    (long long)npoints * nsample < (1ll << 31),
                "ogc_group_linear_bwd: bad dimensions");
    const int T = npoints * nsample;
    if (b == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(grad_y && idx && rel && grad_p && dwx, "ogc_group_linear_bwd: null pointer");
    if (!aligned16(rel) || !aligned16(grad_y) || !aligned16(idx)) {
        ogc_set_error("ogc_group_linear_bwd: tensors must be 16-byte aligned");
        return OGC_ERR_UNSUPPORTED;
    }
    return group_bwd("ogc_group_linear_bwd", b, m, n, T, grad_y, idx, grad_p, stream, -1, rel, dwx);
}

extern "C" int ogc_group_concat_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                     const int *idx, float *grad_points, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0 &&
                    (long long)npoints * nsample < (1ll << 31),
                "ogc_group_concat_grad: bad dimensions");
    const int T = npoints * nsample;
    if (b == 0 || c == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(grad_out, "ogc_group_concat_grad: null pointer");
    return group_bwd("ogc_group_concat_grad", b, c, n, T, grad_out + 3 * (size_t)T, idx, grad_points, stream,
                     (long long)(3 + c) * T);
}

// ---- the grouping gradient as a GATHER over transposed neighbour lists ------------------------------------------------------
// grad_points[b, c, j] = sum over the positions t with idx[b, t] == j of grad_out[b, c, t].  The scatter form above is bound
// by LDS float atomics (ds_add_f32: ~0.4 lane-atomics per cycle and CU, measured: 0.50 ms for a tensor the loads alone
// stream in 0.09 ms).  The lists "which positions point at j" depend on coordinates only, so they are built once per
// neighbour tensor (ogc_group_reverse, part of a step's geometry plan) and shared by all channels: the positions are cut into
// chunks of tc (one chunk of one channel plane = tc floats fits LDS), rev_start[b][chunk * n + j] .. [chunk * n + j + 1]
// delimits the entries of point j inside chunk `chunk`, rev_pos holds the positions relative to their chunk (16 bits).
// The gradient kernel gives a workgroup one (b, c) plane: chunk by chunk it stages the plane in LDS with coalesced loads
// (the next chunk's loads are in flight meanwhile) and every thread adds up the entries of ITS points — registers, no atomics,
// one writer per output (the caller need not zero grad_points).
namespace {

constexpr int GR_THREADS = 512;

// One workgroup per (sample, chunk): the chunk's tc positions are counted into an LDS histogram over the n points (integer LDS
// atomics), the histogram is scanned in place, and the positions are dealt to their lists through LDS cursors — no global
// atomics, one launch.  The lists of chunk `ch` occupy rev_pos[ch * tc ...); rev_start has n + 1 entries per chunk.
constexpr int GRB_THREADS = 1024;

// Runs: neighbour rows end in long runs of ONE index (kNN rows clamped to the nearest neighbour beyond the radius, ball-query
// rows padded with the first hit) — the lists of those points would be tens of entries long per row and a few threads would
// walk them while their wavefronts wait.  Inside every aligned group of 16 positions a run of equal indices is represented by
// its FIRST position only (`heads`: one bit per position, set for run heads); the gradient kernel adds the run up into that
// position while it stages the values.  T must be a multiple of 16.
__global__ __launch_bounds__(GRB_THREADS) void group_reverse_kernel(int n, int T, int tc, const int *__restrict__ idx,
                                                                    int *__restrict__ rev_start,
                                                                    unsigned short *__restrict__ rev_pos,
                                                                    unsigned short *__restrict__ heads, int sorted) {
    extern __shared__ int grb_hist[]; // [n] counts, then cursors
    __shared__ int wsum[GRB_THREADS / 64];
    __shared__ int carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int chunk = blockIdx.x, b = blockIdx.y, chunks = gridDim.x;
    const int t0 = chunk * tc, t1 = min(T, t0 + tc);
    const int *id = idx + (size_t)b * T;
    for (int j = t; j < n; j += GRB_THREADS) grb_hist[j] = 0;
    if (t == 0) carry = t0;
    __syncthreads();
    for (int p0 = t0; p0 < t1; p0 += GRB_THREADS) { // (t0, tc, T multiples of 16 and GRB_THREADS of 64: lane = p % 64)
        const int p = p0 + t;
        int j = -1;
        bool head = false;
        if (p < t1) {
            j = id[p];
            head = (p & 15) == 0 || id[p - 1] != j;
        }
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(head);
        if (p < t1 && (lane & 15) == 0) heads[((size_t)b * T + p) >> 4] = (unsigned short)(hm >> lane);
        if (head && j >= 0 && j < n) atomicAdd(&grb_hist[j], 1);
    }
    __syncthreads();
    int *rs = rev_start + ((size_t)b * chunks + chunk) * (n + 1);
    for (int base = 0; base < n; base += GRB_THREADS * 4) { // exclusive scan, four bins per thread and round
        const int e = base + t * 4;
        int v[4], sum = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = e + u < n ? grb_hist[e + u] : 0; sum += v[u]; }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int pre = carry + incl - sum;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e + u < n) { rs[e + u] = pre; grb_hist[e + u] = pre; }
            pre += v[u];
        }
        __syncthreads();
        if (t == GRB_THREADS - 1) carry = pre;
        __syncthreads();
    }
    if (t == 0) rs[n] = carry;
    unsigned short *rp = rev_pos + (size_t)b * T;
    for (int p = t0 + t; p < t1; p += GRB_THREADS) {
        const int j = id[p];
        const bool head = (p & 15) == 0 || id[p - 1] != j;
        if (head && j >= 0 && j < n) rp[atomicAdd(&grb_hist[j], 1)] = (unsigned short)(p - t0);
    }
    if (sorted) { // deterministic mode: the slot an entry got inside its list was a race — every list ascending (insertion, in place)
        __syncthreads();
        for (int j = t; j < n; j += GRB_THREADS) {
            const int lo = rs[j], hi = rs[j + 1];
            for (int i = lo + 1; i < hi; ++i) {
                const unsigned short v = rp[i];
                int k = i - 1;
                while (k >= lo && rp[k] > v) {
                    rp[k + 1] = rp[k];
                    --k;
                }
                rp[k + 1] = v;
            }
        }
    }
}

// NA: points per thread (n <= NA * GR_THREADS).  Per chunk the workgroup stages tc gradient values — a thread brings one
// aligned group of 16 positions and folds every run of equal indices into its first position (`heads`) — AND the chunk's
// list entries (they are contiguous: the lists of chunk ch start at rev_pos[ch * tc]) in LDS with coalesced loads: walking
// the lists out of global memory costs an L2 round trip per list step.  tc = 16 * GR_THREADS.
// INTERP: the gradient of three_interpolate — position t = 3 i + k stands for grad_out[b, c, i] * weight[b, i, k] (the plane
// has T / 3 values, the products are formed while staging).
// GT: element type of grad_out (float / ogc_bf16: act_io.h; the INTERP form is fp32 only).
// DWX (not with INTERP): the plane is the gradient of a grouped first layer's output (ogc_group_linear_fwd), and the three
// coordinate columns of that layer's weight gradient — dwx[ch][k] = sum over samples and positions of grad_out * rel[b, k, pos] —
// are accumulated from the values while they are staged (`weight` then holds rel (b, 3, T); 192 bytes of rel per thread and chunk,
// shared by the sample's planes through the L2) instead of by a three-channel weight-gradient launch that reads grad_out again.
template <int NA, bool INTERP, typename GT = float, bool DWX = false>
__global__ __launch_bounds__(GR_THREADS) void group_bwd_rev_kernel(int c, int n, int T, int tc, long long go_bstride,
                                                                   const GT *__restrict__ grad_out,
                                                                   const int *__restrict__ rev_start,
                                                                   const unsigned short *__restrict__ rev_pos,
                                                                   const unsigned short *__restrict__ heads,
                                                                   const float *__restrict__ weight,
                                                                   float *__restrict__ grad_points,
                                                                   float *__restrict__ dwx = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float gr_plane[]; // [tc] gradient values, then [tc] 16-bit positions
    unsigned short *gr_pos = reinterpret_cast<unsigned short *>(gr_plane + tc);
    const int t = threadIdx.x, ch = blockIdx.x, b = blockIdx.y;
    const int chunks = (T + tc - 1) / tc;
    const GT *g = grad_out + (size_t)b * go_bstride + (size_t)ch * (INTERP ? T / 3 : T);
    const float *wt = INTERP ? weight + (size_t)b * T : nullptr;
    const float *relb = DWX ? weight + (size_t)b * 3 * T : nullptr;
    float wx0 = 0.f, wx1 = 0.f, wx2 = 0.f;
    const int *rs = rev_start + (size_t)b * chunks * (n + 1);
    const unsigned short *rp = rev_pos + (size_t)b * T;
    const unsigned short *hd = heads + ((size_t)b * T >> 4);
    float4 pre[4];
    uint4 prp[2]; // sixteen 16-bit list entries
    unsigned hmask = 0xFFFFu;
    auto fetch = [&](int chunk) {
        const int p = t * 16, tt = chunk * tc + p;
        const bool in = p < tc && tt < T;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) {
                if constexpr (INTERP) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wt + tt + 4 * u);
                    const int q = tt + 4 * u;
                    pre[u] = make_float4(w4.x * ogc_ld1(g + q / 3), w4.y * ogc_ld1(g + (q + 1) / 3), w4.z * ogc_ld1(g + (q + 2) / 3),
                                         w4.w * ogc_ld1(g + (q + 3) / 3));
                } else {
                    pre[u] = ogc_ld4(g + tt + 4 * u);
                    if constexpr (DWX) {
                        const float4 r0 = *reinterpret_cast<const float4 *>(relb + tt + 4 * u);
                        const float4 r1 = *reinterpret_cast<const float4 *>(relb + T + tt + 4 * u);
                        const float4 r2 = *reinterpret_cast<const float4 *>(relb + 2 * (size_t)T + tt + 4 * u);
                        wx0 = fmaf(pre[u].w, r0.w, fmaf(pre[u].z, r0.z, fmaf(pre[u].y, r0.y, fmaf(pre[u].x, r0.x, wx0))));
                        wx1 = fmaf(pre[u].w, r1.w, fmaf(pre[u].z, r1.z, fmaf(pre[u].y, r1.y, fmaf(pre[u].x, r1.x, wx1))));
                        wx2 = fmaf(pre[u].w, r2.w, fmaf(pre[u].z, r2.z, fmaf(pre[u].y, r2.y, fmaf(pre[u].x, r2.x, wx2))));
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) prp[u] = in ? *reinterpret_cast<const uint4 *>(rp + tt + 8 * u) : make_uint4(0u, 0u, 0u, 0u);
        hmask = in ? hd[tt >> 4] : 0xFFFFu;
    };
    float acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = 0.0f;
    // the bounds of my points' lists in a chunk travel one chunk ahead like the values (2 NA loads out of the L2: read after the
    // barrier they were a dependent round trip in front of every chunk's walk)
    // (up to sixteen points per thread: with thirty-two the 64 registers of bounds spill)
    constexpr bool PFB = NA <= 16;
    int nlo[NA], nhi[NA];
    auto bounds = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int j = t + k * GR_THREADS;
            nlo[k] = nhi[k] = chunk * tc;
            if (j < n) {
                nlo[k] = rs[(size_t)chunk * (n + 1) + j];
                nhi[k] = rs[(size_t)chunk * (n + 1) + j + 1];
            }
        }
    };
    fetch(0);
    if (PFB) bounds(0);
    for (int chunk = 0; chunk < chunks; ++chunk) {
        {
            float v[16] = {pre[0].x, pre[0].y, pre[0].z, pre[0].w, pre[1].x, pre[1].y, pre[1].z, pre[1].w,
                           pre[2].x, pre[2].y, pre[2].z, pre[2].w, pre[3].x, pre[3].y, pre[3].z, pre[3].w};
#pragma unroll
            for (int u = 15; u >= 1; --u) v[u - 1] += ((hmask >> u) & 1u) ? 0.0f : v[u]; // a run accumulates into its head
            const int p = t * 16;
            if (p < tc) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    *reinterpret_cast<float4 *>(gr_plane + p + 4 * u) = make_float4(v[4 * u], v[4 * u + 1], v[4 * u + 2], v[4 * u + 3]);
                *reinterpret_cast<uint4 *>(gr_pos + p) = prp[0];
                *reinterpret_cast<uint4 *>(gr_pos + p + 8) = prp[1];
            }
        }
        __syncthreads();
        const int base = chunk * tc;
        if (!PFB) bounds(chunk);
        int cur[NA], end[NA], longest = 0;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            cur[k] = nlo[k] - base;
            end[k] = nhi[k] - base;
            longest = max(longest, end[k] - cur[k]);
        }
        if (chunk + 1 < chunks) { // in flight while the lists of this chunk are walked
            fetch(chunk + 1);
            if (PFB) bounds(chunk + 1);
        }
        for (int step = 0; step < longest; ++step) { // the NA lists of a thread advance together: NA independent reads
            float v[NA];
#pragma unroll
            for (int k = 0; k < NA; ++k) v[k] = cur[k] + step < end[k] ? gr_plane[gr_pos[cur[k] + step]] : 0.0f;
#pragma unroll
            for (int k = 0; k < NA; ++k) acc[k] += v[k];
        }
        __syncthreads();
    }
    float *gp = grad_points + ((size_t)b * c + ch) * n;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int j = t + k * GR_THREADS;
        if (j < n) gp[j] = acc[k];
    }
    if constexpr (DWX) { // the plane's share of dwx[ch][0..2]: wave sums, then one atomic per wavefront and column
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            wx0 += __shfl_down(wx0, off, 64);
            wx1 += __shfl_down(wx1, off, 64);
            wx2 += __shfl_down(wx2, off, 64);
        }
        if ((t & 63) == 0) {
            unsafeAtomicAdd(dwx + ch * 3, wx0);
            unsafeAtomicAdd(dwx + ch * 3 + 1, wx1);
            unsafeAtomicAdd(dwx + ch * 3 + 2, wx2);
        }
    }
}

} // namespace

extern "C" int ogc_group_reverse_chunk(int n, int npoints, int nsample) {
    // positions per chunk: 512 threads x one group of 16 positions (32 KiB of values + 16 KiB of list entries in LDS: three
    // workgroups per CU)
    if (n < 1 || (long long)npoints * nsample < 1) return 0;
    return 8192;
}

extern "C" int ogc_group_reverse(int b, int n, int npoints, int nsample, const int *idx, int *rev_start,
                                 unsigned short *rev_pos, unsigned short *heads, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 1 && npoints >= 0 && nsample >= 0 && (long long)npoints * nsample < (1ll << 31),
                "ogc_group_reverse: bad dimensions");
    const int T = npoints * nsample;
    if (b == 0 || T == 0) return OGC_OK;
    OGC_REQUIRE(idx && rev_start && rev_pos && heads, "ogc_group_reverse: null pointer");
    if (n > 16384 || b > 65535 || T % 16 != 0) {
        ogc_set_error("ogc_group_reverse: n <= 16384 (an LDS histogram over the points), b <= 65535, positions %% 16 == 0");
        return OGC_ERR_UNSUPPORTED;
    }
    const int tc = ogc_group_reverse_chunk(n, npoints, nsample);
    const int chunks = (T + tc - 1) / tc;
    if ((size_t)n * sizeof(int) > 60 * 1024) { // n = 16384: 64 KiB of histogram + the kernel's static words
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(group_reverse_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * (int)sizeof(int));
            once = true;
        }
    }
    hipLaunchKernelGGL(group_reverse_kernel, dim3(chunks, b), dim3(GRB_THREADS), (size_t)n * sizeof(int),
                       (hipStream_t)stream, n, T, tc, idx, rev_start, rev_pos, heads, ogc_deterministic() ? 1 : 0);
    OGC_CHECK_LAUNCH("ogc_group_reverse");
    return OGC_OK;
}

namespace {
template <typename GT>
int group_points_grad_rev_impl(int b, int c, int n, int npoints, int nsample, const GT *grad_out, const int *rev_start,
                               const unsigned short *rev_pos, const unsigned short *heads, float *grad_points,
                               ogc_stream_t stream, const float *rel = nullptr, float *dwx = nullptr) {
    OGC_REQUIRE(b >= 0 && c >= 0 && n >= 1 && npoints >= 0 && nsample >= 0 && (long long)npoints * nsample < (1ll << 31),
                "ogc_group_points_grad_rev: bad dimensions");
    const int T = npoints * nsample;
    if (b == 0 || c == 0) return OGC_OK;
    OGC_REQUIRE(grad_out && rev_start && rev_pos && heads && grad_points, "ogc_group_points_grad_rev: null pointer");
    if (T % 16 != 0 || ((uintptr_t)grad_out & ogc_act_mask<GT>()) != 0 || !aligned16(rev_pos) || n > 32 * GR_THREADS || b > 65535) {
        ogc_set_error("ogc_group_points_grad_rev: needs npoints * nsample %% 16 == 0, 16-byte aligned tensors and n <= 16384");
        return OGC_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (T == 0) {
        if (ogc_zero_async(grad_points, sizeof(float) * (size_t)b * c * n, s) != hipSuccess) return OGC_ERR_LAUNCH;
        return OGC_OK;
    }
    const int tc = ogc_group_reverse_chunk(n, npoints, nsample);
    const size_t lds = (size_t)tc * (sizeof(float) + sizeof(unsigned short));
    const long long go_bstride = (long long)c * T;
    dim3 grid(c, b);
    if (dwx) {
        if (!rel || !aligned16(rel) || ogc_zero_async(dwx, sizeof(float) * 3 * (size_t)c, s) != hipSuccess) {
            ogc_set_error("ogc_group_points_grad_rev_dwx: rel missing / misaligned, or the fill of dwx failed");
            return OGC_ERR_INVALID_ARG;
        }
    }
#define GR_LAUNCH(NAV)                                                                                                         \
    do {                                                                                                                       \
        if (dwx)                                                                                                               \
            hipLaunchKernelGGL((group_bwd_rev_kernel<NAV, false, GT, true>), grid, dim3(GR_THREADS), lds, s, c, n, T, tc,       \
                               go_bstride, grad_out, rev_start, rev_pos, heads, rel, grad_points, dwx);                        \
        else                                                                                                                   \
            hipLaunchKernelGGL((group_bwd_rev_kernel<NAV, false, GT>), grid, dim3(GR_THREADS), lds, s, c, n, T, tc, go_bstride, \
                               grad_out, rev_start, rev_pos, heads, nullptr, grad_points);                                     \
    } while (0)
    if (n <= GR_THREADS) GR_LAUNCH(1);
    else if (n <= 2 * GR_THREADS) GR_LAUNCH(2);
    else if (n <= 4 * GR_THREADS) GR_LAUNCH(4);
    else if (n <= 8 * GR_THREADS) GR_LAUNCH(8);
    else if (n <= 16 * GR_THREADS) GR_LAUNCH(16);
    else GR_LAUNCH(32);
#undef GR_LAUNCH
    OGC_CHECK_LAUNCH("ogc_group_points_grad_rev");
    return OGC_OK;
}
} // namespace

extern "C" int ogc_group_points_grad_rev(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                         const int *rev_start, const unsigned short *rev_pos,
                                         const unsigned short *heads, float *grad_points, ogc_stream_t stream) {
    return group_points_grad_rev_impl<float>(b, c, n, npoints, nsample, grad_out, rev_start, rev_pos, heads, grad_points, stream);
}

// ... and, from the same pass, the three coordinate columns of a grouped first layer's weight gradient: dwx (c, 3) = sum over samples
// and positions of grad_out[b, ch, pos] * rel[b, k, pos] (rel (b, 3, npoints, nsample) as for ogc_group_linear_fwd; dwx is zeroed
// here) — what ogc_conv1x1_wgrad(rel, grad_out) computes by reading grad_out a second time.
extern "C" int ogc_group_points_grad_rev_dwx(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                             const int *rev_start, const unsigned short *rev_pos, const unsigned short *heads,
                                             const float *rel, float *grad_points, float *dwx, ogc_stream_t stream) {
    OGC_REQUIRE(rel && dwx, "ogc_group_points_grad_rev_dwx: null pointer");
    return group_points_grad_rev_impl<float>(b, c, n, npoints, nsample, grad_out, rev_start, rev_pos, heads, grad_points, stream, rel,
                                             dwx);
}

extern "C" int ogc_group_points_grad_rev_dwx_h(int b, int c, int n, int npoints, int nsample, const ogc_bf16_t *grad_out,
                                               const int *rev_start, const unsigned short *rev_pos, const unsigned short *heads,
                                               const float *rel, float *grad_points, float *dwx, ogc_stream_t stream) {
    OGC_REQUIRE(rel && dwx, "ogc_group_points_grad_rev_dwx_h: null pointer");
    return group_points_grad_rev_impl<ogc_bf16>(b, c, n, npoints, nsample, grad_out, rev_start, rev_pos, heads, grad_points, stream,
                                                rel, dwx);
}

extern "C" int ogc_group_points_grad_rev_h(int b, int c, int n, int npoints, int nsample, const ogc_bf16_t *grad_out,
                                           const int *rev_start, const unsigned short *rev_pos,
                                           const unsigned short *heads, float *grad_points, ogc_stream_t stream) {
    return group_points_grad_rev_impl<ogc_bf16>(b, c, n, npoints, nsample, grad_out, rev_start, rev_pos, heads, grad_points, stream);
}

// grad_out_bstride: floats between the (c, n) planes of consecutive samples (c n for a dense tensor; larger when grad_out is a
// channel slice of a wider gradient, as autograd hands it to the interpolation underneath a concatenation)
extern "C" int ogc_three_interpolate_grad_rev_bs(int b, int c, int n, int m, const float *grad_out, long long grad_out_bstride,
                                                 const float *weight, const int *rev_start, const unsigned short *rev_pos,
                                                 const unsigned short *heads, float *grad_points, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 1 && (long long)n * 3 < (1ll << 31), "ogc_three_interpolate_grad_rev: bad dimensions");
    OGC_REQUIRE(grad_out_bstride >= (long long)c * n, "ogc_three_interpolate_grad_rev: batch stride of grad_out below c n");
    if (b == 0 || c == 0) return OGC_OK;
    OGC_REQUIRE(grad_out && weight && rev_start && rev_pos && heads && grad_points, "ogc_three_interpolate_grad_rev: null pointer");
    const int T = 3 * n;
    if (T % 16 != 0 || !aligned16(weight) || !aligned16(rev_pos) || m > 32 * GR_THREADS || b > 65535) {
        ogc_set_error("ogc_three_interpolate_grad_rev: needs 3 n %% 16 == 0, 16-byte aligned tensors and m <= 16384");
        return OGC_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (T == 0) {
        if (ogc_zero_async(grad_points, sizeof(float) * (size_t)b * c * m, s) != hipSuccess) return OGC_ERR_LAUNCH;
        return OGC_OK;
    }
    const int tc = ogc_group_reverse_chunk(m, n, 3);
    const size_t lds = (size_t)tc * (sizeof(float) + sizeof(unsigned short));
    const long long go_bstride = grad_out_bstride;
    dim3 grid(c, b);
#define GR_LAUNCH(NAV)                                                                                                        \
    hipLaunchKernelGGL((group_bwd_rev_kernel<NAV, true>), grid, dim3(GR_THREADS), lds, s, c, m, T, tc, go_bstride, grad_out, \
                       rev_start, rev_pos, heads, weight, grad_points)
    if (m <= GR_THREADS) GR_LAUNCH(1);
    else if (m <= 2 * GR_THREADS) GR_LAUNCH(2);
    else if (m <= 4 * GR_THREADS) GR_LAUNCH(4);
    else if (m <= 8 * GR_THREADS) GR_LAUNCH(8);
    else if (m <= 16 * GR_THREADS) GR_LAUNCH(16);
    else GR_LAUNCH(32);
#undef GR_LAUNCH
    OGC_CHECK_LAUNCH("ogc_three_interpolate_grad_rev");
    return OGC_OK;
}

extern "C" int ogc_three_interpolate_grad_rev(int b, int c, int n, int m, const float *grad_out, const float *weight,
                                              const int *rev_start, const unsigned short *rev_pos,
                                              const unsigned short *heads, float *grad_points, ogc_stream_t stream) {
    return ogc_three_interpolate_grad_rev_bs(b, c, n, m, grad_out, (long long)(c > 0 ? c : 0) * (n > 0 ? n : 0), weight, rev_start,
                                             rev_pos, heads, grad_points, stream);
}
