// fps.hip — iterative furthest-point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (reference: pointnet2/src/sampling_gpu.cu:93-253).
// The reference re-reads xyz and temp from global memory every round and runs an 11-barrier
// shared-memory tree per round.  Here one workgroup (up to 16 wavefronts) owns one cloud and keeps
// the cloud AND the running min-distances in registers for the whole run (N <= 16384); a round is
//   VALU update -> DPP wave max -> LDS (16 floats) -> barrier -> tie resolution -> barrier.
//
// Tie order.  The reference's winner among equal maxima is fixed by its reduction shape: the
// strided per-thread scan keeps the smallest k (strict '>', sampling_gpu.cu:136-137) and each tree
// step keeps the left operand on ties (__update, :86-91).  Unrolled, that is a total order: among
// points with the maximal value the winner minimises
//       rank(k) = bitrev_{log2 bs}(k % bs) * S + k / bs,   S = ceil(N / bs),
// with bs = min(1024, 2^floor(log2 N)) the reference's block size (cuda_utils.h:10-14).  The kernel
// reduces (value, rank) explicitly, so its own launch shape is free to differ from the reference's.
#include <math.h>

#include "ogc_common.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned fps_rank(int k, int bs_mask, int bs_shift, int S) {
    const unsigned tid = (unsigned)k & (unsigned)bs_mask;
    const unsigned rev = bs_shift ? (__brev(tid) >> (32 - bs_shift)) : 0u;
    return rev * (unsigned)S + ((unsigned)k >> bs_shift);
}

// Register-resident variant: THREADS lanes, PTS points per lane (point k = t + j*THREADS).
template <int PTS, int THREADS>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(int n, int m, int bs_shift,
                                                          const float *__restrict__ xyz,
                                                          float *__restrict__ temp,
                                                          int *__restrict__ idxs) {
    constexpr int NW = THREADS / OGC_WAVE;
    __shared__ float s_wmax[2][16];
    __shared__ u64 s_best[2];

    const int t = threadIdx.x;
    const int b = blockIdx.x;
    const float *__restrict__ dataset = xyz + (size_t)b * n * 3;
    float *tmp = temp + (size_t)b * n;
    int *out = idxs + (size_t)b * m;

    const int bs_mask = (1 << bs_shift) - 1;
    const int S = (n + bs_mask) >> bs_shift;

    float px[PTS], py[PTS], pz[PTS], td[PTS];
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        const int k = t + j * THREADS;
        if (k < n) {
            px[j] = dataset[k * 3 + 0];
            py[j] = dataset[k * 3 + 1];
            pz[j] = dataset[k * 3 + 2];
            td[j] = tmp[k];
        } else {
            px[j] = py[j] = pz[j] = 0.0f;
            td[j] = -1.0f; // padding never wins (real values are >= 0)
        }
    }
    if (t == 0) {
        out[0] = 0;
        s_best[0] = ~0ull;
        s_best[1] = ~0ull;
    }
    __syncthreads();

    int old = 0;
    for (int r = 1; r < m; ++r) {
        const int par = r & 1;
        const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
        float tmax = -1.0f;
#pragma unroll
        for (int j = 0; j < PTS; ++j) {
            const float d = ogc_sqdist(px[j], py[j], pz[j], x1, y1, z1);
            // padding lanes keep -1: fminf(d, -1) = -1
            td[j] = fminf(d, td[j]);
            tmax = fmaxf(tmax, td[j]);
        }
        const float wmax = ogc_wave_max_f32(tmax);
        if ((t & (OGC_WAVE - 1)) == 0) s_wmax[par][t >> 6] = wmax;
        __syncthreads();
        float gmax = s_wmax[par][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) gmax = fmaxf(gmax, s_wmax[par][w]);
        if (tmax == gmax) {
            u64 best = ~0ull;
#pragma unroll
            for (int j = 0; j < PTS; ++j) {
                const int k = t + j * THREADS;
                if (td[j] == gmax) {
                    const u64 key = ((u64)fps_rank(k, bs_mask, bs_shift, S) << 32) | (unsigned)k;
                    best = key < best ? key : best;
                }
            }
            atomicMin(&s_best[par], best);
        }
        if (t == 0) s_best[par ^ 1] = ~0ull;
        __syncthreads();
        old = __builtin_amdgcn_readfirstlane((int)(unsigned)s_best[par]);
        if (t == 0) out[r] = old;
    }
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        const int k = t + j * THREADS;
        if (k < n) tmp[k] = td[j];
    }
}

// Large-N fallback (N > 16384): same rounds, but xyz/temp stay in global memory (L2-resident).
__global__ __launch_bounds__(1024) void fps_mem_kernel(int n, int m, int bs_shift,
                                                       const float *__restrict__ xyz,
                                                       float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ float s_wmax[2][16];
    __shared__ u64 s_best[2];
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    const float *__restrict__ dataset = xyz + (size_t)b * n * 3;
    float *tmp = temp + (size_t)b * n;
    int *out = idxs + (size_t)b * m;
    const int bs_mask = (1 << bs_shift) - 1;
    const int S = (n + bs_mask) >> bs_shift;
    if (t == 0) {
        out[0] = 0;
        s_best[0] = ~0ull;
        s_best[1] = ~0ull;
    }
    __syncthreads();
    int old = 0;
    for (int r = 1; r < m; ++r) {
        const int par = r & 1;
        const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
        float tmax = -1.0f;
        u64 tbest = ~0ull;
        for (int k = t; k < n; k += 1024) {
            const float d = ogc_sqdist(dataset[k * 3 + 0], dataset[k * 3 + 1], dataset[k * 3 + 2], x1, y1, z1);
            const float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            const u64 key = ((u64)fps_rank(k, bs_mask, bs_shift, S) << 32) | (unsigned)k;
            if (d2 > tmax) { tmax = d2; tbest = key; }
            else if (d2 == tmax && key < tbest) tbest = key;
        }
        const float wmax = ogc_wave_max_f32(tmax);
        if ((t & 63) == 0) s_wmax[par][t >> 6] = wmax;
        __syncthreads();
        float gmax = s_wmax[par][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) gmax = fmaxf(gmax, s_wmax[par][w]);
        if (tmax == gmax) atomicMin(&s_best[par], tbest);
        if (t == 0) s_best[par ^ 1] = ~0ull;
        __syncthreads();
        old = __builtin_amdgcn_readfirstlane((int)(unsigned)s_best[par]);
        if (t == 0) out[r] = old;
    }
}

// cuda_utils.h:10-14 opt_n_threads(), evaluated through double log() exactly like the reference.
int fps_ref_block_shift(int work_size) {
    int pow_2 = (int)(log((double)work_size) / log(2.0));
    if (pow_2 > 10) pow_2 = 10;
    if (pow_2 < 0) pow_2 = 0;
    return pow_2;
}

} // namespace

#define FPS_LAUNCH(PTS, THREADS)                                                                  \
    hipLaunchKernelGGL((fps_reg_kernel<PTS, THREADS>), dim3(b), dim3(THREADS), 0, (hipStream_t)stream, \
                       n, m, shift, xyz, temp, idx)

extern "C" int ogc_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                           ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ogc_furthest_point_sampling: negative dimension");
    if (b == 0 || m <= 0) return OGC_OK; // sampling_gpu.cu:98
    OGC_REQUIRE(n >= 1, "ogc_furthest_point_sampling: n must be >= 1 when m > 0");
    OGC_REQUIRE(xyz && temp && idx, "ogc_furthest_point_sampling: null pointer");
    OGC_REQUIRE((long long)b * n * 3 < (1ll << 31), "ogc_furthest_point_sampling: xyz exceeds 32-bit indexing");
    const int shift = fps_ref_block_shift(n);
    if (n <= 64) FPS_LAUNCH(1, 64);
    else if (n <= 128) FPS_LAUNCH(1, 128);
    else if (n <= 256) FPS_LAUNCH(1, 256);
    else if (n <= 512) FPS_LAUNCH(1, 512);
    else if (n <= 1024) FPS_LAUNCH(1, 1024);
    else if (n <= 2048) FPS_LAUNCH(2, 1024);
    else if (n <= 4096) FPS_LAUNCH(4, 1024);
    else if (n <= 8192) FPS_LAUNCH(8, 1024);
    else if (n <= 16384) FPS_LAUNCH(16, 1024);
    else
        hipLaunchKernelGGL(fps_mem_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, n, m, shift, xyz,
                           temp, idx);
    OGC_CHECK_LAUNCH("ogc_furthest_point_sampling");
    return OGC_OK;
}
