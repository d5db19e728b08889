// knn.hip — brute-force exact k-nearest-neighbour search for gfx950.
//
// Replaces knn_kernel_fast / three_nn_kernel_fast (reference: pointnet2/src/interpolate_gpu.cu:9-57,
// :81-124).  The reference keeps a sorted `double best[200]` per thread and shifts it on every hit;
// on a 64-wide wavefront that makes every lane pay for any lane's insertion.  This kernel:
//
//   * one lane = one query, one workgroup = one wavefront (no barriers at all);
//   * the candidate point is wave-uniform, so it is fetched with scalar loads and fed to the VALU
//     as SGPR operands (no LDS / vector-memory traffic in the scan loop);
//   * a lane's current k best live in an LDS max-heap of 64-bit keys (dist_bits << 32 | index);
//     a candidate is admitted only if d < tau (tau = heap top once full, +inf before), which is
//     exactly the reference's strict `d < best[k-1]` test — ties keep the lower index because
//     candidates arrive in index order;
//   * admitted candidates go to a small per-lane pending buffer (one predicated ds_write); the
//     heap is only touched when some lane's buffer is full, so the divergent part is amortised;
//   * at the end the heap is heap-sorted in place (ascending (dist, index)) and written out with
//     fully coalesced stores.
//
// "k smallest by (dist, index)" is what the reference's stable insertion computes, so results
// are bit-identical: squared distance in fp32 with the reference's rounding sequence, non-finite
// distances never selected, idx=0 / dist2=+inf tail when fewer than k candidates qualify.
#include <stdlib.h>

#include "ogc_common.h"
#include "grid.h"

namespace {

typedef unsigned long long u64;

constexpr int KNN_BUF = 16;     // pending candidates per lane between heap updates
constexpr int KNN_GROUP = 8;    // candidates evaluated back to back (ILP) before one wave-level test
constexpr int KNN_LSTRIDE = 65; // LDS row stride (in elements) -> conflict-free both ways

__device__ __forceinline__ u64 knn_pack(float d, int i) {
    return ((u64)__float_as_uint(d) << 32) | (unsigned)i;
}
__device__ __forceinline__ float knn_key_dist(u64 key) { return __uint_as_float((unsigned)(key >> 32)); }

// Insert `key` into the lane's max-heap (column `lane` of heap[][KNN_LSTRIDE]).
__device__ __forceinline__ void knn_heap_insert(u64 *heap, int lane, int k, int &hsize, float &tau,
                                                u64 key) {
    if (hsize < k) { // sift up
        int pos = hsize++;
        while (pos > 0) {
            const int parent = (pos - 1) >> 1;
            const u64 pk = heap[parent * KNN_LSTRIDE + lane];
            if (pk >= key) break;
            heap[pos * KNN_LSTRIDE + lane] = pk;
            pos = parent;
        }
        heap[pos * KNN_LSTRIDE + lane] = key;
        if (hsize == k) tau = knn_key_dist(heap[lane]);
    } else { // replace the root, sift down
        int pos = 0;
        for (;;) {
            int c = 2 * pos + 1;
            if (c >= k) break;
            u64 ck = heap[c * KNN_LSTRIDE + lane];
            if (c + 1 < k) {
                const u64 ck2 = heap[(c + 1) * KNN_LSTRIDE + lane];
                if (ck2 > ck) { ck = ck2; ++c; }
            }
            if (ck <= key) break;
            heap[pos * KNN_LSTRIDE + lane] = ck;
            pos = c;
        }
        heap[pos * KNN_LSTRIDE + lane] = key;
        tau = knn_key_dist(heap[lane]);
    }
}

__device__ __forceinline__ void knn_flush(u64 *heap, const u64 *buf, int lane, int k, int &hsize,
                                          float &tau, int &nbuf) {
    for (int j = 0; j < KNN_BUF; ++j) {
        if (j < nbuf) {
            const u64 key = buf[j * KNN_LSTRIDE + lane];
            if (knn_key_dist(key) < tau) knn_heap_insert(heap, lane, k, hsize, tau, key);
        }
    }
    nbuf = 0;
}

// MODE 0: write squared distances (ogc_knn).  MODE 1: write sqrt(dist2) and clamp indices whose
// distance exceeds `radius` to the nearest neighbour (ogc_knn_clamped).
template <int MODE>
__global__ __launch_bounds__(OGC_WAVE) void knn_heap_kernel(int n, int m, int k, float radius,
                                                            const float *__restrict__ unknown,
                                                            const float *__restrict__ known,
                                                            float *__restrict__ dist_out,
                                                            int *__restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) u64 knn_smem[];
    float *tile = reinterpret_cast<float *>(knn_smem);       // [OGC_TILE_FLOATS] candidate tile
    u64 *heap = knn_smem + OGC_TILE_FLOATS / 2;              // [k][KNN_LSTRIDE]
    u64 *buf = heap + k * KNN_LSTRIDE;                       // [KNN_BUF][KNN_LSTRIDE]
    int *hsizes = (int *)(buf + KNN_BUF * KNN_LSTRIDE); // [64]

    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * OGC_WAVE;
    const int q = q0 + lane;
    const int qc = q < n ? q : n - 1; // idle lanes shadow the last query (never written)

    const float *u = unknown + ((size_t)b * n + qc) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *__restrict__ kn = known + (size_t)b * m * 3;

    float tau = INFINITY;
    int hsize = 0, nbuf = 0;

    // Admitted candidates (d < tau) are appended to the pending buffer in index order; the heap is only touched
    // when some lane's buffer could overflow on the next group (see ogc_scan_candidates for the scan structure).
    auto try_admit = [&](float d, int i) {
        if (d < tau) {
            buf[nbuf * KNN_LSTRIDE + lane] = knn_pack(d, i);
            ++nbuf;
        }
    };
    auto maybe_flush = [&]() {
        if (__builtin_amdgcn_ballot_w64(nbuf > KNN_BUF - KNN_GROUP) != 0) knn_flush(heap, buf, lane, k, hsize, tau, nbuf);
    };
    ogc_scan_candidates(kn, m, ux, uy, uz, tile, lane, [&](const float (&d)[8], int base) {
        if (__builtin_amdgcn_ballot_w64(ogc_min8_f32(d) < tau) == 0) return false; // one branch per group
        unsigned long long mk[8];
        ogc_masks8(d, tau, mk);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (mk[u] != 0) {               // wave-uniform: a scalar branch skips candidates no lane admits
                asm volatile("" ::: "memory");
                try_admit(d[u], base + u);
            }
        maybe_flush();
        return false;
    });
    knn_flush(heap, buf, lane, k, hsize, tau, nbuf);

    // heap sort in place: ascending (dist, index) in heap[0..hsize)
    for (int end = hsize - 1; end > 0; --end) {
        const u64 key = heap[end * KNN_LSTRIDE + lane];
        heap[end * KNN_LSTRIDE + lane] = heap[lane];
        int pos = 0;
        for (;;) {
            int c = 2 * pos + 1;
            if (c >= end) break;
            u64 ck = heap[c * KNN_LSTRIDE + lane];
            if (c + 1 < end) {
                const u64 ck2 = heap[(c + 1) * KNN_LSTRIDE + lane];
                if (ck2 > ck) { ck = ck2; ++c; }
            }
            if (ck <= key) break;
            heap[pos * KNN_LSTRIDE + lane] = ck;
            pos = c;
        }
        heap[pos * KNN_LSTRIDE + lane] = key;
    }
    hsizes[lane] = hsize;
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): LDS writes above visible to the wave
    __builtin_amdgcn_wave_barrier();

    // coalesced write-out: the wave's 64 rows of k entries are contiguous in memory
    const int nq = min(OGC_WAVE, n - q0);
    const size_t base = ((size_t)b * n + q0) * k;
    const int total = nq * k;
    for (int t = lane; t < total; t += OGC_WAVE) {
        const int ql = t / k;
        const int j = t - ql * k;
        const int hs = hsizes[ql];
        float d = INFINITY;
        int id = 0;
        if (j < hs) {
            const u64 key = heap[j * KNN_LSTRIDE + ql];
            d = knn_key_dist(key);
            id = (int)(unsigned)key;
        }
        if (MODE == 1) {
            d = sqrtf(d); // IEEE-correct (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
            if (d > radius && radius >= 0.0f) { // clamped entries carry dist = +inf (see include/ogc_ops.h)
                id = hs > 0 ? (int)(unsigned)heap[ql] : 0;
                d = INFINITY;
            }
        }
        dist_out[base + t] = d;
        idx_out[base + t] = id;
    }
}

// k = 3 in registers (interpolate_gpu.cu:101-121: if / else-if / else-if chain with strict '<').
__global__ __launch_bounds__(OGC_WAVE) void three_nn_kernel(int n, int m, const float *__restrict__ unknown,
                                                            const float *__restrict__ known,
                                                            float *__restrict__ dist2, int *__restrict__ idx) {
    __shared__ __attribute__((aligned(16))) float tile[OGC_TILE_FLOATS];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int q = blockIdx.x * OGC_WAVE + lane;
    const int qc = q < n ? q : n - 1;
    const float *u = unknown + ((size_t)b * n + qc) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    const float *__restrict__ kn = known + (size_t)b * m * 3;

    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    auto update = [&](float d, int i) { // interpolate_gpu.cu:109-120
        const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
        b3 = c2 ? b2 : (c3 ? d : b3);
        i3 = c2 ? i2 : (c3 ? i : i3);
        b2 = c1 ? b1 : (c2 ? d : b2);
        i2 = c1 ? i1 : (c2 ? i : i2);
        b1 = c1 ? d : b1;
        i1 = c1 ? i : i1;
    };
    ogc_scan_candidates(kn, m, ux, uy, uz, tile, lane, [&](const float (&d)[8], int base) {
        if (__builtin_amdgcn_ballot_w64(ogc_min8_f32(d) < b3) == 0) return false;
        unsigned long long mk[8];
        ogc_masks8(d, b3, mk); // b3 only shrinks, so a candidate outside its mask can never enter later
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (mk[u] != 0) {
                asm volatile("" ::: "memory");
                update(d[u], base + u);
            }
        return false;
    });
    if (q < n) {
        float *o = dist2 + ((size_t)b * n + q) * 3;
        int *oi = idx + ((size_t)b * n + q) * 3;
        o[0] = b1; o[1] = b2; o[2] = b3;
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
}

size_t knn_lds_bytes(int k) {
    return OGC_TILE_FLOATS * sizeof(float) + (size_t)(k + KNN_BUF) * KNN_LSTRIDE * sizeof(u64) + OGC_WAVE * sizeof(int);
}

template <int MODE>
int knn_launch(const char *name, int b, int n, int m, int k, float radius, const float *unknown,
               const float *known, float *dist, int *idx, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && m >= 0, "%s: negative dimension (b=%d n=%d m=%d)", name, b, n, m);
    OGC_REQUIRE(k >= 1 && k <= 200, "%s: k=%d outside [1,200] (reference bound, interpolate_gpu.cu:30)",
                name, k);
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(unknown && known && dist && idx, "%s: null pointer", name);
    OGC_REQUIRE((long long)b * n * k < (1ll << 31) && (long long)b * m * 3 < (1ll << 31),
                "%s: tensor exceeds 32-bit indexing", name);
    // Cell-list path (identical results); the all-pairs scan below remains for small clouds and as the fallback.
    // OGC_KNN=brute|grid forces a path (development / tests).
    static const char *mode = getenv("OGC_KNN");
    if (!(mode && mode[0] == 'b')) {
        const int rc = ogc_knn_grid(MODE, b, n, m, k, radius, unknown, known, dist, idx, (hipStream_t)stream);
        if (rc != OGC_ERR_UNSUPPORTED) return rc;
        if (mode && mode[0] == 'g') {
            ogc_set_error("%s: grid path forced but not applicable (m=%d, k=%d)", name, m, k);
            return OGC_ERR_UNSUPPORTED;
        }
    }
    dim3 grid(ogc_divup(n, OGC_WAVE), b);
    hipLaunchKernelGGL(knn_heap_kernel<MODE>, grid, dim3(OGC_WAVE), knn_lds_bytes(k),
                       (hipStream_t)stream, n, m, k, radius, unknown, known, dist, idx);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}

} // namespace

extern "C" int ogc_knn(int b, int n, int m, int k, const float *unknown, const float *known,
                       float *dist2, int *idx, ogc_stream_t stream) {
    return knn_launch<0>("ogc_knn", b, n, m, k, -1.0f, unknown, known, dist2, idx, stream);
}

extern "C" int ogc_knn_clamped(int b, int n, int m, int k, float radius, const float *unknown,
                               const float *known, float *dist, int *idx, ogc_stream_t stream) {
    return knn_launch<1>("ogc_knn_clamped", b, n, m, k, radius, unknown, known, dist, idx, stream);
}

extern "C" int ogc_three_nn(int b, int n, int m, const float *unknown, const float *known,
                            float *dist2, int *idx, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n >= 0 && m >= 0, "ogc_three_nn: negative dimension");
    if (b == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(unknown && known && dist2 && idx, "ogc_three_nn: null pointer");
    {   // known clouds of 1024 points or more: a cell-list search (grid.hip) instead of the all-pairs scan
        const int rc = ogc_three_nn_grid(b, n, m, unknown, known, dist2, idx, (hipStream_t)stream);
        if (rc != OGC_ERR_UNSUPPORTED) return rc;
    }
    dim3 grid(ogc_divup(n, OGC_WAVE), b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(OGC_WAVE), 0, (hipStream_t)stream, n, m, unknown, known,
                       dist2, idx);
    OGC_CHECK_LAUNCH("ogc_three_nn");
    return OGC_OK;
}

// 1 when this library's search kernels evaluate the squared distance as `nvcc --fmad=true` contracts it (libogc_ops_fmad.so), else 0
extern "C" int ogc_distance_contracted(void) {
#ifdef OGC_FMAD
    return 1;
#else
    return 0;
#endif
}
