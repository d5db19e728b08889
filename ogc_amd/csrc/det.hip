// det.hip — the deterministic-gradient mode (ogc_set_deterministic / OGC_DETERMINISTIC=1): sums in a FIXED order.
//
// The reference accumulates its gradients with float atomics (pointnet2/src/group_points_gpu.cu:24, interpolate_gpu.cu:211-213,
// sampling_gpu.cu:62), whose order — and therefore whose last bits — differ from launch to launch; so do this library's fast
// kernels (scatter-adds, split-K weight gradients, per-workgroup partial statistics).  SURVEY.md §5 asks for a switch that
// removes this for tests.  Two mechanisms cover every accumulating kernel that is converted (include/ogc_ops.h lists them):
//
//   * PARTIALS + ORDERED PASS  — a kernel whose workgroups each add ONE partial per output element (statistics, split-K weight
//     gradient tiles) stores the partial into slab `slot` of a scratch buffer instead (slot = the workgroup's index along the
//     split), and ogc_det_reduce_* adds the slabs to the destination in slot order, one thread per element;
//   * GATHER OVER SORTED LISTS — a scatter-add out[idx[t]] += v[t] (grouping / gathering / interpolation gradients, the Chamfer
//     term) becomes: per destination the list of its source positions, sorted ascending, summed by ONE thread in that order.
//
// A test mode: slower than the atomics (ordered passes, lists rebuilt per call), bit-reproducible from run to run and from
// process to process for a fixed input, device and library build.
#include <map>
#include <mutex>
#include <utility>

#include "ogc_common.h"

namespace {
int g_deterministic = 0;
} // namespace

bool ogc_deterministic() { return g_deterministic != 0; }

extern "C" int ogc_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return OGC_OK;
}

extern "C" int ogc_get_deterministic(void) { return g_deterministic; }

// Scratch of the deterministic mode: one grow-only buffer per (device, stream), apart from ogc_workspace (the cell grids of the
// searches live there and must survive the gradient kernels of the same stream).
void *ogc_det_scratch(hipStream_t stream, size_t bytes) {
    struct Slot { void *ptr; size_t size; };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Slot> slots;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Slot &sl = slots[{dev, stream}];
    if (sl.size >= bytes && sl.ptr) return sl.ptr;
    if (sl.ptr) (void)hipFreeAsync(sl.ptr, stream);
    sl.ptr = nullptr;
    sl.size = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    void *p = nullptr;
    if (hipMallocAsync(&p, want, stream) != hipSuccess || !p) {
        (void)hipGetLastError();
        return nullptr;
    }
    sl.ptr = p;
    sl.size = want;
    return p;
}

namespace {

// dst[e] (+)= part[0][e] + part[1][e] + ... in slot order (a left-to-right chain: the order IS the definition)
template <typename T>
__global__ __launch_bounds__(256) void det_reduce_kernel(T *__restrict__ dst, const T *__restrict__ part, int slots, long long n,
                                                         int accumulate) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    T acc = part[e];
    for (int s = 1; s < slots; ++s) acc += part[(long long)s * n + e];
    dst[e] = accumulate ? dst[e] + acc : acc;
}

// ---- lists "which positions t point at destination j", per sample, positions ascending -------------------------------------
__global__ __launch_bounds__(256) void det_count_kernel(long long T, int n, const int *__restrict__ idx, int *__restrict__ start) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T) return;
    const int j = idx[(size_t)b * T + t];
    if (j >= 0 && j < n) atomicAdd(&start[(size_t)b * (n + 1) + j], 1); // integer: the counts do not depend on the order
}

// one workgroup per sample: exclusive scan of n counts in place (n + 1 entries), and a copy as the fill cursor
__global__ __launch_bounds__(1024) void det_scan_kernel(int n, int *__restrict__ start, int *__restrict__ cursor) {
    __shared__ int part[1024];
    const int t = threadIdx.x, b = blockIdx.x;
    int *d = start + (size_t)b * (n + 1);
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

__global__ __launch_bounds__(256) void det_fill_kernel(long long T, int n, const int *__restrict__ idx, int *__restrict__ cursor,
                                                       int *__restrict__ pos) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T) return;
    const int j = idx[(size_t)b * T + t];
    if (j < 0 || j >= n) return;
    const int p = atomicAdd(&cursor[(size_t)b * n + j], 1); // the slot inside the list is arbitrary here: sorted below
    pos[(size_t)b * T + p] = (int)t;
}

// one thread per destination: its list ascending.  Short lists by insertion; long ones (a popular padding index) by heap sort —
// both in place, both with a result that depends on the SET of entries only.
__device__ void det_sort_ints(int *a, int len) {
    if (len <= 32) {
        for (int i = 1; i < len; ++i) {
            const int v = a[i];
            int k = i - 1;
            while (k >= 0 && a[k] > v) {
                a[k + 1] = a[k];
                --k;
            }
            a[k + 1] = v;
        }
        return;
    }
    auto sift = [&](int root, int end) {
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) return;
            if (child + 1 < end && a[child + 1] > a[child]) ++child;
            if (a[root] >= a[child]) return;
            const int tmp = a[root];
            a[root] = a[child];
            a[child] = tmp;
            root = child;
        }
    };
    for (int i = len / 2 - 1; i >= 0; --i) sift(i, len);
    for (int end = len - 1; end > 0; --end) {
        const int tmp = a[0];
        a[0] = a[end];
        a[end] = tmp;
        sift(0, end);
    }
}

__global__ __launch_bounds__(256) void det_sort_kernel(int n, long long T, const int *__restrict__ start, int *__restrict__ pos) {
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= n) return;
    const int *rs = start + (size_t)b * (n + 1);
    det_sort_ints(pos + (size_t)b * T + rs[j], rs[j + 1] - rs[j]);
}

// out[b, ch, j] (+)= sum over the list of j, ascending t, of value(b, ch, t)
//   plain:   value = grad_out[b * go_bstride + ch * T + t]
//   INTERP:  value = grad_out[b * go_bstride + ch * (T / 3) + t / 3] * weight[b * T + t]      (fp32 product, then the sum)
template <bool INTERP>
__global__ __launch_bounds__(256) void det_gather_sum_kernel(int c, int n, long long T, long long go_bstride,
                                                             const float *__restrict__ grad_out, const float *__restrict__ weight,
                                                             const int *__restrict__ start, const int *__restrict__ pos,
                                                             float *__restrict__ out, int accumulate) {
    const int j = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y, b = blockIdx.z;
    if (j >= n) return;
    const int *rs = start + (size_t)b * (n + 1);
    const int *ps = pos + (size_t)b * T;
    const float *g = grad_out + (size_t)b * go_bstride + (size_t)ch * (INTERP ? T / 3 : T);
    const float *w = INTERP ? weight + (size_t)b * T : nullptr;
    float acc = 0.0f;
    for (int e = rs[j]; e < rs[j + 1]; ++e) {
        const int t = ps[e];
        acc += INTERP ? __fmul_rn(g[t / 3], w[t]) : g[t];
    }
    float *o = out + ((size_t)b * c + ch) * n + j;
    *o = accumulate ? *o + acc : acc;
}

} // namespace

hipError_t ogc_det_reduce_f32(float *dst, const float *part, int slots, long long n, int accumulate, hipStream_t s) {
    if (n <= 0 || slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(det_reduce_kernel<float>, dim3(ogc_divup(n, 256)), dim3(256), 0, s, dst, part, slots, n, accumulate);
    return hipGetLastError();
}

hipError_t ogc_det_reduce_f64(double *dst, const double *part, int slots, long long n, int accumulate, hipStream_t s) {
    if (n <= 0 || slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(det_reduce_kernel<double>, dim3(ogc_divup(n, 256)), dim3(256), 0, s, dst, part, slots, n, accumulate);
    return hipGetLastError();
}

// Lists of the positions pointing at every destination, positions ascending: start (b, n + 1) and pos (b, T) in the stream's
// deterministic scratch, `extra` further bytes behind them for the caller (returned through *extra_ptr).
int ogc_det_lists(const char *name, int b, int n, long long T, const int *idx, const int **start_out, const int **pos_out,
                  size_t extra, void **extra_ptr, hipStream_t s) {
    OGC_REQUIRE(b >= 0 && b <= 65535 && n >= 1 && T >= 0 && T < (1ll << 31), "%s (deterministic): bad dimensions", name);
    const size_t n_start = ((size_t)b * (n + 1) + 3) / 4 * 4, n_cur = ((size_t)b * n + 3) / 4 * 4, n_pos = ((size_t)b * T + 3) / 4 * 4;
    char *base = static_cast<char *>(ogc_det_scratch(s, (n_start + n_cur + n_pos) * sizeof(int) + extra + 256));
    if (!base) {
        ogc_set_error("%s (deterministic): no scratch memory", name);
        return OGC_ERR_LAUNCH;
    }
    int *start = reinterpret_cast<int *>(base), *cursor = start + n_start, *pos = cursor + n_cur;
    if (extra_ptr) *extra_ptr = reinterpret_cast<char *>(pos + n_pos);
    *start_out = start;
    *pos_out = pos;
    // (a plain fill, never the step's zero arena: the scratch is not part of it)
    hipLaunchKernelGGL(ogc_zero_kernel, dim3(ogc_divup(n_start, 256) > 2048 ? 2048 : ogc_divup(n_start, 256)), dim3(256), 0, s,
                       reinterpret_cast<uint32_t *>(start), n_start);
    if (T > 0 && b > 0) {
        const dim3 grid(ogc_divup(T, 256), b);
        hipLaunchKernelGGL(det_count_kernel, grid, dim3(256), 0, s, T, n, idx, start);
        hipLaunchKernelGGL(det_scan_kernel, dim3(b), dim3(1024), 0, s, n, start, cursor);
        hipLaunchKernelGGL(det_fill_kernel, grid, dim3(256), 0, s, T, n, idx, cursor, pos);
        hipLaunchKernelGGL(det_sort_kernel, dim3(ogc_divup(n, 256), b), dim3(256), 0, s, n, T, start, pos);
    }
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}

// out (b, c, n) (+)= scatter-add of grad_out over idx (b, T), every sum in ascending position order (see the kernel)
int ogc_det_scatter_add(const char *name, int b, int c, int n, long long T, const int *idx, const float *grad_out,
                        long long go_bstride, const float *weight, int interp, float *out, int accumulate, hipStream_t s) {
    if (b == 0 || c == 0 || n == 0) return OGC_OK;
    OGC_REQUIRE(c <= 65535, "%s (deterministic): more than 65535 channels", name);
    const int *start = nullptr, *pos = nullptr;
    const int rc = ogc_det_lists(name, b, n, T, idx, &start, &pos, 0, nullptr, s);
    if (rc != OGC_OK) return rc;
    const dim3 grid(ogc_divup(n, 256), c, b);
    if (interp)
        hipLaunchKernelGGL(det_gather_sum_kernel<true>, grid, dim3(256), 0, s, c, n, T, go_bstride, grad_out, weight, start, pos, out,
                           accumulate);
    else
        hipLaunchKernelGGL(det_gather_sum_kernel<false>, grid, dim3(256), 0, s, c, n, T, go_bstride, grad_out, weight, start, pos,
                           out, accumulate);
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}
