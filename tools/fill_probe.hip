// How long does it take just to WRITE the ball query's output (16 x 8192 rows of 64 ints = 33.5 MB) from a launch of the
// same shape as ball_query_cells_kernel — the floor under that kernel's duration — and with other launch shapes.
//   hipcc --offload-arch=gfx950 -O3 tools/fill_probe.hip -o tools/_bin/fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NT>
__global__ void fill_rows(int *out, int rows_per_wave, int lds_dummy) {
    extern __shared__ int sm[];
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lds_dummy < 0) sm[lane] = lane;
    const int sub = lane & 3, g = lane >> 2;
    int *o = out + ((size_t)wave * rows_per_wave + g) * 64;
    const int4 v = make_int4(wave, lane, sub, g);
    for (int j = sub * 4; j < 64; j += 16) {
        if (NT) { __builtin_nontemporal_store(v.x, o + j); __builtin_nontemporal_store(v.y, o + j + 1); __builtin_nontemporal_store(v.z, o + j + 2); __builtin_nontemporal_store(v.w, o + j + 3); }
        else *reinterpret_cast<int4 *>(o + j) = v;
    }
}
__global__ void fill_flat(int4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_int4(1, 2, 3, 4);
}
__global__ void empty_kernel(int *out) { if (out == nullptr) out[0] = 1; }
int main() {
    const size_t rows = 16 * 8192;
    int *out; hipMalloc(&out, rows * 64 * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < 50; ++i) launch();
        hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-60s %.2f us per launch\n", name, ms / 50 * 1e3);
    };
    timeit("empty kernel, 8192 x 64 threads", [&] { hipLaunchKernelGGL(empty_kernel, dim3(8192), dim3(64), 0, s, out); });
    timeit("empty kernel, 8192 x 64 threads, 5 KiB LDS", [&] { hipLaunchKernelGGL(empty_kernel, dim3(8192), dim3(64), 5120, s, out); });
    timeit("empty kernel, 256 x 64 threads", [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(64), 0, s, out); });
    timeit("rows, 8192 blocks x 64 threads, 5 KiB LDS", [&] { hipLaunchKernelGGL(fill_rows<0>, dim3(8192), dim3(64), 5120, s, out, 16, 0); });
    timeit("rows, 8192 blocks x 64 threads, 5 KiB LDS, nt stores", [&] { hipLaunchKernelGGL(fill_rows<1>, dim3(8192), dim3(64), 5120, s, out, 16, 0); });
    timeit("rows, 2048 blocks x 256 threads", [&] { hipLaunchKernelGGL(fill_rows<0>, dim3(2048), dim3(256), 0, s, out, 16, 0); });
    timeit("rows, 2048 blocks x 256 threads, nt", [&] { hipLaunchKernelGGL(fill_rows<1>, dim3(2048), dim3(256), 0, s, out, 16, 0); });
    timeit("flat 33.5 MB, 2048 x 256 grid-stride", [&] { hipLaunchKernelGGL(fill_flat, dim3(2048), dim3(256), 0, s, (int4 *)out, rows * 16); });
    timeit("flat 33.5 MB, 8192 x 256", [&] { hipLaunchKernelGGL(fill_flat, dim3(8192), dim3(256), 0, s, (int4 *)out, rows * 16); });
    return 0;
}
