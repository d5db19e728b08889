// Development probe (not part of the product): issue rate of v_mfma_f32_16x16x4_f32 on gfx950 in the shapes the
// convolution kernels use it — 16 accumulators (4 row blocks x 4 column blocks), B operands in registers, A operands
// (a) in registers, (b) one ds_read_b32 per row block and k-step (as conv1x1_gemm_kernel), (c) one ds_read_b128 per
// row block and FOUR k-steps.  Reports cycles per MFMA per wave (s_memtime) for 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(long long *out, int iters, const float *seed) {
    __shared__ float lds[64 * 64 * 4];
    for (int t = threadIdx.x; t < 64 * 64 * 4; t += blockDim.x) lds[t] = seed[t & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15, kk = lane >> 4;
    v4f acc[4][4];
    float4 x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = make_float4(seed[lane + q], seed[lane + q + 64], seed[lane + q + 128], seed[lane + q + 192]);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){0.f, 0.f, 0.f, 0.f};
    float areg[4] = {seed[lane], seed[lane + 1], seed[lane + 2], seed[lane + 3]};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float av[4];
            if (MODE == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = areg[a];
            } else if (MODE == 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = lds[((q + it % 8) * 64 + a * 16 + j) * 4 + kk];
            }
            float4 a4[4];
            if (MODE == 2 && (q & 3) == 0) {
#pragma unroll
                for (int a = 0; a < 4; ++a) a4[a] = *reinterpret_cast<const float4 *>(&lds[(((q >> 2) + it % 8) * 64 + a * 16 + j) * 16 + kk * 4]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float av_ = MODE == 2 ? ((q & 3) == 0 ? a4[a].x : (q & 3) == 1 ? a4[a].y : (q & 3) == 2 ? a4[a].z : a4[a].w) : av[a];
                acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, x[q].x, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, x[q].y, acc[a][1], 0, 0, 0);
                acc[a][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, x[q].z, acc[a][2], 0, 0, 0);
                acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, x[q].w, acc[a][3], 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) s += acc[a][c][0] + acc[a][c][3];
    if (lane == 0) { out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = (long long)s; }
}

template <int MODE>
void run(long long *d, const float *seed, int threads, const char *what) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, seed);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, seed);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double mfma = 128.0 * iters;
    const double tf = 2048.0 * mfma * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD=%d : %.1f cycles per MFMA per wave, %.1f TFLOP/s over %d CUs\n", what, threads / 256, h[0] / mfma, tf, blocks);
}

int main() {
    long long *d;
    float *seed;
    (void)hipMalloc(&d, 16 * 256 * 8);
    (void)hipMalloc(&seed, 4096 * 4);
    for (int pass = 0; pass < 2; ++pass) { // all-zero operands, then random ones: the clock the chip sustains depends on the data
    if (pass == 0) (void)hipMemset(seed, 0, 4096 * 4);
    else {
        float h[4096];
        unsigned r = 12345;
        for (int i = 0; i < 4096; ++i) { r = r * 1664525u + 1013904223u; h[i] = (float)(r >> 8) / 16777216.f - 0.5f; }
        (void)hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    }
    printf(pass ? "random operands\n" : "zero operands\n");
    for (int threads : {256, 512}) {
        run<0>(d, seed, threads, "A operand in registers");
        run<1>(d, seed, threads, "A: ds_read_b32 per k-step");
        run<2>(d, seed, threads, "A: ds_read_b128 per 4 k-steps");
    }
    }
    return 0;
}
