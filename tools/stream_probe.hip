// Development probe (not part of the product): where does the time of the streaming 128 -> 128 forward GEMM go?
// The kernel of ogc_amd/csrc/conv1x1.hip (EXACT, no prologue) with its three phases switchable:
//   LOAD  the next tile's 32 float4 loads per lane,  MFMA  the 1024 MFMAs of a tile,  STORE  its 32 float4 stores.
// B = 16, K = M = 128, hw = 32768 (the shape of tools/bench_ops.py), 256 workgroups of 4 waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../ogc_amd/csrc/conv_stage.h"
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int KQ = 32;

template <bool LOAD, bool MFMA, bool STORE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream(int M, int K, int hw, int ntiles, const float *__restrict__ w,
                                                     const float *__restrict__ in, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float a_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, kk = lane >> 4;
    const int Kq = K >> 2, Mt = M >> 6;
    const int tiles_per_img = hw >> 6;
    const int a_ld = ogc_a_ld(Kq);
    for (int mt = 0; mt < Mt; ++mt) ogc_stage_weight_tile<false, WAVES>(a_lds + (size_t)mt * 64 * a_ld, w, mt * 64, M, K, Kq);
    __syncthreads();
    const int nw = gridDim.x * WAVES;
    const unsigned off_main = (unsigned)(kk * hw + 4 * j) * 4u; // BYTES: scalar base + 32-bit lane offset (saddr form)
    auto load_tile = [&](int t, float4(&x)[KQ]) {
        t = min(t, ntiles - 1);
        const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 64;
        const float *inb = in + (size_t)b * K * hw + p0;
#pragma unroll
        for (int q = 0; q < KQ; ++q)
            x[q] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(inb + (size_t)(q * 4) * hw) + off_main);
    };
    auto compute_store = [&](int t, float4(&x)[KQ]) {
        const int b = t / tiles_per_img, p0 = (t - b * tiles_per_img) * 64;
        float *outb = out + (size_t)b * M * hw + p0;
        const unsigned off_out = (unsigned)(kk * 4 * hw + 4 * j) * 4u;
        for (int mt = 0; mt < Mt; ++mt) {
            const float *at = a_lds + (size_t)mt * 64 * a_ld;
            v4f acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = (v4f){x[a].x, x[c].y, 0.f, 0.f};
            if (MFMA) {
                float av[2][4]; // the A operands of step q + 1 are read from LDS before the MFMAs of step q
#pragma unroll
                for (int a = 0; a < 4; ++a) av[0][a] = at[(a * 16 + j) * a_ld + kk];
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    if (q + 1 < KQ) {
#pragma unroll
                        for (int a = 0; a < 4; ++a) av[(q + 1) & 1][a] = at[(a * 16 + j) * a_ld + (q + 1) * 4 + kk];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][a], x[q].x, acc[a][0], 0, 0, 0);
                        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][a], x[q].y, acc[a][1], 0, 0, 0);
                        acc[a][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][a], x[q].z, acc[a][2], 0, 0, 0);
                        acc[a][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][a], x[q].w, acc[a][3], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (STORE || acc[0][0][0] == 12345.678f) { // (the comparison keeps the accumulators alive without stores)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m0 = mt * 64 + a * 16 + r;
                        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(outb + (size_t)m0 * hw) + off_out) =
                            make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
                    }
            }
        }
    };
    float4 xa[KQ], xb[KQ];
    int t = blockIdx.x * WAVES + wave;
    load_tile(t, xa);
    for (; t + nw < ntiles; t += 2 * nw) {
        if (LOAD) load_tile(t + nw, xb);
        compute_store(t, xa);
        if (LOAD) load_tile(t + 2 * nw, xa);
        compute_store(t + nw, LOAD ? xb : xa);
    }
    if (t < ntiles) compute_store(t, xa);
}

template <bool L, bool F, bool S, int WAVES>
void run(const char *name, int wgs, const float *w, const float *in, float *out) {
    const int B = 16, K = 128, M = 128, hw = 32768, ntiles = B * hw / 64;
    const size_t lds = (size_t)(M / 64) * 64 * ogc_a_ld(K / 4) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&stream<L, F, S, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<L, F, S, WAVES>), dim3(wgs), dim3(WAVES * 64), lds, 0, M, K, hw, ntiles, w, in, out);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream<L, F, S, WAVES>), dim3(wgs), dim3(WAVES * 64), lds, 0, M, K, hw, ntiles, w, in, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/WG %d, %3d WGs: %7.1f us\n", name, WAVES, wgs, ms * 100.f);
}

int main() {
    const size_t n = (size_t)16 * 128 * 32768;
    float *in, *out, *w;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&w, 128 * 128 * 4);
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.f - 0.5f;
    for (size_t o = 0; o < n; o += h.size()) hipMemcpy(in + o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), 128 * 128 * 4, hipMemcpyHostToDevice);
    run<true, true, true, 4>("load + mfma + store", 256, w, in, out);
    run<true, true, false, 4>("load + mfma", 256, w, in, out);
    run<false, true, true, 4>("mfma + store", 256, w, in, out);
    run<false, true, false, 4>("mfma only", 256, w, in, out);
    run<true, false, true, 4>("load + store", 256, w, in, out);
    run<true, false, false, 4>("load only", 256, w, in, out);
    run<false, false, true, 4>("store only", 256, w, in, out);
    run<true, true, true, 8>("load + mfma + store", 256, w, in, out);
    run<true, true, true, 8>("load + mfma + store", 128, w, in, out);
    run<false, true, false, 8>("mfma only", 256, w, in, out);
    run<true, false, true, 8>("load + store", 256, w, in, out);
    return 0;
}
