// Development probe (not part of the product): VALU issue behaviour on gfx950.
// For ILP in {1,2,4,8} independent fp32 chains per lane and W waves per SIMD, report cycles per VALU instruction
// per wave and the per-SIMD instruction rate.  s_memtime ticks are shader cycles (MI355X_MICROARCH.md).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP>
__global__ void spin(long long *out, int iters) {
    float a[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) a[u] = threadIdx.x * 1e-3f + u;
    const float b = 1.0001f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int u = 0; u < ILP; ++u) a[u] = __fmul_rn(a[u], b);
#pragma unroll
            for (int u = 0; u < ILP; ++u) a[u] = __fadd_rn(a[u], b);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int u = 0; u < ILP; ++u) s += a[u];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}
template <int ILP>
void run(long long *d, int threads) {
    long long h[2];
    const int iters = 100000;
    hipLaunchKernelGGL(spin<ILP>, dim3(1), dim3(threads), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double instr = 8.0 * ILP * iters;              // VALU instructions per wave
    const int waves_per_simd = threads / 256 > 0 ? threads / 256 : 1;
    printf("ILP=%d waves/SIMD=%d : %.2f cycles per VALU instr per wave, %.2f cycles per instr per SIMD\n", ILP,
           threads >= 256 ? waves_per_simd : 1, h[0] / instr, h[0] / (instr * (threads >= 256 ? waves_per_simd : 1)));
}
int main() {
    long long *d;
    (void)hipMalloc(&d, 16);
    for (int threads : {64, 256, 512, 1024}) {
        run<1>(d, threads); run<2>(d, threads); run<4>(d, threads); run<8>(d, threads);
    }
    return 0;
}
