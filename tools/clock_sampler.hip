// Development probe: the shader clock while something else keeps the GPU busy.  One wavefront spins for a fixed number
// of dependent VALU instructions and records s_memtime (shader cycles, MI355X_MICROARCH.md) and wall_clock64()
// (constant 100 MHz) around the loop: cycles / wall time = the clock it ran at.
#include <hip/hip_runtime.h>
extern "C" __global__ void clock_sample_kernel(long long *out, int iters) {
    float a = threadIdx.x * 1e-3f;
    const long long w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) a = __fmul_rn(a, 1.0001f);
    const long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)a; }
}
extern "C" int clock_sample(long long *out, int iters, void *stream) {
    hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
