// Development probe: latency (cycles, s_memtime) of the wave-wide reductions an FPS round chains together, one wavefront per
// SIMD, every result feeding the next iteration's input so that nothing overlaps.
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ float wave_max_a(float v) { // fps_wave_max: 4 DPP steps, 4 readlanes, scalar max
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
    const int r0 = __builtin_amdgcn_readlane(x, 0), r1 = __builtin_amdgcn_readlane(x, 16);
    const int r2 = __builtin_amdgcn_readlane(x, 32), r3 = __builtin_amdgcn_readlane(x, 48);
    return __int_as_float(max(max(r0, r1), max(r2, r3)));
}
__device__ __forceinline__ float wave_max_b(float v) { // 4 DPP steps, 2 row broadcasts, ONE readlane
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false)); // row_bcast:15 -> rows 1, 3
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xC, 0xF, false)); // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
__device__ __forceinline__ float wave_max_c(float v) { // everything in vector registers: result in every lane, no SGPR
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_ds_swizzle(x, 0x401F)); // swap rows 0<->1, 2<->3 (xor 16 within 32)
    x = max(x, __shfl_xor(x, 32, 64));                  // ds_bpermute
    return __int_as_float(x);
}
__device__ __forceinline__ unsigned holder_rank(float v, float wm, unsigned rank) { // ballot, count, readlane of the holder
    const unsigned long long h = __builtin_amdgcn_ballot_w64(v == wm);
    if (__popcll(h) == 1) return (unsigned)__builtin_amdgcn_readlane((int)rank, (int)__builtin_ctzll(h));
    return 0u;
}

template <int MODE>
__global__ void probe(long long *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float v = seed + lane * 0.37f + (lane % 7);
    unsigned rk = lane * 2654435761u;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float w;
        if (MODE == 0) w = wave_max_a(v);
        else if (MODE == 1) w = wave_max_b(v);
        else if (MODE == 2) w = wave_max_c(v);
        else { w = wave_max_a(v); rk += holder_rank(v, w, rk); }
        v = v * 0.999f + (w - v) * 1e-6f + (rk & 1); // depends on the result
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)v; }
}

int main() {
    long long *d, h[2];
    (void)hipMalloc(&d, 64);
    const int iters = 20000;
    const char *names[] = {"wave max: 4 DPP + 4 readlane + s_max", "wave max: 6 DPP + 1 readlane", "wave max: 4 DPP + swizzle + bpermute (VGPR only)", "wave max (first form) + ballot / count / readlane of the holder"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d, iters, 1.5f);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d, iters, 1.5f);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d, iters, 1.5f);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, d, iters, 1.5f);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-64s %6.1f cycles per iteration (incl. ~3 dependent VALU ops of the loop)\n", names[mode], (double)h[0] / iters);
    }
    return 0;
}
