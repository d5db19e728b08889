// act_io.h — activations of the shared MLPs stored as fp32 OR bf16 (round 5: `matmul_precision: bf16` with 16-bit activations).
//
// Every kernel that streams an activation (a convolution's raw output y, or the gradient with respect to one) moves FOUR
// consecutive positions of one channel row per lane and step: a float4 for fp32 tensors.  The kernels are templates in the
// element type of those tensors; the two overloads below are the only places that know it.  bf16 = the upper 16 bits of an
// fp32 (round to nearest even on the way out: one v_cvt_pk_bf16_f32 per pair on gfx950), arithmetic stays fp32 everywhere:
// statistics, accumulators, coefficients, and everything that is not the size of an activation.  The `_h` entry points of
// include/ogc_ops.h are the 16-bit instantiations.
#pragma once
#include "ogc_common.h"

typedef unsigned short ogc_bf16; // storage only (== ogc_bf16_t of include/ogc_ops.h)

__device__ __forceinline__ float4 ogc_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ogc_ld4(const ogc_bf16 *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ float ogc_ld1(const float *p) { return *p; }
__device__ __forceinline__ float ogc_ld1(const ogc_bf16 *p) { return __uint_as_float((unsigned)*p << 16); }

__device__ __forceinline__ uint2 ogc_pack4_bf16(const float4 &v) {
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
    union { v2bf_ h[2]; uint2 u; } r;
    r.h[0] = __builtin_convertvector((v2f_){v.x, v.y}, v2bf_);
    r.h[1] = __builtin_convertvector((v2f_){v.z, v.w}, v2bf_);
    return r.u;
}
__device__ __forceinline__ void ogc_st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void ogc_st4(ogc_bf16 *p, const float4 &v) { *reinterpret_cast<uint2 *>(p) = ogc_pack4_bf16(v); }

// four fp32 -> four bf16 MFMA operand values as two explicit v_cvt_pk_bf16_f32 (the vector conversion of ogc_pack_bf16 can leave
// the arrays it reads from in scratch memory when it is used inside a lambda: gemm_chunk.hip, round 5).  The wait states an MFMA
// needs before it reads a VGPR a VALU instruction has just written are INSIDE the asm: the compiler's hazard recogniser does not
// look into an asm block, and without them the matrix pipe reads the operand before the conversion has landed (NaN rows).
typedef short ogc_v4s_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ogc_v4s_ ogc_pack_bf16_rr(float a, float b, float c, float d) {
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    unsigned lo, hi;
    asm("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\ts_nop 1"
        : "=&v"(lo), "=&v"(hi)
        : "v"(a), "v"(b), "v"(c), "v"(d));
    return __builtin_bit_cast(ogc_v4s_, (v2u_){lo, hi});
}

// gfx950's double-depth bf16 MFMA: D = A (16 x 32) . B (32 x 16) + C — lane (i = l & 15, k = l >> 4) supplies row / column i and
// the k-slots 8 k .. 8 k + 7.  Here as "two 16x16x16 operand quads side by side" (slots 8 k .. 8 k + 3 from *0, the rest from *1):
// every product of this library pairs the SAME k-slot of both operands, so how slots map to positions / channels is free as long
// as both operands agree — which they do when both halves are built the same way.
typedef float ogc_v4f_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ogc_v4f_ ogc_mfma_bf16_k32(ogc_v4s_ a0, ogc_v4s_ a1, ogc_v4s_ b0, ogc_v4s_ b1, ogc_v4f_ c) {
    typedef short v8s_ __attribute__((ext_vector_type(8)));
    typedef __bf16 v8bf_ __attribute__((ext_vector_type(8)));
    const v8s_ a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    const v8s_ b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_, a), __builtin_bit_cast(v8bf_, b), c, 0, 0, 0);
}

// the value as it will read back from a tensor of element type T (statistics and extremes of an output are taken over what is
// STORED, so that the norm that follows sees exactly the distribution its mean / rstd describe)
template <typename T>
__device__ __forceinline__ float ogc_as_stored(float v) {
    if constexpr (sizeof(T) == 2) {
        typedef float v2f_ __attribute__((ext_vector_type(2)));
        typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
        union { v2bf_ h; unsigned u; } r;
        r.h = __builtin_convertvector((v2f_){v, 0.f}, v2bf_);
        return __uint_as_float(r.u << 16);
    } else {
        return v;
    }
}

// bytes a lane's four positions occupy: the alignment the entry points ask of an activation tensor
template <typename T>
constexpr uintptr_t ogc_act_mask() { return 4 * sizeof(T) - 1; }
