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
