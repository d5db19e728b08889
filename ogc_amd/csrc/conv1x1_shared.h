// conv1x1_shared.h — what the two translation units of the 1x1 convolution kernels share (conv1x1.hip: forward / input-gradient
// GEMMs; conv1x1_wgrad.hip: weight gradients).  Split in round 5 so that the two halves compile side by side.
#pragma once
#include <stdlib.h>

#include "ogc_common.h"
#include "conv_stage.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));  // four bf16 MFMA operand values

constexpr int WG_WAVES = 4;

// Operand precision of the convolution kernels (ogc_set_matmul_precision): 0 = fp32 operands on
// v_mfma_f32_16x16x4_f32 (default; results are exact fp32 FMA chains), 1 = operands rounded to bf16 (nearest even) on
// v_mfma_f32_16x16x16_bf16, fp32 accumulation, tensors in memory stay fp32 — what autocast(bfloat16) gives the
// reference's Conv2d layers (BASELINE config "OGC-DR ..., bf16").  Layers of 128 channels and more sit on the fp32
// MFMA roof (157 TFLOP/s); with bf16 operands (2.5 PFLOP/s) they fall back onto the HBM roof.  (Defined in conv1x1.hip.)
extern int ogc_g_matmul_bf16;
#define g_matmul_bf16 ogc_g_matmul_bf16

__device__ __forceinline__ v4s ogc_pack_bf16(float a, float b, float c, float d) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    union { v2bf h[2]; v4s s; } u;
    u.h[0] = __builtin_convertvector((v2f){a, b}, v2bf); // one v_cvt_pk_bf16_f32 per pair
    u.h[1] = __builtin_convertvector((v2f){c, d}, v2bf);
    return u.s;
}
