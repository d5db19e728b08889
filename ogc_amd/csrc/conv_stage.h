// Staging of one 64-row tile of the A operand (the layer's weights) in LDS for the MFMA convolution kernels.
//   a_lds[mi * LD + k] = A[m0 + mi][k],  k < 4 Kq,   A[m][k] = TRANSPOSE ? w[k * M + m] : w[m * K + k],
// zeros beyond M rows / K columns.  LD = 4 * odd (ogc_a_ld): lane (j = l & 15, kk = l >> 4) reads its MFMA operand
// A[m0 + 16 a + j][4 q + kk] from bank (j LD + kk) mod 64 — sixteen different multiples of 4 plus kk: conflict-free.
// The obvious loop (one element per thread and round: load, wait, ds_write) costs one L2 round trip PER ROUND — K / 4 of
// them in a row, ~20 us at K = 128, as long as the MFMA work of the tile it feeds.  Here every thread has 8 independent,
// unconditional loads in flight per round (indices clamped into the matrix, padding zeroed by a multiplication), and
// consecutive lanes read consecutive addresses in both orientations (along k for W, along m for W^T).
#pragma once
#include "ogc_common.h"

__host__ __device__ __forceinline__ int ogc_a_ld(int Kq) { return 4 * (Kq | 1); }

template <bool TRANSPOSE, int WAVES>
__device__ __forceinline__ void ogc_stage_weight_tile(float *__restrict__ a_lds, const float *__restrict__ w, int m0, int M,
                                                      int K, int Kq) {
    // (opaque copies: otherwise the address arithmetic below is hoisted out of the caller's loop over tiles and its ~20
    // registers stay live through the MFMA section, which is where the kernels' register count — their occupancy — is set)
    asm volatile("" : "+s"(m0), "+s"(K), "+s"(M), "+s"(Kq));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K4 = Kq * 4, LD = ogc_a_ld(Kq);
    constexpr int R = 64 / WAVES; // rows (W) or columns (W^T) of a 64-slab per wave
    static_assert(R <= 16 && 64 % WAVES == 0, "slab split");
    constexpr int U = R < 8 ? R : 8; // loads in flight per thread
    const char *wb = reinterpret_cast<const char *>(w); // uniform base + 32-bit byte offset per lane: one VGPR per address
    for (int k0 = 0; k0 < K4; k0 += 64) {
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u;
                const int mi = TRANSPOSE ? lane : wave + WAVES * r;
                const int k = k0 + (TRANSPOSE ? wave + WAVES * r : lane);
                const int mc = min(m0 + mi, M - 1), kc = min(k, K - 1);
                const unsigned off = (unsigned)(TRANSPOSE ? kc * M + mc : mc * K + kc) * 4u;
                v[u] = *reinterpret_cast<const float *>(wb + off);
            }
            // nothing may be scheduled across this line: left alone, the compiler (which schedules for register pressure)
            // issues one load, waits for it, stores it, issues the next — the round trips this function exists to overlap
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u;
                const int mi = TRANSPOSE ? lane : wave + WAVES * r;
                const int k = k0 + (TRANSPOSE ? wave + WAVES * r : lane);
                const int m = m0 + mi;
                if (k < K4) a_lds[mi * LD + k] = v[u] * ((m < M && k < K) ? 1.f : 0.f);
            }
        }
    }
}

// The same tile as PACKED bf16 MFMA operands (bf16-operand kernels):  a_bf[(g * 64 + mi) * 4 + kr] = the four bf16 values
// A[m0 + mi][4 (4 g + i) + kr], i = 0..3 — k-slot i of lane group kr of the 16x16x16 MFMA that covers row quads 4 g .. 4 g + 3
// (conv1x1_gemm_kernel "BF").  GQ: compile-time bound on ceil(Kq / 4).  ALL loads of a thread — four per group, GQ groups — are
// issued before the first is used (unconditional, clamped, padding zeroed by a multiplication, as above): the loop this replaces
// (one group per round: load, wait, pack, ds_write) cost one L2 round trip per group and tile — nine at K = 128, four tiles at 256
// rows.  (Used by dgrad_adjoint_kernel: 0.47 -> 0.45 ms at 64 x 64 channels.  In conv1x1_gemm_kernel it measured slower.)
// Consecutive lanes read consecutive addresses in both orientations.
template <bool TRANSPOSE, int WAVES, int GQ>
__device__ __forceinline__ void ogc_stage_weight_tile_bf16(void *__restrict__ a_bf_, const float *__restrict__ w, int m0, int M,
                                                           int K, int Kq) {
    typedef short v4s_ __attribute__((ext_vector_type(4)));
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    typedef __bf16 v2bf_ __attribute__((ext_vector_type(2)));
    static_assert(WAVES == 4, "one group of 256 (row, k-slot group) pairs per round of the workgroup");
    asm volatile("" : "+s"(m0), "+s"(K), "+s"(M), "+s"(Kq));
    v4s_ *a_bf = static_cast<v4s_ *>(a_bf_);
    const int t = threadIdx.x;
    const int mi = TRANSPOSE ? (t & 63) : ((t >> 2) & 63), kr = TRANSPOSE ? (t >> 6) : (t & 3);
    const int m = m0 + mi, mc = min(m, M - 1);
    const int Gq = (Kq + 3) >> 2;
    float v[GQ][4];
#pragma unroll
    for (int g = 0; g < GQ; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * (4 * g + i) + kr, kc = min(k, K - 1);
            v[g][i] = TRANSPOSE ? w[(size_t)kc * M + mc] : w[(size_t)mc * K + kc];
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        if (g < Gq) {
            float z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = v[g][i] * ((m < M && 4 * (4 * g + i) + kr < K) ? 1.f : 0.f);
            union { v2bf_ h[2]; v4s_ s; } u;
            u.h[0] = __builtin_convertvector((v2f_){z[0], z[1]}, v2bf_);
            u.h[1] = __builtin_convertvector((v2f_){z[2], z[3]}, v2bf_);
            a_bf[(g * 64 + mi) * 4 + kr] = u.s;
        }
    }
}
