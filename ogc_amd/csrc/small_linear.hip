// small_linear.hip — nn.Linear on a few hundred rows, forward and backward, one launch each.
//
// Reference: utils/transformer_util.py:5-62 (the decoder layer's projections and feed-forward network act on the
// K = 10 slot embeddings per sample: 160 rows of 128 features at the C4 shapes).  The vendor library runs each of these
// 5-MFLOP products as one or two workgroups of a 128 x 160 tile kernel, ~15 us apiece, and the backward pass of one
// layer is two of them plus a reduction for the bias: on the slot branch, a serial chain that the set-abstraction
// backward has to wait for, that is ~35 us per layer where the arithmetic needs three.  Here
//     forward : y = x W^T + b                                     (r x o)
//     backward: dx = dy W (r x i),  dW = dy^T x (o x i),  db = sum_r dy        — all three in ONE launch
// as 32 x 32 output tiles (256 threads, 2 x 2 outputs each, operands staged through LDS in 64-deep slices; plain fp32
// FMAs — there is no matrix-core shape worth filling at this size).  The workgroups of a backward launch are split
// between the dx tiles and the dW tiles; the dW tiles of the first column block also produce db.
#include "ogc_common.h"

namespace {

constexpr int SL_T = 32;   // tile edge
constexpr int SL_K = 64;   // depth of a staged slice
constexpr int SL_E = SL_T * SL_K / 256;   // elements of a slice per thread and operand

// slice [k0, k0 + SL_K) of the operands of tile (m0, n0) into registers (zeros outside the matrices); the
// faster-varying thread index follows the operand's unit stride
__device__ __forceinline__ void sl_fetch(int M, int N, int K, const float *__restrict__ a, long am, long ak,
                                         const float *__restrict__ b, long bk, long bn, int m0, int n0, int k0,
                                         float (&ra)[SL_E], float (&rb)[SL_E]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < SL_E; ++j) {
        const int e = t + j * 256;
        int mm, kk;
        if (ak == 1) { kk = e % SL_K; mm = e / SL_K; } else { mm = e % SL_T; kk = e / SL_T; }
        const int gm = m0 + mm, gk = k0 + kk;
        ra[j] = (gm < M && gk < K) ? a[gm * am + gk * ak] : 0.f;
        int nn, kb;
        if (bn == 1) { nn = e % SL_T; kb = e / SL_T; } else { kb = e % SL_K; nn = e / SL_K; }
        const int gn = n0 + nn, gkb = k0 + kb;
        rb[j] = (gn < N && gkb < K) ? b[gkb * bk + gn * bn] : 0.f;
    }
}

// C[m, n] (+ bias[n]) = sum_k A(m, k) B(k, n) for the 32 x 32 tile (tm, tn); A(m, k) = a[m * am + k * ak],
// B(k, n) = b[k * bk + n * bn].  colsum != null: also colsum[m] = sum_k A(m, k) (written by the tn == 0 tiles).
// The next slice is fetched into registers while the current one is multiplied out of LDS: one memory latency per
// tile instead of one per slice (the operands are a few hundred KB, the loop is pure latency otherwise).
__device__ __forceinline__ void sl_tile(int M, int N, int K, const float *__restrict__ a, long am, long ak,
                                        const float *__restrict__ b, long bk, long bn, const float *__restrict__ bias,
                                        float *__restrict__ c, int ldc, float *__restrict__ colsum, int tm, int tn) {
    __shared__ float As[SL_K][SL_T + 1];   // [k][m]
    __shared__ float Bs[SL_K][SL_T + 1];   // [k][n]
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = tm * SL_T, n0 = tn * SL_T;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float rs[2] = {0.f, 0.f};
    float ra[SL_E], rb[SL_E];
    sl_fetch(M, N, K, a, am, ak, b, bk, bn, m0, n0, 0, ra, rb);
    for (int k0 = 0; k0 < K; k0 += SL_K) {
#pragma unroll
        for (int j = 0; j < SL_E; ++j) {
            const int e = t + j * 256;
            if (ak == 1) As[e % SL_K][e / SL_K] = ra[j]; else As[e / SL_T][e % SL_T] = ra[j];
            if (bn == 1) Bs[e / SL_T][e % SL_T] = rb[j]; else Bs[e % SL_K][e / SL_K] = rb[j];
        }
        __syncthreads();
        if (k0 + SL_K < K) sl_fetch(M, N, K, a, am, ak, b, bk, bn, m0, n0, k0 + SL_K, ra, rb);
#pragma unroll 16
        for (int kk = 0; kk < SL_K; ++kk) {
            const float a0 = As[kk][ty * 2], a1 = As[kk][ty * 2 + 1];
            const float b0 = Bs[kk][tx * 2], b1 = Bs[kk][tx * 2 + 1];
            acc[0][0] = fmaf(a0, b0, acc[0][0]);
            acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]);
            acc[1][1] = fmaf(a1, b1, acc[1][1]);
            rs[0] += a0;
            rs[1] += a1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gm = m0 + ty * 2 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + tx * 2 + j;
            if (gn < N) c[(size_t)gm * ldc + gn] = acc[i][j] + (bias ? bias[gn] : 0.f);
        }
        if (colsum && tn == 0 && tx == 0) colsum[gm] = rs[i];
    }
}

__global__ __launch_bounds__(256) void small_linear_fwd_kernel(int r, int ni, int no, const float *__restrict__ x,
                                                              const float *__restrict__ w,
                                                              const float *__restrict__ bias, float *__restrict__ y) {
    const int tiles_n = (no + SL_T - 1) / SL_T;
    // y (r x o) = x (r x i) . W^T: A = x, B(k = i, n = o) = w[o * ni + i]
    sl_tile(r, no, ni, x, ni, 1, w, 1, ni, bias, y, no, nullptr, blockIdx.x / tiles_n, blockIdx.x % tiles_n);
}

__global__ __launch_bounds__(256) void small_linear_bwd_kernel(int r, int ni, int no, int dx_tiles,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ w,
                                                              const float *__restrict__ gy, float *__restrict__ gx,
                                                              float *__restrict__ gw, float *__restrict__ gb) {
    const int tiles_i = (ni + SL_T - 1) / SL_T;
    int blk = blockIdx.x;
    if (blk < dx_tiles) {
        // dx (r x i) = dy (r x o) . W (o x i): A = dy, B(k = o, n = i) = w[o * ni + i]
        sl_tile(r, ni, no, gy, no, 1, w, ni, 1, nullptr, gx, ni, nullptr, blk / tiles_i, blk % tiles_i);
        return;
    }
    blk -= dx_tiles;
    // dW (o x i) = dy^T (o x r) . x (r x i): A(m = o, k = r) = gy[r * no + o], B(k = r, n = i) = x[r * ni + i];
    // db[o] = sum_r dy[r, o] is the row sum of A.  Without dW (frozen weight) the tiles have no columns to write.
    sl_tile(no, gw ? ni : 0, r, gy, 1, no, x, ni, 1, nullptr, gw, ni, gb, blk / tiles_i, blk % tiles_i);
}

int sl_check(const char *who, int r, int ni, int no) {
    if (r < 0 || ni <= 0 || no <= 0) {
        ogc_set_error("%s: bad sizes rows=%d in=%d out=%d", who, r, ni, no);
        return OGC_ERR_INVALID_ARG;
    }
    return OGC_OK;
}

} // namespace

extern "C" int ogc_small_linear_fwd(int rows, int n_in, int n_out, const float *x, const float *weight,
                                    const float *bias, float *y, ogc_stream_t stream) {
    const int rc = sl_check("ogc_small_linear_fwd", rows, n_in, n_out);
    if (rc != OGC_OK) return rc;
    if (rows == 0) return OGC_OK;
    OGC_REQUIRE(x && weight && y, "ogc_small_linear_fwd: null pointer");
    const int tiles = ogc_divup(rows, SL_T) * ogc_divup(n_out, SL_T);
    hipLaunchKernelGGL(small_linear_fwd_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, rows, n_in, n_out, x,
                       weight, bias, y);
    OGC_CHECK_LAUNCH("ogc_small_linear_fwd");
    return OGC_OK;
}

extern "C" int ogc_small_linear_bwd(int rows, int n_in, int n_out, const float *x, const float *weight,
                                    const float *grad_y, float *grad_x, float *grad_weight, float *grad_bias,
                                    ogc_stream_t stream) {
    const int rc = sl_check("ogc_small_linear_bwd", rows, n_in, n_out);
    if (rc != OGC_OK) return rc;
    if (rows == 0) {
        // no rows: the parameter gradients are zero
        hipError_t e = hipSuccess;
        if (grad_weight) e = ogc_zero_async(grad_weight, sizeof(float) * (size_t)n_in * n_out, (hipStream_t)stream);
        if (grad_bias && e == hipSuccess) e = ogc_zero_async(grad_bias, sizeof(float) * (size_t)n_out, (hipStream_t)stream);
        OGC_REQUIRE(e == hipSuccess, "ogc_small_linear_bwd: memset failed: %s", hipGetErrorString(e));
        return OGC_OK;
    }
    OGC_REQUIRE(x && weight && grad_y, "ogc_small_linear_bwd: null pointer");
    OGC_REQUIRE(grad_x || grad_weight || grad_bias, "ogc_small_linear_bwd: nothing to compute");
    const int tiles_i = ogc_divup(n_in, SL_T);
    const int dx_tiles = grad_x ? ogc_divup(rows, SL_T) * tiles_i : 0;
    // without dW (frozen weight) the same tiles run with zero columns and only produce db
    const int dw_tiles = (grad_weight || grad_bias) ? ogc_divup(n_out, SL_T) * tiles_i : 0;
    hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(dx_tiles + dw_tiles), dim3(256), 0, (hipStream_t)stream, rows, n_in,
                       n_out, dx_tiles, x, weight, grad_y, grad_x, grad_weight, grad_bias);
    OGC_CHECK_LAUNCH("ogc_small_linear_bwd");
    return OGC_OK;
}
