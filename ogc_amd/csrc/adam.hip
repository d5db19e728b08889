// adam.hip — the NaN rule and the Adam update of a training step as two launches.
//
// Reference: train_seg.py:81-83 / train_flow.py:84-86 (no step when any gradient holds a NaN) followed by
// torch.optim.Adam.step() (train_seg.py:320: Adam with L2 weight decay).  On the GPU the host code used torch's fused
// implementation: _foreach_norm + stack + sum + isnan for the flag, _foreach_add_ on the step counts, three
// multi_tensor_apply launches of the update (31 us each for the 105 tensors / 0.6 M values of segnet_kitti: a handful of
// workgroups), _foreach_sub_ to take the count back on a skipped step — eleven launches, 0.3 ms of a C4 step's main queue
// and 0.5 ms of its launch thread.  Here: one pass over the gradients for the flag, one over everything for the update,
// both over a table of 2048-element chunks so that ~350 workgroups share the work whatever the tensors' sizes.
//
// The tensors stay torch's: parameters, exp_avg, exp_avg_sq and the per-parameter step counts of optimizer.state are
// updated in place, so state_dict() / load_state_dict() see what torch's own step would have left (same formulas as
// ATen's fused kernel: hyper-parameters and bias corrections in double, see adam_update_kernel).
#include <math.h>

#include "ogc_common.h"

namespace {

constexpr int ADAM_THREADS = 256;
constexpr int ADAM_CHUNK = 2048;     // elements per workgroup
constexpr int ADAM_MAX_TENSORS = 320; // gradient pointers travel by value in the kernel arguments (2.5 KiB of the 4)

struct AdamGrads {
    const float *g[ADAM_MAX_TENSORS];
};

// table (device, int64): [0][n] parameter, [1][n] exp_avg, [2][n] exp_avg_sq, [3][n] step count pointers, [4][n] numel
// chunks (device, int32): [c][0] tensor, [c][1] first element
__global__ __launch_bounds__(ADAM_THREADS) void adam_nan_kernel(int n_tensors, const long long *__restrict__ table,
                                                                const int *__restrict__ chunks, AdamGrads grads,
                                                                float *__restrict__ step_snapshot, int *__restrict__ flag) {
    const int ti = chunks[2 * blockIdx.x], off = chunks[2 * blockIdx.x + 1];
    const long long numel = table[4 * (long long)n_tensors + ti];
    const float *g = grads.g[ti];
    if (off == 0 && threadIdx.x == 0)
        step_snapshot[ti] = *reinterpret_cast<const float *>(table[3 * (long long)n_tensors + ti]);
    bool bad = false;
    const long long end = min((long long)off + ADAM_CHUNK, numel);
    for (long long e = off + threadIdx.x; e < end; e += ADAM_THREADS) {
        const float v = g[e];
        bad = bad || v != v;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(ADAM_THREADS) void adam_update_kernel(int n_tensors, const long long *__restrict__ table,
                                                                   const int *__restrict__ chunks, AdamGrads grads,
                                                                   const float *__restrict__ step_snapshot,
                                                                   const int *__restrict__ flag, double lr, double beta1,
                                                                   double beta2, double eps, double weight_decay) {
    if (*flag != 0) return; // a NaN somewhere: no update, no step count (train_seg.py:81-83)
    const int ti = chunks[2 * blockIdx.x], off = chunks[2 * blockIdx.x + 1];
    const long long nt = n_tensors;
    float *p = reinterpret_cast<float *>(table[ti]);
    float *m = reinterpret_cast<float *>(table[nt + ti]);
    float *v = reinterpret_cast<float *>(table[2 * nt + ti]);
    const long long numel = table[4 * nt + ti];
    const float *g = grads.g[ti];
    const float step = step_snapshot[ti] + 1.0f;
    if (off == 0 && threadIdx.x == 0) *reinterpret_cast<float *>(table[3 * nt + ti]) = step;
    // ATen's FusedAdamMathFunctor / adam_math (ATen/native/cuda/fused_adam_utils.cuh of torch 2.10): the hyper-parameters are
    // doubles, the bias corrections are computed in double and handed on as fp32, every expression that mixes a double
    // hyper-parameter with fp32 state is evaluated in double and rounded once on assignment
    const float bc1 = (float)(1.0 - pow(beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    const float step_size = (float)(lr / (double)bc1);
    const long long end = min((long long)off + ADAM_CHUNK, numel);
    for (long long e = off + threadIdx.x; e < end; e += ADAM_THREADS) {
        float param = p[e], grad = g[e], ea = m[e], es = v[e];
        if (weight_decay != 0.0) grad = (float)((double)grad + (double)param * weight_decay);
        ea = (float)(beta1 * (double)ea + (1.0 - beta1) * (double)grad);
        es = (float)(beta2 * (double)es + (1.0 - beta2) * (double)grad * (double)grad);
        const float denom = (float)((double)(sqrtf(es) / bc2_sqrt) + eps);
        param -= step_size * ea / denom;
        p[e] = param;
        m[e] = ea;
        v[e] = es;
    }
}

} // namespace

extern "C" int ogc_adam_max_tensors(void) { return ADAM_MAX_TENSORS; }
extern "C" int ogc_adam_chunk(void) { return ADAM_CHUNK; }

// grad_ptrs: HOST array of n_tensors device pointers (the gradients are new tensors every step; everything else is in
// `table`, built once).  flag (device, int32) must be zero on entry and is 1 afterwards when a gradient held a NaN (the step
// is then skipped on the device: nothing else is written).  step_snapshot: n_tensors floats of scratch.
extern "C" int ogc_adam_step(int n_tensors, int n_chunks, const long long *table, const int *chunks,
                             const void *const *grad_ptrs, float *step_snapshot, int *flag, double lr, double beta1,
                             double beta2, double eps, double weight_decay, ogc_stream_t stream) {
    OGC_REQUIRE(n_tensors >= 0 && n_chunks >= 0, "ogc_adam_step: negative size");
    if (n_tensors == 0 || n_chunks == 0) return OGC_OK;
    OGC_REQUIRE(n_tensors <= ADAM_MAX_TENSORS, "ogc_adam_step: more than %d tensors in one call", ADAM_MAX_TENSORS);
    OGC_REQUIRE(table && chunks && grad_ptrs && step_snapshot && flag, "ogc_adam_step: null pointer");
    AdamGrads grads;
    for (int i = 0; i < n_tensors; ++i) {
        OGC_REQUIRE(grad_ptrs[i], "ogc_adam_step: null gradient pointer");
        grads.g[i] = static_cast<const float *>(grad_ptrs[i]);
    }
    for (int i = n_tensors; i < ADAM_MAX_TENSORS; ++i) grads.g[i] = nullptr;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_nan_kernel, dim3(n_chunks), dim3(ADAM_THREADS), 0, s, n_tensors, table, chunks, grads,
                       step_snapshot, flag);
    hipLaunchKernelGGL(adam_update_kernel, dim3(n_chunks), dim3(ADAM_THREADS), 0, s, n_tensors, table, chunks, grads,
                       step_snapshot, flag, lr, beta1, beta2, eps, weight_decay);
    OGC_CHECK_LAUNCH("ogc_adam_step");
    return OGC_OK;
}
