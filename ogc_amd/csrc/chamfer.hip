// chamfer.hip — the two nearest-neighbour distance terms of the Chamfer loss, and their gradient, as one launch each.
//
// Reference (losses/flow_loss_unsup.py:15-35), with p1 = pc1 + flow:
//     _, idx = knn(1, p1, pc2);  nn1 = grouping_operation(pc2^T, idx).squeeze(-1);  dist1 = (p1^T - nn1).norm(p, dim=1)
//     _, idx = knn(1, pc2, p1);  nn2 = grouping_operation(p1^T, idx).squeeze(-1);   dist2 = (pc2^T - nn2).norm(p, dim=1)
//     loss = (dist1 + dist2).mean()
// i.e. two transposes, two gathers, two differences and two norms around the two 1-NN searches (which stay ogc_knn launches).
// Here, given the two index rows:
//     chamfer_terms        dist1[b, i] = || p1[b, i] - pc2[b, idx12[b, i]] ||_p,   dist2[b, j] = || pc2[b, j] - p1[b, idx21[b, j]] ||_p
//     chamfer_terms_grad   grad_p1[b, i] += g1[b, i] * d dist1 / d p1[b, i];   grad_p1[b, idx21[b, j]] += g2[b, j] * d dist2 / d p1[...]
// (the indices are constants of the differentiation, as `idx.detach()` makes them in the reference; pc2 is data).  p = 2:
// sqrtf((dx*dx + dy*dy) + dz*dz), gradient d / ||d|| (0 where the distance is 0, as torch's norm backward); p = 1: the sum of
// magnitudes, gradient sign(d).  HBM-bound: 24 bytes of coordinates + 4 of index in, 4 out per point and direction.
#include "ogc_common.h"

namespace {

__global__ __launch_bounds__(256) void chamfer_terms_kernel(int n1, int n2, int p, const float *__restrict__ p1,
                                                            const float *__restrict__ pc2, const int *__restrict__ idx12,
                                                            const int *__restrict__ idx21, float *__restrict__ dist1,
                                                            float *__restrict__ dist2) {
    const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n1 + n2) return;
    const float *a = p1 + (size_t)b * n1 * 3, *c = pc2 + (size_t)b * n2 * 3;
    const bool first = t < n1;
    const int i = first ? t : t - n1;
    const int j = first ? idx12[(size_t)b * n1 + i] : idx21[(size_t)b * n2 + i];
    const float *u = first ? a + (size_t)i * 3 : c + (size_t)i * 3;   // the point
    const float *v = first ? c + (size_t)j * 3 : a + (size_t)j * 3;   // its nearest neighbour in the other cloud
    const float dx = u[0] - v[0], dy = u[1] - v[1], dz = u[2] - v[2];
    const float d = p == 1 ? (fabsf(dx) + fabsf(dy)) + fabsf(dz) : sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    (first ? dist1 + (size_t)b * n1 : dist2 + (size_t)b * n2)[i] = d;
}

__global__ __launch_bounds__(256) void chamfer_terms_grad_kernel(int n1, int n2, int p, const float *__restrict__ p1,
                                                                 const float *__restrict__ pc2, const int *__restrict__ idx12,
                                                                 const int *__restrict__ idx21, const float *__restrict__ g1,
                                                                 const float *__restrict__ g2, float *__restrict__ grad_p1) {
    const int b = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n1 + n2) return;
    const float *a = p1 + (size_t)b * n1 * 3, *c = pc2 + (size_t)b * n2 * 3;
    const bool first = t < n1;
    const int i = first ? t : t - n1;
    const int j = first ? idx12[(size_t)b * n1 + i] : idx21[(size_t)b * n2 + i];
    const int k1 = first ? i : j;                                     // the point of p1 involved
    const float *u = a + (size_t)k1 * 3;
    const float *v = c + (size_t)(first ? j : i) * 3;
    // d = (p1 point) - (pc2 point) in both directions: dist2's difference is its negative, and so is its derivative w.r.t. p1
    const float dx = u[0] - v[0], dy = u[1] - v[1], dz = u[2] - v[2];
    const float g = first ? g1[(size_t)b * n1 + i] : g2[(size_t)b * n2 + i];
    float gx, gy, gz;
    if (p == 1) {
        gx = dx > 0.f ? g : (dx < 0.f ? -g : 0.f);
        gy = dy > 0.f ? g : (dy < 0.f ? -g : 0.f);
        gz = dz > 0.f ? g : (dz < 0.f ? -g : 0.f);
    } else {
        const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float s = nrm > 0.f ? g / nrm : 0.f;
        gx = dx * s; gy = dy * s; gz = dz * s;
    }
    float *o = grad_p1 + ((size_t)b * n1 + k1) * 3;
    unsafeAtomicAdd(o, gx);
    unsafeAtomicAdd(o + 1, gy);
    unsafeAtomicAdd(o + 2, gz);
}

// deterministic mode: one thread per point of p1 — its own term (direction 1) first, then the terms of the points of pc2 whose
// nearest neighbour it is (direction 2) in ascending order of those points (lists from ogc_det_lists over idx21)
__global__ __launch_bounds__(256) void chamfer_terms_grad_det_kernel(int n1, int n2, int p, const float *__restrict__ p1,
                                                                     const float *__restrict__ pc2, const int *__restrict__ idx12,
                                                                     const int *__restrict__ start, const int *__restrict__ pos,
                                                                     const float *__restrict__ g1, const float *__restrict__ g2,
                                                                     float *__restrict__ grad_p1) {
    const int b = blockIdx.y, k1 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k1 >= n1) return;
    const float *a = p1 + (size_t)b * n1 * 3, *c = pc2 + (size_t)b * n2 * 3;
    const float *u = a + (size_t)k1 * 3;
    float ax = 0.f, ay = 0.f, az = 0.f;
    auto term = [&](const float *v, float g) {
        const float dx = u[0] - v[0], dy = u[1] - v[1], dz = u[2] - v[2];
        float gx, gy, gz;
        if (p == 1) {
            gx = dx > 0.f ? g : (dx < 0.f ? -g : 0.f);
            gy = dy > 0.f ? g : (dy < 0.f ? -g : 0.f);
            gz = dz > 0.f ? g : (dz < 0.f ? -g : 0.f);
        } else {
            const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
            const float s = nrm > 0.f ? g / nrm : 0.f;
            gx = dx * s; gy = dy * s; gz = dz * s;
        }
        ax += gx; ay += gy; az += gz;
    };
    term(c + (size_t)idx12[(size_t)b * n1 + k1] * 3, g1[(size_t)b * n1 + k1]);
    const int *rs = start + (size_t)b * (n1 + 1);
    const int *ps = pos + (size_t)b * n2;
    for (int e = rs[k1]; e < rs[k1 + 1]; ++e) {
        const int i = ps[e];
        term(c + (size_t)i * 3, g2[(size_t)b * n2 + i]);
    }
    float *o = grad_p1 + ((size_t)b * n1 + k1) * 3;
    o[0] = ax; o[1] = ay; o[2] = az;
}

} // namespace

extern "C" int ogc_chamfer_terms(int b, int n1, int n2, int p, const float *p1, const float *pc2, const int *idx12,
                                 const int *idx21, float *dist1, float *dist2, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n1 >= 0 && n2 >= 0, "ogc_chamfer_terms: negative dimension");
    OGC_REQUIRE(p == 1 || p == 2, "ogc_chamfer_terms: norm must be 1 or 2 (got %d)", p);
    if (b == 0 || n1 + n2 == 0) return OGC_OK;
    OGC_REQUIRE(n1 >= 1 && n2 >= 1, "ogc_chamfer_terms: an empty cloud has no nearest neighbour");
    OGC_REQUIRE(p1 && pc2 && idx12 && idx21 && dist1 && dist2, "ogc_chamfer_terms: null pointer");
    OGC_REQUIRE(b <= 65535 && (long long)b * (n1 > n2 ? n1 : n2) * 3 < (1ll << 31), "ogc_chamfer_terms: exceeds 32-bit indexing");
    hipLaunchKernelGGL(chamfer_terms_kernel, dim3(ogc_divup(n1 + n2, 256), b), dim3(256), 0, (hipStream_t)stream, n1, n2, p, p1, pc2,
                       idx12, idx21, dist1, dist2);
    OGC_CHECK_LAUNCH("ogc_chamfer_terms");
    return OGC_OK;
}

extern "C" int ogc_chamfer_terms_grad(int b, int n1, int n2, int p, const float *p1, const float *pc2, const int *idx12,
                                      const int *idx21, const float *g1, const float *g2, float *grad_p1, ogc_stream_t stream) {
    OGC_REQUIRE(b >= 0 && n1 >= 0 && n2 >= 0, "ogc_chamfer_terms_grad: negative dimension");
    OGC_REQUIRE(p == 1 || p == 2, "ogc_chamfer_terms_grad: norm must be 1 or 2 (got %d)", p);
    if (b == 0 || n1 == 0) return OGC_OK;
    OGC_REQUIRE(n2 >= 1, "ogc_chamfer_terms_grad: an empty cloud has no nearest neighbour");
    OGC_REQUIRE(p1 && pc2 && idx12 && idx21 && g1 && g2 && grad_p1, "ogc_chamfer_terms_grad: null pointer");
    OGC_REQUIRE(b <= 65535 && (long long)b * (n1 > n2 ? n1 : n2) * 3 < (1ll << 31), "ogc_chamfer_terms_grad: exceeds 32-bit indexing");
    hipStream_t s = (hipStream_t)stream;
    if (ogc_deterministic()) {
        const int *start = nullptr, *pos = nullptr;
        const int rc = ogc_det_lists("ogc_chamfer_terms_grad", b, n1, n2, idx21, &start, &pos, 0, nullptr, s);
        if (rc != OGC_OK) return rc;
        hipLaunchKernelGGL(chamfer_terms_grad_det_kernel, dim3(ogc_divup(n1, 256), b), dim3(256), 0, s, n1, n2, p, p1, pc2, idx12,
                           start, pos, g1, g2, grad_p1);
        OGC_CHECK_LAUNCH("ogc_chamfer_terms_grad");
        return OGC_OK;
    }
    if (ogc_zero_async(grad_p1, sizeof(float) * (size_t)b * n1 * 3, s) != hipSuccess) {
        ogc_set_error("ogc_chamfer_terms_grad: zero fill failed");
        return OGC_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(chamfer_terms_grad_kernel, dim3(ogc_divup(n1 + n2, 256), b), dim3(256), 0, s, n1, n2, p, p1, pc2, idx12, idx21,
                       g1, g2, grad_p1);
    OGC_CHECK_LAUNCH("ogc_chamfer_terms_grad");
    return OGC_OK;
}
