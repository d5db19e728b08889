// kabsch.hip — batched 3x3 "Kabsch rotation": R = V diag(1, 1, det(V U^T)) U^T for S = U diag(s) V^T.
//
// Replaces the torch.svd call on (B*K, 3, 3) cross-covariances in the reference's weighted-Kabsch fit
// (losses/seg_loss_unsup.py:44-53).  rocSOLVER's batched SVD costs ~40 launches per call for 40 matrices; here one
// thread handles one matrix with a cyclic Jacobi eigen-solve of S^T S in fp64 (three sweeps reach machine
// precision for 3x3), recovers U column by column (with Gram-Schmidt completion for vanishing singular values) and
// forms R.  The rotation is unique whenever S has rank >= 2 and sigma_2 > sigma_3 is not needed for uniqueness of
// the product V diag(..) U^T; any correct SVD gives the same R up to rounding, so results agree with the reference
// to fp32 accuracy.  A matrix containing NaN/inf yields the identity (the reference's `valid_batches` rule, :38-42).
#include "ogc_common.h"

namespace {

__device__ inline void jacobi_rot(double (&A)[3][3], double (&V)[3][3], int p, int q) {
    if (A[p][q] == 0.0) return;
    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    for (int k = 0; k < 3; ++k) { // A <- A J
        const double akp = A[k][p], akq = A[k][q];
        A[k][p] = c * akp - s * akq;
        A[k][q] = s * akp + c * akq;
    }
    for (int k = 0; k < 3; ++k) { // A <- J^T A
        const double apk = A[p][k], aqk = A[q][k];
        A[p][k] = c * apk - s * aqk;
        A[q][k] = s * apk + c * aqk;
    }
    for (int k = 0; k < 3; ++k) { // V <- V J
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
}

__device__ inline void cross3(const double (&a)[3], const double (&b)[3], double (&o)[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ void kabsch_rotation_kernel(int nb, const float *__restrict__ S_in, float *__restrict__ R_out,
                                       int *__restrict__ valid_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    double S[3][3];
    bool finite = true;
    double scale = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const float v = S_in[i * 9 + r * 3 + c];
            finite = finite && isfinite(v);
            S[r][c] = v;
            scale = fmax(scale, fabs((double)v));
        }
    float *Ro = R_out + i * 9;
    if (valid_out) valid_out[i] = finite ? 1 : 0;
    if (!finite || scale == 0.0) { // ill-posed (NaN) -> identity; all-zero S: any rotation is optimal, torch gives I
        for (int k = 0; k < 9; ++k) Ro[k] = (k % 4 == 0) ? 1.0f : 0.0f;
        return;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) S[r][c] /= scale;
    // A = S^T S, eigen-decomposition A = V diag(l) V^T
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = S[0][r] * S[0][c] + S[1][r] * S[1][c] + S[2][r] * S[2][c];
    for (int sweep = 0; sweep < 6; ++sweep) {
        jacobi_rot(A, V, 0, 1);
        jacobi_rot(A, V, 0, 2);
        jacobi_rot(A, V, 1, 2);
    }
    // sort eigenpairs descending
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (A[ord[b]][ord[b]] > A[ord[a]][ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double Vs[3][3], sig[3];
    for (int j = 0; j < 3; ++j) {
        sig[j] = sqrt(fmax(A[ord[j]][ord[j]], 0.0));
        for (int r = 0; r < 3; ++r) Vs[r][j] = V[r][ord[j]];
    }
    // make V a proper basis (det +1) — the reflection is handled by the diag(1,1,det) factor below, which only
    // depends on det(V) det(U), so flipping a V column together with the matching U column changes nothing.
    double U[3][3];
    const double tiny = 1e-12 * fmax(sig[0], 1e-300);
    int rank = 0;
    for (int j = 0; j < 3; ++j) {
        double u[3];
        for (int r = 0; r < 3; ++r) u[r] = S[r][0] * Vs[0][j] + S[r][1] * Vs[1][j] + S[r][2] * Vs[2][j];
        if (sig[j] > tiny) {
            // re-orthogonalise against previous columns (guards tiny sigma ratios), then normalise
            for (int p = 0; p < j; ++p) {
                const double dp = u[0] * U[0][p] + u[1] * U[1][p] + u[2] * U[2][p];
                for (int r = 0; r < 3; ++r) u[r] -= dp * U[r][p];
            }
            const double nrm = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            if (nrm > tiny) {
                for (int r = 0; r < 3; ++r) U[r][j] = u[r] / nrm;
                rank = j + 1;
                continue;
            }
        }
        break;
    }
    if (rank == 0) { // cannot happen for scale > 0, kept for safety
        for (int k = 0; k < 9; ++k) Ro[k] = (k % 4 == 0) ? 1.0f : 0.0f;
        return;
    }
    if (rank == 1) { // pick any unit vector orthogonal to U[:,0]
        double a[3] = {U[0][0], U[1][0], U[2][0]}, e[3] = {0, 0, 0}, o[3];
        const int m = fabs(a[0]) <= fabs(a[1]) && fabs(a[0]) <= fabs(a[2]) ? 0 : (fabs(a[1]) <= fabs(a[2]) ? 1 : 2);
        e[m] = 1.0;
        cross3(a, e, o);
        const double nrm = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
        for (int r = 0; r < 3; ++r) U[r][1] = o[r] / nrm;
        rank = 2;
    }
    {   // third columns: with d = det(V U^T) the product V diag(1,1,d) U^T equals
        // v1 u1^T + v2 u2^T + (v1 x v2)(u1 x u2)^T  — independent of the sign conventions of v3 / u3.
        double v1[3] = {Vs[0][0], Vs[1][0], Vs[2][0]}, v2[3] = {Vs[0][1], Vs[1][1], Vs[2][1]}, v3[3];
        double u1[3] = {U[0][0], U[1][0], U[2][0]}, u2[3] = {U[0][1], U[1][1], U[2][1]}, u3[3];
        cross3(v1, v2, v3);
        cross3(u1, u2, u3);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Ro[r * 3 + c] = (float)(v1[r] * u1[c] + v2[r] * u2[c] + v3[r] * u3[c]);
    }
}

} // namespace

extern "C" int ogc_kabsch_rotation(int nb, const float *S, float *R, int *valid, ogc_stream_t stream) {
    OGC_REQUIRE(nb >= 0, "ogc_kabsch_rotation: negative batch");
    if (nb == 0) return OGC_OK;
    OGC_REQUIRE(S && R, "ogc_kabsch_rotation: null pointer");
    hipLaunchKernelGGL(kabsch_rotation_kernel, dim3(ogc_divup(nb, 64)), dim3(64), 0, (hipStream_t)stream, nb, S, R,
                       valid);
    OGC_CHECK_LAUNCH("ogc_kabsch_rotation");
    return OGC_OK;
}
