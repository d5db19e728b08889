// small_solvers.hip — the two tiny dense solvers of the OGC losses, on the device so that a training step has no
// host round trip:
//
//   ogc_lsap_maximize   the object-matching step of the invariance loss (losses/seg_loss_unsup.py:234-239 runs
//                       scipy.optimize.linear_sum_assignment(iou, maximize=True) per sample on the host, one
//                       device->host copy + one host->device copy per sample and direction);
//   ogc_sym_eigvals     eigenvalues of small symmetric fp64 matrices (the K x K Gram matrix of the masks, whose
//                       square-rooted eigenvalues sum to the nuclear norm the reference takes from a tall SVD,
//                       losses/seg_loss_unsup.py:300-314).
//
// The assignment is NOT unique when IoUs tie (empty slots give all-zero rows), and the invariance loss depends on
// which of the tied columns is picked, so the kernel restates the exact procedure of the reference's dependency:
// scipy's rectangular_lsap (Crouse, "On implementing 2D rectangular assignment algorithms", 2016; shortest
// augmenting paths with dual variables u, v; candidate columns visited in DESCENDING column order on the first pass
// because `remaining` is filled in reverse; on equal reduced cost a column WITHOUT a row wins; removal from
// `remaining` by swapping in the last entry).  scipy is not vendored in the reference (requirements.txt:1, unpinned);
// tests/test_small_solvers.py pins this restatement against the scipy installed in the image (1.15.3) on tie-heavy
// inputs.  All arithmetic is fp64 on the negated fp32 IoUs, as scipy does after `maximize` negation.
#include "ogc_common.h"

namespace {

constexpr int LSAP_MAX = 64;

// one thread per problem; the state lives in LDS (dynamic indexing of private arrays would go to scratch memory)
__global__ __launch_bounds__(64) void lsap_maximize_kernel(int np, int k, const float *__restrict__ score,
                                                           int *__restrict__ col4row_out) {
    extern __shared__ __attribute__((aligned(8))) unsigned char lsap_smem[];
    const int t = threadIdx.x, prob = blockIdx.x * blockDim.x + t;
    if (prob >= np) return;
    // per-thread slices
    double *base = reinterpret_cast<double *>(lsap_smem) + (size_t)t * (3 * k);
    double *u = base, *v = base + k, *spc = base + 2 * k; // duals, shortest path costs
    short *ibase = reinterpret_cast<short *>(lsap_smem + (size_t)blockDim.x * 3 * k * sizeof(double)) + (size_t)t * (6 * k);
    short *path = ibase, *col4row = ibase + k, *row4col = ibase + 2 * k, *remaining = ibase + 3 * k;
    short *SR = ibase + 4 * k, *SC = ibase + 5 * k;
    const float *sc = score + (size_t)prob * k * k;
    for (int i = 0; i < k; ++i) {
        u[i] = 0.0; v[i] = 0.0;
        path[i] = -1; col4row[i] = -1; row4col[i] = -1;
    }
    bool feasible = true;
    for (int e = 0; e < k * k; ++e) // scipy rejects NaN and -inf costs (= +inf scores) before solving
        if (sc[e] != sc[e] || sc[e] == INFINITY) feasible = false;
    for (int cur = 0; cur < k && feasible; ++cur) {
        // ---- shortest augmenting path from row `cur`
        double min_val = 0.0;
        int num_remaining = k;
        for (int it = 0; it < k; ++it) {
            remaining[it] = (short)(k - it - 1);
            SR[it] = 0; SC[it] = 0;
            spc[it] = INFINITY;
        }
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + (-(double)sc[i * k + j]) - u[i] - v[j];
                if (r < spc[j]) { path[j] = (short)i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            min_val = lowest;
            if (!(min_val < INFINITY)) { feasible = false; break; } // NaN / inf scores: scipy raises; we emit -1
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        if (!feasible) break;
        // ---- dual update
        u[cur] += min_val;
        for (int r = 0; r < k; ++r)
            if (SR[r] && r != cur) u[r] += min_val - spc[col4row[r]];
        for (int j = 0; j < k; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        // ---- augment
        int j = sink;
        while (true) {
            const int r = path[j];
            row4col[j] = (short)r;
            const int prev = col4row[r];
            col4row[r] = (short)j;
            j = prev;
            if (r == cur) break;
        }
    }
    int *o = col4row_out + (size_t)prob * k;
    for (int r = 0; r < k; ++r) o[r] = feasible ? (int)col4row[r] : -1;
}

constexpr int EIG_MAX = 64;

// One wavefront per matrix, Jacobi rotations in the PARALLEL (round-robin tournament) order: a round rotates k / 2
// disjoint index pairs at once — their (c, s) come from the same matrix, the column update A <- A J and the row update
// A <- J^T A each touch every element once, one lane per (row, pair) — so a sweep is k - 1 rounds of three barriers
// instead of k (k - 1) / 2 rotations of five, each paying the fp64 divide / square-root chain (K = 10: 0.12 ms -> 0.03).
__global__ __launch_bounds__(64) void sym_eigvals_kernel(int nb, int k, const double *__restrict__ A_in,
                                                         double *__restrict__ w_out) {
    extern __shared__ __attribute__((aligned(8))) double eig_smem[]; // [k][k+1], then c[n/2], s[n/2]
    const int lane = threadIdx.x, b = blockIdx.x;
    const int ld = k + 1;
    const int n = k + (k & 1), half = n >> 1; // an odd k plays with a dummy index k whose pairs are skipped
    double *rc = eig_smem + (size_t)k * ld, *rs = rc + half;
    const double *A = A_in + (size_t)b * k * k;
    double scale = 0.0;
    for (int e = lane; e < k * k; e += 64) {
        const int r = e / k, c = e % k;
        // symmetrise from the lower triangle, like LAPACK's UPLO='L' (torch.linalg.eigvalsh default)
        const double val = r >= c ? A[r * k + c] : A[c * k + r];
        eig_smem[r * ld + c] = val;
        scale = fabs(val) < INFINITY ? fmax(scale, fabs(val)) : INFINITY; // NaN and inf poison the scale
    }
    for (int off = 32; off > 0; off >>= 1) scale = fmax(scale, __shfl_xor(scale, off, 64));
    __syncthreads();
    const bool finite = scale < INFINITY;
    // pair i of round r (circle method): i = 0: (n - 1, r); else ((r + i) mod (n - 1), (r - i) mod (n - 1))
    auto pair_of = [&](int r, int i, int &p, int &q) {
        int a_ = i == 0 ? n - 1 : (r + i) % (n - 1);
        int b_ = i == 0 ? r : (r - i + (n - 1)) % (n - 1);
        p = min(a_, b_);
        q = max(a_, b_);
    };
    if (finite && scale > 0.0 && k > 1) {
        for (int sweep = 0; sweep < 40; ++sweep) {
            double off2 = 0.0;
            for (int e = lane; e < k * k; e += 64) {
                const int r = e / k, c = e % k;
                if (r != c) { const double x = eig_smem[r * ld + c] / scale; off2 += x * x; }
            }
            for (int o = 32; o > 0; o >>= 1) off2 += __shfl_xor(off2, o, 64);
            if (off2 < 1e-30) break; // relative off-diagonal norm 1e-15: eigenvalue error is second order in it
            for (int r = 0; r < n - 1; ++r) {
                for (int i = lane; i < half; i += 64) { // the rotations of this round
                    int p, q;
                    pair_of(r, i, p, q);
                    double c = 1.0, sn = 0.0;
                    if (q < k) {
                        const double apq = eig_smem[p * ld + q];
                        if (apq != 0.0) {
                            const double app = eig_smem[p * ld + p], aqq = eig_smem[q * ld + q];
                            const double theta = (aqq - app) / (2.0 * apq);
                            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                            c = 1.0 / sqrt(tt * tt + 1.0);
                            sn = tt * c;
                        }
                    }
                    rc[i] = c;
                    rs[i] = sn;
                }
                __syncthreads();
                for (int e = lane; e < k * half; e += 64) { // A <- A J : columns p, q of row `row`
                    const int row = e / half, i = e - row * half;
                    int p, q;
                    pair_of(r, i, p, q);
                    if (q < k && rs[i] != 0.0) {
                        const double xp = eig_smem[row * ld + p], xq = eig_smem[row * ld + q];
                        eig_smem[row * ld + p] = rc[i] * xp - rs[i] * xq;
                        eig_smem[row * ld + q] = rs[i] * xp + rc[i] * xq;
                    }
                }
                __syncthreads();
                for (int e = lane; e < k * half; e += 64) { // A <- J^T A : rows p, q of column `col`
                    const int col = e / half, i = e - col * half;
                    int p, q;
                    pair_of(r, i, p, q);
                    if (q < k && rs[i] != 0.0) {
                        const double xp = eig_smem[p * ld + col], xq = eig_smem[q * ld + col];
                        const double np_ = rc[i] * xp - rs[i] * xq, nq_ = rs[i] * xp + rc[i] * xq;
                        // the rotated pair itself is annihilated exactly
                        eig_smem[p * ld + col] = col == q ? 0.0 : np_;
                        eig_smem[q * ld + col] = col == p ? 0.0 : nq_;
                    }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    // ascending order (rank sort by (value, position)), NaN for non-finite input
    if (lane < k) {
        const double mine = eig_smem[lane * ld + lane];
        int rank = 0;
        for (int j = 0; j < k; ++j) {
            const double o = eig_smem[j * ld + j];
            rank += (o < mine || (o == mine && j < lane)) ? 1 : 0;
        }
        w_out[(size_t)b * k + (finite ? rank : lane)] = finite ? mine : NAN;
    }
}

} // namespace

extern "C" int ogc_lsap_maximize(int np, int k, const float *score, int *col4row, ogc_stream_t stream) {
    OGC_REQUIRE(np >= 0 && k >= 0, "ogc_lsap_maximize: negative size");
    if (np == 0 || k == 0) return OGC_OK;
    OGC_REQUIRE(k <= LSAP_MAX, "ogc_lsap_maximize: more than 64 slots");
    OGC_REQUIRE(score && col4row, "ogc_lsap_maximize: null pointer");
    int threads = np < 64 ? np : 64;
    const int fit = 65536 / (k * (int)(3 * sizeof(double) + 6 * sizeof(short))); // 64 KiB of LDS per workgroup
    if (threads > fit) threads = fit;
    const size_t smem = (size_t)threads * k * (3 * sizeof(double) + 6 * sizeof(short));
    hipLaunchKernelGGL(lsap_maximize_kernel, dim3(ogc_divup(np, threads)), dim3(threads), smem, (hipStream_t)stream, np,
                       k, score, col4row);
    OGC_CHECK_LAUNCH("ogc_lsap_maximize");
    return OGC_OK;
}

extern "C" int ogc_sym_eigvals(int nb, int k, const double *A, double *w, ogc_stream_t stream) {
    OGC_REQUIRE(nb >= 0 && k >= 0, "ogc_sym_eigvals: negative size");
    if (nb == 0 || k == 0) return OGC_OK;
    OGC_REQUIRE(k <= EIG_MAX, "ogc_sym_eigvals: matrix larger than 64 x 64");
    OGC_REQUIRE(A && w, "ogc_sym_eigvals: null pointer");
    hipLaunchKernelGGL(sym_eigvals_kernel, dim3(nb), dim3(64), ((size_t)k * (k + 1) + k + 2) * sizeof(double),
                       (hipStream_t)stream, nb, k, A, w);
    OGC_CHECK_LAUNCH("ogc_sym_eigvals");
    return OGC_OK;
}
