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

// One WAVEFRONT per problem, lane `it` on position `it` of scipy's `remaining` list: the scan over the remaining columns — the
// inner loop of every step of a shortest augmenting path — is one lane-parallel update plus a wave-wide minimum, and the
// tie rule of the sequential scan (the first position attaining the minimum, replaced by every LATER position that attains
// it with a column that has no row yet) becomes two ballots: the highest such later position if there is one, else the
// first.  State in LDS (column- and row-indexed arrays are addressed through `remaining`, a permutation that changes by
// swap-removal).  One thread per problem, as this kernel was first written, spent 65 us of the loss phase's main queue on 16
// problems of 10 x 10; this form ~8.
// minimum over the wavefront, returned to every lane: six DPP steps (within quads, rows, then row broadcasts into the last row)
// on both halves of the double, one 64-bit compare and two selects each, and a readlane of lane 63 — ds_bpermute shuffles made
// this reduction half of a step's latency.
template <int CTRL, int ROW_MASK, bool KEEP_OWN>
__device__ __forceinline__ double lsap_dpp_min(double v) {
    const long long bits = __double_as_longlong(v);
    const int vlo = (int)(unsigned)bits, vhi = (int)(bits >> 32);
    // lanes whose source is outside the wave or whose row is masked keep their own value
    const int lo = __builtin_amdgcn_update_dpp(vlo, vlo, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(vhi, vhi, CTRL, ROW_MASK, 0xF, false);
    const double o = __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
    return o < v ? o : v;
}
__device__ __forceinline__ double lsap_wave_min(double v) {
    v = lsap_dpp_min<0xB1, 0xF, true>(v);  // quad xor 1
    v = lsap_dpp_min<0x4E, 0xF, true>(v);  // quad xor 2
    v = lsap_dpp_min<0x141, 0xF, true>(v); // half-row mirror
    v = lsap_dpp_min<0x140, 0xF, true>(v); // row mirror: every row uniform
    v = lsap_dpp_min<0x142, 0xA, true>(v); // row_bcast:15 into rows 1 and 3
    v = lsap_dpp_min<0x143, 0xC, true>(v); // row_bcast:31 into rows 2 and 3: lane 63 holds the minimum
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)bits, 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

__global__ __launch_bounds__(64) void lsap_maximize_kernel(int np, int k, const float *__restrict__ score,
                                                           int *__restrict__ col4row_out) {
    extern __shared__ __attribute__((aligned(8))) unsigned char lsap_smem[];
    const int lane = threadIdx.x, prob = blockIdx.x;
    double *cost = reinterpret_cast<double *>(lsap_smem); // [k][k]: the negated scores (maximize)
    double *u = cost + (size_t)k * k, *v = u + k, *spc = v + k; // duals, shortest path costs
    short *path = reinterpret_cast<short *>(spc + k), *col4row = path + k, *row4col = col4row + k, *remaining = row4col + k;
    short *SR = remaining + k, *SC = SR + k;
    const float *sc = score + (size_t)prob * k * k;
    bool bad = false;
    for (int e = lane; e < k * k; e += 64) { // scipy rejects NaN and -inf costs (= +inf scores) before solving
        const float c = sc[e];
        bad = bad || c != c || c == INFINITY;
        cost[e] = -(double)c;
    }
    bool feasible = __builtin_amdgcn_ballot_w64(bad) == 0;
    if (lane < k) {
        u[lane] = 0.0; v[lane] = 0.0;
        path[lane] = -1; col4row[lane] = -1; row4col[lane] = -1;
    }
    __syncthreads();
    for (int cur = 0; cur < k && feasible; ++cur) {
        // ---- shortest augmenting path from row `cur`
        if (lane < k) {
            remaining[lane] = (short)(k - lane - 1);
            SR[lane] = 0; SC[lane] = 0;
            spc[lane] = INFINITY;
        }
        __syncthreads();
        double min_val = 0.0;
        int num_remaining = k, sink = -1, i = cur;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            double mine = INFINITY;
            int j = -1;
            bool unassigned = false;
            if (lane < num_remaining) {
                j = remaining[lane];
                const double r = min_val + cost[i * k + j] - u[i] - v[j];
                double s = spc[j];
                if (r < s) { path[j] = (short)i; spc[j] = r; s = r; }
                mine = s;
                unassigned = row4col[j] == -1;
            }
            const double lowest = lsap_wave_min(mine);
            min_val = lowest;
            if (!(min_val < INFINITY)) { feasible = false; break; } // NaN / inf scores: scipy raises; we emit -1
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(lane < num_remaining && mine == lowest);
            const unsigned long long un = __builtin_amdgcn_ballot_w64(lane < num_remaining && mine == lowest && unassigned);
            const int index = un ? 63 - __builtin_clzll(un) : __builtin_ctzll(eq);
            const int jw = __builtin_amdgcn_readlane(j, index); // (index is wave-uniform)
            __syncthreads(); // everybody has read `remaining` and row4col
            const int owner = (int)row4col[jw];
            if (owner == -1) sink = jw; else i = owner;
            if (lane == 0) {
                SC[jw] = 1;
                remaining[index] = remaining[num_remaining - 1];
            }
            --num_remaining;
            __syncthreads();
        }
        if (!feasible) break;
        // ---- dual update (col4row is still the assignment before this row's augmentation)
        if (lane < k) {
            if (lane == cur) u[cur] += min_val;
            else if (SR[lane]) u[lane] += min_val - spc[col4row[lane]];
            if (SC[lane]) v[lane] -= min_val - spc[lane];
        }
        __syncthreads();
        // ---- augment
        if (lane == 0) {
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
        __syncthreads();
    }
    int *o = col4row_out + (size_t)prob * k;
    if (lane < k) o[lane] = feasible ? (int)col4row[lane] : -1;
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
    const size_t smem = ((size_t)k * k + 3 * k) * sizeof(double) + (size_t)6 * k * sizeof(short);
    hipLaunchKernelGGL(lsap_maximize_kernel, dim3(np), dim3(64), smem, (hipStream_t)stream, np, k, score, col4row);
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
