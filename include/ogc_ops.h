/*
 * ogc_ops.h — C ABI of libogc_ops.so, the MI355X (gfx950) implementation of OGC's
 * point-cloud operator stack.
 *
 * This is the drop-in boundary for the reference's native extension `pointnet2_cuda`
 * (PYBIND11_MODULE in pointnet2/src/pointnet2_api.cpp:10-25).  Every entry point below
 * replaces exactly one of the ten pybind functions registered there; the argument order
 * and meaning are the reference's, with `at::Tensor` replaced by raw device pointers and
 * the implicit "current CUDA stream" replaced by an explicit hipStream_t (passed as
 * void* so that this header has no HIP dependency).
 *
 * Contract shared by all entry points (reference: SURVEY.md §8b):
 *   - all tensors are dense, row-major, contiguous device memory; floats are fp32,
 *     indices are int32; dims are passed as int (flat offsets must fit in 32 bits);
 *   - the CALLER allocates every output and scratch buffer and keeps ownership;
 *   - launches are asynchronous on `stream`; no global state; re-entrant;
 *   - return value: OGC_OK (0) or a negative OGC_ERR_* code.  Unlike the reference
 *     (which prints and calls exit(-1), e.g. ball_query_gpu.cu:62-66) nothing here ever
 *     terminates the process; ogc_last_error() gives a thread-local message.
 *
 * Distance arithmetic everywhere is the reference's source expression evaluated in fp32,
 * left to right, WITHOUT fused multiply-add:
 *     d = (ux-x)*(ux-x) + (uy-y)*(uy-y) + (uz-z)*(uz-z)
 * (interpolate_gpu.cu:40, ball_query_gpu.cu:33, sampling_gpu.cu:133).
 */
#ifndef OGC_OPS_H
#define OGC_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ogc_stream_t; /* hipStream_t */
typedef unsigned short ogc_bf16_t; /* one bfloat16 (the upper half of an fp32), storage only: the `_h` entry points */

enum {
    OGC_OK = 0,
    OGC_ERR_INVALID_ARG = -1, /* null pointer, negative dim, k out of range ... */
    OGC_ERR_LAUNCH = -2,      /* hipGetLastError() != hipSuccess after the launch */
    OGC_ERR_UNSUPPORTED = -3  /* shape outside what the kernels handle */
};

/* Library version (major*10000 + minor*100 + patch) and last error text (thread-local).  OGC_VERSION is the version of THIS
 * header; a caller checks ogc_version() == OGC_VERSION before anything else (ogc_amd/_lib.py does): the minor number moves with
 * every change of an existing prototype.  0.2.0: ogc_adam_step takes its five hyper-parameters as double (float before), new
 * entry points ogc_zero_arena_begin / _end, ogc_conv1x1_gemm_any, ogc_conv1x1_wgrad_affine_pooled, ogc_conv1x1_dgrad_pooled.  0.2.1 (the patch number moves with new entry
 * points): ogc_gather_xyz_pair, ogc_flow_advance, ogc_linear_cn, ogc_gru_reset, ogc_gru_blend,
 * ogc_soft_corr_flow, ogc_three_nn_weights; ogc_furthest_point_sampling_chain accepts temp == NULL.  0.2.2: the `_h` entry points
 * (activations of the shared MLPs stored as bf16; see "16-bit activations" at the end of this header).  0.2.4: ogc_set_deterministic /
 * ogc_get_deterministic.  0.2.5: ogc_group_linear_fwd_direct. */
#define OGC_VERSION 205
int ogc_version(void);
/* 0: the squared distance of every search is the reference's SOURCE expression, ((dx*dx) + (dy*dy)) + (dz*dz), one rounding per
 * operation (what all parity tests pin).  1: this is libogc_ops_fmad.so, the same library with the search kernels (FPS, kNN,
 * three-NN, ball query) compiled to evaluate fma(dz, dz, fma(dy, dy, dx*dx)) — what `nvcc --fmad=true`, the reference's build,
 * makes of that expression (pointnet2/setup.py has no -fmad=false); for comparisons against index tensors produced by the real
 * CUDA binary on tie-heavy data.  Opt-in: OGC_FMAD=1 in the environment of ogc_amd. */
int ogc_distance_contracted(void);
/* The deterministic-gradient mode (csrc/det.hip; OGC_DETERMINISTIC=1 in the environment of ogc_amd sets it at load time).  The
 * reference accumulates gradients with float atomics (pointnet2/src/group_points_gpu.cu:24, interpolate_gpu.cu:211-213,
 * sampling_gpu.cu:62): their order, and so the last bits of every sum, change from launch to launch — as do this library's fast
 * kernels.  With the mode on, the entry points below form every sum in a FIXED order (bit-identical results from run to run and
 * from process to process on one device and build), slower, same contracts otherwise:
 *   scatter-adds as gathers over ascending position lists — ogc_group_points_grad, ogc_gather_points_grad, ogc_group_concat_grad,
 *     ogc_group_linear_bwd, ogc_three_interpolate_grad, ogc_chamfer_terms_grad; the transposed lists of ogc_group_reverse and
 *     ogc_reverse_neighbours come out sorted, which fixes the order inside ogc_group_points_grad_rev*, ogc_three_interpolate_grad_rev*
 *     and ogc_neighbour_consistency_bwd;
 *   per-workgroup partials + one ordered pass — the weight gradients ogc_conv1x1_wgrad* (all forms), the BatchNorm statistics and
 *     gradient sums of ogc_batch_norm_fwd / _bwd / _maxpool_fwd / _maxpool_bwd.
 * Not converted (the GroupNorm statistics taken in convolution epilogues, the moment matrices of ogc_conv1x1_wgrad_moments*): the
 * host layers route around them while the mode is on (ogc_amd/fused.py, DETERMINISTIC).  Forward results do not depend on the mode
 * except through BatchNorm's batch statistics (fp64 sums, now in slab order). */
int ogc_set_deterministic(int on);
int ogc_get_deterministic(void);
const char *ogc_last_error(void);

/* ---- furthest point sampling -------------------------------------------------------
 * replaces furthest_point_sampling_wrapper(b, n, m, points, temp, idx)
 *   pointnet2/src/sampling.cpp:38-49 -> sampling_gpu.cu:93-253
 * xyz (b,n,3) f32; temp (b,n) f32 scratch, must arrive filled with 1e10 (pointnet2.py:33),
 * left holding the final min-distances; idx (b,m) i32, idx[:,0] = 0.
 * Tie order among equal maxima reproduces the reference's block reduction for
 * block_size = min(1024, 2^floor(log2 n)) (cuda_utils.h:10-14). */
int ogc_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                int *idx, ogc_stream_t stream);

/* Furthest point sampling along a CHAIN of levels (utils/pointnet2_util.py:22-27 and utils/flowstep3d_util.py:112-118
 * sample level l+1 from the centres of level l, which are stored in sampling order).
 * ties_out (b) i32, optional: the first round of this run in which more than one point attained the maximum (exact
 * fp32 ties: duplicated points, lattices, and — late in a run that keeps a large share of the cloud — chance
 * coincidences), INT_MAX if there was none.
 * ties_in (b) i32, optional: ties_out of the run that PRODUCED this cloud (xyz = its first n samples, in sampling
 * order).  Where ties_in[i] >= m the result is known without sampling: every one of the parent's first m rounds picked
 * the unique farthest point from its earlier samples in the whole cloud, so inside the subset sample r is still the
 * unique farthest point from samples 0..r-1 — the answer is 0, 1, ..., m-1 (written directly; temp is left untouched
 * and ties_out[i] = ties_in[i], so the chain can go on).  Otherwise the rounds are run as usual, so the result is in
 * every case the one ogc_furthest_point_sampling returns.  Clouds of more than 16384 points are always sampled and
 * report ties_out = 0 (nothing known).
 * temp may be NULL for clouds of up to 16384 points (0.2.1): the running minima start at 1e10, stay in registers and are not
 * handed back — what every caller of the reference does with them (pointnet2.py:33-35 allocates, fills and drops temp), without the
 * fill launch and the (b, n) buffer. */
int ogc_furthest_point_sampling_chain(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                      const int *ties_in, int *ties_out, ogc_stream_t stream);

/* ---- gather -------------------------------------------------------------------------
 * replaces gather_points_wrapper(b, c, n, npoints, points, idx, out)
 *   sampling.cpp:11-21 -> sampling_gpu.cu:8-43
 * out[b,c,j] = points[b,c,idx[b,j]];  points (b,c,n), idx (b,npoints), out (b,c,npoints) */
int ogc_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                      float *out, ogc_stream_t stream);

/* replaces gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points)
 *   sampling.cpp:24-35 -> sampling_gpu.cu:46-90
 * grad_points[b,c,idx[b,j]] += grad_out[b,c,j]; caller zero-fills grad_points (pointnet2.py:73) */
int ogc_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                           const int *idx, float *grad_points, ogc_stream_t stream);

/* ---- k nearest neighbours ----------------------------------------------------------
 * replaces knn_wrapper(b, n, m, k, unknown, known, dist2, idx)
 *   interpolate.cpp:26-36 -> interpolate_gpu.cu:9-79
 * unknown (b,n,3), known (b,m,3); dist2 (b,n,k) SQUARED distances ascending, idx (b,n,k).
 * Ties keep the lower index first; if m < k the tail is idx=0, dist2=+inf; non-finite
 * distances are never selected.  1 <= k <= 200 (the reference's scratch-array bound,
 * interpolate_gpu.cu:30-31); larger k returns OGC_ERR_INVALID_ARG instead of overrunning. */
int ogc_knn(int b, int n, int m, int k, const float *unknown, const float *known,
            float *dist2, int *idx, ogc_stream_t stream);

/* replaces three_nn_wrapper(b, n, m, unknown, known, dist2, idx)
 *   interpolate.cpp:14-23 -> interpolate_gpu.cu:81-146 ; k = 3 special case of ogc_knn */
int ogc_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int *idx, ogc_stream_t stream);

/* ---- three-point interpolation -----------------------------------------------------
 * replaces three_interpolate_wrapper(b, c, m, n, points, idx, weight, out)
 *   interpolate.cpp:39-52 -> interpolate_gpu.cu:149-189
 * out[b,c,i] = w[b,i,0]*p[b,c,idx[b,i,0]] + w[b,i,1]*p[..1] + w[b,i,2]*p[..2], left to right
 * points (b,c,m), idx/weight (b,n,3), out (b,c,n) */
int ogc_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                          const float *weight, float *out, ogc_stream_t stream);

/* replaces three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points)
 *   interpolate.cpp:55-68 -> interpolate_gpu.cu:192-232
 * grad_points[b,c,idx[b,i,j]] += grad_out[b,c,i]*w[b,i,j]; caller zero-fills (pointnet2.py:181) */
int ogc_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int *idx, const float *weight, float *grad_points,
                               ogc_stream_t stream);

/* ---- grouping ------------------------------------------------------------------------
 * replaces group_points_wrapper(b, c, n, npoints, nsample, points, idx, out)
 *   group_points.cpp:26-36 -> group_points_gpu.cu:47-86
 * out[b,c,p,s] = points[b,c,idx[b,p,s]]; points (b,c,n), idx (b,npoints,nsample) */
int ogc_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int *idx, float *out, ogc_stream_t stream);

/* replaces group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points)
 *   group_points.cpp:11-23 -> group_points_gpu.cu:8-44
 * grad_points[b,c,idx[b,p,s]] += grad_out[b,c,p,s]; caller zero-fills (pointnet2.py:224) */
int ogc_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int *idx, float *grad_points, ogc_stream_t stream);

/* ---- ball query ------------------------------------------------------------------------
 * replaces ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx)
 *   ball_query.cpp:16-27 -> ball_query_gpu.cu:9-66
 * xyz (b,n,3) candidates, new_xyz (b,m,3) centres, idx (b,m,nsample).  Row = first nsample
 * indices k (ascending) with d2(k) < radius*radius, padded with the first hit; a centre with
 * no hit gets an all-zero row.  The reference's kernel leaves such rows untouched and relies
 * on the caller's pre-zeroed buffer (pointnet2.py:251); this one WRITES EVERY ROW, so idx may
 * be uninitialised memory (a pre-zeroed buffer works unchanged). */
int ogc_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int *idx, ogc_stream_t stream);

/* ==== fused extensions (no counterpart in the reference's native module; they replace
 * Python-level op sequences of the reference, cited per function) ======================= */

/* kNN followed by the radius clamp every reference caller applies in Python
 *   pointnet2/pointnet2.py:283-286, utils/flowstep3d_util.py:42-44,
 *   losses/seg_loss_unsup.py:120-122, losses/flow_loss_unsup.py:57-59:
 *     dist = sqrt(dist2); idx[dist > radius] = idx[..., 0]
 * One launch, no host synchronisation (the reference's boolean-mask assignment syncs).
 * dist (b,n,k) receives sqrt(dist2) (what KNN.forward returns, pointnet2.py:103), idx the
 * clamped indices.  radius < 0 means "no clamp" (QueryAndGroup radius=None).
 * Entries whose index was clamped carry dist = +inf (every reference caller discards the
 * distances after the clamp): with radius >= 0 the search is RADIUS-LIMITED — it stops as
 * soon as the scanned cells cover the radius and the nearest neighbour is known, instead of
 * collecting all k neighbours first (C4's smoothness term: k = 32, r = 1 m holds ~2 points). */
int ogc_knn_clamped(int b, int n, int m, int k, float radius, const float *unknown,
                    const float *known, float *dist, int *idx, ogc_stream_t stream);

/* One cell grid for several radius searches of the same clouds in themselves.  The smoothness term of the OGC loss searches every
 * cloud twice — k nearest within r_knn, then the first nsample within r_ball
 *   losses/seg_loss_unsup.py:120-123 (KnnLoss), :151-152 (BallQLoss), both on (pc, pc)
 * — and ogc_knn_clamped / ogc_ball_query each sort the cloud into cells first (a third to a half of either call).  Here the
 * caller builds the grid ONCE, for the largest radius it will ask for, into memory it owns, and runs the searches on it:
 *   ogc_cell_grid_bytes(b, n)            size of the grid buffer (bytes; 256-byte aligned device memory)
 *   ogc_cell_grid_build(b, n, r, xyz, grid)        cells of edge 1.01 r over each cloud's bounding box; xyz (b,n,3), n >= 1024
 *   ogc_ball_query_cells(b, n, radius, nsample, xyz, grid, grid_radius, idx)       == ogc_ball_query(b, n, n, radius, nsample, xyz, xyz, idx)
 *   ogc_knn_clamped_cells(b, n, k, radius, xyz, grid, grid_radius, dist, idx)      == ogc_knn_clamped(b, n, n, k, radius, xyz, xyz, dist, idx)
 * for any radius <= grid_radius (the radius the grid was built with; larger ones are refused: OGC_ERR_UNSUPPORTED), bit for bit —
 * cells longer than a query needs only add candidates.  The grid stays valid until the caller frees or overwrites it; the k-NN
 * search writes a per-cloud flag into it (rows it left to its second kernel), nothing else does. */
long long ogc_cell_grid_bytes(int b, int n);
int ogc_cell_grid_build(int b, int n, float radius, const float *xyz, void *grid, ogc_stream_t stream);
int ogc_ball_query_cells(int b, int n, float radius, int nsample, const float *xyz, const void *grid, float grid_radius, int *idx,
                         ogc_stream_t stream);
int ogc_knn_clamped_cells(int b, int n, int k, float radius, const float *xyz, void *grid, float grid_radius, float *dist, int *idx,
                          ogc_stream_t stream);

/* The NaN rule and the Adam update of a training step (train_seg.py:81-83 + torch.optim.Adam.step(), train_seg.py:320) as two
 * launches over a chunk table, on torch's own state tensors (ogc_amd/csrc/adam.hip; fp32 parameters).
 *   table  (device, int64, 5 x n_tensors): pointers of the parameters, exp_avg, exp_avg_sq, step counts (0-dim fp32 tensors),
 *          then the element counts;  chunks (device, int32, n_chunks x 2): (tensor, first element) of every
 *          ogc_adam_chunk()-element piece;  grad_ptrs: HOST array of the n_tensors gradient pointers (<= ogc_adam_max_tensors());
 *   flag   (device, int32, zero on entry): 1 afterwards when a gradient held a NaN — the update and the step counts are then
 *          skipped on the device;  step_snapshot: n_tensors floats of scratch.  Hyper-parameters are doubles as in ATen's
 *          fused kernel (they meet the fp32 state in double expressions rounded once). */
int ogc_adam_max_tensors(void);
int ogc_adam_chunk(void);
int ogc_adam_step(int n_tensors, int n_chunks, const long long *table, const int *chunks, const void *const *grad_ptrs,
                  float *step_snapshot, int *flag, double lr, double beta1, double beta2, double eps, double weight_decay,
                  ogc_stream_t stream);

/* Batched 3x3 Kabsch rotation.  Replaces torch.svd + the reflection fix of the weighted-Kabsch fit
 *   losses/seg_loss_unsup.py:44-53:  u,s,v = svd(S); R = v diag(1,1,det(v u^T)) u^T.
 * S (nb,3,3) f32 cross-covariances (P_c^T diag(w) Q_c), R (nb,3,3) f32 out; valid (nb) i32 out or NULL:
 * 0 where S contains NaN/inf (then R = I — the reference's `valid_batches` rule, :38-42), else 1.
 * One thread per matrix, Jacobi eigen-solve of S^T S in fp64. */
int ogc_kabsch_rotation(int nb, const float *S, float *R, int *valid, ogc_stream_t stream);

/* Grouping with relative coordinates in front, in one output tensor.  Replaces, in QueryAndGroup.forward
 *   pointnet2/pointnet2.py:284-296:  grouped_xyz = group(xyz^T, idx) - new_xyz^T[..., None];
 *                                    new_features = cat([grouped_xyz, group(features, idx)], dim=1)
 * (two gathers, a broadcast subtraction and a concatenation copy of the largest activation of the level).
 * xyz (b,n,3), new_xyz (b,npoints,3), points (b,c,n) (NULL iff c == 0), idx (b,npoints,nsample) i32,
 * out (b,3+c,npoints,nsample): channels 0..2 = xyz[idx] - new_xyz, channels 3.. = points[idx]. */
int ogc_group_concat(int b, int c, int n, int npoints, int nsample, const float *xyz, const float *new_xyz,
                     const float *points, const int *idx, float *out, ogc_stream_t stream);

/* Gradient of ogc_group_concat w.r.t. points: scatter-add of channels 3.. of grad_out (b,3+c,npoints,nsample) into
 * grad_points (b,c,n), which the caller zero-fills (as for ogc_group_points_grad). */
int ogc_group_concat_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                          float *grad_points, ogc_stream_t stream);

/* First layer of a set-abstraction MLP without the grouped tensor (utils/pointnet2_util.py:33-40: QueryAndGroup, then the
 * first Conv2d of the SharedMLP).  That layer is linear in [x_j - c_i ; f_j], so its feature part commutes with the
 * gather:  y[b, m, i, j] = P[b, m, idx[b, i, j]] + sum_k wx[m, k] * rel[b, k, i, j],  P = W_f . f per point (b, m, n),
 * rel (b, 3, npoints, nsample) the relative coordinates (ogc_group_concat with c = 0), wx (m, 3) the xyz columns of the
 * weight.  The (b, 3 + c, npoints, nsample) grouped tensor, its read by the convolution and the scatter of its gradient
 * never happen.  groups > 0: also the statistics of y for the GroupNorm that follows, in the layout of
 * ogc_conv1x1_gemm_gnstats (ogc_conv1x1_gn_slots() copies of (b, groups, 2) f64, overwritten).
 * Needs npoints * nsample % 4 == 0 and 16-byte aligned tensors (OGC_ERR_UNSUPPORTED otherwise).
 * ogc_group_linear_bwd reads the gradient tensor ONCE for both results that need it:  grad_p[b, m, idx] += grad_y (the
 * scatter-add of ogc_group_points_grad: grad_p (b, m, n) must be zeroed by the caller) and dwx[m, k] += sum grad_y * rel
 * (dwx (m, 3), zeroed by the caller).  Needs n <= 16384, npoints * nsample >= 4096 and a multiple of 16
 * (OGC_ERR_UNSUPPORTED otherwise: use ogc_group_points_grad and ogc_conv1x1_wgrad on (rel, grad_y)).  The remaining
 * gradients are small per-point GEMMs: d features = W_f^T grad_p, d W_f = sum_b grad_p f^T. */
int ogc_group_linear_fwd(int b, int m, int n, int npoints, int nsample, int groups, const float *P, const int *idx,
                         const float *rel, const float *wx, float *y, double *stats, ogc_stream_t stream);
/* ogc_group_linear_fwd for 1 .. 4 feature channels (an encoder's first level: the features are the coordinates) without the
 * point-wise product P: feats (b, cf, n), w (m, 3 + cf) the layer's weight rows; every output is ONE fused-multiply-add chain over
 * the input channels [rel x, y, z, f_0 ..] ascending — the order of the reference's convolution over cat([grouped_xyz,
 * grouped_features]) (pointnet2/pointnet2.py:286-297, utils/nn_util.py:45-85) and of this library's matrix kernels.  Other
 * arguments, statistics layout and preconditions as ogc_group_linear_fwd. */
int ogc_group_linear_fwd_direct(int b, int m, int cf, int n, int npoints, int nsample, int groups, const float *feats,
                                const int *idx, const float *rel, const float *w, float *y, double *stats, ogc_stream_t stream);
int ogc_group_linear_fwd_direct_h(int b, int m, int cf, int n, int npoints, int nsample, int groups, const float *feats,
                                  const int *idx, const float *rel, const float *w, ogc_bf16_t *y, double *stats, ogc_stream_t stream);
int ogc_group_linear_bwd(int b, int m, int n, int npoints, int nsample, const float *grad_y, const int *idx,
                         const float *rel, float *grad_p, float *dwx, ogc_stream_t stream);

/* The grouping gradient as a gather (round 2): grad_points[b,c,j] = sum of grad_out[b,c,t] over the positions t = p * nsample + s
 * with idx[b,p,s] == j — group_points_gpu.cu:8-25 / ogc_group_points_grad without atomics, every output written once (the
 * caller need not zero grad_points).  The transposed lists depend on idx only: ogc_group_reverse builds them once per
 * neighbour tensor (all channels, every step that reuses the tensor): the positions are cut into chunks of
 * tc = ogc_group_reverse_chunk(n, npoints, nsample); rev_start (b, chunks, n + 1) i32 delimits, per chunk and point, the
 * entries in rev_pos (b, npoints * nsample) u16 (positions relative to their chunk; the lists of chunk ch start at ch * tc;
 * order inside a list unspecified).  Inside every aligned group of 16 positions a run of equal indices (clamped / padded rows) is
 * listed once, by its first position; heads (b, npoints * nsample / 16) u16 has one bit per position, set for run heads — the
 * gradient kernel folds a run into its head while it stages the values.  Indices outside [0, n) are ignored.  n <= 16384,
 * npoints * nsample a multiple of 16. */
int ogc_group_reverse_chunk(int n, int npoints, int nsample);
int ogc_group_reverse(int b, int n, int npoints, int nsample, const int *idx, int *rev_start, unsigned short *rev_pos,
                      unsigned short *heads, ogc_stream_t stream);
int ogc_group_points_grad_rev(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *rev_start,
                              const unsigned short *rev_pos, const unsigned short *heads, float *grad_points,
                              ogc_stream_t stream);

/* The gradient of three_interpolate as the same gather: rev_* = ogc_group_reverse(b, m, n, 3, idx (b,n,3), ...) — position
 * t = 3 i + k stands for grad_out[b,c,i] * weight[b,i,k].  grad_points (b,c,m) is overwritten.  3 n a multiple of 16. */
/* ogc_group_points_grad_rev that also returns, from the same pass over grad_out, the three coordinate columns of a grouped first
 * layer's weight gradient (ogc_group_linear_fwd): dwx (c, 3) = sum_{b, pos} grad_out[b, ch, pos] * rel[b, k, pos] — the result of
 * ogc_conv1x1_wgrad(rel, grad_out) without its second read of grad_out.  dwx is zeroed by the call.  (_h: grad_out in bf16.) */
int ogc_group_points_grad_rev_dwx(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *rev_start,
                                  const unsigned short *rev_pos, const unsigned short *heads, const float *rel,
                                  float *grad_points, float *dwx, ogc_stream_t stream);
int ogc_three_interpolate_grad_rev(int b, int c, int n, int m, const float *grad_out, const float *weight,
                                   const int *rev_start, const unsigned short *rev_pos, const unsigned short *heads,
                                   float *grad_points, ogc_stream_t stream);
/* ... with grad_out a channel slice of a wider gradient (what autograd hands the interpolation underneath the concatenation of
 * PointnetFPModule.forward, utils/pointnet2_util.py:115-117): grad_out_bstride floats between consecutive samples' (c, n) planes,
 * >= c n; rows stay n contiguous floats.  Saves the copy `.contiguous()` would make. */
int ogc_three_interpolate_grad_rev_bs(int b, int c, int n, int m, const float *grad_out, long long grad_out_bstride,
                                      const float *weight, const int *rev_start, const unsigned short *rev_pos,
                                      const unsigned short *heads, float *grad_points, ogc_stream_t stream);

/* The scalar algebra either side of the fused loss terms (losses/seg_loss_unsup.py:353-392: a mean per term and view, summed
 * and weighted; the masks' gradient as the sum of its five consumers'), one launch per direction.  Pointer / size ARRAYS are host
 * memory, read during the call; at most 8 tensors.
 * ogc_view_means:      out[k] = mean of row k, rows of tensor 0 first: tensor i is dense fp32, viewed as (rows[i], len[i]).
 * ogc_view_means_grad: grad[i][r, :] = (weight[k(i, r)] * g_loss[0]) * (1 / len[i]) — the adjoint of the means under
 *                      loss = <weight, out>; weight (sum rows) and g_loss (1) are device memory.
 * ogc_sum_ranges:      out[e] = sum over the parts i with first[i] <= e < first[i] + count[i] of src[i][e - first[i]], in part
 *                      order; 0 where no part covers e.  total = elements of out. */
int ogc_view_means(int parts, const float *const *src, const int *rows, const long long *len, float *out, ogc_stream_t stream);
int ogc_view_means_grad(int parts, float *const *grad, const int *rows, const long long *len, const float *weight,
                        const float *g_loss, ogc_stream_t stream);
int ogc_sum_ranges(int parts, const float *const *src, const long long *first, const long long *count, long long total,
                   float *out, ogc_stream_t stream);

/* Dynamic (rigid-motion) term of the OGC loss, fused.  Replaces DynamicLoss.forward + fit_motion_svd_batch
 *   losses/seg_loss_unsup.py:64-98, :10-61 (K-fold expanded clouds, einsums, ~65 launches per step).
 * ogc_rigid_moments: per (cloud, slot) the weighted moments of p = pc and q = pc2 with weights mask[:, slot], accumulated
 *   in fp64 in ONE pass: mom (vb,k,16) f64 scratch {W, sum w p, sum w q, sum w p q^T}; outputs S (vb*k,3,3) f32 =
 *   sum w (p - pbar)(q - qbar)^T (the input of ogc_kabsch_rotation) and means (vb*k,6) f32 = (pbar, qbar).  A slot with
 *   zero weight gives NaN (as the reference's division does), which ogc_kabsch_rotation flags invalid.
 * ogc_rigid_translation: t = qbar - R pbar; fits flagged invalid get R = I, t = 0 (:40-42, :58-59).  R is updated in place.
 * ogc_rigid_blend: backward == 0: out (vb,n) = || sum_k mask[n,k] (R_k p_n + t_k) - q_n ||_p;  backward != 0:
 *   out (vb,n,k) = d/d mask of that, times grad_out (vb,n) (the fit itself is detached, :91).  p = 1 or 2; k <= 32.
 *   p == 0 (forward only): out (vb,n,3) = sum_k mask[n,k] (R_k p_n + t_k) - q_n, the vector itself — with pc2 = pc the
 *   mask-blended rigid flow of object-aware ICP (oa_icp.py:31-38, :75-83).
 * pc, pc2 (vb,n,3), mask (vb,n,k) point-major, R (vb*k,3,3), t (vb*k,3). */
int ogc_rigid_moments(int vb, int n, int k, const float *pc, const float *pc2, const float *mask, double *mom, float *S,
                      float *means, ogc_stream_t stream);
int ogc_rigid_translation(int total, const float *means, const int *valid, float *R, float *t, ogc_stream_t stream);
int ogc_rigid_blend(int vb, int n, int k, int p, int backward, const float *pc, const float *pc2, const float *mask,
                    const float *R, const float *t, const float *grad_out, float *out, ogc_stream_t stream);

/* Invariance term of the OGC loss, fused.  Replaces match_mask_by_iou's IoU matrix and InvarianceLoss.distance
 *   losses/seg_loss_unsup.py:212-233, :252-262 (arg-max, one-hot tensors, einsums, permutation products).
 * ogc_mask_iou: iou (pb,k,k) f32 of the hard (arg-max, first maximum) segmentations of mask1 / mask2 (pb,n,k):
 *   intersection / clamp(|g in 1| + |p in 2| - intersection, 1e-10); counts (pb,k,k) i32 scratch.
 * ogc_matched_distance: backward == 0: out1 (pb,n) = ||mask1[n] - mask2[n, col12]||_p, out2 (pb,n) =
 *   ||mask2[n] - mask1[n, col21]||_p with col12 / col21 (pb,k) i32 the assignments of ogc_lsap_maximize for iou and
 *   iou^T; backward != 0: out1 / out2 (pb,n,k) = gradients w.r.t. mask1 / mask2 through the FIRST operand of each
 *   distance only (the permuted targets are detached, :259-260), times grad12 / grad21 (pb,n).  p = 1 or 2; k <= 32. */
int ogc_mask_iou(int pb, int n, int k, const float *mask1, const float *mask2, int *counts, float *iou,
                 ogc_stream_t stream);
int ogc_matched_distance(int pb, int n, int k, int p, int backward, const float *mask1, const float *mask2,
                         const int *col12, const int *col21, const float *grad12, const float *grad21, float *out1,
                         float *out2, ogc_stream_t stream);

/* Linear sum assignment (maximise), batched, on the device.  Replaces the host round trip of the invariance loss
 *   losses/seg_loss_unsup.py:234-239:  scipy.optimize.linear_sum_assignment(iou[b], maximize=True)[1] per sample.
 * score (np,k,k) f32 row-major (rows = slots of mask1), col4row (np,k) i32 out: column assigned to each row.
 * Restates scipy's rectangular_lsap procedure (shortest augmenting paths, its column visiting order and its
 * tie-break) in fp64, so tied IoUs (empty slots) resolve exactly as on the host.  A problem containing NaN or
 * +inf scores yields -1 in every entry (scipy raises ValueError).  k <= 64. */
int ogc_lsap_maximize(int np, int k, const float *score, int *col4row, ogc_stream_t stream);

/* Eigenvalues (ascending) of small symmetric fp64 matrices, lower triangle referenced.  Replaces the singular
 * values of the nuclear-norm monitor  losses/seg_loss_unsup.py:300-314  (sum sqrt(eig(M^T M)) == sum svd(M)).
 * A (nb,k,k) f64, w (nb,k) f64 out; cyclic Jacobi, one wavefront per matrix; NaN/inf input -> NaN.  k <= 64. */
int ogc_sym_eigvals(int nb, int k, const double *A, double *w, ogc_stream_t stream);

/* Mask smoothness over neighbour lists, fused.  Replaces, in KnnLoss / BallQLoss
 *   losses/seg_loss_unsup.py:123-129, :152-158:
 *     nn_mask = grouping_operation(mask, idx); loss = (mask.unsqueeze(3) - nn_mask).norm(p, dim=1).mean(dim=-1)
 * mask (b,n,c) f32 POINT-major (the layout the network emits), idx (b,n,k) i32 into the same cloud,
 * out (b,n) f32: out[i] = (1/k) sum_j ||mask[i] - mask[idx[i,j]]||_p, p = 1 or 2. */
int ogc_neighbour_consistency_fwd(int b, int n, int c, int k, int p, const float *mask, const int *idx, float *out,
                                  ogc_stream_t stream);

/* Transposed neighbour lists (coordinates only; lets the gradient above be a gather instead of a scatter-add).
 * idx (b,n,k) i32 -> rev_start (b,n+1) i32: CSR offsets of the edges ARRIVING at each point; rev_src (b,n*k) i32:
 * their source points (order inside a list unspecified; only rev_start[b][n] entries are used).  Edges that cannot
 * contribute are left out and repeated ones merged: a self edge (idx[i][j] == i) has zero gradient; the copies of a
 * row's first entry (ball-query padding, radius-clamped kNN) become ONE edge, marked by bit 31 of its rev_src entry,
 * whose weight is rev_mult[b][i] (b,n) i32 = number of entries of row i equal to idx[i][0].  ws: (b,n) i32 scratch. */
int ogc_reverse_neighbours(int b, int n, int k, const int *idx, int *rev_start, int *rev_src, int *rev_mult, int *ws,
                           ogc_stream_t stream);

/* Gradient of ogc_neighbour_consistency_fwd w.r.t. mask, through both the centre and the neighbour operand
 * (torch's subgradients: sign(0) = 0 for p = 1, 0 where the norm vanishes for p = 2).
 * rev_start / rev_src / rev_mult from ogc_reverse_neighbours(idx); grad_out (b,n) f32, grad_mask (b,n,c) f32 out
 * (overwritten).  c <= 40. */
int ogc_neighbour_consistency_bwd(int b, int n, int c, int k, int p, const float *mask, const int *idx,
                                  const int *rev_start, const int *rev_src, const int *rev_mult, const float *grad_out,
                                  float *grad_mask, ogc_stream_t stream);

/* Soft nearest-neighbour targets of object-aware ICP, fused.  Replaces, per refinement iteration of
 *   oa_icp.py:62-72:  corr = softmax(-cdist(p1, p2) / temperature); corr *= mask1 @ mask2^T;
 *                     corr /= corr.sum(-1).clamp(1e-10); target = corr @ p2
 * (two (b, n1, n2) tensors — 1 GB each at b = 4, n = 8192 — written and re-read several times) by one pass with
 * running-maximum rescaling.  p1 (b,n1,3) warped source points, p2 (b,n2,3), mask1 (b,n1,k), mask2 (b,n2,k) with the
 * slots of mask2 already permuted to mask1's order; target (b,n1,3) out.  Distances use torch.cdist's large-matrix
 * formula (|a|^2 + |b|^2 - 2 a.b, clamped, sqrt).  k <= 32. */
int ogc_soft_nn_target(int b, int n1, int n2, int k, float temperature, const float *p1, const float *p2,
                       const float *mask1, const float *mask2, float *target, ogc_stream_t stream);

/* Attention core of the MaskFormer head's nn.MultiheadAttention layers
 *   utils/transformer_util.py:5-62 (cross-attention slots <- points, self-attention among the slots; 8 heads,
 *   models/segnet_kitti.py:49-51):   out = softmax(scale * Q K^T) V   per (sample, head), fp32.
 * q (b, lq, .), k, v (b, lk, .) are read in place from the projection outputs: row strides ldq / ldk / ldv floats, the
 * pointer is the first column of the operand inside its packed buffer, head i is the column block [i*d, (i+1)*d).
 * out (b, lq, h*d) contiguous, heads merged (what the output projection takes); prob (b, h, lq, lk) the attention
 * probabilities, kept for the backward pass.  d in {16, 32} (OGC_ERR_UNSUPPORTED otherwise); pointers 16-byte aligned,
 * strides multiples of 4.
 * bwd: dout (b, lq, h*d) -> dq / dk / dv written with row strides lddq / lddk / lddv (e.g. the column blocks of one
 * packed (b, l, 3*h*d) gradient); needs lq * lk floats of LDS per workgroup (OGC_ERR_UNSUPPORTED beyond 64 KiB). */
int ogc_attention_fwd(int b, int lq, int lk, int h, int d, float scale, const float *q, int ldq, const float *k,
                      int ldk, const float *v, int ldv, float *out, float *prob, ogc_stream_t stream);
int ogc_attention_bwd(int b, int lq, int lk, int h, int d, float scale, const float *q, int ldq, const float *k,
                      int ldk, const float *v, int ldv, const float *out, const float *prob, const float *dout,
                      float *dq, int lddq, float *dk, int lddk, float *dv, int lddv, ogc_stream_t stream);

/* nn.Linear on a few hundred rows (the decoder layers' projections and feed-forward network act on K slot embeddings
 * per sample, utils/transformer_util.py:5-62; 160 rows x 128 features at C4), one launch per direction, fp32 FMAs:
 *   fwd: y (rows, n_out) = x (rows, n_in) weight^T + bias        weight (n_out, n_in) as nn.Linear stores it; bias or NULL
 *   bwd: grad_x = grad_y weight,  grad_weight = grad_y^T x,  grad_bias = sum over the rows of grad_y
 *        — any of the three outputs may be NULL (not all); all are overwritten.
 * Any sizes are accepted; the kernels are laid out for rows <= ~1000 (32 x 32 output tiles, no split over the rows). */
int ogc_small_linear_fwd(int rows, int n_in, int n_out, const float *x, const float *weight, const float *bias,
                         float *y, ogc_stream_t stream);
int ogc_small_linear_bwd(int rows, int n_in, int n_out, const float *x, const float *weight, const float *grad_y,
                         float *grad_x, float *grad_weight, float *grad_bias, ogc_stream_t stream);

/* Backward of  y = conv(relu(GroupNorm(y_prev)))  inside a SharedMLP (utils/nn_util.py:45-85) WITHOUT the GroupNorm
 * backward passes (ogc_amd/csrc/gn_fused_bwd.hip).  pa, pb (b, cin): the GroupNorm of y_prev as an affine map per sample and
 * channel (ogc_group_norm_coeffs); mask = [pa y_prev + pb > 0] (relu != 0) or 1.
 *   ogc_conv1x1_wgrad_moments: moments (b, 2, cout, cin): H = g_y mask^T and H2 = g_y (mask . y_prev)^T per sample
 *       (overwritten).  hw % 16 == 0.
 *   ogc_gn_moments_combine: grad_w (cout, cin) = sum_b H2 diag(pa_b) + H diag(pb_b); grad_gamma, grad_beta (cin) of the
 *       GroupNorm — ONE buffer of 2 cin floats, grad_beta == grad_gamma + cin; coef (b, cin, 3) = (alpha, c2, c3) of its adjoint
 *       g_prev = alpha mask g_z + c2 y_prev + c3.  mean, rstd (b, groups), gamma (cin), w (cout, cin); cin <= 512.
 *   ogc_conv1x1_dgrad_adjoint: grad_prev (b, cin, hw) = alpha mask (w^T grad_y) + c2 y_prev + c3 — the input gradient of the
 *       convolution carried through ReLU and GroupNorm in the GEMM's epilogue.  hw % 64 == 0, cout <= 160. */
int ogc_conv1x1_wgrad_moments(int b, int cin, int cout, int hw, int relu, const float *y_prev, const float *pa,
                              const float *pb, const float *grad_y, float *moments, ogc_stream_t stream);
int ogc_gn_moments_combine(int b, int cin, int cout, int hw, int groups, const float *moments, const float *w,
                           const float *pa, const float *pb, const float *mean, const float *rstd, const float *gamma,
                           float *grad_w, float *coef, float *grad_gamma, float *grad_beta, ogc_stream_t stream);
int ogc_conv1x1_dgrad_adjoint(int b, int cin, int cout, int hw, int relu, const float *w, const float *grad_y,
                              const float *y_prev, const float *pa, const float *pb, const float *coef, float *grad_prev,
                              ogc_stream_t stream);

/* The tail of a set-abstraction MLP backwards, without the dense gradient in between:  last layer
 *   y = conv(relu(GroupNorm(y_prev)))  followed by  out = max over the neighbourhood of relu(GroupNorm'(y))
 * (utils/pointnet2_util.py:38-42).  The gradient of that pooled GroupNorm w.r.t. y is affine in y except at the arg-max
 * position of every neighbourhood:   g_y[b, ch, pr, j] = fmaf(c2, y, c3) + (j == argmax[b, ch, pr] ? ag : 0).
 *   ogc_group_norm_maxpool_bwd_sparse: ogc_group_norm_maxpool_bwd with  coef2 (b, c, 2) = (c2, c3)  and
 *       inj (b, c, p, 2) = (ag, argmax as its bit pattern)  in place of grad_x (b, c, p, s); grad_gamma / grad_beta as there.
 *       x_at_argmax (b, c, p) or NULL: x[b, c, pr, argmax] where the forward pass kept it (yext of
 *       ogc_conv1x1_gemm_affine_pool) — the sums pass then gathers from x only for channels whose scale rstd * gamma
 *       is zero (x itself is always required).
 *   ogc_conv1x1_wgrad_moments_pooled / ogc_conv1x1_dgrad_adjoint_pooled: ogc_conv1x1_wgrad_moments / _dgrad_adjoint of the
 *       convolution that wrote y, taking (y, coef2, inj) for grad_y and rebuilding it with the expression above while they
 *       load y — results identical, bit for bit, to the dense sequence; the pass that reads y and writes grad_y (the size of
 *       the layer's activation each way) is gone.  hw = centres * nsample, nsample in {16, 32, 64}. */
int ogc_group_norm_maxpool_bwd_sparse(int b, int c, int p, int s, int groups, int relu, const float *x,
                                      const float *x_at_argmax, const float *gamma, const float *mean, const float *rstd, const float *out, const int *argmax,
                                      const float *grad_out, float *coef2, float *inj, float *grad_gamma,
                                      float *grad_beta, double *ws, ogc_stream_t stream);
int ogc_conv1x1_wgrad_moments_pooled(int b, int cin, int cout, int hw, int relu, int nsample, const float *y_prev,
                                     const float *pa, const float *pb, const float *y, const float *coef2,
                                     const float *inj, float *moments, ogc_stream_t stream);
int ogc_conv1x1_dgrad_adjoint_pooled(int b, int cin, int cout, int hw, int relu, int nsample, const float *w, const float *y,
                                     const float *coef2, const float *inj, const float *y_prev, const float *pa,
                                     const float *pb, const float *coef, float *grad_prev, ogc_stream_t stream);

/* A whole per-neighbourhood MLP and its max-pool in one launch, for INFERENCE
 *   utils/flowstep3d_util.py:57-66 (FlowEmbedding, the correlation layer) and :126-138 (set abstraction):
 *     for conv, bn in zip(mlp_convs, mlp_bns): x = relu(bn(conv(x)));   x = max(x, -1)
 * with BatchNorm in evaluation mode, i.e. an affine map per output channel that the caller folds into the weights:
 *   wt_l (c_{l-1} rounded up to 4, c_l) = TRANSPOSED folded weights (rows beyond c_{l-1} zero), b_l (c_l) folded biases,
 *   x (b, c0, p, nsample) -> out (b, c_last, p) = max over nsample of relu(W3' relu(W2' relu(W1' x + b1) + b2) + b3);
 *   c3 == 0: two layers.  fp32 on v_mfma_f32_16x16x4_f32; the activations between the layers never leave the register
 *   file.  ogc_mlp_chain_pool_supported() says whether a kernel exists for a shape (1) or the caller runs its layers
 *   one by one (0). */
int ogc_mlp_chain_pool_supported(int c0, int c1, int c2, int c3, int nsample);
int ogc_mlp_chain_pool(int b, int c0, int c1, int c2, int c3, int p, int nsample, const float *x, const float *wt1,
                       const float *b1, const float *wt2, const float *b2, const float *wt3, const float *b3, float *out,
                       ogc_stream_t stream);

/* The FlowStep3D correlation layer after its neighbour search, in one launch, for INFERENCE
 *   utils/flowstep3d_util.py:53-66 (FlowEmbedding.forward): group pos2 and feat2 by idx, subtract pos1, concatenate
 *   [pos2[idx] - pos1 (3), feat2[idx] (cf), feat1 repeated (cf)] -> 3 x (conv, BatchNorm(eval), ReLU) -> max over nsample.
 * pos1 (b,3,n1), pos2 (b,3,n2), feat1 (b,cf,n1), feat2 (b,cf,n2), idx (b,n1,nsample) = the clamped neighbour indices
 * (ogc_knn_clamped); weights as for ogc_mlp_chain_pool (wt1 has 3 + 2 cf rows, padded to a multiple of 4); out (b,c3,n1).
 * The (b, 3 + 2 cf, n1, nsample) tensor is never formed: its rows are gathered while the first layer's operands load. */
int ogc_corr_layer_pool_supported(int cf, int c1, int c2, int c3, int nsample);
int ogc_corr_layer_pool(int b, int cf, int c1, int c2, int c3, int n1, int n2, int nsample, const float *pos1,
                        const float *pos2, const float *feat1, const float *feat2, const int *idx, const float *wt1,
                        const float *b1, const float *wt2, const float *b2, const float *wt3, const float *b3, float *out,
                        ogc_stream_t stream);

/* Mask read-out of the segmentation nets
 *   models/segnet_kitti.py:85-88 (segnet_sapien.py / segnet_ogcdr.py :77-80):
 *   mask = softmax_k( F.normalize(feats, dim=1)^T F.normalize(slots, dim=1) / temperature ),  temperature = 0.05.
 * feats (b, d, n) per-point features, slots (b, d, k) object embeddings, mask (b, n, k); fp32, contiguous.  The norms
 * are clamped at 1e-12 as F.normalize does.  k <= 32, d <= 256 (OGC_ERR_UNSUPPORTED otherwise).
 * bwd: grad_mask (b, n, k) -> grad_feats (b, d, n) and grad_slots (b, d, k), both overwritten; mask is the forward
 * output.  ws: scratch of ogc_slot_masks_ws_floats(b, d, n, k) floats (per-workgroup partial sums of the slot
 * gradient, added in a fixed order: the result repeats bit for bit). */
long long ogc_slot_masks_ws_floats(int b, int d, int n, int k);
int ogc_slot_masks_fwd(int b, int d, int n, int k, float temperature, const float *feats, const float *slots,
                       float *mask, ogc_stream_t stream);
int ogc_slot_masks_bwd(int b, int d, int n, int k, float temperature, const float *feats, const float *slots,
                       const float *mask, const float *grad_mask, float *grad_feats, float *grad_slots, float *ws,
                       ogc_stream_t stream);

/* Fused GroupNorm (+ ReLU) forward / backward.  Replaces the nn.GroupNorm -> ReLU(inplace) tail of every
 * Conv2d block of the segmentation nets' SharedMLPs
 *   utils/nn_util.py:6-11 (GroupNorm), :45-85 (_ConvBase ordering), models/segnet_kitti.py:8 (BN_CONFIG).
 * x, y, grad_y, grad_x (b, c, hw) f32 contiguous; gamma, beta, grad_gamma, grad_beta (c); mean, rstd (b*groups)
 * (written by fwd, read by bwd); biased variance, rstd = 1/sqrt(var + eps); relu != 0 applies max(.,0) in fwd
 * and masks grad_y where the output was <= 0 in bwd.
 * ws: caller-allocated scratch holding per-slice partial sums (every entry used is written before it is read: no
 * clearing, no atomics).  fwd: 2*b*groups*ogc_group_norm_stats_slots() doubles; bwd: 2*b*c*ogc_group_norm_bwd_slots()
 * doubles. */
int ogc_group_norm_stats_slots(void);
int ogc_group_norm_bwd_slots(void);
int ogc_group_norm_fwd(int b, int c, int hw, int groups, float eps, int relu, const float *x,
                       const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                       double *ws, ogc_stream_t stream);
int ogc_group_norm_bwd(int b, int c, int hw, int groups, int relu, const float *x, const float *gamma,
                       const float *beta, const float *mean, const float *rstd, const float *grad_y,
                       float *grad_x, float *grad_gamma, float *grad_beta, double *ws, ogc_stream_t stream);

/* GroupNorm (+ ReLU) fused with the max over the neighbourhood — the tail of a set-abstraction MLP
 *   utils/pointnet2_util.py:38-42 (SharedMLP then F.max_pool2d over nsample).
 * x (b, c, p, s) -> out (b, c, p), argmax (b, c, p) i32 (index in [0, s) of the winning neighbour; first index on
 * ties).  s must be a power of two in [4, 256] and x 16-byte aligned (OGC_ERR_UNSUPPORTED otherwise).
 * bwd: grad_out (b, c, p) -> grad_x (b, c, p, s), grad_gamma, grad_beta; scratch sizes as for ogc_group_norm_*. */
int ogc_group_norm_maxpool_fwd(int b, int c, int p, int s, int groups, float eps, int relu, const float *x,
                               const float *gamma, const float *beta, float *out, int *argmax, float *mean,
                               float *rstd, double *ws, ogc_stream_t stream);
int ogc_group_norm_maxpool_bwd(int b, int c, int p, int s, int groups, int relu, const float *x,
                               const float *gamma, const float *mean, const float *rstd, const float *out,
                               const int *argmax, const float *grad_out, float *grad_x, float *grad_gamma,
                               float *grad_beta, double *ws, ogc_stream_t stream);
/* ogc_group_norm_maxpool_bwd with x_at_argmax (b, c, p) = x[b, c, pr, argmax[b, c, pr]] handed in (same source as above). */
int ogc_group_norm_maxpool_bwd_ext(int b, int c, int p, int s, int groups, int relu, const float *x,
                                   const float *x_at_argmax, const float *gamma, const float *mean, const float *rstd,
                                   const float *out, const int *argmax, const float *grad_out, float *grad_x,
                                   float *grad_gamma, float *grad_beta, double *ws, ogc_stream_t stream);

/* Fused BatchNorm2d (+ ReLU) (+ max over the neighbourhood), forward / backward — the F.relu(bn(conv(.))) chains and
 * the trailing .max(dim=-1) of the FlowStep3D blocks
 *   utils/flowstep3d_util.py:57-66 (FlowEmbedding), :126-138 (PointNetSetAbstraction), :181-183 (feature propagation).
 * nn.BatchNorm2d semantics.  training != 0: batch statistics per channel over (b, hw), biased variance,
 * rstd = 1/sqrt(var + eps); running_mean / running_var (c), when non-NULL, are updated in place with `momentum` and the
 * unbiased variance.  training == 0: running statistics (required) are used, and the backward is the affine map's.
 * x, y, grad_y, grad_x (b, c, hw) [maxpool: x (b, c, p, s) -> out, argmax (b, c, p); s a power of two in [4, 256]];
 * gamma, beta, grad_gamma, grad_beta, mean, rstd (c) (mean / rstd written by fwd, read by bwd).
 * stats / slots: optional statistics supplied by the producing kernel (`slots` copies of a (c, 2) f64 accumulator), else
 * NULL / 0 and ws (2c f64) is used.  bwd ws: 3c f64.
 * training == 0 with mean == rstd == NULL (inference, nothing kept for a backward pass): the affine map is formed from the
 * running statistics inside the apply kernel itself — one launch. */
int ogc_batch_norm_fwd(int b, int c, int hw, float eps, int relu, int training, float momentum, const float *x,
                       const float *gamma, const float *beta, float *running_mean, float *running_var, float *y,
                       float *mean, float *rstd, double *ws, const double *stats, int slots, ogc_stream_t stream);
int ogc_batch_norm_bwd(int b, int c, int hw, int relu, int training, const float *x, const float *gamma,
                       const float *beta, const float *mean, const float *rstd, const float *grad_y, float *grad_x,
                       float *grad_gamma, float *grad_beta, double *ws, ogc_stream_t stream);
int ogc_batch_norm_maxpool_fwd(int b, int c, int p, int s, float eps, int relu, int training, float momentum,
                               const float *x, const float *gamma, const float *beta, float *running_mean,
                               float *running_var, float *out, int *argmax, float *mean, float *rstd, double *ws,
                               const double *stats, int slots, ogc_stream_t stream);
int ogc_batch_norm_maxpool_bwd(int b, int c, int p, int s, int relu, int training, const float *x, const float *gamma,
                               const float *mean, const float *rstd, const float *out, const int *argmax,
                               const float *grad_out, float *grad_x, float *grad_gamma, float *grad_beta, double *ws,
                               ogc_stream_t stream);

/* Weight gradient of a 1x1 convolution (bias-free), NCHW, fp32, on the fp32 MFMA pipe:
 *   dw[co, ci] = sum_b sum_p dy[b, co, p] * x[b, ci, p]
 * Replaces the weight-gradient half of the Conv2d(1x1) layers of SharedMLP (utils/nn_util.py:45-85, :155-172), which
 * MIOpen computes through two full NCHW->NHWC transposes.  x (b, cin, hw), dy (b, cout, hw), dw (cout, cin) — dw is
 * overwritten.  hw must be a multiple of 16 and x, dy 16-byte aligned (OGC_ERR_UNSUPPORTED otherwise). */
int ogc_conv1x1_wgrad(int b, int cin, int cout, int hw, const float *x, const float *dy, float *dw,
                      ogc_stream_t stream);

/* Forward / input gradient of the same 1x1 convolution as one fp32-MFMA GEMM over NCHW tensors:
 *   out[b, m, p] = sum_k A[m, k] * in[b, k, p]
 * forward:        transpose_a = 0, A = w (M = Cout, K = Cin),  in = x,  out = y
 * input gradient: transpose_a = 1, A = w^T (M = Cin, K = Cout), in = dy, out = dx      (w is always (Cout, Cin))
 * Requires hw % 64 == 0, K <= 160 and 16-byte aligned tensors (OGC_ERR_UNSUPPORTED otherwise -> vendor conv). */
int ogc_conv1x1_gemm(int b, int M, int K, int hw, int transpose_a, const float *w, const float *in,
                     float *out, ogc_stream_t stream);

/* The forward convolution above, also producing the first pass of the GroupNorm that follows every Conv2d of a
 * SharedMLP (utils/nn_util.py:45-85): per (batch, group) the sum and the sum of squares of the outputs, fp64.
 * stats: ogc_conv1x1_gn_slots() copies of a (b, groups, 2) f64 accumulator, slot-major (the copies spread the atomics
 * over cache lines; their sum is the statistic) — overwritten.  Pass it to ogc_group_norm_fwd_stats /
 * ogc_group_norm_maxpool_fwd_stats, which then skip their own statistics pass over the tensor.
 * Requires, beyond ogc_conv1x1_gemm: groups <= 32, (M / groups) % 4 == 0 and K <= 100 (OGC_ERR_UNSUPPORTED otherwise:
 * wider layers run the plain GEMM and the GroupNorm entry points compute their own statistics). */
int ogc_conv1x1_gn_slots(void);
/* 1 if the forward convolution can produce the next GroupNorm's statistics on the way for this shape (ogc_conv1x1_gemm_gnstats,
 * affine = 0; ogc_conv1x1_gemm_affine with groups > 0, affine = 1): K <= 100, or a shape of the streaming kernel (fp32
 * operands, 100 < K <= 160, hw a multiple of 64, b * hw / 64 >= 2048). */
int ogc_conv1x1_gemm_stats_supported(int b, int M, int K, int hw, int affine);
/* 1 when ogc_conv1x1_gemm takes this shape on its streaming kernel (K in 101 .. 160, >= 2048 position tiles, fp32 operands):
 * the shapes for which the host layers prefer it to a vendor GEMM for a plain product */
int ogc_conv1x1_gemm_stream_supported(int b, int M, int K, int hw);
/* The same product for ANY reduction length and row count (ogc_amd/csrc/gemm_chunk.hip: K walked in chunks, the weights'
 * chunk in LDS, the input chunk in registers; few positions: the wavefronts of a workgroup split the rows): the Conv1d of
 * the feature-propagation modules (utils/pointnet2_util.py:96-120 — 384 -> 128 on 1024 points ...) and the input gradient
 * of layers wider than 160 channels, which went to the vendor library before.  hw % 64 == 0; operands as
 * ogc_set_matmul_precision says (fp32 operands whatever the switch until 0.2.2). */
int ogc_conv1x1_gemm_any(int b, int M, int K, int hw, int transpose_a, const float *w, const float *in, float *out,
                         ogc_stream_t stream);

/* Operand precision of ogc_conv1x1_gemm* / ogc_conv1x1_wgrad* (process-wide; returns the previous setting).
 * 0 (default): fp32 operands on v_mfma_f32_16x16x4_f32 — exact fp32 FMA chains, the parity mode.
 * 1: operands rounded to bf16 (nearest even) on v_mfma_f32_16x16x16_bf16, fp32 accumulation, tensors in memory stay
 *    fp32 — the arithmetic torch.autocast(bfloat16) gives the reference's Conv2d layers (BASELINE config "OGC-DR ...,
 *    bf16"); GroupNorm statistics, losses and everything else stay fp32. */
int ogc_set_matmul_precision(int bf16);
int ogc_conv1x1_gemm_gnstats(int b, int M, int K, int hw, int groups, const float *w, const float *in, float *out,
                             double *stats, ogc_stream_t stream);

/* Deferred normalisation: inside a SharedMLP the GroupNorm (+ ReLU) output of one layer is only ever read by the next
 * layer's convolution, so it need not be written.  ogc_group_norm_coeffs turns the statistics of layer l (supplied by
 * the producing convolution, or computed here from x when stats == NULL; ws: 2*b*groups*ogc_group_norm_stats_slots() f64 then) into mean / rstd
 * (b*groups) for the backward pass and the per-(b, channel) affine map a, bb (b*c):  GroupNorm(x)[b, ch] = a*x + bb.
 * ogc_conv1x1_gemm_affine is the forward convolution of layer l+1 on act(a*in + bb) applied while loading (relu != 0:
 * act = ReLU; groups > 0: also the statistics of ITS output, as ogc_conv1x1_gemm_gnstats, same restrictions);
 * ogc_conv1x1_wgrad_affine is that layer's weight gradient with the same recomputed operand.  The input gradient is
 * ogc_conv1x1_gemm(transpose_a = 1) followed by ogc_group_norm_bwd on (in, mean, rstd), unchanged. */
int ogc_group_norm_coeffs(int b, int c, int hw, int groups, float eps, const float *x, const float *gamma,
                          const float *beta, const double *stats, int slots, double *ws, float *mean, float *rstd,
                          float *a, float *bb, ogc_stream_t stream);
int ogc_conv1x1_gemm_affine(int b, int M, int K, int hw, int relu, int groups, const float *w, const float *in,
                            const float *pa, const float *pb, float *out, double *stats, ogc_stream_t stream);
int ogc_conv1x1_wgrad_affine(int b, int cin, int cout, int hw, int relu, const float *x, const float *pa,
                             const float *pb, const float *dy, float *dw, ogc_stream_t stream);
/* The LAST layer of a set-abstraction MLP backwards for layers wider than the fused GroupNorm path takes (64 channels), with the
 * gradient of its pooled GroupNorm in the sparse form of ogc_group_norm_maxpool_bwd_sparse below: g_y (the size of the layer's
 * activation) is rebuilt from (y, coef2, inj) while y is loaded instead of being written by one kernel and read by two.
 *   ogc_conv1x1_wgrad_affine_pooled: ogc_conv1x1_wgrad_affine with (y, coef2, inj) for dy;
 *   ogc_conv1x1_dgrad_pooled:        grad_z[b] = w^T . g_y[b]  (ogc_amd/csrc/gemm_chunk.hip), w (cout, cin); needs >= 1024 tiles of
 *                                    64 positions (OGC_ERR_UNSUPPORTED otherwise), hw % 64 == 0.
 * nsample in {16, 32, 64} dividing hw; fp32 operands.  Reference: utils/pointnet2_util.py:38-42, utils/nn_util.py:45-85. */
int ogc_conv1x1_wgrad_affine_pooled(int b, int cin, int cout, int hw, int relu, int nsample, const float *x, const float *pa,
                                    const float *pb, const float *y, const float *coef2, const float *inj, float *dw,
                                    ogc_stream_t stream);
int ogc_conv1x1_dgrad_pooled(int b, int cin, int cout, int hw, int nsample, const float *w, const float *y, const float *coef2,
                             const float *inj, float *grad_z, ogc_stream_t stream);

/* ogc_group_norm_fwd / ogc_group_norm_maxpool_fwd with the statistics supplied (`slots` copies, see above). */
int ogc_group_norm_fwd_stats(int b, int c, int hw, int groups, float eps, int relu, const float *x,
                             const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                             const double *stats, int slots, ogc_stream_t stream);
int ogc_group_norm_maxpool_fwd_stats(int b, int c, int p, int s, int groups, float eps, int relu, const float *x,
                                     const float *gamma, const float *beta, float *out, int *argmax, float *mean,
                                     float *rstd, const double *stats, int slots, ogc_stream_t stream);

/* Max-pool of a set-abstraction MLP without re-reading its last layer's output
 *   utils/pointnet2_util.py:38-42: new_features = mlp(grouped); F.max_pool2d(new_features, [1, nsample]).
 * The last layer is conv -> GroupNorm -> ReLU, and y -> act(a*y + bb) is monotone per (batch, channel): the pooled
 * value is the activation of the LARGEST raw output of the neighbourhood when a >= 0 and of the SMALLEST when a < 0,
 * and sign(a) = sign(gamma) is known before the convolution runs.
 * ogc_conv1x1_gemm_affine_pool is ogc_conv1x1_gemm_affine (groups >= 1: statistics are always produced) whose
 * positions are (centre, neighbour) pairs, hw = centres * nsample with nsample in {16, 32, 64}; next_gamma (M) is the
 * scale of the GroupNorm that follows.  Next to out and stats it writes yext (b, M, centres) — per row the largest
 * output of each neighbourhood where next_gamma >= 0, the smallest where it is negative — and aext, the index in
 * [0, nsample) of the first neighbour attaining it.  ogc_group_norm_pool_extremes turns those and the statistics into
 * exactly what ogc_group_norm_maxpool_fwd_stats computes from the full tensor — out (b, c, p), argmax, mean, rstd —
 * so the backward pass (ogc_group_norm_maxpool_bwd) is unchanged.  (Two DIFFERENT raw values that round to the same
 * activation are told apart here and not there; with gamma == 0 both report neighbour 0.) */
int ogc_conv1x1_gemm_affine_pool(int b, int M, int K, int hw, int relu, int groups, int nsample, const float *w,
                                 const float *in, const float *pa, const float *pb, const float *next_gamma,
                                 float *out, double *stats, float *yext, int *aext, ogc_stream_t stream);
int ogc_group_norm_pool_extremes(int b, int c, int p, int s, int groups, float eps, int relu, const float *yext,
                                 const int *aext, const float *gamma, const float *beta, float *out, int *argmax,
                                 float *mean, float *rstd, const double *stats, int slots, ogc_stream_t stream);

/* The two nearest-neighbour distance terms of the Chamfer loss and their gradient (fused extension; replaces the gather /
 * difference / norm sequence of losses/flow_loss_unsup.py:24-35 around the two knn(1, ., .) searches, which stay ogc_knn calls).
 * p1 (b, n1, 3) = pc1 + flow, pc2 (b, n2, 3), idx12 (b, n1) = nearest point of pc2 for every point of p1, idx21 (b, n2) the other
 * way round; p = 1 or 2 (the reference's loss_norm).  dist1 (b, n1), dist2 (b, n2) as the reference's dist1 / dist2.
 * _grad: grad_p1 (b, n1, 3) = gradient of sum(g1 * dist1) + sum(g2 * dist2) w.r.t. p1 with the indices held constant
 * (`idx.detach()` in the reference); zeroed by the entry point, accumulated with fp32 atomics. */
int ogc_chamfer_terms(int b, int n1, int n2, int p, const float *p1, const float *pc2, const int *idx12, const int *idx21,
                      float *dist1, float *dist2, ogc_stream_t stream);
int ogc_chamfer_terms_grad(int b, int n1, int n2, int p, const float *p1, const float *pc2, const int *idx12, const int *idx21,
                           const float *g1, const float *g2, float *grad_p1, ogc_stream_t stream);

/* The element-wise glue of FlowStep3D's refinement loop in inference (fused extensions; csrc/flow_step.hip).  Each replaces a
 * run of framework operators of models/flownet_kitti.py on tensors of a few thousand elements, with the same fp32 operations in
 * the same order (one rounding each, nothing contracted) — except ogc_linear_cn and ogc_soft_corr_flow, whose dot products are
 * FMA chains in an order of their own (the reference calls addmm / normalises and calls bmm): those two agree with the reference
 * within tolerance, not to the bit:
 *   ogc_gather_xyz_pair  out (b, 3, m) = xyz (b, 3, n)[:, :, idx (b, m)] and out_t (b, m, 3) = its transpose
 *                        (gather_operation + transpose(1, 2).contiguous(), utils/flowstep3d_util.py:110-118);
 *   ogc_flow_advance     d = delta * scale (no multiplication when scale == 1), new = cur + d, flow = new - ref, all (b, 3, n);
 *                        out_delta, out_new_t (b, n, 3) and out_flow may be null (flownet_kitti.py:229-231, :245-250: scale is
 *                        the fp32 rounding of 1 / (k_decay_fact * it + 1), as torch divides by a Python scalar);
 *   ogc_linear_cn        y (b, cout, n) = weight (cout, cin) x (b, cin, n) + bias, cout <= 4 — nn.Linear between two
 *                        transposes (flownet_kitti.py:19, :38), fp32 FMA chains over the four quarters of cin, added in order; bias may be null;
 *   ogc_gru_reset        hx (b, c + cx, n) = cat([h, x]); rc (b, c, n, s) the reset gate's un-pooled convolution output, batch
 *                        stride rc_batch_stride floats: out (b, c + cx, n) = cat([sigmoid(max_s rc) * h, x]) (:147-149);
 *   ogc_gru_blend        z = sigmoid(max_s zc), q = tanh(max_s qc), out (b, c, n) = (1 - z) * h + z * q (:147, :149-150); zc, qc
 *                        (b, c, n, s) and h (b, c, n) with their batch strides in floats.
 *   ogc_soft_corr_flow   GlobalCorrLayer's dense soft correspondence and coarse flow (:53-70): w_ij = exp(-(1 - cos(f1_i, f2_j)) /
 *                        (exp(*epsilon) + 0.03)) where (|p_i|^2 + |q_j|^2) - 2 p_i.q_j < support, else 0; flow (b, 3, n1) =
 *                        sum_j w_ij q_j / (sum_j w_ij + 1e-8) - p_i.  pc1 (b, 3, n1), pc2 (b, 3, n2), f1 (b, c, n1), f2 (b, c, n2),
 *                        c a multiple of 4 up to 256; epsilon: the layer's parameter on the device (no host read).
 *   ogc_three_nn_weights weight (b, n, 3) from ogc_three_nn's squared distances dist2 (b, n, 3): r_k = 1 / max(sqrt(d2_k), 1e-10)
 *                        (mode 0, utils/flowstep3d_util.py:169-170) or 1 / (sqrt(d2_k) + 1e-8) (mode 1, utils/pointnet2_util.py:99-101),
 *                        weight_k = r_k / ((r_0 + r_1) + r_2).
 * When s is a multiple of 4: gate pointers 16-byte aligned and gate batch strides multiples of 4 floats (other s: no requirement). */
int ogc_gather_xyz_pair(int b, int n, int m, const float *xyz, const int *idx, float *out, float *out_t, ogc_stream_t stream);
int ogc_flow_advance(int b, int n, float scale, const float *cur, const float *delta, const float *ref, float *out_delta,
                     float *out_new, float *out_new_t, float *out_flow, ogc_stream_t stream);
int ogc_linear_cn(int b, int cin, int cout, int n, const float *x, const float *weight, const float *bias, float *y,
                  ogc_stream_t stream);
int ogc_three_nn_weights(int b, int n, int mode, const float *dist2, float *weight, ogc_stream_t stream);
int ogc_soft_corr_flow(int b, int n1, int n2, int c, float support, const float *epsilon, const float *pc1, const float *pc2,
                       const float *f1, const float *f2, float *flow, ogc_stream_t stream);
int ogc_gru_reset(int b, int c, int cx, int n, int s, const float *rc, long long rc_batch_stride, const float *hx, float *out,
                  ogc_stream_t stream);
int ogc_gru_blend(int b, int c, int n, int s, const float *zc, long long zc_batch_stride, const float *qc,
                  long long qc_batch_stride, const float *h, long long h_batch_stride, float *out, ogc_stream_t stream);

/* One zero fill per training step (fused extension; nothing in the reference to replace: its Python zero-fills every gradient
 * buffer it hands to the native module, pointnet2/pointnet2.py:73,181,224).  ogc_zero_arena_begin fills [base, base + bytes)
 * with zeros by ONE launch on `stream`; until ogc_zero_arena_end every operator of this library that would zero an
 * accumulator lying inside that region, on that stream, skips its own fill.  The caller hands every byte of the region to at
 * most one operator between the two calls.  bytes and base multiples of 4. */
int ogc_zero_arena_begin(void *base, int bytes, ogc_stream_t stream);
int ogc_zero_arena_end(void);

/* ---- 16-bit activations (`matmul_precision: bf16` of BASELINE config 2, verdict row g2) ------------------------------------------
 * Inside a set-abstraction MLP (reference: utils/nn_util.py:45-85 on models/segnet_ogcdr.py:26-41) the only tensors of the size
 * of an activation are the convolutions' raw outputs y and the gradients with respect to them; under bf16 operands every kernel
 * that streams one is bound by its bytes.  The `_h` entry points below are the entry points of the same name with those tensors
 * (and only those) stored as bf16 — ogc_bf16_t, round to nearest even on the way out; every argument is otherwise unchanged, and
 * so is the arithmetic: fp32 accumulation, fp64 statistics, fp32 coefficients, fp32 pooled outputs.  Statistics and
 * neighbourhood extremes a kernel produces next to a 16-bit output are those of the ROUNDED values, so that the norm that
 * follows describes exactly the tensor it reads.  They need ogc_set_matmul_precision(1) (operands are bf16 as well —
 * including the pooled forms, which keep fp32 operands for fp32 tensors) and 8-byte aligned tensors; OGC_ERR_UNSUPPORTED
 * otherwise.  ogc_conv1x1_wgrad_xf_h: x fp32 (the relative coordinates of a grouped first layer), dy bf16. */
int ogc_group_linear_fwd_h(int b, int m, int n, int npoints, int nsample, int groups, const float *P, const int *idx,
                           const float *rel, const float *wx, ogc_bf16_t *y, double *stats, ogc_stream_t stream);
/* ... with P stored point-major, Pt (b, n, m): a position's channels are contiguous and one lane fetches them with 16-byte loads
 * (a quarter of the cache-line lookups of the channel-major gather, which bound the 16-bit form).  m % 64 == 0, m / groups in
 * {16, 32, 64} (or groups == 0), an even position count; y bit-identical to ogc_group_linear_fwd_h on the transposed P. */
int ogc_group_linear_fwd_pt_h(int b, int m, int n, int npoints, int nsample, int groups, const float *Pt, const int *idx,
                              const float *rel, const float *wx, ogc_bf16_t *y, double *stats, ogc_stream_t stream);
int ogc_group_points_grad_rev_h(int b, int c, int n, int npoints, int nsample, const ogc_bf16_t *grad_out, const int *rev_start,
                                const unsigned short *rev_pos, const unsigned short *heads, float *grad_points,
                                ogc_stream_t stream);
int ogc_group_points_grad_rev_dwx_h(int b, int c, int n, int npoints, int nsample, const ogc_bf16_t *grad_out, const int *rev_start,
                                    const unsigned short *rev_pos, const unsigned short *heads, const float *rel,
                                    float *grad_points, float *dwx, ogc_stream_t stream);
int ogc_conv1x1_gemm_h(int b, int M, int K, int hw, int transpose_a, const float *w, const ogc_bf16_t *in, ogc_bf16_t *out,
                       ogc_stream_t stream);
int ogc_conv1x1_gemm_affine_h(int b, int M, int K, int hw, int relu, int groups, const float *w, const ogc_bf16_t *in,
                              const float *pa, const float *pb, ogc_bf16_t *out, double *stats, ogc_stream_t stream);
int ogc_conv1x1_gemm_affine_pool_h(int b, int M, int K, int hw, int relu, int groups, int nsample, const float *w,
                                   const ogc_bf16_t *in, const float *pa, const float *pb, const float *next_gamma,
                                   ogc_bf16_t *out, double *stats, float *yext, int *aext, ogc_stream_t stream);
int ogc_conv1x1_wgrad_xf_h(int b, int cin, int cout, int hw, const float *x, const ogc_bf16_t *dy, float *dw,
                           ogc_stream_t stream);
int ogc_conv1x1_wgrad_affine_h(int b, int cin, int cout, int hw, int relu, const ogc_bf16_t *x, const float *pa,
                               const float *pb, const ogc_bf16_t *dy, float *dw, ogc_stream_t stream);
int ogc_conv1x1_wgrad_affine_pooled_h(int b, int cin, int cout, int hw, int relu, int nsample, const ogc_bf16_t *x,
                                      const float *pa, const float *pb, const ogc_bf16_t *y, const float *coef2,
                                      const float *inj, float *dw, ogc_stream_t stream);
int ogc_conv1x1_dgrad_pooled_h(int b, int cin, int cout, int hw, int nsample, const float *w, const ogc_bf16_t *y,
                               const float *coef2, const float *inj, ogc_bf16_t *grad_z, ogc_stream_t stream);
int ogc_conv1x1_wgrad_moments_h(int b, int cin, int cout, int hw, int relu, const ogc_bf16_t *y_prev, const float *pa,
                                const float *pb, const ogc_bf16_t *grad_y, float *moments, ogc_stream_t stream);
int ogc_conv1x1_wgrad_moments_pooled_h(int b, int cin, int cout, int hw, int relu, int nsample, const ogc_bf16_t *y_prev,
                                       const float *pa, const float *pb, const ogc_bf16_t *y, const float *coef2,
                                       const float *inj, float *moments, ogc_stream_t stream);
int ogc_conv1x1_dgrad_adjoint_h(int b, int cin, int cout, int hw, int relu, const float *w, const ogc_bf16_t *grad_y,
                                const ogc_bf16_t *y_prev, const float *pa, const float *pb, const float *coef,
                                ogc_bf16_t *grad_prev, ogc_stream_t stream);
int ogc_conv1x1_dgrad_adjoint_pooled_h(int b, int cin, int cout, int hw, int relu, int nsample, const float *w,
                                       const ogc_bf16_t *y, const float *coef2, const float *inj, const ogc_bf16_t *y_prev,
                                       const float *pa, const float *pb, const float *coef, ogc_bf16_t *grad_prev,
                                       ogc_stream_t stream);
int ogc_group_norm_bwd_h(int b, int c, int hw, int groups, int relu, const ogc_bf16_t *x, const float *gamma,
                         const float *beta, const float *mean, const float *rstd, const ogc_bf16_t *grad_y,
                         ogc_bf16_t *grad_x, float *grad_gamma, float *grad_beta, double *ws, ogc_stream_t stream);
int ogc_group_norm_maxpool_bwd_sparse_h(int b, int c, int p, int s, int groups, int relu, const ogc_bf16_t *x,
                                        const float *x_at_argmax, const float *gamma, const float *mean, const float *rstd,
                                        const float *out, const int *argmax, const float *grad_out, float *coef2, float *inj,
                                        float *grad_gamma, float *grad_beta, double *ws, ogc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OGC_OPS_H */
