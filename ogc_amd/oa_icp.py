"""Object-aware ICP: scene-flow refinement from object masks (reference: oa_icp.py:16-84).

``weighted_kabsch`` and ``object_aware_icp`` keep the reference's signatures and results.  The per-slot clouds
are broadcast views instead of K-fold ``repeat`` copies; on the GPU the soft correspondence step of every iteration
is one fused kernel (no (B, N, N) tensors); the round driver (oa_icp.py:87-236, dataset I/O) is outside the hot path.
"""
import torch

from .losses.seg_loss_unsup import fit_motion_svd_batch, interpolate_mask_by_flow, match_mask_by_iou


def _rigid_flow_fused(pc, target, mask):
    """_rigid_flow_per_object(pc, target - pc, mask^T) as five launches of the loss's rigid-fit kernels (fused.rigid_residual's:
    fp64 moments in one pass, Kabsch, translation) and one blend that returns the vector sum_k m_k (R_k p + t_k) - p
    (ogc_rigid_blend with p = 0) — instead of ~25 framework launches on K-fold expanded clouds (two of them 0.3-ms batched
    3 x 3 products: 0.6 ms of the 0.8 ms an iteration took at B = 4, N = 8192 once the soft-NN step ran on the matrix cores).
    pc, target (B, N, 3) contiguous fp32; mask (B, N, K) contiguous."""
    from .pointnet2 import pointnet2 as _api
    from .utils.zero_arena import zeroed_empty
    nat = _api._native
    B, N, K = mask.shape
    dev = mask.device
    mom = zeroed_empty(B * K * 16, torch.float64, dev)
    S = torch.empty(B * K, 3, 3, dtype=torch.float32, device=dev)
    means = torch.empty(B * K, 6, dtype=torch.float32, device=dev)
    nat.rigid_moments_wrapper(B, N, K, pc, target, mask, mom, S, means)
    R = torch.empty_like(S)
    valid = torch.empty(B * K, dtype=torch.int32, device=dev)
    nat.kabsch_rotation_wrapper(B * K, S, R, valid)
    t = torch.empty(B * K, 3, dtype=torch.float32, device=dev)
    nat.rigid_translation_wrapper(B * K, means, valid, R, t)
    out = torch.empty(B, N, 3, dtype=torch.float32, device=dev)
    nat.rigid_blend_wrapper(B, N, K, 0, 0, pc, pc, mask, R, t, None, out)
    return out


def _rigid_flow_per_object(pc, flow, mask_t):
    """Fit one rigid motion per (sample, slot) to ``flow`` and return the mask-blended rigid flow.
    pc, flow (B, N, 3); mask_t (B, K, N) -> (B, N, 3).  Reference: oa_icp.py:24-38 / :75-83."""
    n_batch, n_object, n_point = mask_t.shape
    pc_rep = pc.unsqueeze(1).expand(-1, n_object, -1, -1).reshape(n_batch * n_object, n_point, 3)
    flow_rep = flow.unsqueeze(1).expand(-1, n_object, -1, -1).reshape(n_batch * n_object, n_point, 3)
    object_R, object_t = fit_motion_svd_batch(pc_rep, pc_rep + flow_rep, mask_t.reshape(n_batch * n_object, n_point))
    pc_transformed = torch.einsum('bij,bnj->bni', object_R, pc_rep) + object_t.unsqueeze(1)
    pc_transformed = pc_transformed.reshape(n_batch, n_object, n_point, 3)
    return torch.einsum('bkn,bkni->bni', mask_t, pc_transformed) - pc


def weighted_kabsch(pc, flow, mask):
    """pc, flow (B, N, 3), mask (B, N, K) -> object-wise rigid flow (B, N, 3). Reference: oa_icp.py:16-38."""
    return _rigid_flow_per_object(pc, flow, mask.transpose(1, 2))


def object_aware_icp(pc1, pc2, flow, mask1, mask2, icp_iter=10, temperature=0.01):
    """Alternate soft nearest-neighbour correspondences (restricted to consistent objects) and per-object
    rigid fits.  pc1, pc2, flow (B, N, 3); mask1, mask2 (B, N, K) -> refined flow (B, N, 3).
    Reference: oa_icp.py:41-84."""
    # align the slot order of frame 2 to frame 1
    mask2_interpolated = interpolate_mask_by_flow(pc1, pc2, mask1, flow)
    perm = match_mask_by_iou(mask2_interpolated, mask2)
    mask2 = torch.einsum('bij,bnj->bni', perm, mask2)

    mask1_t = mask1.transpose(1, 2)
    from .pointnet2 import pointnet2 as _api
    fused = getattr(_api._native, "soft_nn_target_wrapper", None)
    if (fused is not None and pc1.is_cuda and mask1.shape[2] <= 32
            and not any(t.requires_grad for t in (pc1, pc2, flow, mask1, mask2))):
        # one pass per iteration, nothing of size (B, N1, N2) is materialised (ogc_soft_nn_target)
        B, N1, K = mask1.shape
        N2 = pc2.shape[1]
        pc2c, m1c, m2c = pc2.contiguous().float(), mask1.contiguous().float(), mask2.contiguous().float()
        target = torch.empty(B, N1, 3, dtype=torch.float32, device=pc1.device)
        fit = all(getattr(_api._native, n, None) is not None for n in
                  ("rigid_moments_wrapper", "kabsch_rotation_wrapper", "rigid_translation_wrapper", "rigid_blend_wrapper"))
        fit = fit and K <= 32 and pc1.dtype == torch.float32
        pc1c = pc1.contiguous()
        for _ in range(icp_iter):
            fused(B, N1, N2, K, temperature, (pc1 + flow).contiguous().float(), pc2c, m1c, m2c, target)
            flow = _rigid_flow_fused(pc1c, target, m1c) if fit else _rigid_flow_per_object(pc1, target - pc1, mask1_t)
        return flow

    consistency12 = torch.einsum('bmk,bnk->bmn', mask1, mask2)   # object-consistency scores (B, N1, N2)
    for _ in range(icp_iter):
        corr12 = (-torch.cdist(pc1 + flow, pc2) / temperature).softmax(-1)
        corr12 = corr12 * consistency12
        corr12 = corr12 / corr12.sum(-1, keepdim=True).clamp(1e-10)
        flow = torch.einsum('bmn,bnj->bmj', corr12, pc2) - pc1
        flow = _rigid_flow_per_object(pc1, flow, mask1_t)
    return flow
