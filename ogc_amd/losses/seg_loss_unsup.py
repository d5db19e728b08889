"""Unsupervised OGC segmentation losses (reference: losses/seg_loss_unsup.py) on the HIP operators.

Same classes, constructor arguments, forward signatures, returned values and ``loss_dict`` keys.  What changed
is how the values are computed on the device:
  * ``fit_motion_svd_batch`` never materialises the (B*K, N, N) ``diag_embed`` (10.7 GB at B*K=40, N=8192;
    reference :36) — the weighted cross-covariance is formed as P_c^T (w * Q_c);
  * it has no host synchronisation (the reference's ``valid_batches.any()`` and boolean indexing sync);
  * kNN + radius clamp is one fused launch (``knn_radius_clamp``);
  * ``UnsupervisedOGCLoss`` gathers its monitored scalars with a single device->host copy instead of six
    ``.item()`` calls.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from ..pointnet2.pointnet2 import ball_query, grouping_operation, knn, knn_radius_clamp


def _kabsch_rotation(S, valid, eye):
    """R = V diag(1, 1, det(V U^T)) U^T for S = U diag(s) V^T (reference :44-53).  On the GPU this is one launch
    of the HIP 3x3 kernel (ogc_kabsch_rotation); otherwise torch's SVD, like the reference."""
    from ..pointnet2 import pointnet2 as _api
    fused = getattr(_api._native, "kabsch_rotation_wrapper", None)
    if fused is not None and S.is_cuda:
        S = S.detach().contiguous().float()
        R = torch.empty_like(S)
        fused(S.shape[0], S, R)
        return R
    S_safe = torch.where(valid.view(-1, 1, 1), S, eye)
    u, _, vh = torch.linalg.svd(S_safe)
    v = vh.transpose(1, 2)
    det = torch.det(v.bmm(u.transpose(1, 2)))
    diag = torch.ones_like(S[..., 0])
    diag[:, 2] = det
    return v.bmm(torch.diag_embed(diag).bmm(u.transpose(1, 2)))


def fit_motion_svd_batch(pc1, pc2, mask=None):
    """Weighted Kabsch: rigid (R, t) minimising sum_n w_n |R p_n + t - q_n|^2 per batch item.
    pc1, pc2 (B, N, 3), mask (B, N) -> R (B, 3, 3), t (B, 3).  Reference: seg_loss_unsup.py:10-61.
    Items whose covariance contains NaN (e.g. an all-zero mask) get the identity transform."""
    if mask is None:
        pc1_mean = pc1.mean(dim=1, keepdim=True)
        pc2_mean = pc2.mean(dim=1, keepdim=True)
        weighted_q = pc2 - pc2_mean
    else:
        denom = mask.sum(dim=1, keepdim=True)                                       # (B, 1)
        pc1_mean = (torch.einsum('bnd,bn->bd', pc1, mask) / denom).unsqueeze(1)     # (B, 1, 3)
        pc2_mean = (torch.einsum('bnd,bn->bd', pc2, mask) / denom).unsqueeze(1)
        weighted_q = (pc2 - pc2_mean) * mask.unsqueeze(-1)                          # diag(w) @ Q_c, row by row
    S = (pc1 - pc1_mean).transpose(1, 2).bmm(weighted_q)                            # (B, 3, 3)

    valid = ~torch.isnan(S).flatten(1).any(dim=1)                                   # (B,)
    eye = torch.eye(3, device=pc1.device, dtype=S.dtype).expand_as(S)
    R = _kabsch_rotation(S, valid, eye)
    t = pc2_mean.squeeze(1) - R.bmm(pc1_mean.transpose(1, 2)).squeeze(2)

    R = torch.where(valid.view(-1, 1, 1), R, eye)
    t = torch.where(valid.view(-1, 1), t, torch.zeros_like(t))
    return R, t


class DynamicLoss(nn.Module):
    """Per-slot rigid fit of the flow; mask-blended transformed cloud vs pc + flow. Reference: :64-98."""

    def __init__(self, loss_norm=2):
        super().__init__()
        self.loss_norm = loss_norm

    def forward(self, pc, mask, flow):
        # pc (B, N, 3), mask (B, N, K), flow (B, N, 3) -> scalar
        n_batch, n_point, n_object = mask.size()
        pc2 = pc + flow
        mask_t = mask.transpose(1, 2)                                               # (B, K, N)
        pc_rep = pc.unsqueeze(1).expand(-1, n_object, -1, -1).reshape(n_batch * n_object, n_point, 3)
        pc2_rep = pc2.unsqueeze(1).expand(-1, n_object, -1, -1).reshape(n_batch * n_object, n_point, 3)
        with torch.no_grad():  # the transformed cloud is detached in the reference (:91)
            object_R, object_t = fit_motion_svd_batch(pc_rep, pc2_rep, mask_t.reshape(n_batch * n_object, n_point))
            pc_transformed = torch.einsum('bij,bnj->bni', object_R, pc_rep) + object_t.unsqueeze(1)
            pc_transformed = pc_transformed.reshape(n_batch, n_object, n_point, 3)
        blended = (mask_t.unsqueeze(-1) * pc_transformed).sum(1)
        return (blended - pc2).norm(p=self.loss_norm, dim=-1).mean()

    def forward_views(self, pcs, masks, flows):
        """The per-view losses [forward(pc_v, mask_v, flow_v) for v] computed with ONE batched Kabsch fit over all
        views (the views are independent samples, so concatenating them along the batch changes no value)."""
        n_view, n_batch = len(pcs), pcs[0].shape[0]
        pc, mask, flow = torch.cat(pcs), torch.cat(masks), torch.cat(flows)
        n_point, n_object = mask.shape[1], mask.shape[2]
        pc2 = pc + flow
        from ..fused import rigid_residual, rigid_residual_available
        if rigid_residual_available(mask, self.loss_norm) and not (pc.requires_grad or flow.requires_grad):
            per_point = rigid_residual(pc, pc2, mask, self.loss_norm)               # (V*B, N), fused kernels
            return list(per_point.view(n_view, n_batch * n_point).mean(dim=1))
        mask_t = mask.transpose(1, 2)
        pc_rep = pc.unsqueeze(1).expand(-1, n_object, -1, -1).reshape(-1, n_point, 3)
        pc2_rep = pc2.unsqueeze(1).expand(-1, n_object, -1, -1).reshape(-1, n_point, 3)
        with torch.no_grad():
            object_R, object_t = fit_motion_svd_batch(pc_rep, pc2_rep, mask_t.reshape(-1, n_point))
            pc_transformed = torch.einsum('bij,bnj->bni', object_R, pc_rep) + object_t.unsqueeze(1)
            pc_transformed = pc_transformed.reshape(-1, n_object, n_point, 3)
        blended = (mask_t.unsqueeze(-1) * pc_transformed).sum(1)
        per_point = (blended - pc2).norm(p=self.loss_norm, dim=-1)                  # (V*B, N)
        return list(per_point.view(n_view, n_batch * n_point).mean(dim=1))


def _neighbour_consistency(mask, idx, k, cross_entropy, loss_norm):
    """mask (B, K, N) vs its values at neighbour indices idx (B, N, k). Reference: :123-129, :152-158."""
    nn_mask = grouping_operation(mask, idx.detach())
    if cross_entropy:
        target = mask.unsqueeze(3).expand(-1, -1, -1, k).detach()
        loss = F.binary_cross_entropy(nn_mask, target, reduction='none').sum(dim=1).mean(dim=-1)
    else:
        loss = (mask.unsqueeze(3) - nn_mask).norm(p=loss_norm, dim=1).mean(dim=-1)
    return loss.mean()


def _neighbour_consistency_views(mask, idx, k, cross_entropy, loss_norm, n_view):
    """Same as _neighbour_consistency for `n_view` views concatenated along the batch: one value per view."""
    nn_mask = grouping_operation(mask, idx.detach())
    if cross_entropy:
        target = mask.unsqueeze(3).expand(-1, -1, -1, k).detach()
        loss = F.binary_cross_entropy(nn_mask, target, reduction='none').sum(dim=1).mean(dim=-1)
    else:
        loss = (mask.unsqueeze(3) - nn_mask).norm(p=loss_norm, dim=1).mean(dim=-1)   # (V*B, N)
    return loss.view(n_view, -1).mean(dim=1)


class KnnLoss(nn.Module):
    """Mask smoothness over the k nearest neighbours (those beyond ``radius`` replaced by the nearest).
    Reference: :101-129."""

    def __init__(self, k, radius, cross_entropy=False, loss_norm=1, **kwargs):
        super().__init__()
        self.k = k
        self.radius = radius
        self.cross_entropy = cross_entropy
        self.loss_norm = loss_norm

    def forward(self, pc, mask):
        mask = mask.permute(0, 2, 1).contiguous()
        _, idx = knn_radius_clamp(self.k, self.radius, pc, pc)
        return _neighbour_consistency(mask, idx, self.k, self.cross_entropy, self.loss_norm)


def _shared_grid_searches(pc, k, r_knn, nsample, r_ball):
    """Both neighbour searches of the smoothness term — k nearest within r_knn (clamped), first nsample within r_ball, each of the
    clouds in themselves — on ONE cell grid built for the larger radius (ogc_cell_grid_build: each search alone sorts the clouds
    into cells first, a third to a half of its time).  Same index tensors as knn_radius_clamp / ball_query, bit for bit
    (tests/test_ops_gpu.py::test_shared_cell_grid_searches).  (None, None) where the grid does not apply."""
    from ..pointnet2 import pointnet2 as _api
    cg = getattr(_api._native, "CellGrid", None)
    if (cg is None or r_knn is None or r_ball is None or not pc.is_cuda or pc.dtype != torch.float32
            or not cg.applies(pc, max(r_knn, r_ball)) or not (r_knn > 0) or pc.shape[1] <= 4 * k or nsample > 512):
        return None, None
    grid = cg(pc, max(r_knn, r_ball))
    B, N, _ = pc.shape
    dist = torch.empty(B, N, k, dtype=torch.float32, device=pc.device)
    idx_knn = torch.empty(B, N, k, dtype=torch.int32, device=pc.device)
    grid.knn_clamped(k, r_knn, dist, idx_knn)
    idx_ball = torch.empty(B, N, nsample, dtype=torch.int32, device=pc.device)
    grid.ball_query(r_ball, nsample, idx_ball)
    return idx_knn, idx_ball


class BallQLoss(nn.Module):
    """Mask smoothness over ball-query neighbours. Reference: :132-158."""

    def __init__(self, k, radius, cross_entropy=False, loss_norm=1, **kwargs):
        super().__init__()
        self.k = k
        self.radius = radius
        self.cross_entropy = cross_entropy
        self.loss_norm = loss_norm

    def forward(self, pc, mask):
        mask = mask.permute(0, 2, 1).contiguous()
        idx = ball_query(self.radius, self.k, pc.contiguous(), pc.contiguous())
        return _neighbour_consistency(mask, idx, self.k, self.cross_entropy, self.loss_norm)


class SmoothLoss(nn.Module):
    """Reference: :161-180."""

    def __init__(self, w_knn, w_ball_q, knn_loss_params, ball_q_loss_params):
        super().__init__()
        self.knn_loss = KnnLoss(**knn_loss_params)
        self.ball_q_loss = BallQLoss(**ball_q_loss_params)
        self.w_knn = w_knn
        self.w_ball_q = w_ball_q

    def forward(self, pc, mask):
        return (self.w_knn * self.knn_loss(pc, mask)) + (self.w_ball_q * self.ball_q_loss(pc, mask))

    def plan_views(self, pcs, stacked=None):
        """Neighbour indices of all views (coordinates only; may run ahead on a side stream): ONE kNN and ONE
        ball-query launch over the concatenated views (B*V clouds fill the GPU far better than V launches of B), plus
        the transposed lists the fused gradient kernel gathers over.  stacked: the views as one (V B, N, 3) tensor, when the
        caller holds them that way (then nothing is concatenated)."""
        pc = stacked.contiguous() if stacked is not None else torch.cat(list(pcs)).contiguous()
        kl, bl = self.knn_loss, self.ball_q_loss
        idx_knn, idx_ball = _shared_grid_searches(pc, kl.k, kl.radius, bl.k, bl.radius)
        if idx_knn is None:
            _, idx_knn = knn_radius_clamp(kl.k, kl.radius, pc, pc)
            idx_ball = ball_query(bl.radius, bl.k, pc, pc)
        plan = {"knn": idx_knn, "ball": idx_ball}
        if pc.is_cuda:
            from ..fused import reverse_neighbours
            from ..pointnet2 import pointnet2 as _api
            if getattr(_api._native, "reverse_neighbours_wrapper", None) is not None:
                plan["knn_rev"] = reverse_neighbours(idx_knn)
                plan["ball_rev"] = reverse_neighbours(idx_ball)
        return plan

    def forward_views(self, pcs, masks, geometry=None):
        """[forward(pc_v, mask_v) for v], evaluated on the views concatenated along the batch."""
        from ..fused import neighbour_consistency, neighbour_consistency_available
        n_view = len(pcs)
        if geometry is None:
            geometry = self.plan_views(pcs)
        elif not isinstance(geometry, dict):  # a Pending from a side stream
            geometry = geometry.get()
        kl, bl = self.knn_loss, self.ball_q_loss
        terms = []
        mask_pm = torch.cat(list(masks))                                   # (V*B, N, K) point-major
        mask_cm = None
        for name, cfg in (("knn", kl), ("ball", bl)):
            if name + "_rev" in geometry and neighbour_consistency_available(mask_pm, cfg.loss_norm, cfg.cross_entropy):
                per_point = neighbour_consistency(mask_pm, geometry[name], geometry[name + "_rev"], cfg.loss_norm)
                terms.append(per_point.view(n_view, -1).mean(dim=1))
            else:
                if mask_cm is None:
                    mask_cm = mask_pm.permute(0, 2, 1).contiguous()
                terms.append(_neighbour_consistency_views(mask_cm, geometry[name], cfg.k, cfg.cross_entropy,
                                                          cfg.loss_norm, n_view))
        return list(self.w_knn * terms[0] + self.w_ball_q * terms[1])


def interpolate_mask_by_flow(pc1, pc2, mask1, flow1, k=1):
    """Mask of pc2 interpolated from the k nearest points of pc1 + flow1. Reference: :183-209."""
    warped_pc1 = (pc1 + flow1).contiguous()
    dist, idx = knn(k, pc2.contiguous(), warped_pc1)
    grouped = grouping_operation(mask1.transpose(1, 2).contiguous(), idx.detach())   # (B, K, N, k)
    if k == 1:
        out = grouped.squeeze(-1)
    else:
        inv = 1.0 / dist.clamp(min=1e-10)
        weight = inv / inv.sum(dim=2, keepdim=True)
        out = (weight.unsqueeze(1) * grouped).sum(dim=-1)
    return out.transpose(1, 2)


def _iou_matrix(mask1, mask2):
    """Pairwise IoU (B, K, K) of the hard (arg-max) segmentations. Reference: :220-233."""
    n_object = mask1.size(2)
    eye = torch.eye(n_object, dtype=torch.float32, device=mask1.device)
    onehot1 = eye[mask1.argmax(-1).detach()]
    onehot2 = eye[mask2.argmax(-1).detach()]
    intersection = torch.einsum('bng,bnp->bgp', onehot1, onehot2)
    union = onehot1.sum(dim=1).unsqueeze(-1) + onehot2.sum(dim=1, keepdim=True) - intersection
    return intersection / union.clamp(1e-10)


def _assign_columns(iou):
    """(..., K, K) IoUs -> (..., K) int64 column matched to each row, maximising the total IoU with scipy's
    tie-breaking.  On the GPU: one launch of ogc_lsap_maximize (no host round trip); otherwise scipy itself, as in
    the reference (:234-239)."""
    from ..pointnet2 import pointnet2 as _api
    fused = getattr(_api._native, "lsap_maximize_wrapper", None)
    K = iou.size(-1)
    if fused is not None and iou.is_cuda:
        flat = iou.detach().reshape(-1, K, K).contiguous().float()
        cols = torch.empty(flat.shape[:2], dtype=torch.int32, device=iou.device)
        fused(flat.size(0), K, flat, cols)
        return cols.view(iou.shape[:-1]).long()
    host = iou.detach().reshape(-1, K, K).cpu().numpy()
    cols = np.stack([linear_sum_assignment(m, maximize=True)[1] for m in host], 0)
    return torch.from_numpy(cols).to(iou.device).view(iou.shape[:-1])


def match_mask_by_iou(mask1, mask2):
    """Hungarian matching of the hard segmentations by IoU -> permutation matrices (B, K, K).
    Reference: :212-240 (scipy on the host, one sync per sample; here on the device)."""
    eye = torch.eye(mask1.size(2), dtype=torch.float32, device=mask1.device)
    return eye[_assign_columns(_iou_matrix(mask1, mask2))]


def match_mask_pairs_both_ways(pairs):
    """For every (mask_a, mask_b): (match_mask_by_iou(a, b), match_mask_by_iou(b, a)) — the IoU of the reverse
    direction is the transpose — solved in ONE batched launch for all pairs, samples and directions (the reference
    makes a device->host->device round trip per sample and direction)."""
    ious = torch.stack([_iou_matrix(a, b) for a, b in pairs])                 # (P, B, K, K)
    both = torch.stack([ious, ious.transpose(2, 3)], 1)                       # (P, 2, B, K, K)
    eye = torch.eye(pairs[0][0].size(2), dtype=torch.float32, device=pairs[0][0].device)
    perms = eye[_assign_columns(both)]                                        # (P, 2, B, K, K)
    return [(perms[i, 0], perms[i, 1]) for i in range(len(pairs))]


class InvarianceLoss(nn.Module):
    """Reference: :243-280."""

    def __init__(self, cross_entropy=False, loss_norm=2):
        super().__init__()
        self.cross_entropy = cross_entropy
        self.loss_norm = loss_norm

    def distance(self, mask1, mask2):
        if self.cross_entropy:
            loss = F.binary_cross_entropy(mask1, mask2, reduction='none').sum(dim=1)
        else:
            loss = (mask1 - mask2).norm(p=self.loss_norm, dim=-1)
        return loss.mean()

    def _from_perms(self, mask1, mask2, perm2, perm1):
        target_mask1 = torch.einsum('bij,bnj->bni', perm2, mask2).detach()
        target_mask2 = torch.einsum('bij,bnj->bni', perm1, mask1).detach()
        return self.distance(mask1, target_mask1) + self.distance(mask2, target_mask2)

    def forward(self, mask1, mask2):
        (perm2, perm1), = match_mask_pairs_both_ways([(mask1, mask2)])
        return self._from_perms(mask1, mask2, perm2, perm1)

    def forward_pairs(self, pairs):
        """[forward(a, b) for (a, b) in pairs]: all pairs, samples and directions matched in one batch."""
        from ..fused import matched_distance_available, matched_distances
        if matched_distance_available(pairs[0][0], self.loss_norm, self.cross_entropy):
            n_pair = len(pairs)
            d12, d21 = matched_distances(torch.cat([a for a, _ in pairs]), torch.cat([b for _, b in pairs]),
                                         self.loss_norm)                        # (P*B, N) each
            per_pair = d12.view(n_pair, -1).mean(dim=1) + d21.view(n_pair, -1).mean(dim=1)
            return list(per_pair)
        perms = match_mask_pairs_both_ways(pairs)
        return [self._from_perms(a, b, p2, p1) for (a, b), (p2, p1) in zip(pairs, perms)]


class EntropyLoss(nn.Module):
    """Reference: :283-297."""

    def forward(self, mask, epsilon=1e-5):
        return (-(mask * torch.log(mask.clamp(epsilon))).sum(dim=-1)).mean()


def _sym_eigvals(gram):
    """Eigenvalues of (B, K, K) symmetric fp64 matrices: the HIP Jacobi kernel on the GPU (torch.linalg.eigvalsh
    checks LAPACK's `info` on the host, i.e. synchronises), torch otherwise."""
    from ..pointnet2 import pointnet2 as _api
    fused = getattr(_api._native, "sym_eigvals_wrapper", None)
    if fused is not None and gram.is_cuda:
        gram = gram.contiguous()
        w = torch.empty(gram.shape[:2], dtype=torch.float64, device=gram.device)
        fused(gram.size(0), gram.size(1), gram, w)
        return w
    return torch.linalg.eigvalsh(gram)


class RankLoss(nn.Module):
    """Mean nuclear norm of the (N, K) masks. Reference: :300-314."""

    def forward(self, mask):
        # sum of singular values of the (N, K) mask = sum_i sqrt(eig_i(M^T M)); the K x K Gram matrix is formed and
        # decomposed in fp64, so the value matches an fp32 SVD of M to fp32 accuracy without a tall-matrix SVD
        # (rocSOLVER's bidiagonalisation of (B, 8192, 10) is ~300 tiny launches per step).
        m = mask.detach()
        B, N, K = m.shape
        chunk = 512 if N % 512 == 0 else N
        # fp32 products over short chunks, fp64 accumulation across chunks (an fp64 batched GEMM of this shape is ~50x
        # slower on the GPU and an fp32 sum over all N would lose digits)
        mc = m.reshape(B, N // chunk, chunk, K)
        # (bmm, not einsum('bcnk,bcnl->bckl'): the same batched product for less than half the launch-thread time)
        m3 = mc.reshape(B * (N // chunk), chunk, K)
        gram = torch.bmm(m3.transpose(1, 2), m3).view(B, N // chunk, K, K).double().sum(dim=1)
        sv = _sym_eigvals(gram).clamp_min(0).sqrt()
        return sv.sum(dim=1).mean().to(mask.dtype)


_COEFF_CACHE = {}


def _term_coefficients(n_view, aug, w_knn, w_ball, weights, device):
    """(4, L) matrix taking the vector [dynamic per view (V) | k-NN term per view (V) | ball term per view (V) | matched distances
    1->2 per pair (2) | 2->1 per pair (2)] to [dynamic, smooth, invariance, weighted sum] as the reference combines them
    (seg_loss_unsup.py:353-392: sums over the views, halved when augmented views double them).  Cached per configuration: the
    weights only change at the start steps, and building a device tensor from host numbers is a (blocking) copy."""
    key = (n_view, bool(aug), float(w_knn), float(w_ball), tuple(float(w) for w in weights), str(device))
    hit = _COEFF_CACHE.get(key)
    if hit is None:
        L = 3 * n_view + (4 if aug else 0)
        scale = 0.5 if aug else 1.0
        rows = [[0.0] * L for _ in range(4)]
        for v in range(n_view):
            rows[0][v] = scale
            rows[1][n_view + v] = scale * float(w_knn)
            rows[1][2 * n_view + v] = scale * float(w_ball)
        if aug:
            for j in range(4):
                rows[2][3 * n_view + j] = 1.0
        wd, ws, wi = (float(w) for w in weights)
        rows[3] = [wd * a + ws * b_ + wi * c for a, b_, c in zip(rows[0], rows[1], rows[2])]
        hit = torch.tensor(rows, dtype=torch.float32, device=device)
        if len(_COEFF_CACHE) > 64:
            _COEFF_CACHE.clear()
        _COEFF_CACHE[key] = hit
    return hit


class _CombineTerms(torch.autograd.Function):
    """loss = <last row of coeff, v>; its gradient w.r.t. v is that row, scaled.  (The monitored terms are the other rows —
    formed off the critical path by _monitored_terms, and NOT as a matrix product: a NaN in one term — a NaN flow makes the
    dynamic term NaN, the step is then skipped — must not leak into the others through 0 x NaN.)"""

    @staticmethod
    def forward(ctx, v, coeff):
        ctx.save_for_backward(coeff)
        return torch.dot(coeff[3], v)

    @staticmethod
    def backward(ctx, g_loss):
        coeff, = ctx.saved_tensors
        return coeff[3] * g_loss, None


LOSS_GLUE = __import__("os").environ.get("OGC_LOSS_GLUE", "1") != "0"   # the two Functions below instead of framework operators
_GLUE_PARTS = __import__("os").environ.get("OGC_LOSS_GLUE_PARTS", "means,fork").split(",")   # (bisecting)


def _glue_available(*tensors):
    from .. import pointnet2_cuda as nat
    return (LOSS_GLUE and getattr(nat, "view_means_wrapper", None) is not None
            and all(t.is_cuda and t.dtype is torch.float32 for t in tensors))


class _ViewMeansCombine(torch.autograd.Function):
    """(loss, v) with v = cat([part.view(rows, -1).mean(1) for part in parts]) and loss = <coeff[3], v> — the per-view means of
    every term and their weighted sum (:353-392) as ONE kernel and a dot product, and backward ONE kernel that writes every
    term's gradient ((coeff * g) * (1 / len), MeanBackward's own arithmetic) instead of a broadcast division per term
    (csrc/loss_glue.hip).  v is returned for the monitors and carries no gradient."""

    @staticmethod
    def forward(ctx, coeff, rows, *parts):
        from .. import pointnet2_cuda as nat
        parts = [p.contiguous() for p in parts]
        v = torch.empty(sum(rows), dtype=torch.float32, device=parts[0].device)
        nat.view_means_wrapper(parts, list(rows), v)
        ctx.save_for_backward(coeff)
        ctx.rows, ctx.shapes = list(rows), [p.shape for p in parts]
        ctx.mark_non_differentiable(v)
        return torch.dot(coeff[3], v), v

    @staticmethod
    def backward(ctx, g_loss, _g_v=None):
        from .. import pointnet2_cuda as nat
        coeff, = ctx.saved_tensors
        grads = [torch.empty(s, dtype=torch.float32, device=coeff.device) for s in ctx.shapes]
        nat.view_means_grad_wrapper(grads, ctx.rows, coeff[3], g_loss.contiguous())
        return (None, None) + tuple(grads)


class _MaskConsumers(torch.autograd.Function):
    """The stacked masks handed to their five consumers — three take all views, the invariance term the first and the second
    half — so that the five gradients are added by ONE kernel (ogc_sum_ranges) instead of four additions and two zero-filled
    embeddings of the halves."""

    @staticmethod
    def forward(ctx, mask, half):
        ctx.half_elems = half * mask[0].numel()
        ctx.shape = mask.shape
        ctx.set_materialize_grads(False)
        return mask.view_as(mask), mask.view_as(mask), mask.view_as(mask), mask[:half], mask[half:]

    @staticmethod
    def backward(ctx, *grads):
        from .. import pointnet2_cuda as nat
        firsts = [0, 0, 0, 0, ctx.half_elems]
        have = [(g.contiguous(), f) for g, f in zip(grads, firsts) if g is not None]
        if not have:
            return None, None
        out = torch.empty(ctx.shape, dtype=torch.float32, device=have[0][0].device)
        nat.sum_ranges_wrapper([g for g, _ in have], [f for _, f in have], out)
        return out, None


def _monitored_terms(v, coeff):
    """[dynamic, smooth, invariance] from the per-view vector: every row sums only the entries it has a coefficient for."""
    rows = coeff[:3]
    return torch.where(rows != 0, rows * v.detach()[None], torch.zeros((), device=v.device)).sum(dim=1)


class PendingLossDict:
    """The monitored scalars of one loss evaluation, copied to the host asynchronously (one non-blocking copy of a
    stacked tensor).  ``resolve()`` returns the reference's ``loss_dict`` of Python floats (:407-412)."""
    KEYS = ('dynamic', 'smooth', 'invariance', 'entropy', 'rank', 'sum')

    def __init__(self, terms):
        from ..utils.streams import HostScalars
        self._keys = list(terms)
        self._scalars = HostScalars(torch.stack([terms[k].detach().float().reshape(()) for k in self._keys]))
        self._dict = None

    def resolve(self):
        if self._dict is None:
            d = dict(zip(self._keys, self._scalars.get()))
            d.setdefault('invariance', 0)
            self._dict = {k: d[k] for k in self.KEYS}
        return self._dict


class UnsupervisedOGCLoss(nn.Module):
    """Weighted sum of dynamic / smooth / invariance terms (+ entropy and rank for monitoring).
    Reference: :317-409; returns ``(loss, loss_dict)`` with keys dynamic, smooth, invariance, entropy, rank, sum."""

    def __init__(self, dynamic_loss, smooth_loss, invariance_loss, entropy_loss, rank_loss,
                 weights=[10.0, 0.1, 0.1], start_steps=[0, 0, 0]):
        super().__init__()
        self.dynamic_loss = dynamic_loss
        self.smooth_loss = smooth_loss
        self.invariance_loss = invariance_loss
        self.w_dynamic, self.w_smooth, self.w_invariance = weights
        self.start_step_dynamic, self.start_step_smooth, self.start_step_invariance = start_steps
        self.entropy_loss = entropy_loss
        self.rank_loss = rank_loss

    def step_lossw(self, it, weight, start_step=0):
        return 0 if it < start_step else weight

    def plan_geometry(self, pcs, aug_transform=False):
        """Coordinate-only part of the loss (neighbour searches of the smooth term) for forward(..., geometry=...)."""
        n_view = 4 if aug_transform else 2
        if hasattr(self.smooth_loss, "plan_views"):
            return self.smooth_loss.plan_views(list(pcs)[:n_view])
        return None

    takes_stacked_views = True   # forward(..., stacked=(pc, mask, flow)): the views stacked along the batch, as ONE tensor each

    def _stacked_terms(self, stacked, n_view, geometry, aug_transform, weights):
        """The three differentiable terms from the view-major tensors (V B, N, .) — no concatenation of views, and the scalar
        algebra of the reference (per-view means -> sums over views -> weighted sum, :353-392) as ONE small matrix product
        instead of ~20 scalar kernels forward and as many backward.  None where a term has no fused kernel."""
        from ..fused import (matched_distance_available, matched_distances, neighbour_consistency,
                             neighbour_consistency_available, rigid_residual, rigid_residual_available)
        pc, mask, flow = stacked
        dl, sl, il = self.dynamic_loss, self.smooth_loss, self.invariance_loss
        kl, bl = getattr(sl, "knn_loss", None), getattr(sl, "ball_q_loss", None)
        if geometry is not None and not isinstance(geometry, dict):
            geometry = geometry.get()   # a Pending from a side stream
        if (kl is None or bl is None or not mask.is_cuda or pc.requires_grad or flow.requires_grad
                or not isinstance(dl, DynamicLoss) or not rigid_residual_available(mask, dl.loss_norm)
                or not neighbour_consistency_available(mask, kl.loss_norm, kl.cross_entropy)
                or not neighbour_consistency_available(mask, bl.loss_norm, bl.cross_entropy)
                or (aug_transform and not matched_distance_available(mask, il.loss_norm, il.cross_entropy))):
            return None
        if geometry is None:
            geometry = sl.plan_views(list(pc.view((n_view, -1) + tuple(pc.shape[1:])).unbind(0)), stacked=pc)
        if "knn_rev" not in geometry or "ball_rev" not in geometry:
            return None
        half = mask.shape[0] // 2   # pairs (view 0, view 2), (view 1, view 3): the first two views against the last two
        glue = _glue_available(mask)
        if glue and aug_transform and mask.requires_grad and "fork" in _GLUE_PARTS:
            m_dyn, m_knn, m_ball, m_lo, m_hi = _MaskConsumers.apply(mask, half)
        else:
            m_dyn = m_knn = m_ball = mask
            m_lo, m_hi = mask[:half], mask[half:]
        terms = [rigid_residual(pc, pc + flow, m_dyn, dl.loss_norm),
                 neighbour_consistency(m_knn, geometry["knn"], geometry["knn_rev"], kl.loss_norm),
                 neighbour_consistency(m_ball, geometry["ball"], geometry["ball_rev"], bl.loss_norm)]
        rows = [n_view] * 3
        if aug_transform:
            terms += list(matched_distances(m_lo, m_hi, il.loss_norm))
            rows += [2, 2]
        coeff = _term_coefficients(n_view, aug_transform, sl.w_knn, sl.w_ball_q, weights, mask.device)
        if glue and "means" in _GLUE_PARTS and all(t.numel() % r == 0 and t.numel() > 0 for t, r in zip(terms, rows)):
            loss, v = _ViewMeansCombine.apply(coeff, tuple(rows), *terms)
            return loss, v, coeff
        v = torch.cat([t.view(r, -1).mean(dim=1) for t, r in zip(terms, rows)])
        return _CombineTerms.apply(v, coeff), v, coeff

    def forward(self, pcs, masks, flows, step_w=False, it=0, aug_transform=False, geometry=None, sync=True, stacked=None):
        # pcs / masks / flows: lists of 2 (or 4 with aug_transform) tensors (B, N, 3) / (B, N, K) / (B, N, 3)
        assert len(pcs) == len(masks) == len(flows), "Inconsistent number of frames!"
        n_view = 4 if aug_transform else 2
        pcs, masks, flows = list(pcs)[:n_view], list(masks)[:n_view], list(flows)[:n_view]
        assert len(pcs) == n_view

        def weight(w, start):
            return self.step_lossw(it, weight=w, start_step=start) if step_w else w

        if stacked is not None and stacked[1].shape[0] == n_view * masks[0].shape[0]:
            fast = self._stacked_terms(stacked, n_view, geometry, aug_transform,
                                       (weight(self.w_dynamic, self.start_step_dynamic), weight(self.w_smooth, self.start_step_smooth),
                                        weight(self.w_invariance, self.start_step_invariance)))
            if fast is not None:
                loss, v, coeff = fast
                return self._with_monitors(loss, {'sum': loss}, masks, aug_transform, sync, lazy=(v, coeff))

        def total(vals):
            # the reference's association: (v1 + v2), then += (v3 + v4), then * 0.5  (:358-361)
            out = vals[0] + vals[1]
            if aug_transform:
                out = 0.5 * (out + (vals[2] + vals[3]))
            return out

        terms = {}
        if hasattr(self.dynamic_loss, "forward_views"):
            l_dynamic = total(self.dynamic_loss.forward_views(pcs, masks, flows))
        else:
            l_dynamic = total([self.dynamic_loss(p, m, f) for p, m, f in zip(pcs, masks, flows)])
        terms['dynamic'] = l_dynamic
        loss = weight(self.w_dynamic, self.start_step_dynamic) * l_dynamic

        if hasattr(self.smooth_loss, "forward_views"):
            l_smooth = total(self.smooth_loss.forward_views(pcs, masks, geometry))
        else:
            l_smooth = total([self.smooth_loss(p, m) for p, m in zip(pcs, masks)])
        terms['smooth'] = l_smooth
        loss = loss + weight(self.w_smooth, self.start_step_smooth) * l_smooth

        if aug_transform:
            if hasattr(self.invariance_loss, "forward_pairs"):
                inv_a, inv_b = self.invariance_loss.forward_pairs([(masks[0], masks[2]), (masks[1], masks[3])])
                l_invariance = inv_a + inv_b
            else:
                l_invariance = self.invariance_loss(masks[0], masks[2]) + self.invariance_loss(masks[1], masks[3])
            terms['invariance'] = l_invariance
            loss = loss + weight(self.w_invariance, self.start_step_invariance) * l_invariance

        terms['sum'] = loss
        return self._with_monitors(loss, terms, masks, aug_transform, sync)

    def _with_monitors(self, loss, terms, masks, aug_transform, sync, lazy=None):
        def total(vals):
            out = vals[0] + vals[1]
            if aug_transform:
                out = 0.5 * (out + (vals[2] + vals[3]))
            return out

        def monitors():
            with torch.no_grad():  # monitoring only (:394-405)
                if lazy is not None:   # the differentiable terms' values, from the vector the loss was formed from
                    values = _monitored_terms(*lazy)
                    terms['dynamic'], terms['smooth'] = values[0], values[1]
                    if aug_transform:
                        terms['invariance'] = values[2]
                terms['entropy'] = total([self.entropy_loss(m) for m in masks])
                terms['rank'] = total([self.rank_loss(m) for m in masks])
            return PendingLossDict(terms)

        if loss.is_cuda and not sync:
            # the monitored values feed nothing downstream: compute and ship them on a side stream so that the
            # backward pass does not queue behind ~80 small launches
            from ..utils.streams import side_stream
            side = side_stream(loss.device, "monitor")
            side.wait_stream(torch.cuda.current_stream())
            for t in list(masks) + list(terms.values()) + ([lazy[0]] if lazy is not None else []):
                t.record_stream(side)
            with torch.cuda.stream(side):
                pending = monitors()
        else:
            pending = monitors()
        return loss, (pending.resolve() if sync else pending)


class UnsupervisedOGCLossSingleFrame(UnsupervisedOGCLoss):
    """The Waymo variant (reference: train_seg_waymo.py:244-334): the trainer keeps every other view
    (`pcs[:, ::2]`, :59 — Waymo only has backward flow), so a sample is ONE frame, plus its augmented twin once
    augmentation is on.  Dynamic / smooth / entropy / rank are averaged over the one or two views; the invariance term
    pairs the frame with its twin.  Same ``loss_dict`` keys."""

    takes_stacked_views = False   # (one or two views per sample: the view-major fast path of the two-frame loss does not apply)

    def plan_geometry(self, pcs, aug_transform=False):
        n_view = 2 if aug_transform else 1
        if hasattr(self.smooth_loss, "plan_views"):
            return self.smooth_loss.plan_views(list(pcs)[:n_view])
        return None

    def forward(self, pcs, masks, flows, step_w=False, it=0, aug_transform=False, geometry=None, sync=True):
        assert len(pcs) == len(masks) == len(flows), "Inconsistent number of frames!"
        n_view = 2 if aug_transform else 1
        pcs, masks, flows = list(pcs)[:n_view], list(masks)[:n_view], list(flows)[:n_view]
        assert len(pcs) == n_view

        def weight(w, start):
            return self.step_lossw(it, weight=w, start_step=start) if step_w else w

        def total(vals):  # (v1 + v2) * 0.5 with augmentation, v1 alone without (:287-291, :300-304)
            return 0.5 * (vals[0] + vals[1]) if aug_transform else vals[0]

        terms = {}
        l_dynamic = total(self.dynamic_loss.forward_views(pcs, masks, flows))
        terms['dynamic'] = l_dynamic
        loss = weight(self.w_dynamic, self.start_step_dynamic) * l_dynamic
        l_smooth = total(self.smooth_loss.forward_views(pcs, masks, geometry))
        terms['smooth'] = l_smooth
        loss = loss + weight(self.w_smooth, self.start_step_smooth) * l_smooth
        if aug_transform:
            l_invariance = self.invariance_loss.forward_pairs([(masks[0], masks[1])])[0]
            terms['invariance'] = l_invariance
            loss = loss + weight(self.w_invariance, self.start_step_invariance) * l_invariance
        terms['sum'] = loss

        def monitors():
            with torch.no_grad():
                terms['entropy'] = total([self.entropy_loss(m) for m in masks])
                terms['rank'] = total([self.rank_loss(m) for m in masks])
            return PendingLossDict(terms)

        if loss.is_cuda and not sync:
            from ..utils.streams import side_stream
            side = side_stream(loss.device, "monitor")
            side.wait_stream(torch.cuda.current_stream())
            for t in list(masks) + list(terms.values()):
                t.record_stream(side)
            with torch.cuda.stream(side):
                pending = monitors()
        else:
            pending = monitors()
        return loss, (pending.resolve() if sync else pending)
