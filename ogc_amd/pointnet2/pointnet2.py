"""Operator API of the PointNet++ stack — same public surface as the reference's
``pointnet2/pointnet2.py`` (seven ``autograd.Function`` classes with their ``.apply`` aliases,
``gather_nd``, ``QueryAndGroup``, ``GroupAll``), so ``from pointnet2.pointnet2 import *`` callers
(reference: utils/pointnet2_util.py:5, utils/flowstep3d_util.py:4, losses/seg_loss_unsup.py:7,
losses/flow_loss_unsup.py:4) work unchanged.

Differences from the reference, none of them visible in results:
  * outputs are allocated on the input's device (``torch.empty(..., device=x.device)``) instead of
    the hard-coded ``torch.cuda.FloatTensor`` (reference pointnet2.py:32-33,61,99-100,...);
  * the native module is ``ogc_amd.pointnet2_cuda`` (HIP kernels behind a C ABI) — there is no CPU
    implementation in the product; tests may substitute ``_native`` with the CPU oracle;
  * ``QueryAndGroup`` fuses kNN + sqrt + radius clamp into one launch (no boolean-mask host sync,
    reference pointnet2.py:283-286).
"""
from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import pointnet2_cuda as _native  # the drop-in for the reference's `pointnet2_cuda`

__all__ = [
    "gather_nd",
    "FurthestPointSampling", "furthest_point_sample",
    "GatherOperation", "gather_operation",
    "KNN", "knn",
    "ThreeNN", "three_nn",
    "ThreeInterpolate", "three_interpolate",
    "GroupingOperation", "grouping_operation",
    "BallQuery", "ball_query",
    "QueryAndGroup", "GroupAll",
    "knn_radius_clamp",
]


def _new(ref: torch.Tensor, shape, dtype, fill=None) -> torch.Tensor:
    t = torch.empty(shape, dtype=dtype, device=ref.device)
    if fill is not None:
        t.fill_(fill)
    return t


def gather_nd(points: torch.Tensor, idx: torch.Tensor, t=False):
    """Index rows (t=False: points (B,N,C), idx (B,M)) or columns (t=True: points (B,C,N)).
    Reference: pointnet2.py:10-14."""
    if t:
        return points.gather(2, idx.unsqueeze(1).expand(-1, points.size(1), -1))
    return points.gather(1, idx.unsqueeze(2).expand(-1, -1, points.size(2)))


class FurthestPointSampling(Function):
    """Reference: pointnet2.py:17-42 -> furthest_point_sampling_wrapper."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        # xyz (B, N, 3) -> (B, npoint) int32; first index is 0; temp starts at 1e10 (pointnet2.py:33)
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = _new(xyz, (B, npoint), torch.int32)
        temp = _new(xyz, (B, N), torch.float32, fill=1e10)
        _native.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


def furthest_point_sample_chain(xyz: torch.Tensor, npoint: int, parent_ties=None):
    """furthest_point_sample for a level of a sampling chain (not in the reference): -> (idx (B, npoint) int32, ties).
    `parent_ties`: the `ties` returned for the level that produced `xyz` — `xyz` must be that level's sampled centres in
    sampling order.  Samples whose parent run had a unique farthest point in each of its first `npoint` rounds need no
    rounds at all here (the answer is 0..npoint-1, see include/ogc_ops.h); the others are sampled as usual, so the indices are always those of
    furthest_point_sample.  Without the native entry point this is furthest_point_sample and ties is None."""
    chain = getattr(_native, "furthest_point_sampling_chain_wrapper", None)
    if chain is None or not xyz.is_cuda:
        return furthest_point_sample(xyz, npoint), None
    assert xyz.is_contiguous()
    B, N, _ = xyz.size()
    output = _new(xyz, (B, npoint), torch.int32)
    # (the running minima: the kernels keep them in registers and nobody reads them afterwards — no buffer, no fill launch;
    # above 16384 points they live in memory)
    temp = _new(xyz, (B, N), torch.float32, fill=1e10) if N > 16384 else None
    ties = torch.empty(B, dtype=torch.int32, device=xyz.device)
    chain(B, N, npoint, xyz, temp, output, parent_ties, ties)
    return output, ties


class GatherOperation(Function):
    """Reference: pointnet2.py:45-78 -> gather_points_wrapper / gather_points_grad_wrapper."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        # features (B, C, N), idx (B, npoint) -> (B, C, npoint)
        assert features.is_contiguous()
        assert idx.is_contiguous()
        B, npoint = idx.size()
        _, C, N = features.size()
        output = _new(features, (B, C, npoint), torch.float32)
        _native.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.for_backwards = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_features = _new(grad_out, (B, C, N), torch.float32, fill=0.0)
        _native.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class KNN(Function):
    """Reference: pointnet2.py:81-109 -> knn_wrapper.  Returns (sqrt(dist2), idx)."""

    @staticmethod
    def forward(ctx, k: int, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        # unknown (B, N, 3), known (B, M, 3) -> dist (B, N, k) ascending L2, idx (B, N, k) int32
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = _new(unknown, (B, N, k), torch.float32)
        idx = _new(unknown, (B, N, k), torch.int32)
        ctx.mark_non_differentiable(idx)
        fused = getattr(_native, "knn_clamped_wrapper", None)
        if fused is not None and unknown.is_cuda:
            # the square root inside the search kernel (correctly rounded, like CUDA's sqrtf the reference's torch.sqrt
            # runs; torch.sqrt on ROCm is not, so the distances would differ from the reference's in the last bit)
            fused(B, N, m, k, -1.0, unknown, known, dist2, idx)
            return dist2, idx
        _native.knn_wrapper(B, N, m, k, unknown, known, dist2, idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


knn = KNN.apply


class ThreeNN(Function):
    """Reference: pointnet2.py:112-140 -> three_nn_wrapper.  Returns (sqrt(dist2), idx)."""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous()
        assert known.is_contiguous()
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = _new(unknown, (B, N, 3), torch.float32)
        idx = _new(unknown, (B, N, 3), torch.int32)
        _native.three_nn_wrapper(B, N, m, unknown, known, dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """Reference: pointnet2.py:143-187 -> three_interpolate_wrapper / ..._grad_wrapper."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor, rev=None) -> torch.Tensor:
        # features (B, C, M), idx (B, n, 3), weight (B, n, 3) -> (B, C, n)
        # rev (optional, not in the reference): fused.group_reverse(idx, M) — the gradient becomes a gather
        assert features.is_contiguous()
        assert idx.is_contiguous()
        assert weight.is_contiguous()
        B, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        ctx.rev = rev
        output = _new(features, (B, c, n), torch.float32)
        _native.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.three_interpolate_for_backward
        B, c, n = grad_out.size()
        rev = getattr(ctx, "rev", None)
        if rev is not None and getattr(_native, "three_interpolate_grad_rev_wrapper", None) is not None:
            grad_features = _new(grad_out, (B, c, m), torch.float32)
            sliced = getattr(_native, "three_interpolate_grad_rev_sliced_wrapper", None)
            if (sliced is not None and B > 1 and not grad_out.is_contiguous() and grad_out.dtype is torch.float32
                    and grad_out.stride(2) == 1 and grad_out.stride(1) == n and grad_out.stride(0) >= c * n):
                # the gradient of cat([interpolated, skip features]) arrives as a channel slice: read where it lies
                sliced(B, c, n, m, grad_out, weight, rev[0], rev[1], rev[2], grad_features)
            else:
                _native.three_interpolate_grad_rev_wrapper(B, c, n, m, grad_out.contiguous(), weight, rev[0], rev[1], rev[2],
                                                           grad_features)
            return grad_features, None, None, None
        grad_features = _new(grad_out, (B, c, m), torch.float32, fill=0.0)
        _native.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """Reference: pointnet2.py:190-230 -> group_points_wrapper / group_points_grad_wrapper."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        # features (B, C, N), idx (B, npoint, nsample) -> (B, C, npoint, nsample)
        assert features.is_contiguous()
        assert idx.is_contiguous()
        idx = idx.int()  # pointnet2.py:203
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = _new(features, (B, C, nfeatures, nsample), torch.float32)
        _native.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.for_backwards = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = _new(grad_out, (B, C, N), torch.float32, fill=0.0)
        _native.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class GroupConcat(Function):
    """cat([group(xyz^T, idx) - new_xyz^T[..., None], group(features, idx)], dim=1) written once (ogc_group_concat);
    the op sequence of QueryAndGroup.forward (reference pointnet2.py:284-296).  Coordinates carry no gradient."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx):
        B, N, _ = xyz.size()
        _, npoint, nsample = idx.size()
        C = features.size(1)
        out = _new(features, (B, 3 + C, npoint, nsample), torch.float32)
        _native.group_concat_wrapper(B, C, N, npoint, nsample, xyz.contiguous(), new_xyz.contiguous(),
                                     features.contiguous(), idx, out)
        ctx.for_backwards = (idx, N, C)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, N, C = ctx.for_backwards
        B, _, npoint, nsample = grad_out.size()
        grad_features = _new(grad_out, (B, C, N), torch.float32, fill=0.0)
        _native.group_concat_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return None, None, grad_features, None


class BallQuery(Function):
    """Reference: pointnet2.py:233-260 -> ball_query_wrapper (note the (B, N, npoint) arg order)."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        # xyz (B, N, 3) candidates, new_xyz (B, npoint, 3) centres -> idx (B, npoint, nsample) int32
        assert new_xyz.is_contiguous()
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        # the reference pre-zeroes idx (pointnet2.py:251) because its kernel leaves rows without a hit untouched; the
        # HIP kernels write every row themselves (empty rows as zeros), so on the GPU the 4 * npoint * nsample byte
        # fill (32 MB and a launch per call at C4) is skipped
        writes_all = getattr(_native, "BALL_QUERY_WRITES_ALL_ROWS", False) and xyz.is_cuda and B * npoint * nsample > 0
        idx = _new(xyz, (B, npoint, nsample), torch.int32, fill=None if writes_all else 0)
        _native.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


def knn_radius_clamp(k: int, radius, unknown: torch.Tensor, known: torch.Tensor):
    """``dist, idx = knn(k, unknown, known)`` followed by the clamp every reference caller applies:
    ``idx[dist > radius] = idx[:, :, 0]`` (pointnet2.py:283-286, flowstep3d_util.py:42-44,
    seg_loss_unsup.py:120-122, flow_loss_unsup.py:57-59); ``radius=None`` skips the clamp.
    One fused launch when the native module offers it, otherwise knn + torch.where."""
    unknown = unknown.contiguous()
    known = known.contiguous()
    fused = getattr(_native, "knn_clamped_wrapper", None)
    if fused is not None:
        B, N, _ = unknown.size()
        dist = _new(unknown, (B, N, k), torch.float32)
        idx = _new(unknown, (B, N, k), torch.int32)
        fused(B, N, known.size(1), k, -1.0 if radius is None else float(radius), unknown, known, dist, idx)
        return dist, idx
    dist, idx = knn(k, unknown, known)
    if radius is not None:
        idx = torch.where(dist > radius, idx[:, :, :1], idx)
    return dist, idx


class QueryAndGroup(nn.Module):
    """kNN grouping with radius clamp + centre subtraction + feature concat.
    Reference: pointnet2.py:263-301 (ball_query there is commented out in favour of kNN)."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None, neighbours=None,
                idx=None):
        # xyz (B, N, 3), new_xyz (B, npoint, 3), features (B, C, N)
        # -> new_features (B, 3 + C, npoint, nsample), grouped_xyz (B, 3, npoint, nsample)
        # `neighbours` (optional, not in the reference): an un-clamped (dist, idx) = knn(nsample, new_xyz, xyz) shared
        # by several groupers of a multi-scale level, which differ only in the radius of the clamp.
        # `idx` (optional, not in the reference): the clamped neighbour indices themselves, from a geometry plan.
        if idx is not None:
            pass
        elif neighbours is None:
            _, idx = knn_radius_clamp(self.nsample, self.radius, new_xyz, xyz)
        else:
            dist, idx = neighbours
            if self.radius is not None:
                idx = torch.where(dist > self.radius, idx[:, :, :1], idx)
        if (features is not None and self.use_xyz and getattr(_native, "group_concat_wrapper", None) is not None
                and not xyz.requires_grad and not new_xyz.requires_grad):
            new_features = GroupConcat.apply(xyz, new_xyz, features, idx.int().contiguous())
            return new_features, new_features[:, :3]
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx) - new_xyz.transpose(1, 2).unsqueeze(-1)

        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features, grouped_xyz


class GroupAll(nn.Module):
    """Reference: pointnet2.py:304-327."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        # xyz (B, N, 3), features (B, C, N) -> (B, C + 3, 1, N); new_xyz is ignored
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz, grouped_xyz
        grouped_features = features.unsqueeze(2)
        if self.use_xyz:
            return torch.cat([grouped_xyz, grouped_features], dim=1), grouped_xyz
        return grouped_features, grouped_xyz
