"""FlowStep3D building blocks (reference: utils/flowstep3d_util.py): the local correlation layer
``FlowEmbedding``, the set-abstraction flavour with optional cached FPS indices, and 3-NN feature
propagation.  Same names / constructor arguments / ``state_dict`` keys (``mlp_convs.{i}``, ``mlp_bns.{i}``)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2.pointnet2 import (GroupAll, QueryAndGroup, ball_query, furthest_point_sample, gather_operation,
                                   grouping_operation, knn, knn_radius_clamp, three_nn)


def _norm2d(channels, use_instance_norm):
    return nn.InstanceNorm2d(channels, affine=True) if use_instance_norm else nn.BatchNorm2d(channels)


class FlowEmbedding(nn.Module):
    """The correlation layer: for every point of cloud 1, its ``nsample`` nearest points of cloud 2 (clamped to
    ``radius``), features [pos2 - pos1, feat2, feat1] -> MLP -> max.  Reference: flowstep3d_util.py:7-66."""

    def __init__(self, radius, nsample, in_channel, mlp, pooling='max', corr_func='concat', knn=True,
                 use_instance_norm=False):
        super().__init__()
        self.radius = radius
        self.nsample = nsample
        self.knn = knn
        self.pooling = pooling
        self.corr_func = corr_func
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        if corr_func == 'concat':
            last_channel = in_channel * 2 + 3
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1, bias=False))
            self.mlp_bns.append(_norm2d(out_channel, use_instance_norm))
            last_channel = out_channel

    def forward(self, pos1, pos2, feature1, feature2):
        # pos1, pos2 (B, 3, N); feature1, feature2 (B, C, N) -> pos1, (B, mlp[-1], N)
        pos1_t = pos1.permute(0, 2, 1).contiguous()
        pos2_t = pos2.permute(0, 2, 1).contiguous()
        B, N, _ = pos1_t.shape
        if self.knn:
            _, idx = knn_radius_clamp(self.nsample, self.radius, pos1_t, pos2_t)     # :42-44
        else:
            # The reference's branch (:48-51) unpacks two values from ball_query, which returns one tensor:
            # it raises there too.  Kept as an explicit error instead of silently diverging.
            raise NotImplementedError("FlowEmbedding(knn=False) is dead code in the reference "
                                      "(utils/flowstep3d_util.py:48 cannot run)")

        pos_diff = grouping_operation(pos2, idx) - pos1.view(B, -1, N, 1)            # (B, 3, N, S)
        feat2_grouped = grouping_operation(feature2, idx)                            # (B, C, N, S)
        if self.corr_func == 'concat':
            feat_diff = torch.cat([feat2_grouped, feature1.view(B, -1, N, 1).expand(-1, -1, -1, self.nsample)], dim=1)
        feat1_new = torch.cat([pos_diff, feat_diff], dim=1)                          # (B, 2C+3, N, S)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            feat1_new = F.relu(bn(conv(feat1_new)))
        return pos1, feat1_new.max(dim=-1)[0]


class PointNetSetAbstraction(nn.Module):
    """Reference: flowstep3d_util.py:69-138.  ``npoint == N`` is legal: FPS then returns a permutation and the
    output features are in FPS order (callers rely on this quirk)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, return_fps=False, use_xyz=True,
                 use_act=True, act=F.relu, mean_aggr=False, use_instance_norm=False):
        super().__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.group_all = group_all
        self.use_xyz = use_xyz
        self.use_act = use_act
        self.mean_aggr = mean_aggr
        self.act = act
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = (in_channel + 3) if use_xyz else in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1, bias=False))
            self.mlp_bns.append(_norm2d(out_channel, use_instance_norm))
            last_channel = out_channel
        self.queryandgroup = GroupAll(self.use_xyz) if group_all else QueryAndGroup(radius, nsample, self.use_xyz)
        self.return_fps = return_fps

    def forward(self, xyz, points, fps_idx=None):
        # xyz (B, 3, N), points (B, D, N) -> new_xyz (B, 3, S), new_points (B, D', S) [, fps_idx (B, S)]
        xyz = xyz.contiguous()
        xyz_t = xyz.permute(0, 2, 1).contiguous()
        if (not self.group_all) and (self.npoint != -1):
            if fps_idx is None:
                fps_idx = furthest_point_sample(xyz_t, self.npoint)
            new_xyz = gather_operation(xyz, fps_idx)
        else:
            new_xyz = xyz
        new_points, _ = self.queryandgroup(xyz_t, new_xyz.transpose(2, 1).contiguous(), points)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_points = self.act(bn(conv(new_points))) if self.use_act else conv(new_points)
        new_points = new_points.mean(dim=-1) if self.mean_aggr else new_points.max(dim=-1)[0]
        if self.return_fps:
            return new_xyz, new_points, fps_idx
        return new_xyz, new_points


class PointNetFeaturePropogation(nn.Module):
    """3-NN inverse-distance upsampling (distances clamped below at 1e-10), optional skip + Conv1d/BN MLP.
    Reference: flowstep3d_util.py:141-184."""

    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.apply_mlp = mlp is not None
        last_channel = in_channel
        if self.apply_mlp:
            for out_channel in mlp:
                self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
                self.mlp_bns.append(nn.BatchNorm1d(out_channel))
                last_channel = out_channel

    def forward(self, pos1, pos2, feature1, feature2):
        # pos1 (B, 3, N) dense, pos2 (B, 3, S) sparse, feature1 (B, D1, N) or None, feature2 (B, D2, S)
        pos1_t = pos1.permute(0, 2, 1).contiguous()
        pos2_t = pos2.permute(0, 2, 1).contiguous()
        B, _, N = pos1.shape
        dists, idx = three_nn(pos1_t, pos2_t)
        weight = 1.0 / dists.clamp(min=1e-10)                                         # :169-170
        weight = weight / weight.sum(dim=-1, keepdim=True)
        interpolated = (grouping_operation(feature2, idx) * weight.view(B, 1, N, 3)).sum(dim=-1)
        feat_new = interpolated if feature1 is None else torch.cat([interpolated, feature1], dim=1)
        if self.apply_mlp:
            for conv, bn in zip(self.mlp_convs, self.mlp_bns):
                feat_new = F.relu(bn(conv(feat_new)))
        return feat_new
