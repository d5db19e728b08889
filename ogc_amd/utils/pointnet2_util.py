"""PointNet++ set-abstraction / feature-propagation modules used by the segmentation nets
(reference: utils/pointnet2_util.py).  Same class names, constructor arguments, forward signatures and
``state_dict`` layout (``groupers``, ``mlps.{i}.layer{j}...``, ``mlp.layer{j}...``)."""
import torch
import torch.nn as nn

from ..pointnet2.pointnet2 import (GroupAll, QueryAndGroup, furthest_point_sample, gather_nd, knn_radius_clamp,
                                   three_interpolate, three_nn)
from .nn_util import SharedMLP


class _PointnetSAModuleBase(nn.Module):
    """FPS -> gather centres -> per scale [group -> shared MLP -> max over the neighbourhood] -> concat.
    Reference: pointnet2_util.py:9-49."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz, features=None, return_inds=False):
        # xyz (B, N, 3), features (B, C, N) -> new_xyz (B, npoint, 3), new_features (B, sum(mlp[-1]), npoint)
        new_xyz, new_inds = None, None
        if self.npoint is not None:
            new_inds = furthest_point_sample(xyz, self.npoint).long()
            new_xyz = gather_nd(xyz, new_inds)  # == gather on the transposed cloud, transposed back (:22-27)

        pooled = []
        shared = {}  # nsample -> un-clamped kNN, computed once per level (the scales only differ in the clamp radius)
        n_same = {}
        for grouper in self.groupers:
            if isinstance(grouper, QueryAndGroup):
                n_same[grouper.nsample] = n_same.get(grouper.nsample, 0) + 1
        for grouper, mlp in zip(self.groupers, self.mlps):
            if isinstance(grouper, QueryAndGroup) and n_same[grouper.nsample] > 1:
                if grouper.nsample not in shared:
                    shared[grouper.nsample] = knn_radius_clamp(grouper.nsample, None, new_xyz, xyz)
                grouped = grouper(xyz, new_xyz, features, neighbours=shared[grouper.nsample])[0]
            else:
                grouped = grouper(xyz, new_xyz, features)[0]  # (B, C', npoint, nsample)
            pooled.append(mlp.forward_maxpool(grouped))       # shared MLP, then max over nsample (:38-42)
        new_features = torch.cat(pooled, dim=1)
        if return_inds:
            return new_xyz, new_features, new_inds
        return new_xyz, new_features


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale grouping.  NOTE: like the reference (pointnet2_util.py:69-71) the first entry of every
    ``mlps[i]`` list is incremented IN PLACE by 3 when ``use_xyz`` — callers pass fresh literals."""

    def __init__(self, npoint, radii, nsamples, mlps, bn, use_xyz=True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(QueryAndGroup(radius, nsample, use_xyz=use_xyz) if npoint is not None
                                 else GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(SharedMLP(mlp_spec, bn=bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale SA (npoint = radius = nsample = None gives group-all). Reference: pointnet2_util.py:76-88."""

    def __init__(self, mlp, npoint, radius, nsample, bn, use_xyz=True):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], bn=bn, use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """3-NN inverse-distance interpolation + skip concat + shared MLP. Reference: pointnet2_util.py:91-120."""

    def __init__(self, mlp, bn):
        super().__init__()
        self.mlp = SharedMLP(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        # unknown (B, n, 3), known (B, m, 3), unknow_feats (B, C1, n), known_feats (B, C2, m) -> (B, mlp[-1], n)
        if known is not None:
            dist, idx = three_nn(unknown.contiguous(), known.contiguous())
            dist_recip = 1.0 / (dist + 1e-8)                                  # :99
            weight = dist_recip / dist_recip.sum(dim=2, keepdim=True)         # :100-101
            interpolated = three_interpolate(known_feats.contiguous(), idx.contiguous(), weight.contiguous())
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            interpolated = torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(interpolated.unsqueeze(-1)).squeeze(-1)
