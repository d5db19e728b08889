"""PointNet++ set-abstraction / feature-propagation modules used by the segmentation nets
(reference: utils/pointnet2_util.py).  Same class names, constructor arguments, forward signatures and
``state_dict`` layout (``groupers``, ``mlps.{i}.layer{j}...``, ``mlp.layer{j}...``)."""
import torch
import torch.nn as nn

from ..pointnet2.pointnet2 import (GroupAll, QueryAndGroup, furthest_point_sample, furthest_point_sample_chain, gather_nd,
                                   knn_radius_clamp,
                                   three_interpolate, three_nn)
from .nn_util import SharedMLP


# Levels 2+ of an encoder re-sample the previous level's centres; when that level's rounds were free of exact fp32 ties
# the result is known in advance (DESIGN.md "FPS along a chain of levels") and the rounds are skipped.  bench.py turns
# this off for a few extra steps to report what a step costs when every level has to run all its rounds (clouds with
# duplicated points).
FPS_CHAIN_SHORTCUT = True


class _PointnetSAModuleBase(nn.Module):
    """FPS -> gather centres -> per scale [group -> shared MLP -> max over the neighbourhood] -> concat.
    Reference: pointnet2_util.py:9-49."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def plan_geometry(self, xyz, parent_ties=None):
        """Everything of this level that depends on coordinates only: FPS indices, the sampled centres and the
        un-clamped kNN of every neighbourhood size in use (the scales of a multi-scale level differ only in the
        radius of the clamp, so one search serves them all).  May be evaluated ahead of time on a side stream."""
        if self.npoint is None:
            return None
        plan = self.plan_sampling(xyz, parent_ties)
        plan["idx"] = self.plan_neighbours(xyz, plan["new_xyz"])
        plan["rev"] = self.plan_reverse(plan["idx"], xyz.shape[1])
        return plan

    def plan_reverse(self, idx, n):
        """Transposed neighbour lists (which (centre, slot) positions point at each source point) of every scale, for the
        gather form of the grouping gradient; only where a gradient will be asked for."""
        from .. import fused as _fused
        group_reverse = _fused.group_reverse
        if not torch.is_grad_enabled():  # (not self.training: a plan made ahead may precede the switch to train())
            return [None] * len(idx)
        want = lambda j: j is not None and j.is_cuda and j.shape[1] * j.shape[2] >= _fused.GROUP_GRAD_GATHER_MIN_FANIN * n
        return [group_reverse(j.int().contiguous(), n) if want(j) else None for j in idx]

    def plan_sampling(self, xyz, parent_ties=None):
        """The sequential part: FPS indices and the sampled centres (:22-27).  parent_ties: `ties` of the plan whose
        `new_xyz` this `xyz` is (the previous level of the encoder) — lets tie-free clouds skip the sampling rounds."""
        new_inds, ties = furthest_point_sample_chain(xyz, self.npoint, parent_ties if FPS_CHAIN_SHORTCUT else None)
        new_inds = new_inds.long()
        new_xyz = gather_nd(xyz, new_inds)  # == gather on the transposed cloud, transposed back (:22-27)
        return {"new_inds": new_inds, "new_xyz": new_xyz, "ties": ties}

    def plan_neighbours(self, xyz, new_xyz):
        """Neighbour lists of every scale around the sampled centres."""
        knn, idx = {}, []
        # one search per neighbourhood size, clamped at the LARGEST radius of the scales that share it: what lies beyond
        # is replaced by the nearest neighbour in every one of them, and a radius-limited search stops early
        reach = {}
        for grouper in self.groupers:
            if isinstance(grouper, QueryAndGroup):
                r = reach.get(grouper.nsample, 0.0)
                reach[grouper.nsample] = None if (r is None or grouper.radius is None) else max(r, grouper.radius)
        for grouper in self.groupers:
            if not isinstance(grouper, QueryAndGroup):
                idx.append(None)
                continue
            if grouper.nsample not in knn:
                knn[grouper.nsample] = knn_radius_clamp(grouper.nsample, reach[grouper.nsample], new_xyz, xyz)
            dist, nn_idx = knn[grouper.nsample]
            if grouper.radius is not None:  # the clamp of pointnet2.py:283-286, per scale
                nn_idx = torch.where(dist > grouper.radius, nn_idx[:, :, :1], nn_idx)
            idx.append(nn_idx.contiguous())
        return idx

    def forward(self, xyz, features=None, return_inds=False, geometry=None):
        # xyz (B, N, 3), features (B, C, N) -> new_xyz (B, npoint, 3), new_features (B, sum(mlp[-1]), npoint)
        # `geometry` (optional, not in the reference): the result of plan_geometry(xyz), possibly still pending on a
        # side stream.
        if geometry is None:
            geometry = self.plan_geometry(xyz)
        elif hasattr(geometry, "get"):
            geometry = geometry.get()
        if geometry is not None and "idx" not in geometry:  # only the sampling was planned ahead
            geometry = dict(geometry, idx=self.plan_neighbours(xyz, geometry["new_xyz"]))
        if geometry is not None and "rev" not in geometry:
            geometry = dict(geometry, rev=self.plan_reverse(geometry["idx"], xyz.shape[1]))
        new_xyz = geometry["new_xyz"] if geometry is not None else None
        new_inds = geometry["new_inds"] if geometry is not None else None

        from ..fused import first_level_plain, grouped_first_layer, grouped_first_layer_available
        pooled = []
        for i, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            if isinstance(grouper, QueryAndGroup):
                nn_idx = geometry["idx"][i]
                head = mlp.first_layer() if (grouper.use_xyz and xyz.is_cuda and not first_level_plain(features)) else None
                if head is not None and grouped_first_layer_available(xyz, new_xyz, features, nn_idx, head[0], head[1]):
                    # grouping + first convolution without the (B, 3 + C, npoint, nsample) tensor in between
                    pooled.append(mlp.forward_maxpool(None, first=lambda conv, gn, j=nn_idx, rv=geometry["rev"][i]:
                                                      grouped_first_layer(xyz, new_xyz, features, j, conv, gn, rv,
                                                                          act16=len(mlp) >= 2)))
                    continue
                grouped = grouper(xyz, new_xyz, features, idx=nn_idx)[0]
            else:
                grouped = grouper(xyz, new_xyz, features)[0]  # (B, C', npoint, nsample)
            pooled.append(mlp.forward_maxpool(grouped))       # shared MLP, then max over nsample (:38-42)
        new_features = torch.cat(pooled, dim=1) if len(pooled) != 1 else pooled[0]   # (one scale: nothing to join, no copy)
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

    @staticmethod
    def plan_geometry(unknown, known):
        """3-NN indices and inverse-distance weights (coordinates only; may run ahead on a side stream)."""
        dist, idx = three_nn(unknown.contiguous(), known.contiguous())
        dist_recip = 1.0 / (dist + 1e-8)                                  # :99
        weight = dist_recip / dist_recip.sum(dim=2, keepdim=True)         # :100-101
        idx, weight = idx.contiguous(), weight.contiguous()
        rev = None
        if torch.is_grad_enabled() and idx.is_cuda and (3 * idx.shape[1]) % 16 == 0:
            from ..fused import group_reverse
            rev = group_reverse(idx.int() if idx.dtype != torch.int32 else idx, known.shape[1])
        return {"idx": idx, "weight": weight, "rev": rev}

    def forward(self, unknown, known, unknow_feats, known_feats, geometry=None):
        # unknown (B, n, 3), known (B, m, 3), unknow_feats (B, C1, n), known_feats (B, C2, m) -> (B, mlp[-1], n)
        if known is not None:
            if geometry is None:
                geometry = self.plan_geometry(unknown, known)
            elif hasattr(geometry, "get"):
                geometry = geometry.get()
            interpolated = three_interpolate(known_feats.contiguous(), geometry["idx"], geometry["weight"],
                                             geometry.get("rev"))
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            interpolated = torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(interpolated.unsqueeze(-1)).squeeze(-1)
