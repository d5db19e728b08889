"""FlowStep3D building blocks (reference: utils/flowstep3d_util.py): the local correlation layer
``FlowEmbedding``, the set-abstraction flavour with optional cached FPS indices, and 3-NN feature
propagation.  Same names / constructor arguments / ``state_dict`` keys (``mlp_convs.{i}``, ``mlp_bns.{i}``)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..pointnet2.pointnet2 import (GroupAll, QueryAndGroup, ball_query, furthest_point_sample, furthest_point_sample_chain,
                                   gather_operation, grouping_operation, knn, knn_radius_clamp, three_interpolate, three_nn)


class geometry_memo:
    """Scope in which set-abstraction layers share coordinate-only work.  FlowStep3D applies ten different SA layers
    to the SAME coordinates (`pc1_l_loc[2]`) in every refinement iteration, and every one of them re-runs a full FPS
    (2048 sequential rounds) and a kNN on identical inputs (reference: models/flownet_kitti.py:231-250, SURVEY.md
    §3.2).  FPS(xyz, npoint) and kNN(k, new_xyz, xyz) are pure functions of the coordinates, and the k nearest of a
    larger search are a prefix of it, so inside this scope they are computed once per distinct coordinate tensor.
    Keys hold a reference to the tensor, so a cached entry can never alias a recycled allocation."""
    _active = None

    def __enter__(self):
        self._prev = geometry_memo._active
        geometry_memo._active = {}
        return self

    def __exit__(self, *exc):
        geometry_memo._active = self._prev

    @staticmethod
    def entry(xyz):
        memo = geometry_memo._active
        if memo is None:
            return None
        key = (xyz.data_ptr(), xyz._version, tuple(xyz.shape))
        hit = memo.get(key)
        if hit is None or hit["ref"] is not xyz:
            hit = {"ref": xyz, "xyz_t": None, "fps": {}, "new_xyz": {}, "knn": {}, "three_nn": {}, "ties": None}
            memo[key] = hit
        return hit

    @staticmethod
    def transposed(xyz):
        """xyz (B, 3, N) -> (B, N, 3) contiguous, once per coordinate tensor inside a scope."""
        hit = geometry_memo.entry(xyz)
        if hit is None:
            return xyz.permute(0, 2, 1).contiguous()
        if hit["xyz_t"] is None:
            hit["xyz_t"] = xyz.permute(0, 2, 1).contiguous()
        return hit["xyz_t"]

    @staticmethod
    def note_transposed(xyz, xyz_t):
        """The producer of `xyz` (B, 3, N) already holds its (B, N, 3) copy: the consumers (the next set-abstraction layer, the
        correlation layer) find it instead of making their own."""
        hit = geometry_memo.entry(xyz)
        if hit is not None and hit["xyz_t"] is None:
            hit["xyz_t"] = xyz_t

    @staticmethod
    def note_chain(xyz, ties):
        """`xyz` (B, 3, n) holds the centres of a level of an FPS chain IN SAMPLING ORDER and `ties` (B,) int32 is what
        furthest_point_sample_chain returned for that level.  A set-abstraction layer that samples this cloud again — the
        reference does so with npoint == n on the coarse levels (models/flownet_kitti.py:16, :30, :127, :144: a full run
        of n sequential rounds whose answer is a permutation) — then continues the chain: the rounds the parent run decided
        without a tie are not run again (the answer is 0, 1, ... for them, ogc_furthest_point_sampling_chain)."""
        if ties is None:
            return
        hit = geometry_memo.entry(xyz)
        if hit is not None:
            hit["ties"] = ties.contiguous()


def _shared_mlp(x, convs, norms, pool):
    """[relu(norm(conv(x))) for every layer] (+ max over the last dimension): the MLP tail of every FlowStep3D block
    (reference flowstep3d_util.py:64-66, :134-136, :181-183), through the fused conv / BatchNorm kernels on the GPU."""
    from ..fused import conv_norm_act, mlp_chain_pool, mlp_chain_pool_available
    n = len(convs)
    if pool and mlp_chain_pool_available(x, convs, norms):
        return mlp_chain_pool(x, convs, norms)   # inference: the whole chain and the max in one launch
    for i, (conv, norm) in enumerate(zip(convs, norms)):
        x = conv_norm_act(x, conv, norm, relu=True, maxpool=pool and i == n - 1)
    if pool and n == 0:
        x = x.max(dim=-1)[0]
    return x


class _FoldAware(nn.Module):
    """Blocks whose evaluation-mode path runs on weights folded ahead of time (fused.mlp_chain_pool / corr_layer_pool): entering
    training mode invalidates what was folded (fused.note_training_mode)."""

    def train(self, mode=True):
        if mode:
            from ..fused import note_training_mode
            note_training_mode()
        return super().train(mode)


def _norm2d(channels, use_instance_norm):
    return nn.InstanceNorm2d(channels, affine=True) if use_instance_norm else nn.BatchNorm2d(channels)


class FlowEmbedding(_FoldAware):
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
        pos1_t = geometry_memo.transposed(pos1)   # (the layer that made pos1 left its transpose in the memo; pos2 never moves)
        pos2_t = geometry_memo.transposed(pos2)
        B, N, _ = pos1_t.shape
        if self.knn:
            _, idx = knn_radius_clamp(self.nsample, self.radius, pos1_t, pos2_t)     # :42-44
        else:
            # The reference's branch (:48-51) unpacks two values from ball_query, which returns one tensor:
            # it raises there too.  Kept as an explicit error instead of silently diverging.
            raise NotImplementedError("FlowEmbedding(knn=False) is dead code in the reference "
                                      "(utils/flowstep3d_util.py:48 cannot run)")

        from ..fused import corr_layer_pool, corr_layer_pool_available
        if self.corr_func == 'concat' and corr_layer_pool_available(feature1, feature2, idx, self.mlp_convs, self.mlp_bns):
            # inference: grouping, concatenation, the three layers and the max in one launch
            return pos1, corr_layer_pool(pos1, pos2, feature1, feature2, idx, self.mlp_convs, self.mlp_bns)
        pos_diff = grouping_operation(pos2, idx) - pos1.view(B, -1, N, 1)            # (B, 3, N, S)
        feat2_grouped = grouping_operation(feature2, idx)                            # (B, C, N, S)
        if self.corr_func == 'concat':
            feat_diff = torch.cat([feat2_grouped, feature1.view(B, -1, N, 1).expand(-1, -1, -1, self.nsample)], dim=1)
        feat1_new = torch.cat([pos_diff, feat_diff], dim=1)                          # (B, 2C+3, N, S)
        return pos1, _shared_mlp(feat1_new, self.mlp_convs, self.mlp_bns, pool=True)


class PointNetSetAbstraction(_FoldAware):
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

    def forward(self, xyz, points, fps_idx=None, pool=True, grouped_only=False):
        # xyz (B, 3, N), points (B, D, N) -> new_xyz (B, 3, S), new_points (B, D', S) [, fps_idx (B, S)]
        # pool=False (blocks without activation only): new_points stay (B, D', S, nsample), the max over the neighbours is the
        # caller's (the GRU folds it into its gate kernels, flow_glue.gru_reset / gru_blend).  grouped_only=True: stop in front of
        # the MLP and return (new_xyz, the grouped input (B, 3 + D, S, nsample)) — two blocks on the same input share it
        from .. import flow_glue
        assert pool or not (self.use_act or self.mean_aggr), "pool=False is for the blocks without activation"
        xyz = xyz.contiguous()
        memo = geometry_memo.entry(xyz)
        if memo is not None and memo["xyz_t"] is not None:
            xyz_t = memo["xyz_t"]
        else:
            xyz_t = xyz.permute(0, 2, 1).contiguous()
            if memo is not None:
                memo["xyz_t"] = xyz_t
        sampled = (not self.group_all) and (self.npoint != -1)
        given_idx = fps_idx is not None
        if sampled:
            if fps_idx is None:
                if memo is not None and self.npoint in memo["fps"]:
                    fps_idx = memo["fps"][self.npoint]
                else:
                    if memo is not None and memo["ties"] is not None and xyz_t.is_cuda:
                        fps_idx, _ = furthest_point_sample_chain(xyz_t, self.npoint, memo["ties"])
                    else:
                        fps_idx = furthest_point_sample(xyz_t, self.npoint)
                    if memo is not None:
                        memo["fps"][self.npoint] = fps_idx
            new_xyz_t = None
            if memo is not None and not given_idx and self.npoint in memo["new_xyz"]:
                new_xyz = memo["new_xyz"][self.npoint]
            else:
                if flow_glue.available(xyz, what="gather_pair"):
                    new_xyz, new_xyz_t = flow_glue.gather_xyz_pair(xyz, fps_idx)   # both layouts of the centres in one launch
                else:
                    new_xyz = gather_operation(xyz, fps_idx)
                if memo is not None and not given_idx:
                    memo["new_xyz"][self.npoint] = new_xyz
                    if new_xyz_t is not None:
                        memo.setdefault("new_xyz_t", {})[self.npoint] = new_xyz_t
        else:
            new_xyz = xyz
        # the centres as (B, S, 3), once per call (and once per forward for coordinates the memo knows)
        if not sampled:
            new_xyz_t = xyz_t
        elif memo is not None and not given_idx and self.npoint in memo.setdefault("new_xyz_t", {}):
            new_xyz_t = memo["new_xyz_t"][self.npoint]
        else:
            if new_xyz_t is None:
                new_xyz_t = new_xyz.transpose(2, 1).contiguous()
            if memo is not None and not given_idx:
                memo["new_xyz_t"][self.npoint] = new_xyz_t
        if sampled:
            geometry_memo.note_transposed(new_xyz, new_xyz_t)   # whoever takes new_xyz as ITS coordinates finds the transpose
        neighbours = None
        if memo is not None and sampled and not given_idx and isinstance(self.queryandgroup, QueryAndGroup):
            # un-clamped kNN shared between layers: the k nearest are a prefix of any larger search on the same inputs
            k = self.nsample
            cached = memo["knn"].get(self.npoint)
            if cached is None or cached[1].shape[2] < k:
                kk = min(max(k, 32), xyz_t.shape[1])
                cached = knn_radius_clamp(max(kk, k), None, new_xyz_t, xyz_t)
                memo["knn"][self.npoint] = cached
            # (the prefix as its own contiguous pair, once per row length: the recurrent blocks ask for the same few lengths on
            # the same coordinates in every iteration — two copy launches per layer call otherwise)
            prefix = memo["knn"].get((self.npoint, k))
            if prefix is None or prefix[2] is not cached[1]:
                prefix = (cached[0][:, :, :k].contiguous(), cached[1][:, :, :k].contiguous(), cached[1])
                memo["knn"][(self.npoint, k)] = prefix
            neighbours = prefix[:2]
        if neighbours is not None:
            new_points, _ = self.queryandgroup(xyz_t, new_xyz_t, points, neighbours=neighbours)
        else:
            new_points, _ = self.queryandgroup(xyz_t, new_xyz_t, points)
        if grouped_only:
            return new_xyz, new_points
        if self.use_act and self.act is F.relu:
            new_points = _shared_mlp(new_points, self.mlp_convs, self.mlp_bns, pool=not self.mean_aggr)
            if self.mean_aggr:
                new_points = new_points.mean(dim=-1)
        else:
            from ..fused import pointwise_conv
            for conv, bn in zip(self.mlp_convs, self.mlp_bns):
                new_points = self.act(bn(pointwise_conv(new_points, conv))) if self.use_act else \
                    pointwise_conv(new_points, conv)
            if not pool:
                pass
            elif self.mean_aggr:
                new_points = new_points.mean(dim=-1)
            elif new_points.requires_grad:
                new_points = new_points.max(dim=-1)[0]      # (its backward sends the gradient to ONE arg-max, as the reference's)
            else:
                new_points = torch.amax(new_points, dim=-1)  # inference: the same values without the index half of the reduction
        if self.return_fps:
            return new_xyz, new_points, fps_idx
        return new_xyz, new_points


class PointNetFeaturePropogation(_FoldAware):
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
        # the three neighbours and their weights depend on the coordinates only: FlowStep3D upsamples the coarse flow from
        # the same centres to the same points in every iteration (models/flownet_kitti.py:224, :249)
        memo = geometry_memo.entry(pos1)
        hit = memo["three_nn"].get(id(pos2)) if memo is not None else None
        if hit is None or hit[0] is not pos2 or hit[3] != (pos2.data_ptr(), pos2._version):
            from .. import flow_glue
            if flow_glue.available(pos1, pos2, what="three_nn_w"):
                # inference: the transposed copies from the memo, sqrt / clamp / reciprocal / sum / division as one launch
                idx, weight = flow_glue.three_nn_with_weights(geometry_memo.transposed(pos1), geometry_memo.transposed(pos2))
            else:
                dists, idx = three_nn(pos1.permute(0, 2, 1).contiguous(), pos2.permute(0, 2, 1).contiguous())
                weight = 1.0 / dists.clamp(min=1e-10)                                     # :169-170
                weight = (weight / weight.sum(dim=-1, keepdim=True)).contiguous()
            hit = (pos2, idx, weight, (pos2.data_ptr(), pos2._version))
            if memo is not None:
                memo["three_nn"][id(pos2)] = hit
        _, idx, weight, _ = hit
        # sum_k w_k f[idx_k] (:171): one kernel, evaluated as (w0 f0 + w1 f1) + w2 f2 without contraction
        interpolated = three_interpolate(feature2.contiguous(), idx, weight)
        feat_new = interpolated if feature1 is None else torch.cat([interpolated, feature1], dim=1)
        if self.apply_mlp:
            feat_new = _shared_mlp(feat_new, self.mlp_convs, self.mlp_bns, pool=False)
        return feat_new
