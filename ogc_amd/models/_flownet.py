"""Shared implementation of the three ``FlowStep3D`` variants (reference: models/flownet_kitti.py,
models/flownet_sapien.py, models/flownet_ogcdr.py — same recurrent structure, different widths /
neighbourhood sizes / encoder depth).  Sub-module and attribute names follow the reference so that its
checkpoints (``encoder_loc.sa1.mlp_convs.0.weight``, ``global_corr_layer.epsilon``, ...) load unchanged.
"""
import torch
import torch.nn as nn

from .. import flow_glue
from ..pointnet2.pointnet2 import furthest_point_sample_chain, gather_operation
from ..utils.flowstep3d_util import (FlowEmbedding, PointNetFeaturePropogation, PointNetSetAbstraction,
                                     geometry_memo)


def joint_fps(xyz_a, xyz_b, npoints, parent_ties=None, return_ties=False, all_ties=None):
    """FPS chains of TWO clouds in one launch per level.  Sampling is per cloud, so stacking both clouds along the
    batch gives each its own indices unchanged, but the sequential rounds (one workgroup per cloud) are paid once
    instead of twice.  Every level samples from the previous level's centres, which are stored in sampling order: a
    cloud whose parent run never saw a tie needs no rounds at all from the second level on
    (furthest_point_sample_chain); `parent_ties` continues a chain started elsewhere (the clouds must then be that
    chain's last centres).  xyz_* (B, 3, N) -> ([idx level 1, idx level 2, ...] for a, the same for b[, ties])."""
    B = xyz_a.shape[0]
    level = torch.cat([xyz_a, xyz_b])                                   # (2B, 3, N)
    idx_a, idx_b, ties = [], [], parent_ties
    for npoint in npoints:
        idx, ties = furthest_point_sample_chain(level.permute(0, 2, 1).contiguous(), npoint, ties)
        if all_ties is not None:
            all_ties.append(ties)   # (2B,) of this level, for geometry_memo.note_chain
        idx_a.append(idx[:B].contiguous())
        idx_b.append(idx[B:].contiguous())
        level = gather_operation(level.contiguous(), idx)
    return (idx_a, idx_b, ties) if return_ties else (idx_a, idx_b)


def _note_levels(pc_a, pc_b, level_ties):
    """Tell the geometry memo that level i >= 1 of both clouds' pyramids came out of a joint_fps chain (level_ties[i - 1]:
    the (2B,) ties of that level, cloud a first): a set-abstraction layer sampling such a level AGAIN (npoint == n on the
    coarse levels) continues the chain instead of running all its rounds (geometry_memo.note_chain)."""
    for i, ties in enumerate(level_ties, start=1):
        if ties is None or i >= len(pc_a):
            continue
        B = pc_a[i].shape[0]
        geometry_memo.note_chain(pc_a[i], ties[:B])
        geometry_memo.note_chain(pc_b[i], ties[B:])


def _stacked(a, b):
    return None if a is None else [torch.cat([x, y]) for x, y in zip(a, b)]


def _sa(npoint, nsample, in_channel, mlp, inorm, **kw):
    return PointNetSetAbstraction(npoint=npoint, radius=None, nsample=nsample, in_channel=in_channel, mlp=mlp,
                                  group_all=False, use_instance_norm=inorm, **kw)


def _fc_channel_major(fc, x):
    """nn.Linear along the channel axis of x (B, C, N) -> (B, 3, N): between two transposed copies in the reference
    (flownet_kitti.py:19, :38), one launch on the channel-major tensor in inference."""
    if flow_glue.available(x, fc.weight, fc.bias, what="linear_cn") and fc.weight.shape[0] <= 4 and x.is_contiguous():
        return flow_glue.linear_cn(x, fc.weight, fc.bias)
    return fc(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()


class Flow0Regressor(nn.Module):
    """Reference: flownet_kitti.py:6-19."""

    def __init__(self, npoint, use_instance_norm, cfg):
        super().__init__()
        w = cfg["width"]
        self.sa1 = _sa(int(npoint / 4), cfg["reg_nsample"], w, [w, w, w], use_instance_norm)
        self.fc = torch.nn.Linear(w, 3)

    def forward(self, pc1_l_loc, corr_feats):
        _, x = self.sa1(pc1_l_loc[2], corr_feats)
        return _fc_channel_major(self.fc, x)


class FlowRegressor(nn.Module):
    """Reference: flownet_kitti.py:22-38."""

    def __init__(self, npoint, use_instance_norm, cfg):
        super().__init__()
        w = cfg["width"]
        self.sa1 = _sa(int(npoint / 4), cfg["reg_nsample"], w, [w, w, w], use_instance_norm)
        self.sa2 = _sa(int(npoint / 4), cfg["reg_nsample"], w, [w, w, w], use_instance_norm)
        self.fc = torch.nn.Linear(w, 3)

    def forward(self, pc1_l_loc, corr_feats):
        _, x = self.sa1(pc1_l_loc[2], corr_feats)
        _, x = self.sa2(pc1_l_loc[2], x)
        return _fc_channel_major(self.fc, x)


class GlobalCorrLayer(nn.Module):
    """Dense soft correlation between the coarsest levels of both clouds, then upsampling of the resulting
    coarse flow through FP/SA stages.  Reference: flownet_kitti.py:41-81 (3 levels) and
    flownet_sapien.py:41-76 (2 levels)."""

    def __init__(self, npoint, use_instance_norm, cfg):
        super().__init__()
        self.support_th = 10 ** 2  # 10 m
        self.epsilon = torch.nn.Parameter(torch.zeros(1))
        self.fp0 = PointNetFeaturePropogation(in_channel=3, mlp=[])
        stages = cfg["glob_corr_sa"]  # [(div, nsample, in_channel, mlp), ...] coarse -> fine
        self.n_stage = len(stages)
        for i, (div, nsample, cin, mlp) in enumerate(stages, start=1):
            setattr(self, "sa%d" % i, _sa(int(npoint / div), nsample, cin, list(mlp), use_instance_norm))
            setattr(self, "fp%d" % i, PointNetFeaturePropogation(in_channel=mlp[-1], mlp=[]))

    def calc_corr_mat(self, pcloud1, pcloud2, feature1, feature2):
        """Soft correspondence weights between the coarsest levels: pcloud (B, n, 3), feature (B, n, C) -> (B, n1, n2),
        w_ij = exp(-(1 - cos(f1_i, f2_j)) / (e^epsilon + 0.03)) for pairs closer than sqrt(support_th), else 0.
        Same arithmetic as the reference (flownet_kitti.py:53-65): squared distances in the expanded form
        (|p|^2 + |q|^2) - 2 p.q with the inner products from one bmm, features normalised with the 1e-8 guard."""
        temperature = self.epsilon.exp() + 0.03
        sq1 = pcloud1.square().sum(-1, keepdim=True)                       # (B, n1, 1)
        sq2 = pcloud2.square().sum(-1).unsqueeze(1)                        # (B, 1, n2)
        gram = torch.bmm(pcloud1, pcloud2.transpose(1, 2))
        within_reach = ((sq1 + sq2) - 2 * gram) < self.support_th
        unit1 = feature1 / (feature1.square().sum(-1, keepdim=True) + 1e-8).sqrt()
        unit2 = feature2 / (feature2.square().sum(-1, keepdim=True) + 1e-8).sqrt()
        cosine_distance = 1.0 - torch.bmm(unit1, unit2.transpose(1, 2))
        return torch.exp(-cosine_distance / temperature) * within_reach.to(cosine_distance.dtype)

    def forward(self, pc1_l_glob, pc2_l_glob, feats1_glob, feats2_glob):
        top = self.n_stage + 1  # index of the coarsest level in pc*_l_glob
        if (flow_glue.available(pc1_l_glob[top], pc2_l_glob[top], feats1_glob, feats2_glob, self.epsilon, what="soft_corr")
                and flow_glue.soft_corr_flow_supported(feats1_glob)):
            # inference: weights, row sums and the weighted mean of cloud 2 in one launch, on the channel-major tensors
            flow0_cm = flow_glue.soft_corr_flow(pc1_l_glob[top].contiguous(), pc2_l_glob[top].contiguous(), feats1_glob.contiguous(),
                                                feats2_glob.contiguous(), self.epsilon, float(self.support_th))
            return self._upsample(pc1_l_glob, top, flow0_cm)
        pcloud1 = pc1_l_glob[top].permute(0, 2, 1)
        pcloud2 = pc2_l_glob[top].permute(0, 2, 1)
        corr_mat = self.calc_corr_mat(pcloud1, pcloud2, feats1_glob.permute(0, 2, 1), feats2_glob.permute(0, 2, 1))
        row_sum = corr_mat.sum(-1, keepdim=True)
        flow0 = (corr_mat @ pcloud2.contiguous()) / (row_sum + 1e-8) - pcloud1.contiguous()

        return self._upsample(pc1_l_glob, top, flow0.permute(0, 2, 1).contiguous())

    def _upsample(self, pc1_l_glob, top, flow0):
        feats = self.fp0(pc1_l_glob[top - 1], pc1_l_glob[top], None, flow0)
        for i in range(1, self.n_stage + 1):
            level = top - i
            _, feats = getattr(self, "sa%d" % i)(pc1_l_glob[level], feats)
            feats = getattr(self, "fp%d" % i)(pc1_l_glob[level - 1], pc1_l_glob[level], None, feats)
        return feats


class EncoderLoc(nn.Module):
    """Two SA levels (N/2, N/4) that can re-use cached FPS indices. Reference: flownet_kitti.py:84-100."""

    def __init__(self, npoint, use_instance_norm, cfg):
        super().__init__()
        k = cfg["loc_nsample"]
        self.sa1 = _sa(int(npoint / 2), k, 3, [32, 32, 32], use_instance_norm, return_fps=True)
        self.sa2 = _sa(int(npoint / 4), k, 32, [64, 64, 64], use_instance_norm, return_fps=True)

    def forward(self, pc, feature, fps_idx=None):
        fps_idx1 = fps_idx[0] if fps_idx is not None else None
        pc_l1, feat_l1, fps_idx1 = self.sa1(pc, feature, fps_idx=fps_idx1)
        fps_idx2 = fps_idx[1] if fps_idx is not None else None
        pc_l2, feat_l2, fps_idx2 = self.sa2(pc_l1, feat_l1, fps_idx=fps_idx2)
        return [pc, pc_l1, pc_l2], feat_l2, [fps_idx1, fps_idx2]


class EncoderGlob(nn.Module):
    """Reference: flownet_kitti.py:103-118 (3 SA) / flownet_sapien.py:97-109 (2 SA)."""

    def __init__(self, npoint, use_instance_norm, cfg):
        super().__init__()
        self.n_sa = len(cfg["glob_enc"])
        for i, (div, nsample, cin, mlp) in enumerate(cfg["glob_enc"], start=1):
            setattr(self, "sa%d" % i, _sa(int(npoint / div), nsample, cin, list(mlp), use_instance_norm))

    def npoints(self):
        return [getattr(self, "sa%d" % i).npoint for i in range(1, self.n_sa + 1)]

    def forward(self, pc, feature, fps_idx=None):
        pc_l = [pc]
        for i in range(1, self.n_sa + 1):
            pc_i, feature = getattr(self, "sa%d" % i)(pc_l[-1], feature,
                                                      fps_idx=None if fps_idx is None else fps_idx[i - 1])
            pc_l.append(pc_i)
        return pc_l, feature


class H0Net(nn.Module):
    """Reference: flownet_kitti.py:121-132."""

    def __init__(self, npoint, use_instance_norm, cfg):
        super().__init__()
        w, k = cfg["width"], cfg["h0_nsample"]
        self.sa1 = _sa(int(npoint / 4), k, 64, [w, w, w], use_instance_norm)
        self.sa2 = _sa(int(npoint / 4), k, w, [w], use_instance_norm, use_act=False)

    def forward(self, pc, feature):
        _, feat_l1 = self.sa1(pc, feature)
        _, feat_l2 = self.sa2(pc, feat_l1)
        return feat_l2


class GRU(nn.Module):
    """Gated recurrent unit whose gates are set-abstraction convolutions (nsample = 4).
    Reference: flownet_kitti.py:135-151."""

    def __init__(self, npoint, hidden_dim, input_dim, use_instance_norm):
        super().__init__()
        in_ch = hidden_dim + input_dim
        for name in ("convz", "convr", "convq"):
            setattr(self, name, _sa(int(npoint / 4), 4, in_ch, [hidden_dim], use_instance_norm, use_act=False))

    def forward(self, h, x, pc):
        # x: a tensor (B, input_dim, N), or the list of tensors whose concatenation along the channels it is
        parts = list(x) if isinstance(x, (list, tuple)) else [x]
        hx = torch.cat([h] + parts, dim=1)
        if flow_glue.available(hx, *[p for m in (self.convz, self.convr, self.convq) for p in m.parameters()], what="gru"):
            # inference: the gates' max over the neighbours, their activations and the products around them in two launches
            # (sigmoid, mul, cat | sigmoid, tanh, 1 - z, two products, sum — and the three maxima — otherwise)
            # the update and reset gates read the same grouped input: one grouping and one product with both weights stacked
            from ..fused import conv1x1_inference
            c = h.shape[1]
            zr = conv1x1_inference(self.convz(pc, hx, grouped_only=True)[1],
                                   flow_glue.stacked_weight(self.convz.mlp_convs[0].weight, self.convr.mlp_convs[0].weight))
            qc = self.convq(pc, flow_glue.gru_reset(zr, hx, c, rc_channel0=c), pool=False)[1]
            return flow_glue.gru_blend(zr, qc, hx, c, zc_channel0=0)
        x = parts[0] if len(parts) == 1 else hx[:, h.shape[1]:]
        z = torch.sigmoid(self.convz(pc, hx)[1])
        r = torch.sigmoid(self.convr(pc, hx)[1])
        q = torch.tanh(self.convq(pc, torch.cat([r * h, x], dim=1))[1])
        return (1 - z) * h + z * q


class NoGRU(nn.Module):
    """Reference: flownet_kitti.py:154-163 (unused by FlowStep3D, kept for API parity)."""

    def __init__(self, npoint, hidden_dim, input_dim, use_instance_norm):
        super().__init__()
        self.conv = _sa(int(npoint / 4), 4, input_dim, [hidden_dim], use_instance_norm)

    def forward(self, x, pc):
        return self.conv(pc, x)[1]


class FlowStep3DBase(nn.Module):
    """Reference forward: models/flownet_kitti.py:209-252."""

    def __init__(self, cfg, npoint, use_instance_norm, loc_flow_nn, loc_flow_rad, k_decay_fact):
        super().__init__()
        w = cfg["width"]
        self.k_decay_fact = k_decay_fact
        self.use_instance_norm = use_instance_norm
        self.encoder_loc = EncoderLoc(npoint, use_instance_norm, cfg)
        self.encoder_glob = EncoderGlob(npoint, use_instance_norm, cfg)
        self.global_corr_layer = GlobalCorrLayer(npoint, use_instance_norm, cfg)
        self.h0_net = H0Net(npoint, use_instance_norm, cfg)
        self.flow0_regressor = Flow0Regressor(npoint, use_instance_norm, cfg)
        self.flow_regressor = FlowRegressor(npoint, use_instance_norm, cfg)
        self.local_corr_layer = FlowEmbedding(radius=loc_flow_rad, nsample=loc_flow_nn, in_channel=64, mlp=[w, w, w],
                                              pooling='max', corr_func='concat', use_instance_norm=use_instance_norm)
        self.gru = GRU(npoint, hidden_dim=w, input_dim=w + 64 + 16 + 3, use_instance_norm=use_instance_norm)
        k1, k2 = cfg["flow_conv_nsample"]
        self.flow_conv1 = _sa(int(npoint / 4), k1, 3, [32, 32, 32], use_instance_norm)
        self.flow_conv2 = _sa(int(npoint / 4), k2, 32, [16, 16, 16], use_instance_norm)
        self.flow_up_sample = PointNetFeaturePropogation(in_channel=3, mlp=[])

    def calc_glob_corr(self, pc1_loc, feats1_loc, pc2_loc, feats2_loc, parent_ties=None):
        # parent_ties: the local encoder's sampling chain, which these centres end (they are in sampling order)
        fps1 = fps2 = None
        level_ties = []
        if pc1_loc.is_cuda and pc1_loc.shape == pc2_loc.shape:
            fps1, fps2 = joint_fps(pc1_loc, pc2_loc, self.encoder_glob.npoints(), parent_ties, all_ties=level_ties)
        if fps1 is not None and self._two_clouds_per_call():
            B = pc1_loc.shape[0]
            pc_l, feats = self.encoder_glob(torch.cat([pc1_loc, pc2_loc]), torch.cat([feats1_loc, feats2_loc]),
                                            _stacked(fps1, fps2))
            pc1_l_glob, pc2_l_glob = [p[:B] for p in pc_l], [p[B:] for p in pc_l]
            feats1_glob, feats2_glob = feats[:B], feats[B:]
        else:
            pc1_l_glob, feats1_glob = self.encoder_glob(pc1_loc, feats1_loc, fps1)
            pc2_l_glob, feats2_glob = self.encoder_glob(pc2_loc, feats2_loc, fps2)
        _note_levels(pc1_l_glob, pc2_l_glob, level_ties)
        return self.global_corr_layer(pc1_l_glob, pc2_l_glob, feats1_glob, feats2_glob)

    def _two_clouds_per_call(self):
        """May an encoder see both clouds of a pair as one batch of 2B?  Every operator of a set-abstraction block works per
        cloud and per position except BatchNorm in training mode, whose statistics are those of the call (the reference
        makes one call per cloud, models/flownet_kitti.py:213-214): so in evaluation mode, or with instance norm.  The
        launches of the second cloud disappear, and at B = 1 a launch over 8192 points fills half of the chip at best."""
        return (not self.training) or self.use_instance_norm

    def calc_h0(self, feats1_loc, pc):
        return torch.tanh(self.h0_net(pc, feats1_loc))

    def get_x(self, feats1_loc_new, corr_feats, flow, pc, parts=False):
        # parts=True: the four tensors instead of their concatenation (the GRU concatenates them with its state in one go)
        _, flow_feats = self.flow_conv1(pc, flow)
        _, flow_feats = self.flow_conv2(pc, flow_feats)
        if parts:
            return [feats1_loc_new, corr_feats, flow_feats, flow]
        return torch.cat([feats1_loc_new, corr_feats, flow_feats, flow], dim=1)

    def get_x_slim(self, feats1_loc_new, corr_feats):
        return torch.cat([feats1_loc_new, corr_feats], dim=1)

    def plan_geometry_async(self, pc1, pc2, after=None):
        """The sampling chains of the local encoder for a FUTURE pair (pc*, (B, N, 3)), queued on a side stream: furthest point
        sampling is the one part of a forward pass that depends on coordinates only AND is sequential — 4096 + 2048 rounds for
        an 8192-point pair, 3.4 ms of one workgroup per cloud, at the head of the step's critical path — so a trainer that
        already holds the next batch lets it run underneath the current step's dense kernels (train_step.flow_train_step,
        `next_batch=`), as the segmentation step does (models/_segnet.py).  Returns the handle `forward(..., geometry=)` takes.
        after: an event marking the pair ready (the side stream then waits for that only)."""
        from ..utils.streams import launch_on_side, side_stream
        if not (pc1.is_cuda and pc1.shape == pc2.shape):
            return None
        stream = side_stream(pc1.device, "flownet-geometry")
        pc1.record_stream(stream)
        pc2.record_stream(stream)

        def chains():
            level_ties = []
            a, b_ = pc1.permute(0, 2, 1).contiguous(), pc2.permute(0, 2, 1).contiguous()
            idx1, idx2, ties = joint_fps(a, b_, [self.encoder_loc.sa1.npoint, self.encoder_loc.sa2.npoint], return_ties=True,
                                         all_ties=level_ties)
            return {"fps1": idx1, "fps2": idx2, "ties": ties, "level_ties": level_ties, "shape": tuple(pc1.shape)}

        return launch_on_side(stream, chains, after=after)

    def forward(self, pc1, pc2, feature1, feature2, iters=1, geometry=None):
        # pc*, feature* (B, N, 3) -> list of `iters` flow predictions, each (B, N, 3)
        # geometry: plan_geometry_async(pc1, pc2) made earlier (the same clouds: sampling depends on nothing else)
        with geometry_memo():  # FPS / kNN on unchanged coordinates are computed once per forward
            return self._forward(pc1, pc2, feature1, feature2, iters, geometry)

    def _forward(self, pc1, pc2, feature1, feature2, iters, geometry=None):
        flow_predictions = []
        planned = geometry.get() if geometry is not None else None
        if planned is not None and planned["shape"] != tuple(pc1.shape):
            planned = None
        pc1 = pc1.permute(0, 2, 1).contiguous()
        pc2 = pc2.permute(0, 2, 1).contiguous()
        feature1 = feature1.permute(0, 2, 1).contiguous()
        feature2 = feature2.permute(0, 2, 1).contiguous()

        fps_idx1 = fps_idx2 = loc_ties = None
        level_ties = []
        if planned is not None:
            fps_idx1, fps_idx2, loc_ties, level_ties = planned["fps1"], planned["fps2"], planned["ties"], planned["level_ties"]
        elif pc1.is_cuda and pc1.shape == pc2.shape:  # both sampling chains in one launch per level
            fps_idx1, fps_idx2, loc_ties = joint_fps(pc1, pc2, [self.encoder_loc.sa1.npoint, self.encoder_loc.sa2.npoint],
                                                     return_ties=True, all_ties=level_ties)
        if fps_idx1 is not None and self._two_clouds_per_call():
            B = pc1.shape[0]
            pc_l, feats, _ = self.encoder_loc(torch.cat([pc1, pc2]), torch.cat([feature1, feature2]), _stacked(fps_idx1, fps_idx2))
            pc1_l_loc, pc2_l_loc = [pc1] + [p[:B] for p in pc_l[1:]], [pc2] + [p[B:] for p in pc_l[1:]]
            feats1_loc, feats2_loc = feats[:B], feats[B:]
        else:
            pc1_l_loc, feats1_loc, fps_idx1 = self.encoder_loc(pc1, feature1, fps_idx1)
            pc2_l_loc, feats2_loc, _ = self.encoder_loc(pc2, feature2, fps_idx2)
        _note_levels(pc1_l_loc, pc2_l_loc, level_ties)

        corr_feats = self.calc_glob_corr(pc1_l_loc[-1], feats1_loc, pc2_l_loc[-1], feats2_loc, loc_ties)
        flow0_lr = self.flow0_regressor(pc1_l_loc, corr_feats)
        flow0 = self.flow_up_sample(pc1_l_loc[0], pc1_l_loc[2], None, flow0_lr)
        flow_predictions.append(flow0.permute(0, 2, 1))

        h = self.calc_h0(feats1_loc, pc1_l_loc[-1])

        if iters > 1 and flow_glue.available(pc1, flow0, flow0_lr, what="advance"):
            return self._refine_inference(flow_predictions, pc1, pc1_l_loc, pc2_l_loc, feats2_loc, fps_idx1, flow0, flow0_lr, h, iters)
        pc1_new = pc1 + flow0.detach()
        pc1_new_lr = pc1_l_loc[2] + flow0_lr.detach()
        for it in range(iters - 1):
            pc1_new = pc1_new.detach()
            pc1_new_lr = pc1_new_lr.detach()
            flow_lr = pc1_new_lr - pc1_l_loc[2]

            pc1_new_l_loc, feats1_loc_new, _ = self.encoder_loc(pc1_new, pc1_new, fps_idx1)
            _, corr_feats = self.local_corr_layer(pc1_new_l_loc[-1], pc2_l_loc[-1], feats1_loc_new, feats2_loc)

            x = self.get_x(feats1_loc_new, corr_feats, flow_lr, pc=pc1_l_loc[2], parts=True)
            h = self.gru(h=h, x=x, pc=pc1_l_loc[-1])
            delta_flow_lr = self.flow_regressor(pc1_l_loc, h) / (self.k_decay_fact * it + 1)
            pc1_new_lr = pc1_new_lr + delta_flow_lr

            delta_flow = self.flow_up_sample(pc1_l_loc[0], pc1_l_loc[2], None, delta_flow_lr)
            pc1_new = pc1_new + delta_flow
            flow_predictions.append((pc1_new - pc1).permute(0, 2, 1))
        return flow_predictions

    def _refine_inference(self, flow_predictions, pc1, pc1_l_loc, pc2_l_loc, feats2_loc, fps_idx1, flow0, flow0_lr, h, iters):
        """The refinement loop above (flownet_kitti.py:229-250) when nothing is differentiated: the same operators on the same values,
        with the sums / differences / scaling of the two running clouds as one launch each (flow_glue.flow_advance), which also
        leaves the (B, N, 3) copy of the moved cloud for the encoder of the next iteration."""
        # pc1 + flow0 (and its transpose), pc1_l_loc[2] + flow0_lr and the first low-resolution flow, (that sum) - pc1_l_loc[2]
        _, pc1_new, pc1_new_t, _ = flow_glue.flow_advance(pc1, flow0, None, want_t=True, want_flow=False)
        _, pc1_new_lr, _, flow_lr = flow_glue.flow_advance(pc1_l_loc[2], flow0_lr, pc1_l_loc[2])
        for it in range(iters - 1):
            geometry_memo.note_transposed(pc1_new, pc1_new_t)
            pc1_new_l_loc, feats1_loc_new, _ = self.encoder_loc(pc1_new, pc1_new, fps_idx1)
            _, corr_feats = self.local_corr_layer(pc1_new_l_loc[-1], pc2_l_loc[-1], feats1_loc_new, feats2_loc)

            x = self.get_x(feats1_loc_new, corr_feats, flow_lr, pc=pc1_l_loc[2], parts=True)
            h = self.gru(h=h, x=x, pc=pc1_l_loc[-1])
            delta_flow_lr, pc1_new_lr, _, flow_lr = flow_glue.flow_advance(
                pc1_new_lr, self.flow_regressor(pc1_l_loc, h), pc1_l_loc[2], divisor=self.k_decay_fact * it + 1, want_delta=True)

            delta_flow = self.flow_up_sample(pc1_l_loc[0], pc1_l_loc[2], None, delta_flow_lr)
            last = it == iters - 2
            _, pc1_new, pc1_new_t, flow = flow_glue.flow_advance(pc1_new, delta_flow, pc1, want_t=not last)
            flow_predictions.append(flow.permute(0, 2, 1))
        return flow_predictions
