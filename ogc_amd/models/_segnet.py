"""Shared implementation of the three ``MaskFormer3D`` variants (reference: models/segnet_kitti.py,
models/segnet_sapien.py, models/segnet_ogcdr.py — they differ only in the encoder/decoder table)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import fused
from ..utils import subgraph
from ..utils.nn_util import Seq
from ..utils.pointnet2_util import PointnetFPModule, PointnetSAModule, PointnetSAModuleMSG
from ..utils.transformer_util import MaskFormerHead

BN_CONFIG = {"class": "GroupNorm", "num_groups": 4}


# Priority of the side stream the next batch's geometry plan runs on (0 = as the launch stream, -1 = high).  Its kernels are
# short and latency-bound (an FPS round chain on one workgroup per cloud, searches of a few thousand queries); with the same
# priority as the dense kernels their workgroups wait for a free slot behind thousands of the main queue's.
import os as _os
GEOMETRY_STREAM_PRIORITY = int(_os.environ.get("OGC_GEOMETRY_PRIORITY", "0"))


class MaskFormer3DBase(nn.Module):
    """PointNet++ encoder/decoder -> per-point embedding; MaskFormer head -> K slot embeddings;
    mask = softmax_K(cos(point, slot) / 0.05).  Reference forward: models/segnet_kitti.py:62-89.

    ``sa_specs``: list of dicts, the first (multi-scale) with keys div, radii, nsamples, mlps and the others
    (single-scale) with div, radius, nsample, mlp; ``fp_specs``: list of MLP channel lists (finest first).
    """

    def __init__(self, sa_specs, fp_specs, n_slot, n_point, use_xyz, bn, n_transformer_layer,
                 transformer_embed_dim, transformer_input_pos_enc):
        super().__init__()
        self.SA_modules = nn.ModuleList()
        for spec in sa_specs:
            npoint = int(n_point / spec["div"])
            if "radii" in spec:
                self.SA_modules.append(PointnetSAModuleMSG(
                    npoint=npoint, radii=list(spec["radii"]), nsamples=list(spec["nsamples"]),
                    mlps=[list(m) for m in spec["mlps"]], use_xyz=use_xyz, bn=bn))
            else:
                self.SA_modules.append(PointnetSAModule(
                    npoint=npoint, radius=spec["radius"], nsample=spec["nsample"], mlp=list(spec["mlp"]),
                    use_xyz=use_xyz, bn=bn))
        self.FP_modules = nn.ModuleList(PointnetFPModule(mlp=list(m), bn=bn) for m in fp_specs)

        self.MF_head = MaskFormerHead(
            n_slot=n_slot, input_dim=256, n_transformer_layer=n_transformer_layer,
            transformer_embed_dim=transformer_embed_dim, transformer_n_head=8,
            transformer_hidden_dim=transformer_embed_dim, input_pos_enc=transformer_input_pos_enc)
        self.object_mlp = Seq(transformer_embed_dim).conv1d(transformer_embed_dim, bn=bn).conv1d(64, activation=None)

    overlap_geometry = True  # run the coordinate-only work (FPS, kNN, 3-NN) on a side stream on the GPU

    def _plan(self, pc):
        """Geometry of all levels, in dependency order (each level samples from the previous level's centres)."""
        sa_plans, l_pc = [], [pc]
        ties = None  # level l+1 samples from the centres of level l, stored in sampling order (FPS chain)
        for sa in self.SA_modules:
            g = sa.plan_geometry(l_pc[-1], ties)
            ties = g.get("ties")
            sa_plans.append(g)
            l_pc.append(g["new_xyz"])
        fp_plans = [fp.plan_geometry(l_pc[i], l_pc[i + 1]) for i, fp in enumerate(self.FP_modules)]
        return sa_plans, fp_plans

    def plan_geometry_async(self, pc, after=None):
        """Queue the coordinate-only work of a forward pass on `pc` (FPS, kNN, 3-NN of every level) on the geometry
        side stream and return the handles `forward(..., geometry=...)` takes.  By default the side stream first waits
        for the work already queued on the current stream; with `after` (an event marking `pc` ready) it waits for that
        event only, so the plans of a FUTURE batch can run underneath the current step's dense kernels."""
        from ..utils.streams import Pending, side_stream
        n_sa, n_fp = len(self.SA_modules), len(self.FP_modules)
        sa_geo, fp_geo = [None] * n_sa, [None] * n_fp
        stream = side_stream(pc.device, "segnet-geometry", priority=GEOMETRY_STREAM_PRIORITY)
        if after is None:
            stream.wait_stream(torch.cuda.current_stream())
        else:
            stream.wait_event(after)
        pc.record_stream(stream)
        with torch.cuda.stream(stream):  # one event per level so that SA1 can start as soon as ITS plan is ready
            l_last, ties = pc, None  # level l+1 samples from the centres of level l, stored in sampling order
            for i, sa in enumerate(self.SA_modules):
                g = sa.plan_geometry(l_last, ties)
                ties = g.get("ties")
                ev = torch.cuda.Event()
                ev.record(stream)
                sa_geo[i] = Pending(g, ev)
                l_last = g["new_xyz"]
            cents = [pc] + [p._value["new_xyz"] for p in sa_geo]
            for i, fp in enumerate(self.FP_modules):
                g = fp.plan_geometry(cents[i], cents[i + 1])
                ev = torch.cuda.Event()
                ev.record(stream)
                fp_geo[i] = Pending(g, ev)
        return sa_geo, fp_geo

    overlap_head = True      # run the slot branch on a side stream underneath the feature-propagation stack

    # The slot branch is ~45 launches forwards and ~115 backwards on (B, K, E) tensors, each a few microseconds of GPU time and
    # ~15 of launch-thread time — 2 ms of the ~11 ms of Python a C4 step costs, on a thread that is level with the GPU.  In
    # training it therefore runs as a pair of HIP graphs (utils/subgraph.py): one launch each way, the same kernels in the same
    # order.
    graph_slot_branch = True

    def _slots_eager(self, coarse_feats, coarse_pc):
        slot = self.MF_head(coarse_feats.transpose(1, 2), coarse_pc)      # (B, K, D)
        return self.object_mlp(slot.transpose(1, 2))                      # (B, 64, K)

    def _slots(self, coarse_feats, coarse_pc):
        if self.graph_slot_branch and coarse_feats.requires_grad:
            return subgraph.run(self, "slots", self._slots_eager, (coarse_feats, coarse_pc), parts=(self.MF_head, self.object_mlp))
        return self._slots_eager(coarse_feats, coarse_pc)

    def forward(self, pc, point_feats, geometry=None):
        # pc (B, N, 3), point_feats (B, N, 3) -> mask (B, N, K);  geometry: plan_geometry_async(pc) made earlier
        n_sa, n_fp = len(self.SA_modules), len(self.FP_modules)
        sa_geo, fp_geo = [None] * n_sa, [None] * n_fp
        if geometry is not None:
            sa_geo, fp_geo = geometry
        elif pc.is_cuda and self.overlap_geometry:
            sa_geo, fp_geo = self.plan_geometry_async(pc)

        l_pc, l_feats = [pc], [point_feats.transpose(1, 2).contiguous()]
        for i, sa in enumerate(self.SA_modules):
            li_pc, li_feats = sa(l_pc[-1], l_feats[-1], geometry=sa_geo[i])
            l_pc.append(li_pc)
            l_feats.append(li_feats)
        # The slot branch (MaskFormer head + object MLP) reads only the coarsest level, the feature-propagation stack
        # does not write it: the two meet at the mask read-out.  The slot branch is ~100 launches of a few microseconds
        # on (B, K, E) tensors that leave the GPU almost empty, so on the GPU it runs on a side stream underneath the
        # decoder's dense kernels — in the forward pass here and, because autograd replays every node on the stream
        # its forward ran on, in the backward pass as well.
        branch = None
        if pc.is_cuda and self.overlap_head:
            from ..utils.streams import side_stream
            cur = torch.cuda.current_stream(pc.device)
            branch = side_stream(pc.device, "segnet-slots")
            branch.wait_stream(cur)
            l_feats[-1].record_stream(branch)
            l_pc[-1].record_stream(branch)
            with torch.cuda.stream(branch):
                slot = self._slots(l_feats[-1], l_pc[-1])
        # decoder: coarsest -> finest, FP_modules[i] produces level i
        for i in range(n_fp - 1, -1, -1):
            l_feats[i] = self.FP_modules[i](l_pc[i], l_pc[i + 1], l_feats[i], l_feats[i + 1], geometry=fp_geo[i])
        if branch is None:
            slot = self._slots(l_feats[-1], l_pc[-1])
        else:
            cur.wait_stream(branch)
            slot.record_stream(cur)
        if fused.slot_masks_available(l_feats[0], slot):
            return fused.slot_masks(l_feats[0], slot, 0.05)
        logits = torch.einsum('bdn,bdk->bnk', F.normalize(l_feats[0], dim=1), F.normalize(slot, dim=1)) / 0.05
        return logits.softmax(dim=-1)
