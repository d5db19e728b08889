"""Unsupervised FlowStep3D losses (reference: losses/flow_loss_unsup.py): Chamfer distance through 1-NN
lookups and flow smoothness over kNN / ball-query neighbourhoods, on the HIP operators."""
import torch
import torch.nn as nn

from ..pointnet2.pointnet2 import ball_query, grouping_operation, knn, knn_radius_clamp


class _ChamferTerms(torch.autograd.Function):
    """(dist1, dist2) of the Chamfer loss from the warped cloud, the target cloud and the two 1-NN index rows: one launch forward,
    one backward (csrc/chamfer.hip) instead of two transposes, two gathers, two differences and two norms each way.  The indices
    are constants of the differentiation, as ``idx.detach()`` makes them in the reference (:26,30); pc2 is data."""

    @staticmethod
    def forward(ctx, warped, target, idx12, idx21, norm):
        from ..pointnet2 import pointnet2 as _api
        b, n1, n2 = warped.size(0), warped.size(1), target.size(1)
        dist1 = torch.empty(b, n1, device=warped.device)
        dist2 = torch.empty(b, n2, device=warped.device)
        _api._native.chamfer_terms_wrapper(b, n1, n2, norm, warped, target, idx12, idx21, dist1, dist2)
        ctx.save_for_backward(warped, target, idx12, idx21)
        ctx.norm = norm
        return dist1, dist2

    @staticmethod
    def backward(ctx, g1, g2):
        from ..pointnet2 import pointnet2 as _api
        warped, target, idx12, idx21 = ctx.saved_tensors
        b, n1, n2 = warped.size(0), warped.size(1), target.size(1)
        g1 = torch.zeros(b, n1, device=warped.device) if g1 is None else g1.contiguous()
        g2 = torch.zeros(b, n2, device=warped.device) if g2 is None else g2.contiguous()
        grad = torch.empty_like(warped)
        _api._native.chamfer_terms_grad_wrapper(b, n1, n2, ctx.norm, warped, target, idx12, idx21, g1, g2, grad)
        return grad, None, None, None, None


class ChamferLoss(nn.Module):
    """Bidirectional nearest-neighbour distance between pc1 + flow and pc2. Reference: :7-35."""

    def __init__(self, loss_norm=2):
        super().__init__()
        self.loss_norm = loss_norm

    def forward(self, pc1, pc2, flow):
        # pc1, pc2, flow (B, N, 3) -> scalar
        target = pc2.contiguous()
        warped = (pc1 + flow).contiguous()
        idx12 = knn(1, warped, target)[1].detach()          # nearest point of pc2 for every warped point (:24-25)
        idx21 = knn(1, target, warped)[1].detach()          # ... and of the warped cloud for every point of pc2 (:29-30)
        from ..pointnet2 import pointnet2 as _api
        fused = (warped.is_cuda and self.loss_norm in (1, 2) and not target.requires_grad and warped.dtype == torch.float32
                 and getattr(_api._native, "chamfer_terms_wrapper", None) is not None)
        if fused:
            dist1, dist2 = _ChamferTerms.apply(warped, target, idx12.squeeze(-1).contiguous(), idx21.squeeze(-1).contiguous(),
                                               self.loss_norm)
        else:  # the operator sequence (CPU-oracle tests, other norms, a target cloud that wants a gradient)
            channels_first = lambda t: t.transpose(1, 2).contiguous()  # noqa: E731
            dist1 = (channels_first(warped) - grouping_operation(channels_first(target), idx12).squeeze(-1)).norm(p=self.loss_norm, dim=1)
            dist2 = (channels_first(target) - grouping_operation(channels_first(warped), idx21).squeeze(-1)).norm(p=self.loss_norm, dim=1)
        return (dist1 + dist2).mean()


class KnnLoss(nn.Module):
    """Flow smoothness over the k nearest neighbours (clamped to ``radius``). Reference: :38-62."""

    def __init__(self, k, radius, loss_norm=1):
        super().__init__()
        self.k = k
        self.radius = radius
        self.loss_norm = loss_norm

    def forward(self, pc, flow):
        flow = flow.permute(0, 2, 1).contiguous()
        _, idx = knn_radius_clamp(self.k, self.radius, pc.contiguous(), pc.contiguous())
        nn_flow = grouping_operation(flow, idx.detach())
        return (flow.unsqueeze(3) - nn_flow).norm(p=self.loss_norm, dim=1).mean()


class BallQLoss(nn.Module):
    """Flow smoothness over ball-query neighbours. Reference: :65-87."""

    def __init__(self, k, radius, loss_norm=1):
        super().__init__()
        self.k = k
        self.radius = radius
        self.loss_norm = loss_norm

    def forward(self, pc, flow):
        pc = pc.contiguous()
        flow = flow.permute(0, 2, 1).contiguous()
        idx = ball_query(self.radius, self.k, pc, pc)
        nn_flow = grouping_operation(flow, idx.detach())
        return (flow.unsqueeze(3) - nn_flow).norm(p=self.loss_norm, dim=1).mean()


class SmoothLoss(nn.Module):
    """Reference: :90-109."""

    def __init__(self, w_knn, w_ball_q, knn_loss_params, ball_q_loss_params):
        super().__init__()
        self.knn_loss = KnnLoss(**knn_loss_params)
        self.ball_q_loss = BallQLoss(**ball_q_loss_params)
        self.w_knn = w_knn
        self.w_ball_q = w_ball_q

    def plan(self, pc):
        """Neighbour lists of `pc` (and their transposes for the fused gradient) — coordinates only.  The reference
        searches them again for every refinement iteration's prediction (:126-130) although pc1 never changes."""
        pc = pc.contiguous()
        kl, bl = self.knn_loss, self.ball_q_loss
        from .seg_loss_unsup import _shared_grid_searches
        idx_knn, idx_ball = _shared_grid_searches(pc, kl.k, kl.radius, bl.k, bl.radius)   # one cell grid for both searches
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

    def forward(self, pc, flow, plan=None):
        if plan is None:
            return (self.w_knn * self.knn_loss(pc, flow)) + (self.w_ball_q * self.ball_q_loss(pc, flow))
        from ..fused import neighbour_consistency, neighbour_consistency_available
        terms = []
        flow_cm = None
        for name, cfg in (("knn", self.knn_loss), ("ball", self.ball_q_loss)):
            if name + "_rev" in plan and neighbour_consistency_available(flow, cfg.loss_norm, False):
                terms.append(neighbour_consistency(flow.contiguous(), plan[name], plan[name + "_rev"],
                                                   cfg.loss_norm).mean())
            else:
                if flow_cm is None:
                    flow_cm = flow.permute(0, 2, 1).contiguous()
                nn_flow = grouping_operation(flow_cm, plan[name].detach())
                terms.append((flow_cm.unsqueeze(3) - nn_flow).norm(p=cfg.loss_norm, dim=1).mean())
        return self.w_knn * terms[0] + self.w_ball_q * terms[1]


class PendingFlowLossDict:
    """loss_dict of one evaluation on its way to the host (one non-blocking copy); resolve() -> dict of floats."""

    def __init__(self, monitored):
        from ..utils.streams import HostScalars
        self._keys = [k for k, _ in monitored]
        self._scalars = HostScalars(torch.stack([v.detach().float().reshape(()) for _, v in monitored]))
        self._dict = None

    def resolve(self):
        if self._dict is None:
            self._dict = dict(zip(self._keys, self._scalars.get()))
        return self._dict


class UnsupervisedFlowStep3DLoss(nn.Module):
    """Per-iteration weighted Chamfer + smoothness. Reference: :112-140; ``loss_dict`` keys
    ``chamfer_loss_#i``, ``smooth_loss_#i``, ``sum`` (gathered with one device->host copy)."""

    def __init__(self, chamfer_loss, smooth_loss, weights=[0.75, 0.25], iters_w=[1.0]):
        super().__init__()
        self.chamfer_loss = chamfer_loss
        self.smooth_loss = smooth_loss
        self.w_chamfer, self.w_smooth = weights
        self.iters_w = iters_w

    def forward(self, pc1, pc2, flow_preds, sync=True, extra=None):
        # extra: [(name, scalar tensor)] appended to the monitored dict (e.g. the EPE terms the reference's trainer adds)
        assert len(flow_preds) == len(self.iters_w)
        monitored, loss = [], 0
        plan = self.smooth_loss.plan(pc1) if hasattr(self.smooth_loss, "plan") else None
        for i, flow_pred in enumerate(flow_preds):
            chamfer_i = self.chamfer_loss(pc1, pc2, flow_pred)
            smooth_i = self.smooth_loss(pc1, flow_pred, plan) if plan is not None else self.smooth_loss(pc1, flow_pred)
            monitored += [('chamfer_loss_#%d' % i, chamfer_i), ('smooth_loss_#%d' % i, smooth_i)]
            loss = loss + self.iters_w[i] * (self.w_chamfer * chamfer_i + self.w_smooth * smooth_i)
        monitored.append(('sum', loss))
        if extra:
            monitored += list(extra)
        pending = PendingFlowLossDict(monitored)
        return loss, (pending.resolve() if sync else pending)
