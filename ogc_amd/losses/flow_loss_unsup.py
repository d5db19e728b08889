"""Unsupervised FlowStep3D losses (reference: losses/flow_loss_unsup.py): Chamfer distance through 1-NN
lookups and flow smoothness over kNN / ball-query neighbourhoods, on the HIP operators."""
import torch
import torch.nn as nn

from ..pointnet2.pointnet2 import ball_query, grouping_operation, knn, knn_radius_clamp


class ChamferLoss(nn.Module):
    """Bidirectional nearest-neighbour distance between pc1 + flow and pc2. Reference: :7-35."""

    def __init__(self, loss_norm=2):
        super().__init__()
        self.loss_norm = loss_norm

    def forward(self, pc1, pc2, flow):
        # pc1, pc2, flow (B, N, 3) -> scalar
        pc2 = pc2.contiguous()
        pc2_t = pc2.transpose(1, 2).contiguous()
        pc1 = (pc1 + flow).contiguous()
        pc1_t = pc1.transpose(1, 2).contiguous()
        _, idx = knn(1, pc1, pc2)
        nn1 = grouping_operation(pc2_t, idx.detach()).squeeze(-1)
        dist1 = (pc1_t - nn1).norm(p=self.loss_norm, dim=1)
        _, idx = knn(1, pc2, pc1)
        nn2 = grouping_operation(pc1_t, idx.detach()).squeeze(-1)
        dist2 = (pc2_t - nn2).norm(p=self.loss_norm, dim=1)
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

    def forward(self, pc, flow):
        return (self.w_knn * self.knn_loss(pc, flow)) + (self.w_ball_q * self.ball_q_loss(pc, flow))


class UnsupervisedFlowStep3DLoss(nn.Module):
    """Per-iteration weighted Chamfer + smoothness. Reference: :112-140; ``loss_dict`` keys
    ``chamfer_loss_#i``, ``smooth_loss_#i``, ``sum`` (gathered with one device->host copy)."""

    def __init__(self, chamfer_loss, smooth_loss, weights=[0.75, 0.25], iters_w=[1.0]):
        super().__init__()
        self.chamfer_loss = chamfer_loss
        self.smooth_loss = smooth_loss
        self.w_chamfer, self.w_smooth = weights
        self.iters_w = iters_w

    def forward(self, pc1, pc2, flow_preds):
        assert len(flow_preds) == len(self.iters_w)
        monitored, loss = [], 0
        for i, flow_pred in enumerate(flow_preds):
            chamfer_i = self.chamfer_loss(pc1, pc2, flow_pred)
            smooth_i = self.smooth_loss(pc1, flow_pred)
            monitored += [('chamfer_loss_#%d' % i, chamfer_i), ('smooth_loss_#%d' % i, smooth_i)]
            loss = loss + self.iters_w[i] * (self.w_chamfer * chamfer_i + self.w_smooth * smooth_i)
        monitored.append(('sum', loss))
        values = torch.stack([v.detach().float().reshape(()) for _, v in monitored]).tolist()
        return loss, {k: v for (k, _), v in zip(monitored, values)}
