"""One optimisation step of unsupervised OGC segmentation training — the body of the reference's
``Trainer._train_it`` (train_seg.py:47-86) — factored out so that the benchmark, the smoke test and the
training driver share it.  Semantics kept: views are flattened into the batch for the network, the loss is
called with ``step_w=True, it=it*b``, and the optimiser step is skipped when any gradient contains NaN
(train_seg.py:81-83; here one fused check and one host sync instead of one per parameter)."""
import torch

from .losses.seg_loss_unsup import (DynamicLoss, EntropyLoss, InvarianceLoss, RankLoss, SmoothLoss,
                                    UnsupervisedOGCLoss)

KITTI_LOSS = dict(  # config/seg/kittisf/kittisf_unsup.yaml:40-56
    weights=[10.0, 0.1, 0.1], start_steps=[0, 100, 1000],
    dynamic_loss_params=dict(loss_norm=2),
    smooth_loss_params=dict(w_knn=3., w_ball_q=1., knn_loss_params=dict(k=32, radius=1., loss_norm=1),
                            ball_q_loss_params=dict(k=64, radius=2., loss_norm=1)),
    invariance_loss_params=dict(loss_norm=2))

SAPIEN_LOSS = dict(  # config/seg/sapien/sapien_unsup.yaml
    weights=[10.0, 0.1, 0.1], start_steps=[0, 0, 0],
    dynamic_loss_params=dict(loss_norm=2),
    smooth_loss_params=dict(w_knn=3., w_ball_q=1., knn_loss_params=dict(k=8, radius=0.1, loss_norm=1),
                            ball_q_loss_params=dict(k=16, radius=0.2, loss_norm=1)),
    invariance_loss_params=dict(loss_norm=2))


def build_criterion(cfg):
    return UnsupervisedOGCLoss(DynamicLoss(**cfg["dynamic_loss_params"]), SmoothLoss(**cfg["smooth_loss_params"]),
                               InvarianceLoss(**cfg["invariance_loss_params"]), EntropyLoss(), RankLoss(),
                               weights=cfg["weights"], start_steps=cfg["start_steps"])


class PendingStep:
    """Outcome of a train_step whose scalars are still on their way to the host.  ``result()`` -> (loss_dict, stepped)."""

    def __init__(self, pending_losses, bad_scalar):
        self._losses, self._bad, self._out = pending_losses, bad_scalar, None

    def result(self):
        if self._out is None:
            losses = self._losses.resolve() if hasattr(self._losses, "resolve") else self._losses
            self._out = (losses, not bool(self._bad.get()[0]))
        return self._out


def make_optimizer(params, lr, weight_decay=0.0):
    """Adam as the reference builds it (train_seg.py:320).  On the GPU the fused implementation is used: the same
    update, and it can skip itself on the device when handed a `found_inf` flag, which is what lets train_step apply
    the reference's NaN-gradient rule (train_seg.py:81-83) without stopping to read the flag on the host."""
    params = list(params)
    fused = bool(params) and all(p.is_cuda for p in params)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=fused)


def train_step(segnet, criterion, optimizer, batch, it, aug_transform, sync=True):
    """batch = (pcs (b,t,n,3), segms (b,t,n), flows (b,t,n,3), valids), already on the device.
    Returns (loss_dict, stepped); with sync=False a PendingStep whose result() gives the same pair later, so the
    host can queue the next step while this one still runs (no host synchronisation inside the step when the
    optimizer is fused — see make_optimizer)."""
    from .utils.streams import HostScalars
    segnet.train()
    optimizer.zero_grad(set_to_none=True)
    pcs, segms, flows, _ = batch
    b, t, n = segms.size()
    flat = pcs.view(b * t, n, -1).contiguous()
    pcs_l = [pcs[:, tt].contiguous() for tt in range(t)]
    flows_l = [flows[:, tt].contiguous() for tt in range(t)]
    loss_geometry = None
    if pcs.is_cuda and hasattr(criterion, "plan_geometry"):
        # the smooth term's neighbour searches depend on coordinates only: start them now on a side stream so they
        # overlap the network's forward pass instead of serialising after it
        from .utils.streams import launch_on_side, side_stream
        loss_geometry = launch_on_side(side_stream(pcs.device, "loss-geometry"),
                                       lambda: criterion.plan_geometry(pcs_l, aug_transform))
    masks = segnet(flat, flat).view(b, t, n, -1)
    masks_l = [masks[:, tt].contiguous() for tt in range(t)]
    kw = {"geometry": loss_geometry} if loss_geometry is not None else {}
    loss, losses = criterion(pcs_l, masks_l, flows_l, step_w=True, it=it * b, aug_transform=aug_transform, sync=False,
                             **kw)
    loss.backward()
    grads = [p.grad for p in segnet.parameters() if p.grad is not None]
    bad = torch.isnan(torch.stack(torch._foreach_norm(grads)).sum())  # NaN anywhere -> NaN norm
    # under DDP the all-reduced gradients make this decision identical on all ranks
    if getattr(optimizer, "_step_supports_amp_scaling", False) and bad.is_cuda:
        # fused optimizer: the kernel itself skips the update (and the step count) when found_inf == 1
        optimizer.grad_scale = None
        optimizer.found_inf = bad.float().reshape(())
        try:
            optimizer.step()
        finally:
            del optimizer.grad_scale
            del optimizer.found_inf
        pending = PendingStep(losses, HostScalars(bad.reshape(1)))
    else:
        skip = bool(bad)  # host sync
        if not skip:
            optimizer.step()
        pending = PendingStep(losses, HostScalars(torch.tensor([skip])))
    return pending.result() if sync else pending
