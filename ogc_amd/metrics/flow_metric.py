"""Scene-flow metrics (reference: metrics/flow_metric.py:4-25 `eval_flow`, train_flow.py:18-30 `epe_metric`), computed
on the device the flows live on; the reference copies both tensors to the host first and reads four `.item()`s."""
import torch


def flow_metrics(gt_flow, flow_pred, epe_norm_thresh=0.05, eps=1e-10):
    """(EPE3D, Acc3DS, Acc3DR, Outliers3D) as one 4-element tensor on gt_flow's device (no synchronisation)."""
    gt_flow, flow_pred = gt_flow.detach(), flow_pred.detach().to(gt_flow.device)
    epe_norm = torch.norm(flow_pred - gt_flow, dim=2)
    relative_err = epe_norm / (torch.norm(gt_flow, dim=2) + eps)
    acc_strict = torch.logical_or(epe_norm < epe_norm_thresh, relative_err < 0.05).float().mean()
    acc_relax = torch.logical_or(epe_norm < 2 * epe_norm_thresh, relative_err < 0.1).float().mean()
    outlier = torch.logical_or(epe_norm > 6 * epe_norm_thresh, relative_err > 0.1).float().mean()
    return torch.stack([epe_norm.mean(), acc_strict, acc_relax, outlier])


def eval_flow(gt_flow, flow_pred, epe_norm_thresh=0.05, eps=1e-10):
    """The reference's signature: four Python floats (one device->host copy)."""
    return tuple(flow_metrics(gt_flow, flow_pred, epe_norm_thresh, eps).tolist())


def epe_terms(gt_flow, flow_preds):
    """[('epe3d_#i', scalar tensor)] for the iterative predictions of FlowStep3D (train_flow.py:18-30), on the device."""
    gt_flow = gt_flow.detach()
    return [('epe3d_#%d' % i, torch.norm(p.detach() - gt_flow, dim=2).mean()) for i, p in enumerate(flow_preds)]


def epe_metric(gt_flow, flow_preds):
    names, values = zip(*epe_terms(gt_flow, flow_preds))
    return dict(zip(names, torch.stack(values).tolist()))
