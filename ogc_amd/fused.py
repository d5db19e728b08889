"""Host-side wrappers of the fused HIP extensions that replace Python-level op sequences of the reference's
layers (no counterpart in its native module; see include/ogc_ops.h "fused extensions")."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .pointnet2 import pointnet2 as _api


class _GroupNormAct(Function):
    """y = act(GroupNorm(x)) with act = ReLU or identity — one autograd node, two launches forward, three backward.
    Reference sequence: nn.GroupNorm then nn.ReLU(inplace=True) (utils/nn_util.py:6-11, :45-85)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu):
        nat = _api._native
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        hw = x.numel() // max(B * C, 1)
        y = torch.empty_like(x)
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = torch.empty(2 * B * groups, dtype=torch.float64, device=x.device)
        nat.group_norm_fwd_wrapper(B, C, hw, groups, eps, relu, x, weight.detach().contiguous(),
                                   bias.detach().contiguous(), y, mean, rstd, ws)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.cfg = (groups, relu, hw)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        nat = _api._native
        x, weight, bias, mean, rstd = ctx.saved_tensors
        groups, relu, hw = ctx.cfg
        B, C = x.shape[0], x.shape[1]
        grad_y = grad_y.contiguous()
        grad_x = torch.empty_like(x)
        gw = torch.empty_like(weight)
        gb = torch.empty_like(bias)
        ws = torch.empty(2 * B * C + B * groups, dtype=torch.float64, device=x.device)
        nat.group_norm_bwd_wrapper(B, C, hw, groups, relu, x, weight.detach().contiguous(), bias.detach().contiguous(),
                                   mean, rstd, grad_y, grad_x, gw, gb, ws)
        return grad_x, gw, gb, None, None, None


def group_norm_act(x, gn: torch.nn.GroupNorm, relu: bool):
    """GroupNorm followed by an optional ReLU.  HIP-fused on the GPU (fp32); the plain torch composition
    otherwise (that is what the reference runs)."""
    if (x.is_cuda and x.dtype == torch.float32 and gn.affine
            and getattr(_api._native, "group_norm_fwd_wrapper", None) is not None):
        return _GroupNormAct.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)
    y = F.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps)
    return F.relu(y) if relu else y
