"""Accuracy of fused.conv_norm_act (conv1x1 -> BatchNorm -> ReLU [-> max]) against the same chain in float64."""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: F401,E402
from ogc_amd import fused  # noqa: E402


def rel(a, b):
    return float((a.double() - b).norm() / b.norm().clamp_min(1e-300))


def run(B, cin, cout, P, S, training, relu, maxpool, seed=0):
    g = torch.Generator().manual_seed(seed)
    conv = nn.Conv2d(cin, cout, 1, bias=False)
    bn = nn.BatchNorm2d(cout)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(cout, generator=g) * 0.2 - 0.1)
        bn.running_mean.copy_(torch.rand(cout, generator=g) * 0.2 - 0.1)
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    x = torch.randn(B, cin, P, S, generator=g)
    conv64, bn64 = nn.Conv2d(cin, cout, 1, bias=False).double(), nn.BatchNorm2d(cout).double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    bn64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    for m in (conv, bn, conv64, bn64):
        m.train(bool(training))
    x64 = x.double().requires_grad_(True)
    y64 = bn64(conv64(x64))
    if relu:
        y64 = F.relu(y64)
    if maxpool:
        y64 = y64.max(-1)[0]
    tgt = torch.randn(y64.shape, generator=g, dtype=torch.float64)
    (y64 * tgt).sum().backward()
    out = {}
    for mode in ("fused", "torch32"):
        c, b_ = conv.cuda(), bn.cuda()
        for p in list(c.parameters()) + list(b_.parameters()):
            p.grad = None
        xc = x.cuda().requires_grad_(True)
        if mode == "fused":
            y = fused.conv_norm_act(xc, c, b_, relu=relu, maxpool=maxpool)
        else:
            y = b_(c(xc))
            y = F.relu(y) if relu else y
            y = y.max(-1)[0] if maxpool else y
        (y * tgt.float().cuda()).sum().backward()
        out[mode] = (rel(y.detach().cpu(), y64.detach()), rel(xc.grad.cpu(), x64.grad), rel(c.weight.grad.cpu(), conv64.weight.grad),
                     rel(b_.weight.grad.cpu(), bn64.weight.grad), rel(b_.bias.grad.cpu(), bn64.bias.grad))
    print("B%d %d->%d P%d S%d train=%d relu=%d max=%d" % (B, cin, cout, P, S, training, relu, maxpool))
    for mode, v in out.items():
        print("   %-8s y %.1e dx %.1e dW %.1e dgamma %.1e dbeta %.1e" % ((mode,) + v))


if __name__ == "__main__":
    for training in (0, 1):
        for relu, mp in ((1, 0), (1, 1), (0, 0)):
            run(2, 32, 64, 256, 16, training, relu, mp)
            run(2, 131, 128, 512, 16, training, relu, mp)
            run(1, 64, 64, 2048, 32, training, relu, mp)
