"""Backward of a set-abstraction MLP's tail (last convolution + pooled GroupNorm) as one autograd node with the gradient in sparse
form (fused._NormActConvPool) against the two-node sequence, at the C4 shapes:   python tools/pool_tail_compare.py"""
import torch

import ogc_amd  # noqa: F401
from ogc_amd import fused

SHAPES = [(16, 32, 32, 2048, 64), (16, 32, 64, 2048, 64), (16, 64, 128, 1024, 64), (16, 128, 256, 512, 64)]


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


if __name__ == "__main__":
    width = fused.FUSED_GN_BACKWARD_MAX_WIDTH
    for B, cin, cout, P, S in SHAPES:
        y_prev = torch.randn(B, cin, P, S, device="cuda")
        gn, gn2 = torch.nn.GroupNorm(4, cin).cuda(), torch.nn.GroupNorm(4, cout).cuda()
        conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
        probe = torch.randn(B, cout, P, device="cuda")
        res = {}
        for name, one_node, w in (("two nodes", False, width), ("two nodes, moment matrices", False, 160),
                                  ("one node (sparse gradient)", True, 160)):
            fused.FUSED_GN_BACKWARD_MAX_WIDTH = w
            if cout > w and w != width:
                continue
            if one_node and not fused.norm_act_conv_pool_available(y_prev, gn, conv, gn2):
                continue
            yp = y_prev.clone().requires_grad_(True)
            if one_node:
                out = fused.norm_act_conv_pool(yp, None, gn, True, conv, gn2, True)
            else:
                y, stats, ext = fused.norm_act_conv(yp, None, gn, True, conv, gn2, pool=S)
                out = fused.group_norm_act_maxpool(y, gn2, True, stats, ext)
            res[name] = bench(lambda: torch.autograd.grad(out, [yp, conv.weight, gn.weight, gn2.weight], probe,
                                                           retain_graph=True))
        fused.FUSED_GN_BACKWARD_MAX_WIDTH = width
        print("%3d -> %3d  P=%d S=%d   " % (cin, cout, P, S) + "   ".join("%s %.3f ms" % kv for kv in res.items()), flush=True)
