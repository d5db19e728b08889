"""Backward of conv(relu(GroupNorm(y_prev))) through the moment matrices (ogc_conv1x1_wgrad_moments ->
ogc_gn_moments_combine -> ogc_conv1x1_dgrad_adjoint, csrc/gn_fused_bwd.hip) against the same op sequence evaluated in
float64 by torch (the reference's layers: utils/nn_util.py:45-85), and against the separate GroupNorm-backward passes it
replaces: input gradient, convolution weight gradient, GroupNorm weight / bias gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


@pytest.mark.parametrize("B,cin,cout,hw,groups,relu", [
    (2, 32, 32, 2048, 4, True), (3, 32, 64, 4096, 4, True), (2, 64, 128, 1024, 4, True), (2, 128, 128, 2048, 4, True),
    (2, 16, 24, 640, 4, True), (1, 100, 160, 512, 4, True), (2, 64, 64, 1024, 8, False), (16, 32, 32, 8192, 4, True)])
def test_fused_gn_backward(B, cin, cout, hw, groups, relu):
    import ogc_amd  # noqa: F401
    from ogc_amd import fused
    g = torch.Generator().manual_seed(B * 1000 + cin + cout)
    y_prev = (torch.randn(B, cin, hw, 1, generator=g) * 1.5 + 0.3).cuda()
    gn = torch.nn.GroupNorm(groups, cin).cuda()
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    with torch.no_grad():
        gn.weight.copy_(torch.rand(cin, generator=g) + 0.5)
        gn.weight[::5] *= -1.0                                     # negative scales too
        gn.bias.copy_(torch.rand(cin, generator=g) - 0.5)
    probe = torch.randn(B, cout, hw, 1, generator=g).cuda()

    width = fused.FUSED_GN_BACKWARD_MAX_WIDTH
    fused.FUSED_GN_BACKWARD_MAX_WIDTH = 160          # the kernels are tested at every width they accept

    def run(flag):
        fused.FUSED_GN_BACKWARD = flag
        yp = y_prev.clone().requires_grad_(True)
        for p in list(gn.parameters()) + list(conv.parameters()):
            p.grad = None
        y, _ = fused.norm_act_conv(yp, None, gn, relu, conv)
        (y * probe).sum().backward()
        return y.detach(), yp.grad, conv.weight.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()

    try:
        assert fused.norm_act_conv_available(y_prev, gn, conv)
        new = run(True)
        old = run(False)
    finally:
        fused.FUSED_GN_BACKWARD, fused.FUSED_GN_BACKWARD_MAX_WIDTH = True, width
    # float64 truth
    yp = y_prev.double().requires_grad_(True)
    z = F.group_norm(yp, groups, gn.weight.double(), gn.bias.double(), gn.eps)
    z = F.relu(z) if relu else z
    y = F.conv2d(z, conv.weight.double())
    gw64 = gn.weight.double().detach().requires_grad_(True)
    gb64 = gn.bias.double().detach().requires_grad_(True)
    cw64 = conv.weight.double().detach().requires_grad_(True)
    z2 = F.group_norm(yp, groups, gw64, gb64, gn.eps)
    y2 = F.conv2d(F.relu(z2) if relu else z2, cw64)
    truth = torch.autograd.grad((y2 * probe.double()).sum(), [yp, cw64, gw64, gb64])
    assert torch.equal(new[0], old[0])
    names = ("grad_prev", "grad_conv_weight", "grad_gn_weight", "grad_gn_bias")
    for name, n_, o_, t_ in zip(names, new[1:], old[1:], truth):
        e_new, e_old = _rel(n_, t_), _rel(o_, t_)
        # as close to the exact gradient as the separate passes (which accumulate their sums in fp64), within a small factor
        assert e_new <= max(4 * e_old, 2e-6), (name, e_new, e_old)
        assert e_new < 1e-5, (name, e_new)


@pytest.mark.parametrize("B,cin,cout,P,S,groups,groups2,relu2", [
    (2, 32, 32, 128, 64, 4, 4, True), (3, 32, 64, 96, 64, 4, 8, True), (2, 16, 32, 40, 16, 4, 4, True),
    (2, 64, 64, 66, 32, 8, 4, True), (1, 24, 48, 8, 64, 4, 4, False), (16, 32, 64, 2048, 64, 4, 4, True), (4, 64, 128, 256, 64, 4, 4, True)])
def test_pooled_tail_backward(B, cin, cout, P, S, groups, groups2, relu2):
    """Last SharedMLP layer + pooled GroupNorm as one autograd node (fused._NormActConvPool: the gradient w.r.t. the
    convolution's output stays in sparse form, ogc_group_norm_maxpool_bwd_sparse / *_pooled) against the two-node sequence
    that writes it out: the kernels rebuild the same fp32 expression element by element, so the results agree up to the
    order in which workgroups add their partial moment matrices (float atomics; the pooled GroupNorm's own results are identical)."""
    import ogc_amd  # noqa: F401
    from ogc_amd import fused
    g = torch.Generator().manual_seed(B * 1000 + cin + cout + P)
    y_prev = (torch.randn(B, cin, P, S, generator=g) * 1.5 + 0.3).cuda()
    gn = torch.nn.GroupNorm(groups, cin).cuda()
    gn2 = torch.nn.GroupNorm(groups2, cout).cuda()
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    with torch.no_grad():
        for m in (gn, gn2):
            m.weight.copy_(torch.rand(m.num_channels, generator=g) + 0.5)
            m.weight[::5] *= -1.0                                  # negative scales: the pooled value is the minimum's image
            m.bias.copy_(torch.rand(m.num_channels, generator=g) - 0.5)
    probe = torch.randn(B, cout, P, generator=g).cuda()
    params = list(gn.parameters()) + list(conv.parameters()) + list(gn2.parameters())
    width = fused.FUSED_GN_BACKWARD_MAX_WIDTH
    fused.FUSED_GN_BACKWARD_MAX_WIDTH = 160          # the kernels are tested beyond the widths the product uses them at

    def run(one_node):
        yp = y_prev.clone().requires_grad_(True)
        for p in params:
            p.grad = None
        if one_node:
            assert fused.norm_act_conv_pool_available(yp, gn, conv, gn2)
            out = fused.norm_act_conv_pool(yp, None, gn, True, conv, gn2, relu2)
        else:
            y, stats, extremes = fused.norm_act_conv(yp, None, gn, True, conv, gn2, pool=S)
            out = fused.group_norm_act_maxpool(y, gn2, relu2, stats, extremes)
        (out * probe).sum().backward()
        return [out.detach(), yp.grad] + [p.grad.clone() for p in params]

    try:
        new, old = run(True), run(False)
    finally:
        fused.FUSED_GN_BACKWARD_MAX_WIDTH = width
    for name, n_, o_ in zip(("out", "grad_prev", "gn.weight", "gn.bias", "conv.weight", "gn2.weight", "gn2.bias"), new, old):
        if name in ("out", "gn2.weight", "gn2.bias"):
            assert torch.equal(n_, o_), (name, _rel(n_, o_))
        else:
            assert _rel(n_, o_) < 2e-6, (name, _rel(n_, o_))


@pytest.mark.parametrize("B,cin,cout,P,S,groups,groups2,relu2", [
    (16, 64, 128, 256, 64, 4, 4, True), (16, 128, 256, 128, 64, 4, 4, True), (8, 64, 128, 512, 32, 4, 8, True),
    (4, 96, 160, 1024, 16, 4, 4, False), (16, 131, 128, 128, 64, 1, 4, True), (4, 32, 96, 512, 64, 4, 4, True)])
def test_wide_pooled_tail_backward(B, cin, cout, P, S, groups, groups2, relu2):
    """The same one-node tail for layers WIDER than the moment-matrix path takes (round 4: SA2 64 -> 128, SA3 128 -> 256 at C4):
    the plain weight- and input-gradient kernels in their pooled forms (ogc_conv1x1_wgrad_affine_pooled, ogc_conv1x1_dgrad_pooled:
    g_y rebuilt from (y, coef2, inj) on load) against the two-node sequence that writes the dense g_y.  Same fp32 expression
    element by element: the pooled GroupNorm's results are identical, the rest agrees up to summation order."""
    import ogc_amd  # noqa: F401
    from ogc_amd import fused
    g = torch.Generator().manual_seed(B * 1000 + cin + cout + P)
    y_prev = (torch.randn(B, cin, P, S, generator=g) * 1.5 + 0.3).cuda()
    gn = torch.nn.GroupNorm(groups, cin).cuda()
    gn2 = torch.nn.GroupNorm(groups2, cout).cuda()
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    with torch.no_grad():
        for m in (gn, gn2):
            m.weight.copy_(torch.rand(m.num_channels, generator=g) + 0.5)
            m.weight[::5] *= -1.0
            m.bias.copy_(torch.rand(m.num_channels, generator=g) - 0.5)
    probe = torch.randn(B, cout, P, generator=g).cuda()
    params = list(gn.parameters()) + list(conv.parameters()) + list(gn2.parameters())

    def run(one_node):
        yp = y_prev.clone().requires_grad_(True)
        for p in params:
            p.grad = None
        if one_node:
            assert fused.WIDE_POOL_BACKWARD and fused.norm_act_conv_pool_available(yp, gn, conv, gn2)
            out = fused.norm_act_conv_pool(yp, None, gn, True, conv, gn2, relu2)
        else:
            y, stats, extremes = fused.norm_act_conv(yp, None, gn, True, conv, gn2, pool=S)
            out = fused.group_norm_act_maxpool(y, gn2, relu2, stats, extremes)
        (out * probe).sum().backward()
        return [out.detach(), yp.grad] + [p.grad.clone() for p in params]

    new, old = run(True), run(False)
    for name, n_, o_ in zip(("out", "grad_prev", "gn.weight", "gn.bias", "conv.weight", "gn2.weight", "gn2.bias"), new, old):
        if name in ("out", "gn2.weight", "gn2.bias"):
            assert torch.equal(n_, o_), (name, _rel(n_, o_))
        else:
            assert _rel(n_, o_) < 3e-6, (name, _rel(n_, o_))
