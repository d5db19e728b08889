"""Max-pool of a set-abstraction MLP from the neighbourhood extremes its last convolution records
(ogc_conv1x1_gemm_affine_pool + ogc_group_norm_pool_extremes) against the pass over the full tensor
(ogc_conv1x1_gemm_affine + ogc_group_norm_maxpool_fwd_stats): same pooled values bit for bit, input gradients to the last bits
(hence the same arg-max element in every neighbourhood), parameter gradients to rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mlp(cin, widths):
    from ogc_amd.utils.nn_util import SharedMLP
    from ogc_amd.models._segnet import BN_CONFIG
    torch.manual_seed(cin + sum(widths))
    mlp = SharedMLP([cin] + widths, bn=BN_CONFIG).cuda()
    with torch.no_grad():   # both signs of the norm's scale, one exact zero
        for m in mlp.modules():
            if isinstance(m, torch.nn.GroupNorm):
                m.weight.normal_(0.0, 1.0)
                m.bias.normal_(0.0, 0.5)
                m.weight[1] = 0.0
    return mlp


def _run(mlp, x, w, pooled_conv):
    import ogc_amd.pointnet2.pointnet2 as api
    nat = api._native
    saved = getattr(nat, "conv1x1_gemm_affine_pool_wrapper", None)
    if not pooled_conv:
        nat.conv1x1_gemm_affine_pool_wrapper = None
    try:
        for p in mlp.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        out = mlp.forward_maxpool(xi)
        (out * w).sum().backward()
        return [out.detach(), xi.grad] + [p.grad.clone() for p in mlp.parameters()]
    finally:
        nat.conv1x1_gemm_affine_pool_wrapper = saved


@pytest.mark.parametrize("B,cin,widths,P,S", [(4, 32, [32, 64], 256, 64), (2, 16, [32, 32, 64], 128, 32),
                                               (3, 8, [16, 32], 64, 16), (16, 32, [32, 32], 2048, 64),
                                               (1, 64, [64, 128], 64, 64),
                                               # wide layers: the streaming kernel's epilogue (K > 100, >= 2048 position tiles)
                                               (4, 128, [128, 256], 512, 64), (8, 64, [128, 128], 256, 64)])
def test_pool_from_extremes_is_bit_identical(B, cin, widths, P, S, monkeypatch):
    import ogc_amd  # noqa: F401
    from ogc_amd import fused
    monkeypatch.setattr(fused, "POOL_EXTREMES_WIDE", True)   # (off by default: correct but not faster at K > 100)
    mlp = _mlp(cin, widths)
    torch.manual_seed(B + P)
    x = torch.randn(B, cin, P, S, device="cuda")
    x[..., S // 2:] = x[..., :1]          # duplicated neighbours, as the ball query pads short neighbourhoods
    w = torch.randn(B, widths[-1], P, device="cuda")
    calls = []
    import ogc_amd.pointnet2.pointnet2 as api
    orig = api._native.group_norm_pool_extremes_wrapper

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)

    api._native.group_norm_pool_extremes_wrapper = spy
    try:
        got = _run(mlp, x, w, True)
    finally:
        api._native.group_norm_pool_extremes_wrapper = orig
    assert calls, "the extremes path did not run"
    want = _run(mlp, x, w, False)
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        if i == 0:     # pooled output: bit for bit
            assert torch.equal(a, b), (i, (a - b).abs().max().item())
        elif i == 1:   # input gradient: same arg-max element in every neighbourhood (a different one would move whole entries);
            # its GroupNorm sums now come from moment matrices accumulated with fp32 atomics (csrc/gn_fused_bwd.hip), so
            # the last bits depend on the order of those additions
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item(), (i, (a - b).abs().max().item())
        else:          # parameter gradients: the weight-gradient kernel adds its partial sums with atomics
            assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item() + 1e-6, (i, (a - b).abs().max().item())


def test_unsupported_neighbourhood_falls_back():
    """nsample = 8 is not offered by the pooled convolution: the full-tensor pass runs and the result is the same op."""
    import ogc_amd  # noqa: F401
    mlp = _mlp(8, [16, 32])
    x = torch.randn(2, 8, 64, 8, device="cuda")
    out = mlp.forward_maxpool(x)
    ref = mlp(x).max(dim=-1)[0]
    assert torch.allclose(out, ref, atol=1e-5)
