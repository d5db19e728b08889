"""Grouping + first convolution of a set-abstraction MLP without the grouped tensor (ogc_group_linear_fwd) against the
op sequence it replaces: QueryAndGroup's concatenation followed by the 1x1 convolution, outputs, GroupNorm statistics and
gradients; and a whole SA level with and without it."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,npoint,ns,C,M,groups", [(2, 256, 64, 16, 3, 32, 4), (3, 1024, 128, 64, 96, 64, 4),
                                                      (2, 512, 32, 64, 128, 128, 4), (1, 300, 17, 16, 5, 24, 0)])
def test_matches_group_then_conv(B, N, npoint, ns, C, M, groups):
    import ogc_amd  # noqa: F401
    from ogc_amd.fused import _GroupedFirstLayer
    from ogc_amd.pointnet2 import pointnet2 as api
    g = torch.Generator().manual_seed(N + C)
    xyz = ((torch.rand(B, N, 3, generator=g) - 0.5) * 40).cuda()
    new_xyz = xyz[:, :npoint].contiguous()
    feats = torch.randn(B, C, N, generator=g).cuda()
    idx = torch.randint(0, N, (B, npoint, ns), generator=g, dtype=torch.int32).cuda()
    weight = (torch.randn(M, 3 + C, 1, 1, generator=g) / (3 + C) ** 0.5).cuda()
    probe = torch.randn(B, M, npoint, ns, generator=g).cuda()

    f_ref, w_ref = feats.clone().requires_grad_(True), weight.clone().requires_grad_(True)
    grouped = api.GroupConcat.apply(xyz, new_xyz, f_ref, idx)
    y_ref = F.conv2d(grouped, w_ref)
    (y_ref * probe).sum().backward()

    f_new, w_new = feats.clone().requires_grad_(True), weight.clone().requires_grad_(True)
    y_new, stats = _GroupedFirstLayer.apply(xyz, new_xyz, f_new, idx, w_new, groups)
    (y_new * probe).sum().backward()

    scale = y_ref.abs().max().item()
    assert (y_new - y_ref).abs().max().item() <= 2e-6 * scale + 1e-6
    for a, b in ((f_new.grad, f_ref.grad), (w_new.grad, w_ref.grad)):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-6
    if groups:
        slots = stats.numel() // (2 * B * groups)
        s = stats.view(slots, B, groups, 2).sum(0)
        yg = y_ref.detach().double().view(B, groups, -1)
        assert torch.allclose(s[..., 0], yg.sum(-1), rtol=1e-5, atol=1e-5 * yg.abs().sum(-1).max().item())
        assert torch.allclose(s[..., 1], (yg * yg).sum(-1), rtol=1e-5)
    else:
        assert stats is None


def test_sa_level_with_and_without_the_fusion(monkeypatch):
    import ogc_amd  # noqa: F401
    import ogc_amd.fused as fused
    from ogc_amd.utils.pointnet2_util import PointnetSAModuleMSG
    torch.manual_seed(0)
    bn = {"class": "GroupNorm", "num_groups": 4}
    sa = PointnetSAModuleMSG(npoint=128, radii=[1.0, 2.0], nsamples=[16, 32], mlps=[[8, 32, 32], [8, 32, 64]], bn=bn).cuda()
    xyz = ((torch.rand(3, 512, 3) - 0.5) * 20).cuda()
    feats = torch.randn(3, 8, 512).cuda()
    outs = []
    for enabled in (True, False):
        if not enabled:
            monkeypatch.setattr(fused, "grouped_first_layer_available", lambda *a, **k: False)
        for p in sa.parameters():
            p.grad = None
        f = feats.clone().requires_grad_(True)
        new_xyz, new_feats = sa(xyz, f)
        new_feats.square().mean().backward()
        outs.append([new_feats.detach(), f.grad] + [p.grad.clone() for p in sa.parameters()])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-7


@pytest.mark.parametrize("cf,m,groups", [(3, 32, 4), (3, 64, 4), (1, 16, 0), (4, 32, 2)])
def test_direct_first_layer_is_one_chain_over_the_concatenated_channels(cf, m, groups):
    """ogc_group_linear_fwd_direct (few feature channels: an encoder's first level) against the layer as the reference runs it —
    QueryAndGroup's concatenation, then the 1x1 convolution over [rel (3), features (cf)] on this library's matrix kernel
    (ogc_conv1x1_gemm: v_mfma_f32_16x16x4_f32, input channels ascending): the same fused-multiply-add chain, so the same BITS;
    and its GroupNorm statistics against the values it stored."""
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator().manual_seed(100 * cf + m)
    B, N, npoint, nsample = 2, 1024, 256, 32
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
    new_xyz = xyz[:, :npoint].contiguous()
    feats = torch.randn(B, cf, N, generator=g).cuda()
    idx = torch.randint(0, N, (B, npoint, nsample), generator=g, dtype=torch.int32).cuda()
    w = torch.randn(m, 3 + cf, generator=g).cuda()
    T = npoint * nsample
    rel = torch.empty(B, 3, npoint, nsample, device="cuda")
    nat.group_concat_wrapper(B, 0, N, npoint, nsample, xyz, new_xyz, None, idx, rel)
    y = torch.empty(B, m, npoint, nsample, device="cuda")
    stats = torch.zeros(nat.conv1x1_gn_slots() * B * groups * 2, dtype=torch.float64, device="cuda") if groups else None
    nat.group_linear_fwd_direct_wrapper(B, m, cf, N, npoint, nsample, groups, feats, idx, rel, w, y, stats)
    grouped = torch.empty(B, 3 + cf, npoint, nsample, device="cuda")
    nat.group_concat_wrapper(B, cf, N, npoint, nsample, xyz, new_xyz, feats, idx, grouped)
    ref = torch.empty(B, m, npoint, nsample, device="cuda")
    nat.conv1x1_gemm_wrapper(B, m, 3 + cf, T, 0, w, grouped, ref)
    exact = torch.einsum("oc,bcps->bops", w.double(), grouped.double())
    print("cf=%d m=%d: %d of %d outputs differ in the last bits from the matrix kernel's" % (cf, m, int((y != ref).sum()), y.numel()))
    assert torch.equal(y, ref)
    assert ((y.double() - exact).norm() / exact.norm()).item() <= 1e-6
    if groups:
        s = stats.view(-1, B, groups, 2).sum(0)
        yg = y.double().view(B, groups, -1)
        # (a thread adds its outputs in fp32 before the fp64 tree: 1e-6 relative)
        torch.testing.assert_close(s[..., 0], yg.sum(-1), rtol=1e-6, atol=1e-3)
        torch.testing.assert_close(s[..., 1], (yg * yg).sum(-1), rtol=1e-6, atol=1e-3)
