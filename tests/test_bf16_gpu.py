"""bf16-operand mode of the 1x1-convolution kernels (ogc_set_matmul_precision(1)): every variant against the same
product computed in fp64 from operands rounded to bf16 the way the kernels round them (nearest even) — i.e. the only
difference left is the fp32 accumulation order — and the fp32 mode unchanged after switching back."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16_mode():
    import ogc_amd  # noqa: F401
    from ogc_amd.pointnet2 import pointnet2 as api
    nat = api._native
    previous = nat.set_matmul_precision("bf16")
    try:
        yield nat
    finally:
        nat.set_matmul_precision(previous)


def r(t):
    return t.bfloat16().double()


def close(got, want, what, rel=2e-5):
    scale = want.abs().max().item() + 1e-30
    err = (got.double() - want).abs().max().item()
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("B,cin,cout,hw", [(2, 6, 32, 256), (3, 32, 64, 1024), (2, 99, 64, 512), (2, 131, 128, 256),
                                           (1, 128, 256, 192), (2, 160, 40, 128), (2, 64, 64, 64)])
def test_bf16_gemm_and_wgrad(bf16_mode, B, cin, cout, hw):
    nat = bf16_mode
    g = torch.Generator().manual_seed(B * 1000 + cin)
    x = torch.randn(B, cin, hw, generator=g).cuda()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).cuda()
    dy = torch.randn(B, cout, hw, generator=g).cuda()
    y = torch.empty(B, cout, hw, device="cuda")
    nat.conv1x1_gemm_wrapper(B, cout, cin, hw, 0, w, x, y)
    close(y, torch.einsum("mk,bkp->bmp", r(w), r(x)), "forward")
    if cout <= 160:
        dx = torch.empty_like(x)
        nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, dy, dx)
        close(dx, torch.einsum("mk,bmp->bkp", r(w), r(dy)), "input gradient")
    dw = torch.empty(cout, cin, device="cuda")
    nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, dw)
    close(dw, torch.einsum("bmp,bkp->mk", r(dy), r(x)), "weight gradient")
    # previous GroupNorm + ReLU folded into the operand load: the operand is rounded AFTER the affine map
    pa, pb = (torch.rand(B * cin, generator=g) + 0.5).cuda(), torch.randn(B * cin, generator=g).cuda()
    # the kernels apply the map as ONE fused multiply-add (a single rounding to fp32) before rounding to bf16
    act = torch.relu((pa.view(B, cin, 1).double() * x.double() + pb.view(B, cin, 1).double()).float())
    nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, 0, w, x, pa, pb, y, None)
    close(y, torch.einsum("mk,bkp->bmp", r(w), r(act)), "forward with folded norm")
    nat.conv1x1_wgrad_affine_wrapper(B, cin, cout, hw, 1, x, pa, pb, dy, dw)
    close(dw, torch.einsum("bmp,bkp->mk", r(dy), r(act)), "weight gradient with folded norm")
    if cin <= 100 and cout % 16 == 0:
        slots = nat.conv1x1_gn_slots()
        stats = torch.empty(slots * B * 4 * 2, dtype=torch.float64, device="cuda")
        nat.conv1x1_gemm_gnstats_wrapper(B, cout, cin, hw, 4, w, x, y, stats)
        want = torch.einsum("mk,bkp->bmp", r(w), r(x))
        close(y, want, "forward with statistics")
        s = stats.view(slots, B, 4, 2).sum(0)
        wg = want.reshape(B, 4, -1)
        close(s[..., 0], wg.sum(-1), "group sums", rel=1e-4 * float(wg.abs().sum(-1).max() / (wg.sum(-1).abs().max() + 1e-30)))
        close(s[..., 1], (wg * wg).sum(-1), "group sums of squares", rel=1e-4)


def test_fp32_mode_is_untouched_by_the_switch():
    import ogc_amd  # noqa: F401
    from ogc_amd.pointnet2 import pointnet2 as api
    nat = api._native
    x, w = torch.randn(2, 64, 256, device="cuda"), torch.randn(32, 64, device="cuda")
    y0, y1, yb = (torch.empty(2, 32, 256, device="cuda") for _ in range(3))
    nat.conv1x1_gemm_wrapper(2, 32, 64, 256, 0, w, x, y0)
    assert nat.set_matmul_precision("bf16") == "fp32"
    nat.conv1x1_gemm_wrapper(2, 32, 64, 256, 0, w, x, yb)
    assert nat.set_matmul_precision("fp32") == "bf16"
    nat.conv1x1_gemm_wrapper(2, 32, 64, 256, 0, w, x, y1)
    assert torch.equal(y0, y1)
    rel = ((yb - y0).abs().max() / y0.abs().max()).item()
    assert 1e-5 < rel < 3e-2, rel   # bf16 operands: visibly different, by about 2^-8 per product
    with pytest.raises(ValueError):
        nat.set_matmul_precision("fp16")


def test_shared_mlp_trains_in_bf16_mode(bf16_mode):
    """Forward + backward of a SharedMLP (deferred normalisation, fused pooling) in bf16 mode stay close to fp32."""
    from ogc_amd.utils.nn_util import SharedMLP
    torch.manual_seed(0)
    mlp = SharedMLP([6, 64, 64, 128], bn={"class": "GroupNorm", "num_groups": 4}).cuda()
    x = torch.randn(4, 6, 256, 64, device="cuda", requires_grad=True)
    outs = {}
    for mode in ("bf16", "fp32"):
        bf16_mode.set_matmul_precision(mode)
        for p in mlp.parameters():
            p.grad = None
        x.grad = None
        y = mlp.forward_maxpool(x)
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        outs[mode] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mlp.parameters()]
    bf16_mode.set_matmul_precision("bf16")
    # the arg-max of the pooling may move between near-equal neighbours, so gradients are compared in norm
    for a, b in zip(outs["bf16"], outs["fp32"]):
        assert torch.isfinite(a).all()
        assert ((a - b).norm() / (b.norm() + 1e-12)).item() < 0.1


@pytest.mark.parametrize("B,cin,cout,hw", [(4, 448, 256, 2048), (2, 131, 128, 4096), (16, 256, 128, 32768), (3, 67, 64, 1024), (2, 224, 64, 2048)])
def test_bf16_gemm_any(bf16_mode, B, cin, cout, hw):
    """ogc_conv1x1_gemm_any follows the operand switch since the end of round 5 (it kept fp32 operands before, so that a bf16
    configuration mixed the two): both orientations against fp64 on operands rounded to bf16."""
    nat = bf16_mode
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, cin, hw, generator=g).cuda()
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).cuda()
    y = torch.empty(B, cout, hw, device="cuda")
    nat.conv1x1_gemm_any_wrapper(B, cout, cin, hw, 0, w, x, y)
    close(y, torch.einsum("mk,bkp->bmp", r(w), r(x)), "forward")
    dy = torch.randn(B, cout, hw, generator=g).cuda()
    dx = torch.empty(B, cin, hw, device="cuda")
    nat.conv1x1_gemm_any_wrapper(B, cin, cout, hw, 1, w, dy, dx)
    close(dx, torch.einsum("mk,bmp->bkp", r(w), r(dy)), "input gradient")
