"""The fused mask read-out (ogc_slot_masks_fwd / _bwd) against the reference's op sequence
(models/segnet_kitti.py:85-88: normalize, einsum, / 0.05, softmax) evaluated in fp64, forward and both gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(feats, slots, w):
    f = feats.double().requires_grad_(True)
    s = slots.double().requires_grad_(True)
    logits = torch.einsum('bdn,bdk->bnk', F.normalize(f, dim=1), F.normalize(s, dim=1)) / 0.05
    mask = logits.softmax(dim=-1)
    (mask * w.double()).sum().backward()
    return mask.detach(), f.grad, s.grad


def _fused(feats, slots, w):
    from ogc_amd.fused import slot_masks, slot_masks_available
    f = feats.clone().requires_grad_(True)
    s = slots.clone().requires_grad_(True)
    assert slot_masks_available(f, s)
    mask = slot_masks(f, s, 0.05)
    (mask * w).sum().backward()
    return mask.detach(), f.grad, s.grad


def _close(got, want, tol):
    scale = want.abs().max().item() + 1e-30
    err = (got.double() - want).abs().max().item()
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("B,D,N,K", [(16, 64, 8192, 10), (2, 64, 1000, 8), (3, 32, 130, 17), (1, 100, 77, 1),
                                     (2, 16, 128, 32), (1, 256, 300, 32), (4, 64, 1, 5)])
def test_matches_the_op_sequence(B, D, N, K):
    import ogc_amd  # noqa: F401
    torch.manual_seed(B * 1000 + N + K)
    feats = torch.randn(B, D, N, device="cuda") * 0.7
    slots = torch.randn(B, D, K, device="cuda") * 1.3
    w = torch.randn(B, N, K, device="cuda")
    want = _reference(feats, slots, w)
    got = _fused(feats, slots, w)
    assert got[0].shape == (B, N, K)
    assert torch.allclose(got[0].sum(-1), torch.ones(B, N, device="cuda"), atol=1e-5)
    _close(got[0], want[0], 2e-5)
    _close(got[1], want[1], 5e-5)
    _close(got[2], want[2], 5e-5)


def test_zero_features_and_zero_slots_take_the_clamped_branch():
    """F.normalize divides by max(|x|, 1e-12): an all-zero point or slot gives cosines of 0 and a gradient g / eps."""
    import ogc_amd  # noqa: F401
    torch.manual_seed(5)
    feats = torch.randn(2, 64, 200, device="cuda")
    feats[0, :, 17] = 0.0
    feats[1, :, 199] = 0.0
    slots = torch.randn(2, 64, 6, device="cuda")
    slots[1, :, 3] = 0.0
    w = torch.randn(2, 200, 6, device="cuda")
    want = _reference(feats, slots, w)
    got = _fused(feats, slots, w)
    _close(got[0], want[0], 2e-5)
    assert torch.isfinite(got[1]).all() and torch.isfinite(got[2]).all()
    # the clamped rows have gradients ~1e12 times the others: compare them on their own scale
    for b, p in ((0, 17), (1, 199)):
        _close(got[1][b, :, p], want[1][b, :, p], 5e-5)
    live = torch.ones(2, 200, dtype=torch.bool, device="cuda")
    live[0, 17] = live[1, 199] = False
    _close(got[1].permute(0, 2, 1)[live], want[1].permute(0, 2, 1)[live], 5e-5)
    _close(got[2][1, :, 3], want[2][1, :, 3], 5e-5)
    keep = torch.ones(2, 6, dtype=torch.bool, device="cuda")
    keep[1, 3] = False
    _close(got[2].permute(0, 2, 1)[keep], want[2].permute(0, 2, 1)[keep], 5e-5)


def test_backward_repeats_bit_for_bit():
    import ogc_amd  # noqa: F401
    torch.manual_seed(9)
    feats = torch.randn(4, 64, 4096, device="cuda")
    slots = torch.randn(4, 64, 10, device="cuda")
    w = torch.randn(4, 4096, 10, device="cuda")
    a = _fused(feats, slots, w)
    b = _fused(feats, slots, w)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_model_uses_the_fused_read_out():
    """MaskFormer3D's forward goes through ogc_slot_masks_fwd on the GPU and agrees with the op sequence."""
    import ogc_amd  # noqa: F401
    from ogc_amd import fused
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    torch.manual_seed(3)
    net = MaskFormer3D(n_slot=6, n_point=512, transformer_embed_dim=128).cuda()
    pc = torch.rand(2, 512, 3, device="cuda") * 4
    calls = []
    orig = fused.slot_masks

    def spy(*a, **k):
        calls.append(1)
        return orig(*a, **k)

    fused.slot_masks = spy
    try:
        got = net(pc, pc)
    finally:
        fused.slot_masks = orig
    assert calls
    avail = fused.slot_masks_available
    fused.slot_masks_available = lambda *a: False
    try:
        want = net(pc, pc)
    finally:
        fused.slot_masks_available = avail
    assert torch.allclose(got, want, atol=2e-5), (got - want).abs().max()
