"""ogc_chamfer_terms / _grad (csrc/chamfer.hip) against the reference's operator sequence (losses/flow_loss_unsup.py:15-35: two
1-NN searches, gathers, differences, norms) on the same inputs: the distance terms bit for bit, the gradient w.r.t. the flow
to fp32 rounding of the atomically accumulated sums; duplicates (zero distances), both norms, clouds of different sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sequence(pc1, pc2, flow, norm):
    """The reference's statements on this package's operators (what ChamferLoss ran until round 5)."""
    from ogc_amd.pointnet2.pointnet2 import grouping_operation, knn
    pc2 = pc2.contiguous()
    pc2_t = pc2.transpose(1, 2).contiguous()
    p1 = (pc1 + flow).contiguous()
    p1_t = p1.transpose(1, 2).contiguous()
    _, idx = knn(1, p1, pc2)
    dist1 = (p1_t - grouping_operation(pc2_t, idx.detach()).squeeze(-1)).norm(p=norm, dim=1)
    _, idx = knn(1, pc2, p1)
    dist2 = (pc2_t - grouping_operation(p1_t, idx.detach()).squeeze(-1)).norm(p=norm, dim=1)
    return dist1, dist2


@pytest.mark.parametrize("norm", [1, 2])
@pytest.mark.parametrize("B,n1,n2,scale", [(2, 2048, 2048, (60, 4, 80)), (3, 700, 1300, (1, 1, 1)), (1, 8192, 8192, (60, 4, 80))])
def test_terms_and_gradient_match_the_operator_sequence(B, n1, n2, scale, norm):
    from ogc_amd.losses.flow_loss_unsup import _ChamferTerms
    from ogc_amd.pointnet2.pointnet2 import knn
    g = torch.Generator().manual_seed(n1 + norm)
    sc = torch.tensor(scale, dtype=torch.float32)
    pc1 = ((torch.rand(B, n1, 3, generator=g) - 0.5) * sc).cuda()
    pc2 = ((torch.rand(B, n2, 3, generator=g) - 0.5) * sc).cuda()
    flow = (torch.randn(B, n1, 3, generator=g) * 0.05 * sc).cuda()
    k = min(n1, n2) // 8
    pc2[:, :k] = (pc1 + flow)[:, :k]                      # exact hits: zero distances, the gradient's 0 / 0 case
    weights1 = torch.rand(B, n1, generator=g).cuda()
    weights2 = torch.rand(B, n2, generator=g).cuda()
    f_ref = flow.clone().requires_grad_(True)
    d1_ref, d2_ref = _sequence(pc1, pc2, f_ref, norm)
    ((d1_ref * weights1).sum() + (d2_ref * weights2).sum()).backward()
    f_new = flow.clone().requires_grad_(True)
    warped = (pc1 + f_new).contiguous()
    idx12 = knn(1, warped, pc2)[1].squeeze(-1).contiguous()
    idx21 = knn(1, pc2, warped)[1].squeeze(-1).contiguous()
    d1, d2 = _ChamferTerms.apply(warped, pc2.contiguous(), idx12, idx21, norm)
    ((d1 * weights1).sum() + (d2 * weights2).sum()).backward()
    assert torch.equal(d1, d1_ref) and torch.equal(d2, d2_ref)
    assert bool((d1[:, :k] == 0).all())
    scale_g = float(f_ref.grad.abs().max())
    assert float((f_new.grad - f_ref.grad).abs().max()) <= 1e-5 * scale_g, (float((f_new.grad - f_ref.grad).abs().max()), scale_g)
    assert bool(torch.isfinite(f_new.grad).all())


def test_chamfer_loss_module_uses_the_fused_terms_and_matches():
    from ogc_amd.losses.flow_loss_unsup import ChamferLoss
    g = torch.Generator().manual_seed(4)
    pc1 = ((torch.rand(2, 4096, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda()
    pc2 = (pc1[:, torch.randperm(4096, generator=g)] + 0.05 * torch.randn(2, 4096, 3, generator=g).cuda()).contiguous()
    flow = (0.1 * torch.randn(2, 4096, 3, generator=g)).cuda().requires_grad_(True)
    loss = ChamferLoss(loss_norm=2)(pc1, pc2, flow)
    loss.backward()
    f2 = flow.detach().clone().requires_grad_(True)
    d1, d2 = _sequence(pc1, pc2, f2, 2)
    want = (d1 + d2).mean()
    want.backward()
    assert float(abs(loss - want)) <= 1e-6 * float(want)
    assert float((flow.grad - f2.grad).abs().max()) <= 1e-5 * float(f2.grad.abs().max())
