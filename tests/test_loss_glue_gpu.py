"""csrc/loss_glue.hip: the per-view means of the loss terms with their weighted sum, and the masks' gradient as the sum of its
consumers' — against the framework operators they replace (losses/seg_loss_unsup.py `_stacked_terms`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def nat():
    from ogc_amd import pointnet2_cuda
    return pointnet2_cuda


@pytest.mark.parametrize("shapes", [[(4, 32768), (4, 32768), (4, 32768), (2, 32768), (2, 32768)], [(3, 1001), (1, 7)], [(5, 4)]])
def test_view_means_and_their_adjoint(nat, shapes):
    g = torch.Generator(device=DEV).manual_seed(sum(r * l for r, l in shapes))
    parts = [torch.randn(r * l, device=DEV, generator=g) * 3 + 1 for r, l in shapes]
    rows = [r for r, _ in shapes]
    v = torch.full((sum(rows),), 9.0, device=DEV)
    nat.view_means_wrapper(parts, rows, v)
    want = torch.cat([p.double().view(r, -1).mean(1) for p, r in zip(parts, rows)])
    assert (v.double() - want).abs().max().item() <= 1e-7 * (want.abs().max().item() + 1)
    weight = torch.randn(sum(rows), device=DEV, generator=g)
    g_loss = torch.tensor(1.7, device=DEV)
    grads = [torch.full_like(p, 5.0) for p in parts]
    nat.view_means_grad_wrapper(grads, rows, weight, g_loss)
    k = 0
    for got, (r, l) in zip(grads, shapes):
        ref = ((weight[k:k + r] * g_loss)[:, None].expand(r, l) / l).reshape(-1)   # MeanBackward on the weighted sum's gradient
        assert torch.equal(got, ref)
        k += r
    with pytest.raises(RuntimeError):
        nat.view_means_wrapper(parts, [r + 1 for r in rows], v)


@pytest.mark.parametrize("total,ranges", [(16 * 8192 * 10, [(0, 1), (0, 1), (0, 1), (0, 0.5), (0.5, 0.5)]),
                                           (1003, [(0, 1), (5 / 1003, 100 / 1003)]), (64, [(16 / 64, 16 / 64)]), (40, [])])
def test_sum_ranges(nat, total, ranges):
    g = torch.Generator(device=DEV).manual_seed(total)
    firsts = [int(round(a * total)) for a, _ in ranges]
    parts = [torch.randn(int(round(c * total)), device=DEV, generator=g) for _, c in ranges]
    out = torch.full((total,), 3.0, device=DEV)
    nat.sum_ranges_wrapper(parts, firsts, out)
    want = torch.zeros(total, device=DEV)
    for p, f in zip(parts, firsts):   # (the same order of additions: bit-equal)
        want[f:f + p.numel()] += p
    assert torch.equal(out, want)
    if parts:
        with pytest.raises(Exception):
            nat.sum_ranges_wrapper(parts, [total] + firsts[1:], out)


def test_stacked_loss_with_and_without_the_glue_kernels(monkeypatch):
    """UnsupervisedOGCLoss on stacked views: value, monitored terms and the masks' gradient with the glue kernels against the
    framework-operator form of the same algebra."""
    from ogc_amd.losses import seg_loss_unsup as L
    from ogc_amd.train_step import KITTI_LOSS, _SplitViews, _views, build_criterion
    from ogc_amd.utils.synthetic import make_scene_batch
    crit = build_criterion(KITTI_LOSS)
    batch = make_scene_batch(2, 2048, 10, seed=77, outdoor=True, aug=True, device=DEV)
    flat, pcs_l, flows_l, pcs_s, flows_s = _views(batch)
    b, t, n = batch[1].shape
    logits = torch.randn(b, t, n, 10, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    results = []
    for glue in (True, False):
        monkeypatch.setattr(L, "LOSS_GLUE", glue)
        x = logits.clone().requires_grad_(True)
        m = torch.softmax(x, -1)
        m.retain_grad()
        *masks_l, masks_s = _SplitViews.apply(m)
        loss, losses = crit(pcs_l, masks_l, flows_l, step_w=True, it=4000, aug_transform=True, sync=False,
                            stacked=(pcs_s, masks_s, flows_s))
        loss.backward()
        d = losses.resolve() if hasattr(losses, "resolve") else losses
        results.append((loss.detach().double().item(), m.grad.clone(), x.grad.clone(),
                        {k: float(d[k]) for k in ("dynamic", "smooth", "invariance", "sum")}))
    (la, ga, xa, da), (lb, gb, xb, db) = results
    assert abs(la - lb) <= 2e-6 * abs(lb)
    for k in da:
        assert abs(da[k] - db[k]) <= 2e-6 * abs(db[k]) + 1e-9, k
    # the masks' gradient: five consumers' gradients added in a fixed order by the kernel, pairwise in graph order by autograd
    # (measured 1.2e-7 of the largest entry; two runs of the SAME form differ by 3e-8 through the terms' atomic sums)
    worst, norm = (ga - gb).abs().max().item() / gb.abs().max().item(), (ga - gb).double().norm().item() / gb.double().norm().item()
    assert worst <= 1e-6 and norm <= 2e-7, (worst, norm)
    # behind the softmax (g - <m, g> cancels: the two runs of one form differ by 2e-6 of the largest entry there) only loosely
    assert (xa - xb).double().norm().item() <= 3e-4 * xb.double().norm().item()
