"""train_step._fused_adam_step (the fused Adam update without torch's per-parameter Python) against optimizer.step(): same
parameters and state, bit for bit, over steps that include a skipped one (found_inf = 1) and a state reload."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets():
    torch.manual_seed(3)
    a = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.GroupNorm(1, 13), torch.nn.Linear(13, 5)).cuda()
    return a, copy.deepcopy(a)


def test_fast_path_equals_optimizer_step():
    from ogc_amd.train_step import _fused_adam_step, make_optimizer
    a, b = _nets()
    oa, ob = make_optimizer(a.parameters(), 1e-2, weight_decay=1e-4), make_optimizer(b.parameters(), 1e-2, weight_decay=1e-4)
    used = []
    for step in range(6):
        x = torch.randn(4, 7, device="cuda")
        for net in (a, b):
            net.zero_grad(set_to_none=True)
            net(x).square().sum().backward()
        flag = torch.tensor(1.0 if step == 3 else 0.0, device="cuda")
        if step == 2:  # the learning rate moves (a scheduler), the state is reloaded (a resumed run)
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-3
            oa.load_state_dict(copy.deepcopy(oa.state_dict()))
        used.append(_fused_adam_step(oa, flag))
        if not used[-1]:
            oa.grad_scale, oa.found_inf = None, flag
            oa.step()
            del oa.grad_scale, oa.found_inf
        ob.grad_scale, ob.found_inf = None, flag
        ob.step()
        del ob.grad_scale, ob.found_inf
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.equal(pa, pb), step
    assert used[0] is False and all(used[1:]), used       # the first step builds the state through torch
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sa:
        for name in ("step", "exp_avg", "exp_avg_sq"):
            assert torch.equal(sa[k][name], sb[k][name]), (k, name)
    assert float(sa[0]["step"]) == 5.0                     # six steps, one skipped


def test_parameters_without_gradient_fall_back():
    from ogc_amd.train_step import _fused_adam_step, make_optimizer
    a, _ = _nets()
    opt = make_optimizer(a.parameters(), 1e-2)
    x = torch.randn(4, 7, device="cuda")
    a(x).sum().backward()
    opt.step()
    a.zero_grad(set_to_none=True)
    a[0](x).sum().backward()                               # only the first layer has gradients now
    assert _fused_adam_step(opt, None) is False


def test_adam_kernel_equals_optimizer_step():
    """train_step._adam_kernel_step (ogc_adam_step: the NaN rule and the update as two launches on torch's state tensors)
    against torch's fused optimizer.step(): parameters and state to fp32 rounding over steps that include a NaN gradient
    (skipped, step counts untouched), a learning-rate change and a state reload."""
    from ogc_amd.train_step import _adam_kernel_step, make_optimizer
    a, b = _nets()
    oa, ob = make_optimizer(a.parameters(), 1e-2, weight_decay=1e-4), make_optimizer(b.parameters(), 1e-2, weight_decay=1e-4)
    used = []
    for step in range(7):
        x = torch.randn(4, 7, device="cuda")
        for net in (a, b):
            net.zero_grad(set_to_none=True)
            net(x).square().sum().backward()
        if step == 3:
            for net in (a, b):
                net[0].weight.grad[2, 3] = float("nan")
        if step == 2:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 3e-3
            oa.load_state_dict(copy.deepcopy(oa.state_dict()))
        flag = _adam_kernel_step(oa)
        used.append(flag is not None)
        if flag is None:
            oa.step()
        else:
            assert int(flag.item()) == (1 if step == 3 else 0)
        grads = [p.grad for p in b.parameters()]
        bad = torch.isnan(torch.stack(torch._foreach_norm(grads)).sum())
        ob.grad_scale, ob.found_inf = None, bad.float().reshape(())
        ob.step()
        del ob.grad_scale, ob.found_inf
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7), (step, float((pa - pb).abs().max()))
    assert used[0] is False and all(used[1:]), used
    # the kernel evaluates ATen's expressions (double hyper-parameters against fp32 state, rounded once per assignment): after
    # seven steps nearly every parameter is the same fp32 number (what is left is the contraction of ATen's own build)
    same = sum(int((pa == pb).sum()) for pa, pb in zip(a.parameters(), b.parameters()))
    total = sum(pa.numel() for pa in a.parameters())
    print("adam kernel vs torch fused: %d of %d parameters bit-identical after 7 steps" % (same, total))
    assert same >= 0.9 * total, (same, total)
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k in sa:
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == 6.0
        for name in ("exp_avg", "exp_avg_sq"):
            assert torch.allclose(sa[k][name], sb[k][name], rtol=1e-5, atol=2e-6), (k, name, float((sa[k][name] - sb[k][name]).abs().max()))
