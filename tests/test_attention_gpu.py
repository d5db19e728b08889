"""The fused attention core (ogc_attention_fwd / _bwd) against nn.MultiheadAttention itself: same module, same
parameters, outputs and every gradient (inputs, packed projection weights and biases, output projection)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mha, q, k, fused):
    from ogc_amd.fused import multihead_attention
    for p in mha.parameters():
        p.grad = None
    q = q.clone().requires_grad_(True)
    k = q if k is None else k.clone().requires_grad_(True)
    out = multihead_attention(mha, q, k, k) if fused else mha(q, k, k, need_weights=True)[0]
    w = torch.linspace(-1.0, 1.0, out.numel(), device=out.device).view_as(out)
    (out * w).sum().backward()
    grads = [q.grad] + ([] if k is q else [k.grad]) + [p.grad.clone() for p in mha.parameters()]
    return out.detach(), grads


@pytest.mark.parametrize("B,Lq,Lk,E,H", [(16, 10, 512, 128, 8), (3, 10, None, 128, 8), (2, 7, 33, 256, 8),
                                         (1, 16, 900, 128, 8), (2, 1, 5, 64, 2), (4, 8, None, 256, 8)])
def test_matches_multihead_attention(B, Lq, Lk, E, H):
    import ogc_amd  # noqa: F401
    from ogc_amd.fused import attention_core_available
    torch.manual_seed(B * 100 + Lq)
    mha = torch.nn.MultiheadAttention(E, H, batch_first=True).cuda()
    with torch.no_grad():
        mha.in_proj_bias.normal_(0, 0.2)
        mha.out_proj.bias.normal_(0, 0.2)
    q = torch.randn(B, Lq, E, device="cuda")
    k = None if Lk is None else torch.randn(B, Lk, E, device="cuda") * 1.5
    assert attention_core_available(E, H, Lq, Lq if Lk is None else Lk, q)
    want, gwant = _run(mha, q, k, fused=False)
    got, ggot = _run(mha, q, k, fused=True)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), (got - want).abs().max()
    assert len(ggot) == len(gwant)
    for a, b in zip(ggot, gwant):
        scale = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-6, ((a - b).abs().max().item(), scale)


def test_falls_back_outside_the_supported_shapes():
    import ogc_amd  # noqa: F401
    from ogc_amd.fused import attention_core_available, multihead_attention
    mha = torch.nn.MultiheadAttention(64, 8, batch_first=True).cuda()          # heads of 8 columns: not offered
    q = torch.randn(2, 5, 64, device="cuda")
    assert not attention_core_available(64, 8, 5, 5, q)
    assert torch.allclose(multihead_attention(mha, q, q, q), mha(q, q, q, need_weights=False)[0], atol=1e-6)
    big = torch.nn.MultiheadAttention(128, 8, batch_first=True).cuda()
    q, k = torch.randn(1, 16, 128, device="cuda"), torch.randn(1, 4096, 128, device="cuda")
    assert not attention_core_available(128, 8, 16, 4096, q, k)                 # backward would not fit in LDS
    assert torch.allclose(multihead_attention(big, q, k, k), big(q, k, k, need_weights=False)[0], atol=1e-5)


def test_probabilities_are_distributions():
    import ogc_amd  # noqa: F401
    from ogc_amd.pointnet2 import pointnet2 as api
    B, L, N, E, H = 2, 10, 300, 128, 8
    q, kv = torch.randn(B, L, E, device="cuda"), torch.randn(B, N, 2 * E, device="cuda")
    out = torch.empty(B, L, E, device="cuda")
    prob = torch.empty(B, H, L, N, device="cuda")
    api._native.attention_fwd_wrapper(H, 0.25, q, kv[:, :, :E], kv[:, :, E:], out, prob)
    assert torch.allclose(prob.sum(-1), torch.ones(B, H, L, device="cuda"), atol=1e-5) and (prob >= 0).all()
    ref = torch.softmax(torch.einsum("blhd,bnhd->bhln", q.view(B, L, H, 16), kv[:, :, :E].reshape(B, N, H, 16)) * 0.25, -1)
    assert torch.allclose(prob, ref, rtol=1e-4, atol=1e-6)


def test_many_rows_linear_matches_linear():
    """The key / value and memory projections (16 x 512 rows) with the batch-split weight gradient."""
    import ogc_amd  # noqa: F401
    from ogc_amd.fused import _ManyRowsLinear, many_rows_linear
    torch.manual_seed(11)
    x = torch.randn(16, 512, 128, device="cuda")
    lin = torch.nn.Linear(128, 256).cuda()
    w = torch.randn(16, 512, 256, device="cuda")
    outs = []
    for fn in (torch.nn.functional.linear, many_rows_linear):
        xi = x.clone().requires_grad_(True)
        lin.zero_grad()
        y = fn(xi, lin.weight, lin.bias)
        (y * w).sum().backward()
        outs.append((y.detach(), xi.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
    assert isinstance(many_rows_linear(x.requires_grad_(True), lin.weight, lin.bias).grad_fn, _ManyRowsLinear.apply(
        x, lin.weight, lin.bias).grad_fn.__class__)
    for a, b in zip(outs[1], outs[0]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale, ((a - b).abs().max().item(), scale)


@pytest.mark.parametrize("shape,n_out,bias", [((16, 10, 128), 128, True), ((16, 10, 128), 384, True), ((3, 7, 50), 33, False),
                                              ((1, 128), 64, True), ((1000, 96), 130, True), ((2, 1, 1, 5), 1, True)])
def test_small_linear_matches_linear(shape, n_out, bias):
    """ogc_small_linear_fwd / _bwd against F.linear: output, input gradient, weight and bias gradients."""
    import ogc_amd  # noqa: F401
    from ogc_amd.fused import _SmallLinear, small_linear
    torch.manual_seed(sum(shape) + n_out)
    x = torch.randn(*shape, device="cuda")
    lin = torch.nn.Linear(shape[-1], n_out, bias=bias).cuda()
    w = torch.randn(*shape[:-1], n_out, device="cuda")
    outs = []
    for fn in (torch.nn.functional.linear, small_linear):
        xi = x.clone().requires_grad_(True)
        lin.zero_grad()
        y = fn(xi, lin.weight, lin.bias)
        if fn is small_linear:
            assert isinstance(y.grad_fn, type(_SmallLinear.apply(xi, lin.weight, lin.bias).grad_fn))
        (y * w).sum().backward()
        outs.append([y.detach(), xi.grad, lin.weight.grad.clone()] + ([lin.bias.grad.clone()] if bias else []))
    for a, b in zip(outs[1], outs[0]):
        assert a.shape == b.shape
        scale = b.abs().max().item() + 1e-30
        assert (a.double() - b.double()).abs().max().item() <= 2e-5 * scale, ((a - b).abs().max().item(), scale)


def test_small_linear_partial_gradients():
    """Frozen weight (db and dx only), frozen input, nothing but the bias."""
    import ogc_amd  # noqa: F401
    from ogc_amd.fused import small_linear
    torch.manual_seed(2)
    x = torch.randn(16, 10, 128, device="cuda")
    W = torch.randn(70, 128, device="cuda")
    b = torch.randn(70, device="cuda")
    g = torch.randn(16, 10, 70, device="cuda")
    for rx, rw, rb in ((True, False, True), (False, True, True), (False, False, True), (True, False, False)):
        res = []
        for fn in (torch.nn.functional.linear, small_linear):
            xi, Wi, bi = x.clone().requires_grad_(rx), W.clone().requires_grad_(rw), b.clone().requires_grad_(rb)
            fn(xi, Wi, bi).backward(g)
            res.append([t.grad for t in (xi, Wi, bi)])
        for a, r in zip(res[1], res[0]):
            assert (a is None) == (r is None)
            if r is not None:
                assert (a - r).abs().max().item() <= 2e-5 * r.abs().max().item()
