"""16-bit activations (the `_h` entry points of include/ogc_ops.h, csrc/act_io.h): every kernel that streams a raw convolution
output or its gradient, with those tensors stored as bf16, against the fp32 entry point of the same name fed the SAME values
(the bf16 tensor widened) under the same operand precision.  Where the arithmetic in front of the store is the same instruction
sequence the 16-bit output must be the fp32 output rounded to nearest even, bit for bit; reductions that finish with atomics
are compared to the last bits; the pooled forms (bf16 operands instead of fp32 ones) against fp64 with a bf16 tolerance.  Then
one set-abstraction level and the C2 network with the switch on against off."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture()
def nat():
    import ogc_amd  # noqa: F401
    from ogc_amd.pointnet2 import pointnet2 as api
    n = api._native
    previous = n.set_matmul_precision("bf16")
    try:
        yield n
    finally:
        n.set_matmul_precision(previous)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def rel(got, want):
    return float((got.double() - want.double()).norm() / (want.double().norm() + 1e-30))


@pytest.mark.parametrize("B,cin,cout,P,S,groups", [(2, 64, 64, 64, 64, 4), (3, 64, 128, 32, 32, 4), (2, 128, 128, 16, 64, 4),
                                                   (2, 128, 256, 8, 16, 4), (1, 6, 32, 4, 16, 4)])
def test_gemm_affine_h(nat, B, cin, cout, P, S, groups):
    hw = P * S
    xh = rnd(B, cin, hw, seed=cin).to(BF)
    x = xh.float()
    w = rnd(cout, cin, seed=1, scale=cin ** -0.5)
    pa, pb = torch.rand(B * cin, device=DEV) + 0.5, rnd(B * cin, seed=2)
    slots = nat.conv1x1_gn_slots()
    # without statistics: the rounded fp32 output
    y = torch.empty(B, cout, hw, device=DEV)
    yh = torch.empty(B, cout, hw, device=DEV, dtype=BF)
    nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, 0, w, x, pa, pb, y, None)
    nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, 0, w, xh, pa, pb, yh, None)
    assert torch.equal(yh, y.to(BF))
    # with statistics: same output, sums of the STORED values
    st = torch.zeros(slots * B * groups * 2, dtype=torch.float64, device=DEV)
    yh2 = torch.empty_like(yh)
    nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, groups, w, xh, pa, pb, yh2, st)
    assert torch.equal(yh2, yh)
    sums = st.view(slots, B, groups, 2).sum(0)
    v = yh.double().view(B, groups, -1)
    assert torch.allclose(sums[..., 0], v.sum(-1), rtol=1e-6, atol=1e-3 * v.abs().sum(-1).max().item())
    assert torch.allclose(sums[..., 1], (v * v).sum(-1), rtol=1e-6)
    # pooled tail: extremes of the stored values, first position among equals
    gamma = rnd(cout, seed=5)
    st2 = torch.zeros_like(st)
    yext = torch.empty(B, cout, P, device=DEV)
    aext = torch.empty(B, cout, P, dtype=torch.int32, device=DEV)
    yh3 = torch.empty_like(yh)
    nat.conv1x1_gemm_affine_pool_wrapper(B, cout, cin, hw, 1, groups, S, w, xh, pa, pb, gamma, yh3, st2, yext, aext)
    assert torch.equal(yh3, yh)
    assert torch.allclose(st2.view(slots, B, groups, 2).sum(0), sums, rtol=1e-9)
    v4 = yh.float().view(B, cout, P, S)
    sg = torch.where(gamma < 0, -1.0, 1.0).view(1, cout, 1, 1)
    hi, ai = (v4 * sg).max(dim=3)
    assert torch.equal(yext, hi * sg[..., 0])
    first = ((v4 * sg) == hi.unsqueeze(-1)).int().argmax(dim=3).int()
    assert torch.equal(aext, first)
    # plain input-gradient orientation
    g = rnd(B, cout, hw, seed=7).to(BF)
    if cout <= 160:
        dz, dzh = torch.empty(B, cin, hw, device=DEV), torch.empty(B, cin, hw, device=DEV, dtype=BF)
        nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, g.float(), dz)
        nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, g, dzh)
        assert torch.equal(dzh, dz.to(BF))


def _sparse(nat, B, cout, P, S, groups, yh, seed):
    """coef2 / inj of a pooled GroupNorm behind y (any valid values do: the consumers only rebuild from them)."""
    coef2 = rnd(B, cout, 2, seed=seed, scale=0.1)
    inj = torch.empty(B, cout, P, 2, device=DEV)
    inj[..., 0] = rnd(B, cout, P, seed=seed + 1)
    arg = torch.randint(0, S, (B, cout, P), device=DEV, dtype=torch.int32)
    inj[..., 1] = arg.view(torch.float32)
    gy = coef2[:, :, None, None, 0].double() * yh.double().view(B, cout, P, S) + coef2[:, :, None, None, 1].double()
    gy.scatter_add_(3, arg.long().unsqueeze(-1), inj[..., 0:1].double())
    return coef2.contiguous(), inj.contiguous(), gy.view(B, cout, P * S)


@pytest.mark.parametrize("B,cin,cout,P,S", [(2, 64, 64, 64, 64), (2, 64, 32, 32, 32), (3, 32, 64, 16, 16)])
def test_moment_path_h(nat, B, cin, cout, P, S):
    hw, groups = P * S, 4
    yph = rnd(B, cin, hw, seed=3).to(BF)
    gyh = rnd(B, cout, hw, seed=4).to(BF)
    w = rnd(cout, cin, seed=1, scale=cin ** -0.5)
    pa, pb = torch.rand(B * cin, device=DEV) + 0.5, rnd(B * cin, seed=2, scale=0.3)
    m32, m16 = torch.zeros(B, 2, cout, cin, device=DEV), torch.zeros(B, 2, cout, cin, device=DEV)
    nat.conv1x1_wgrad_moments_wrapper(B, cin, cout, hw, 1, yph.float(), pa, pb, gyh.float(), m32)
    nat.conv1x1_wgrad_moments_wrapper(B, cin, cout, hw, 1, yph, pa, pb, gyh, m16)
    assert rel(m16, m32) < 2e-6
    # input gradient + adjoint: bf16 operands in the 16-bit form (the weights are rounded; the stored gradient already is) ->
    # against the fp32 kernel fed the ROUNDED weights: same products, another summation order, then one rounding to bf16
    coef = rnd(B, cin, 3, seed=9, scale=0.2)
    wr = w.to(BF).float()
    o32 = torch.empty(B, cin, hw, device=DEV)
    o16 = torch.empty(B, cin, hw, device=DEV, dtype=BF)
    nat.set_matmul_precision("fp32")
    nat.conv1x1_dgrad_adjoint_wrapper(B, cin, cout, hw, 1, wr, gyh.float(), yph.float(), pa, pb, coef, o32)
    nat.set_matmul_precision("bf16")
    nat.conv1x1_dgrad_adjoint_wrapper(B, cin, cout, hw, 1, w, gyh, yph, pa, pb, coef, o16)
    assert rel(o16, o32) < 3e-3 and float((o16 != o32.to(BF)).float().mean()) < 0.02
    # pooled forms: y in place of the dense gradient, rebuilt in fp32 and rounded to the operand precision
    coef2, inj, gy = _sparse(nat, B, cout, P, S, groups, gyh, 11)
    m32.zero_(), m16.zero_()
    nat.conv1x1_wgrad_moments_pooled_wrapper(B, cin, cout, hw, 1, S, yph.float(), pa, pb, gyh.float(), coef2, inj, m32)
    nat.conv1x1_wgrad_moments_pooled_wrapper(B, cin, cout, hw, 1, S, yph, pa, pb, gyh, coef2, inj, m16)
    assert rel(m16, m32) < 3e-3
    nat.conv1x1_dgrad_adjoint_pooled_wrapper(B, cin, cout, hw, 1, S, w, gyh.float(), coef2, inj, yph.float(), pa, pb, coef, o32)
    nat.conv1x1_dgrad_adjoint_pooled_wrapper(B, cin, cout, hw, 1, S, w, gyh, coef2, inj, yph, pa, pb, coef, o16)
    assert rel(o16, o32) < 6e-3


@pytest.mark.parametrize("B,cin,cout,P,S", [(16, 64, 128, 64, 64), (32, 128, 256, 32, 64), (16, 128, 128, 64, 16)])
def test_wide_path_h(nat, B, cin, cout, P, S):
    """The kernels behind a wide layer / a wide pooled tail: weight gradient (register tiles and the shared 128 x 128 tile), the
    pooled input gradient, the GroupNorm backward pair."""
    hw, groups = P * S, 4
    xh = rnd(B, cin, hw, seed=3).to(BF)
    yh = rnd(B, cout, hw, seed=4).to(BF)
    w = rnd(cout, cin, seed=1, scale=cin ** -0.5)
    pa, pb = torch.rand(B * cin, device=DEV) + 0.5, rnd(B * cin, seed=2, scale=0.3)
    act = torch.relu(pa.view(B, cin, 1) * xh.float() + pb.view(B, cin, 1))
    r = lambda t: t.to(BF).double()
    # dense weight gradient, folded norm: same operands (bf16) either way
    d32, d16 = torch.zeros(cout, cin, device=DEV), torch.zeros(cout, cin, device=DEV)
    nat.conv1x1_wgrad_affine_wrapper(B, cin, cout, hw, 1, xh.float(), pa, pb, yh.float(), d32)
    nat.conv1x1_wgrad_affine_wrapper(B, cin, cout, hw, 1, xh, pa, pb, yh, d16)
    assert rel(d16, d32) < 5e-6
    # pooled weight / input gradient: bf16 operands in the 16-bit form -> against fp64 on rounded operands
    coef2, inj, gy = _sparse(nat, B, cout, P, S, groups, yh, 21)
    nat.conv1x1_wgrad_affine_pooled_wrapper(B, cin, cout, hw, 1, S, xh, pa, pb, yh, coef2, inj, d16)
    want = torch.einsum("bmp,bkp->mk", r(gy.float()), r(act))
    assert rel(d16, want) < 2e-3
    if B * (hw // 64) >= 1024:
        gz = torch.empty(B, cin, hw, device=DEV, dtype=BF)
        nat.conv1x1_dgrad_pooled_wrapper(B, cin, cout, hw, S, w, yh, coef2, inj, gz)
        want = torch.einsum("mk,bmp->bkp", r(w), r(gy.float()))
        assert rel(gz, want) < 4e-3     # (the output's own rounding: 2^-9)
    # GroupNorm backward pair
    gamma, beta = rnd(cin, seed=5) + 1.5, rnd(cin, seed=6)
    mean, rstd = rnd(B * groups, seed=7, scale=0.1), torch.rand(B * groups, device=DEV) + 0.5
    gzh = rnd(B, cin, hw, seed=8).to(BF)
    ws = nat.group_norm_ws(B, cin, groups, True, torch.device(DEV))
    dx32, dx16 = torch.empty(B, cin, hw, device=DEV), torch.empty(B, cin, hw, device=DEV, dtype=BF)
    gw, gb, gw2, gb2 = (torch.empty(cin, device=DEV) for _ in range(4))
    nat.group_norm_bwd_wrapper(B, cin, hw, groups, 1, xh.float(), gamma, beta, mean, rstd, gzh.float(), dx32, gw, gb, ws)
    nat.group_norm_bwd_wrapper(B, cin, hw, groups, 1, xh, gamma, beta, mean, rstd, gzh, dx16, gw2, gb2, ws)
    assert torch.equal(dx16, dx32.to(BF)) and torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_grouped_first_layer_h(nat):
    B, N, P, S, M, groups = 2, 1024, 256, 64, 64, 4
    T = P * S
    Pm = rnd(B, M, N, seed=1)
    idx = torch.randint(0, N, (B, P, S), device=DEV, dtype=torch.int32)
    relc = rnd(B, 3, P, S, seed=2, scale=0.1)
    wx = rnd(M, 3, seed=3)
    slots = nat.conv1x1_gn_slots()
    y, yh = torch.empty(B, M, P, S, device=DEV), torch.empty(B, M, P, S, device=DEV, dtype=BF)
    st, sth = (torch.zeros(slots * B * groups * 2, dtype=torch.float64, device=DEV) for _ in range(2))
    nat.group_linear_fwd_wrapper(B, M, N, P, S, groups, Pm, idx, relc, wx, y, st)
    nat.group_linear_fwd_wrapper(B, M, N, P, S, groups, Pm, idx, relc, wx, yh, sth)
    assert torch.equal(yh, y.to(BF))
    v = yh.double().view(B, groups, -1)
    sums = sth.view(slots, B, groups, 2).sum(0)
    assert torch.allclose(sums[..., 1], (v * v).sum(-1), rtol=1e-6)
    assert torch.allclose(sums[..., 0], v.sum(-1), rtol=1e-6, atol=1e-3 * v.abs().sum(-1).max().item())
    # P point-major: the same y bit for bit, the same statistics up to the order of their additions
    for Mv, gv in ((64, 4), (128, 4), (64, 1), (128, 8)):
        Pv = rnd(B, Mv, N, seed=7)
        wv = rnd(Mv, 3, seed=8)
        ya, yb_ = (torch.empty(B, Mv, P, S, device=DEV, dtype=BF) for _ in range(2))
        sa, sb = (torch.zeros(slots * B * gv * 2, dtype=torch.float64, device=DEV) for _ in range(2))
        nat.group_linear_fwd_wrapper(B, Mv, N, P, S, gv, Pv, idx, relc, wv, ya, sa)
        nat.group_linear_fwd_pt_wrapper(B, Mv, N, P, S, gv, Pv.transpose(1, 2).contiguous(), idx, relc, wv, yb_, sb)
        assert torch.equal(ya, yb_)
        ra, rb_ = sa.view(slots, B, gv, 2).sum(0), sb.view(slots, B, gv, 2).sum(0)
        assert torch.allclose(ra[..., 1], rb_[..., 1], rtol=1e-6) and torch.allclose(ra[..., 0], rb_[..., 0], rtol=1e-6, atol=1e-5 * float(ya.float().abs().sum() / (B * gv)))
    # gradient: gather over the transposed lists, and the three coordinate columns of the weight gradient
    from ogc_amd import fused
    rev = fused.group_reverse(idx, N)
    gh = rnd(B, M, P, S, seed=4).to(BF)
    d32, d16 = torch.empty(B, M, N, device=DEV), torch.empty(B, M, N, device=DEV)
    nat.group_points_grad_rev_wrapper(B, M, N, P, S, gh.float(), rev[0], rev[1], rev[2], d32)
    nat.group_points_grad_rev_wrapper(B, M, N, P, S, gh, rev[0], rev[1], rev[2], d16)
    assert torch.equal(d16, d32)
    w32, w16 = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV)
    nat.conv1x1_wgrad_wrapper(B, 3, M, T, relc, gh.float(), w32)
    nat.conv1x1_wgrad_wrapper(B, 3, M, T, relc, gh, w16)
    assert rel(w16, w32) < 5e-6
    # both from ONE pass over the gradient (ogc_group_points_grad_rev_dwx): the same dP bit for bit, dwx as the fp64 sum (fp32
    # operands here; the separate weight-gradient kernel rounds rel to bf16 under this precision)
    want = torch.einsum("bmt,bkt->mk", gh.double().view(B, M, T), relc.double().view(B, 3, T))
    for gt in (gh, gh.float()):
        dp, dwx = torch.empty(B, M, N, device=DEV), torch.empty(M, 3, device=DEV)
        nat.group_points_grad_rev_dwx_wrapper(B, M, N, P, S, gt, rev[0], rev[1], rev[2], relc, dp, dwx)
        assert torch.equal(dp, d32)
        assert rel(dwx, want) < 1e-5


def test_pooled_sums_h(nat):
    """ogc_group_norm_maxpool_bwd_sparse_h: y is only gathered for channels whose scale is zero."""
    B, C, P, S, groups = 2, 64, 128, 32, 4
    yh = rnd(B, C, P, S, seed=1).to(BF)
    gamma = rnd(C, seed=2)
    gamma[3] = 0.0
    mean, rstd = rnd(B * groups, seed=3, scale=0.1), torch.rand(B * groups, device=DEV) + 0.5
    arg = torch.randint(0, S, (B, C, P), device=DEV, dtype=torch.int32)
    yext = yh.float().gather(3, arg.long().unsqueeze(-1)).squeeze(-1).contiguous()
    out = torch.relu(rnd(B, C, P, seed=4))
    gout = rnd(B, C, P, seed=5)
    res = []
    for y in (yh.float(), yh):
        coef2, inj = torch.empty(B, C, 2, device=DEV), torch.empty(B, C, P, 2, device=DEV)
        gw, gb = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        ws = nat.group_norm_ws(B, C, groups, True, torch.device(DEV))
        nat.group_norm_maxpool_bwd_sparse_wrapper(B, C, P, S, groups, 1, y, gamma, mean, rstd, out, arg, gout, coef2, inj, gw, gb,
                                                  ws, yext)
        res.append((coef2, inj, gw, gb))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def _sa_level(act16, seed=0):
    from ogc_amd import fused
    from ogc_amd.utils.pointnet2_util import PointnetSAModuleMSG
    torch.manual_seed(seed)
    sa = PointnetSAModuleMSG(npoint=256, radii=[0.2, 0.4], nsamples=[64, 64], mlps=[[6, 64, 64, 64], [6, 64, 64, 128]],
                             bn={"class": "GroupNorm", "num_groups": 4}).to(DEV)
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(16, 512, 3, generator=g).to(DEV)    # (fan-in 32 >= 24: the gather form of the grouping gradient)
    feats = torch.randn(16, 6, 512, generator=g).to(DEV).requires_grad_(True)
    saved = fused.ACT16
    fused.ACT16 = act16
    fused.GATE_MISSES.clear()
    try:
        _, out = sa(xyz, feats)
        (out * torch.sin(torch.arange(out.numel(), device=DEV).view_as(out) * 0.37)).sum().backward()
    finally:
        fused.ACT16 = saved
    grads = torch.cat([p.grad.flatten() for p in sa.parameters()])
    return out.detach(), feats.grad.detach(), grads, dict(fused.GATE_MISSES)


def test_set_abstraction_level_16_bit_against_32(nat):
    """One multi-scale level, forward and backward, three ways: fp32 everything / bf16 operands (rounds 2-4) / bf16 operands and
    16-bit activations.  The gradients of this toy objective (random signs on the pooled outputs) are dominated by ReLU and
    max-pool gates that any 2^-9 rounding flips — bf16 OPERANDS alone move them by ~10 % — so the 16-bit run is held to the same
    scale: at most 1.6 x the distance the operand rounding already puts between itself and fp32 (measured: 1.3 x)."""
    nat.set_matmul_precision("fp32")
    o0, f0, g0, _ = _sa_level(False)
    nat.set_matmul_precision("bf16")
    o1, f1, g1, _ = _sa_level(False)
    o2, f2, g2, misses = _sa_level(True)
    assert "act16_leave" not in misses, misses
    assert o2.dtype == torch.float32 and f2.dtype == torch.float32
    assert rel(o2, o1) > 1e-5                                   # 16 bits were in effect
    assert rel(o2, o0) < 1e-2 and rel(o2, o0) < 2.0 * rel(o1, o0)
    assert rel(f2, f0) < 1.6 * rel(f1, f0) and rel(g2, g0) < 1.6 * rel(g1, g0), (rel(f2, f0), rel(f1, f0), rel(g2, g0), rel(g1, g0))
    cos = float((g2.double() * g0.double()).sum() / (g2.double().norm() * g0.double().norm()))
    assert cos > 0.98, cos


def test_activations_are_stored_in_16_bits(nat):
    """The saved tensors of the level's nodes: the big ones are bf16 (half the bytes of the fp32 run)."""
    from ogc_amd import fused
    from ogc_amd.utils.pointnet2_util import PointnetSAModule
    torch.manual_seed(0)
    sa = PointnetSAModule(mlp=[6, 64, 64, 128], npoint=256, radius=0.3, nsample=64,
                          bn={"class": "GroupNorm", "num_groups": 4}).to(DEV)
    xyz = torch.rand(16, 512, 3, device=DEV)
    feats = torch.randn(16, 6, 512, device=DEV, requires_grad=True)
    peak = {}
    for on in (False, True):
        saved, fused.ACT16 = fused.ACT16, on
        try:
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            _, out = sa(xyz, feats)
            out.sum().backward()
            torch.cuda.synchronize()
            peak[on] = torch.cuda.max_memory_allocated() - base
        finally:
            fused.ACT16 = saved
    assert peak[True] < 0.62 * peak[False], peak


@pytest.mark.parametrize("B,cin,cout,P,S", [(32, 64, 64, 256, 64), (32, 64, 128, 256, 64), (32, 128, 256, 512, 32), (64, 128, 128, 128, 64),
                                            (32, 30, 64, 256, 64)])
def test_persistent_kernel_equals_tile_kernel(nat, B, cin, cout, P, S):
    """conv1x1_h.hip (persistent workgroups, weights staged once) against the tile kernel of conv1x1.hip (OGC_GEMM16=0) on launches
    large enough for the former (>= 8192 tiles of 64 positions): identical outputs and extremes, statistics equal up to the order
    of their fp64 additions."""
    hw, groups = P * S, 4
    assert B * (hw // 64) >= 8192
    xh = rnd(B, cin, hw, seed=cin).to(BF)
    w = rnd(cout, cin, seed=1, scale=cin ** -0.5)
    pa, pb = torch.rand(B * cin, device=DEV) + 0.5, rnd(B * cin, seed=2)
    gamma = rnd(cout, seed=5)
    slots = nat.conv1x1_gn_slots()
    g = rnd(B, cout, hw, seed=7).to(BF) if cout <= 160 else None
    res = {}
    for mode in ("0", "1"):
        os.environ["OGC_GEMM16"] = mode
        try:
            y0 = torch.empty(B, cout, hw, device=DEV, dtype=BF)
            nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, 0, w, xh, pa, pb, y0, None)
            st = torch.zeros(slots * B * groups * 2, dtype=torch.float64, device=DEV)
            y1 = torch.empty_like(y0)
            nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, groups, w, xh, pa, pb, y1, st)
            st2 = torch.zeros_like(st)
            y2 = torch.empty_like(y0)
            yext = torch.empty(B, cout, P, device=DEV)
            aext = torch.empty(B, cout, P, dtype=torch.int32, device=DEV)
            nat.conv1x1_gemm_affine_pool_wrapper(B, cout, cin, hw, 1, groups, S, w, xh, pa, pb, gamma, y2, st2, yext, aext)
            dz = da = None
            if g is not None:
                dz = torch.empty(B, cin, hw, device=DEV, dtype=BF)
                nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, g, dz)
                # the adjoint input gradient (persistent for <= 64 reduction channels): y_prev = xh, its own coefficients
                coef = rnd(B, cin, 3, seed=9, scale=0.2)
                da = torch.empty(B, cin, hw, device=DEV, dtype=BF)
                nat.conv1x1_dgrad_adjoint_wrapper(B, cin, cout, hw, 1, w, g, xh, pa, pb, coef, da)
            torch.cuda.synchronize()
            res[mode] = (y0, y1, st.view(slots, B, groups, 2).sum(0), y2, st2.view(slots, B, groups, 2).sum(0), yext, aext, dz, da)
        finally:
            os.environ.pop("OGC_GEMM16", None)
    for k, (a, b) in enumerate(zip(res["0"], res["1"])):
        if a is None:
            continue
        if a.dtype == torch.float64:
            assert torch.allclose(a, b, rtol=1e-11, atol=1e-6), k
        else:
            assert torch.equal(a, b), k
    assert torch.equal(res["1"][0], res["1"][1]) and torch.equal(res["1"][0], res["1"][3])


@pytest.mark.parametrize("npoint,nsample,mlp,B,N", [(256, 16, [6, 32, 32, 64], 16, 1024), (128, 32, [6, 64, 128], 8, 512),
                                                    (64, 64, [6, 48, 48, 48], 4, 256), (256, 16, [6, 64], 4, 512),
                                                    (100, 24, [6, 32, 32], 2, 300)])
def test_other_level_shapes_run_and_agree(nat, npoint, nsample, mlp, B, N):
    """Level shapes away from C2's — 16 / 32 neighbours, widths that are not multiples of 64, a two-layer and a one-layer MLP, a
    neighbourhood size the pooled kernels do not take: whatever mix of 16-bit kernels and widened fall-backs a shape gets, the pooled
    features agree with the fp32-activation run to bf16 accuracy and the gradients are finite and aligned with it."""
    from ogc_amd import fused
    from ogc_amd.utils.pointnet2_util import PointnetSAModule
    res = {}
    for on in (False, True):
        torch.manual_seed(3)
        sa = PointnetSAModule(mlp=list(mlp), npoint=npoint, radius=0.3, nsample=nsample, bn={"class": "GroupNorm", "num_groups": 4}).to(DEV)
        g = torch.Generator().manual_seed(5)
        xyz = torch.rand(B, N, 3, generator=g).to(DEV)
        feats = torch.randn(B, mlp[0], N, generator=g).to(DEV).requires_grad_(True)
        saved, fused.ACT16 = fused.ACT16, on
        try:
            _, out = sa(xyz, feats)
            out.square().sum().backward()
        finally:
            fused.ACT16 = saved
        res[on] = (out.detach(), feats.grad.detach(), torch.cat([p.grad.flatten() for p in sa.parameters()]))
    (o0, f0, g0), (o1, f1, g1) = res[False], res[True]
    assert torch.isfinite(o1).all() and torch.isfinite(f1).all() and torch.isfinite(g1).all()
    assert rel(o1, o0) < 1e-2
    for a, b in ((f1, f0), (g1, g0)):
        cos = float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-30))
        assert cos > 0.99, cos
