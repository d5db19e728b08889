"""flow_glue.py / csrc/flow_step.hip: the fused element-wise kernels of FlowStep3D's inference loop against the framework operator
sequences they replace (models/flownet_kitti.py:135-151, :229-250), and the model with them on against the model with them off."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("B,N,M", [(1, 8192, 4096), (2, 100, 37), (3, 1, 5)])
def test_gather_xyz_pair(B, N, M):
    from ogc_amd import flow_glue
    from ogc_amd.pointnet2.pointnet2 import gather_operation
    xyz = _rand(B, 3, N, seed=1, scale=30.0)
    idx = torch.randint(0, N, (B, M), generator=torch.Generator().manual_seed(2), dtype=torch.int32).cuda()
    out, out_t = flow_glue.gather_xyz_pair(xyz, idx)
    ref = gather_operation(xyz, idx)
    assert torch.equal(out, ref) and torch.equal(out_t, ref.transpose(1, 2).contiguous())


@pytest.mark.parametrize("B,N,divisor", [(1, 8192, 1.0), (2, 2048, 1.8), (1, 77, 3.4), (2, 5, 1), (1, 999, 2.6), (1, 1000, 4.2), (1, 1001, 7.0)])
def test_flow_advance_is_the_operator_sequence(B, N, divisor):
    from ogc_amd import flow_glue
    cur, delta, ref = _rand(B, 3, N, seed=3, scale=20.0), _rand(B, 3, N, seed=4), _rand(B, 3, N, seed=5, scale=20.0)
    d, new, new_t, flow = flow_glue.flow_advance(cur, delta, ref, divisor=divisor, want_delta=True, want_t=True)
    d_ref = delta / divisor                      # (torch: delta * fp32(1 / divisor), the reciprocal taken in double)
    new_ref = cur + d_ref
    assert torch.equal(d, d_ref) and torch.equal(new, new_ref) and torch.equal(flow, new_ref - ref)
    assert torch.equal(new_t, new_ref.permute(0, 2, 1).contiguous())
    d2, new2, t2, f2 = flow_glue.flow_advance(cur, delta, None, want_flow=False)
    assert d2 is None and t2 is None and f2 is None and torch.equal(new2, cur + delta)


@pytest.mark.parametrize("B,cin,cout,N", [(1, 128, 3, 2048), (2, 64, 3, 300), (1, 7, 4, 33), (2, 128, 1, 64)])
def test_linear_cn(B, cin, cout, N):
    from ogc_amd import flow_glue
    x, w, bias = _rand(B, cin, N, seed=6), _rand(cout, cin, seed=7, scale=0.1), _rand(cout, seed=8)
    y = flow_glue.linear_cn(x, w, bias)
    ref = torch.nn.functional.linear(x.double().permute(0, 2, 1), w.double(), bias.double()).permute(0, 2, 1)
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-5)
    y0 = flow_glue.linear_cn(x, w, None)
    torch.testing.assert_close(y0.double(), ref - bias.double().view(1, -1, 1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,Cx,N,S", [(1, 128, 211, 2048, 4), (2, 16, 5, 100, 4), (1, 8, 3, 33, 3), (2, 4, 0, 17, 8)])
def test_gru_gates(B, C, Cx, N, S):
    from ogc_amd import flow_glue
    hx = _rand(B, C + Cx, N, seed=9)
    rc, zc, qc = _rand(B, C, N, S, seed=10, scale=3.0), _rand(B, C, N, S, seed=11, scale=3.0), _rand(B, C, N, S, seed=12, scale=2.0)
    h, x = hx[:, :C], hx[:, C:]
    out = flow_glue.gru_reset(rc, hx, C)
    ref = torch.cat([torch.sigmoid(torch.amax(rc, dim=-1)) * h, x], dim=1)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-7)
    assert torch.equal(out[:, C:], x)
    blend = flow_glue.gru_blend(zc, qc, hx, C)
    z, q = torch.sigmoid(torch.amax(zc, dim=-1)), torch.tanh(torch.amax(qc, dim=-1))
    torch.testing.assert_close(blend, (1 - z) * h + z * q, rtol=1e-6, atol=1e-7)
    # the gates as channel ranges of one wider tensor (update and reset gate from one stacked product)
    zr = torch.cat([zc, rc], dim=1).contiguous()
    assert torch.equal(flow_glue.gru_reset(zr, hx, C, rc_channel0=C), out)
    assert torch.equal(flow_glue.gru_blend(zr, qc, hx, C, zc_channel0=0), blend)
    # a NaN among the neighbours stays a NaN, as torch.amax's
    zc2 = zc.clone()
    zc2[0, 0, 0, S - 1] = float("nan")
    assert torch.isnan(flow_glue.gru_blend(zc2, qc, hx, C)[0, 0, 0])


def test_not_available_when_differentiating_or_on_cpu():
    from ogc_amd import flow_glue
    a = torch.zeros(1, 3, 8, device="cuda")
    assert flow_glue.available(a)
    assert not flow_glue.available(a.clone().requires_grad_())
    with torch.no_grad():
        assert flow_glue.available(a.clone().requires_grad_())
    assert not flow_glue.available(torch.zeros(1, 3, 8))
    assert not flow_glue.available(a.double())


@pytest.mark.parametrize("variant", ["kitti", "sapien"])
def test_flowstep3d_inference_with_and_without_the_glue(variant):
    """Same predictions (to fp32 rounding of the few operations whose summation order differs: the 3-channel linear layer) with the
    fused loop as with the reference's operator sequence."""
    import importlib
    from ogc_amd import flow_glue
    from ogc_amd.utils.synthetic import make_scene_batch
    mod = importlib.import_module("ogc_amd.models.flownet_" + variant)
    torch.manual_seed(0)
    N = 2048 if variant == "kitti" else 512
    net = mod.FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).cuda().eval()
    pcs = make_scene_batch(2, N, 6, seed=3, aug=False, device="cuda", outdoor=variant == "kitti")[0]
    pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
    was = flow_glue.ENABLED
    try:
        with torch.no_grad():
            flow_glue.ENABLED = True
            on = net(pc1, pc2, pc1, pc2, iters=4)
            flow_glue.ENABLED = False
            off = net(pc1, pc2, pc1, pc2, iters=4)
    finally:
        flow_glue.ENABLED = was
    assert len(on) == len(off) == 4
    for a, b in zip(on, off):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,n1,n2,C", [(1, 256, 256, 128), (2, 100, 77, 64), (1, 9, 300, 4)])
def test_soft_corr_flow_is_the_global_correlation(B, n1, n2, C):
    """GlobalCorrLayer's own operator sequence (calc_corr_mat, row sums, weighted mean; flownet_kitti.py:53-70) in fp64 against the
    kernel; clouds wide enough that many pairs lie beyond the 10 m support."""
    from ogc_amd import flow_glue
    from ogc_amd.models.flownet_kitti import FlowStep3D
    layer = FlowStep3D(npoint=512, loc_flow_nn=16, loc_flow_rad=1.5).global_corr_layer.cuda()
    with torch.no_grad():
        layer.epsilon.fill_(0.3)
    p1, p2 = _rand(B, 3, n1, seed=20, scale=8.0), _rand(B, 3, n2, seed=21, scale=8.0)
    f1, f2 = _rand(B, C, n1, seed=22), _rand(B, C, n2, seed=23)
    with torch.no_grad():
        flow = flow_glue.soft_corr_flow(p1, p2, f1, f2, layer.epsilon, float(layer.support_th))
        dl = layer.double()
        q1, q2 = p1.double().permute(0, 2, 1), p2.double().permute(0, 2, 1)
        w = dl.calc_corr_mat(q1, q2, f1.double().permute(0, 2, 1), f2.double().permute(0, 2, 1))
        ref = ((w @ q2) / (w.sum(-1, keepdim=True) + 1e-8) - q1).permute(0, 2, 1)
        frac_out = (w == 0).double().mean().item()
    assert 0.05 < frac_out < 0.95          # the support mask matters in this test
    torch.testing.assert_close(flow.double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,n,m", [(1, 8192, 2048), (2, 300, 50), (1, 5, 3)])
def test_three_nn_with_weights(B, n, m):
    from ogc_amd import flow_glue
    from ogc_amd.pointnet2.pointnet2 import three_nn
    unknown, known = _rand(B, n, 3, seed=30, scale=10.0), _rand(B, m, 3, seed=31, scale=10.0)
    unknown[:, 0] = known[:, 0]          # a coincident pair: distance 0, clamped at 1e-10
    idx, weight = flow_glue.three_nn_with_weights(unknown, known)
    dists, idx_ref = three_nn(unknown, known)
    w = 1.0 / dists.clamp(min=1e-10)
    w = w / w.sum(dim=-1, keepdim=True)
    assert torch.equal(idx, idx_ref)
    torch.testing.assert_close(weight, w, rtol=1e-6, atol=1e-9)
    # mode 1: the segmentation nets' form (utils/pointnet2_util.py:99-101)
    w1 = torch.empty_like(weight)
    from ogc_amd import pointnet2_cuda as nat
    nat.three_nn_weights_wrapper(B, n, 1, dists * dists, w1)
    r = 1.0 / ((dists * dists).sqrt() + 1e-8)
    torch.testing.assert_close(w1, r / r.sum(dim=2, keepdim=True), rtol=1e-6, atol=1e-9)


def test_fps_chain_without_temp_equals_fps():
    """ogc_furthest_point_sampling_chain with temp == NULL (minima kept in registers, starting at 1e10): the indices of the plain
    entry point, for the register kernel, the bucket kernels and a tie lattice."""
    from ogc_amd.pointnet2.pointnet2 import furthest_point_sample, furthest_point_sample_chain
    for B, N, m, seed in [(2, 300, 100, 1), (1, 2048, 512, 2), (2, 8192, 4096, 3), (1, 16384, 1024, 4), (3, 5000, 700, 5)]:
        xyz = _rand(B, N, 3, seed=seed, scale=20.0)
        if seed == 5:
            xyz = torch.round(xyz)       # a lattice: exact distance ties
        idx, ties = furthest_point_sample_chain(xyz, m)
        assert torch.equal(idx, furthest_point_sample(xyz, m)), (B, N, m)


def test_glue_sees_weights_the_fused_optimizer_wrote():
    """Evaluate, train one step (the fused Adam kernel writes parameters through raw pointers: no version counter moves), evaluate
    again: the inference glue must use the NEW weights — its stacked gate weights are cached, and were keyed by version counters
    alone until the flow trainer replay caught the second evaluation running on the first one's weights."""
    from ogc_amd import flow_glue
    from ogc_amd.models.flownet_kitti import FlowStep3D
    from ogc_amd.train_step import make_optimizer
    from ogc_amd.utils.synthetic import make_scene_batch
    torch.manual_seed(0)
    N = 1024
    net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).cuda()
    opt = make_optimizer(net.parameters(), lr=1e-2)
    pcs = make_scene_batch(1, N, 6, seed=3, aug=False, device="cuda", outdoor=True)[0]
    pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()

    def evaluate(glue):
        was = flow_glue.ENABLED
        flow_glue.ENABLED = glue
        try:
            net.eval()
            with torch.no_grad():
                return net(pc1, pc2, pc1, pc2, iters=3)[-1].clone()
        finally:
            flow_glue.ENABLED = was

    first = evaluate(True)
    net.train()
    out = net(pc1, pc2, pc1, pc2, iters=2)
    sum(o.square().mean() for o in out).backward()
    opt.step()
    opt.zero_grad()
    second_on, second_off = evaluate(True), evaluate(False)
    assert float((second_off - first).abs().max()) > 1e-3          # the step really moved the predictions
    torch.testing.assert_close(second_on, second_off, rtol=1e-4, atol=1e-5)
