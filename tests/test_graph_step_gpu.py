"""The training step replayed as one HIP graph (ogc_amd/graph_step.py) against the eager step: same batches in the
same order, same losses step by step (up to the rounding noise of the scatter-add atomics, which both have)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LOSS_TOL, WEIGHT_TOL = 3e-4, 0.01   # (typically 2e-5 / 0.002; the test asked for 2e-3 / 0.05 while hipMemsetAsync nodes fed stale
#                                     bits into the replayed step's accumulators.  3e-4 since round 4: both runs accumulate their
#                                     weight gradients with float atomics — the feature-propagation layers' too now — in an order
#                                     that differs from run to run, five optimizer steps amplify that, and once in ten runs of the
#                                     suite the invariance term of step 4 differed by 1.1e-4)


def _build(npoint):
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer
    torch.manual_seed(10)
    net = MaskFormer3D(n_slot=10, n_point=npoint, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                       transformer_input_pos_enc=False).cuda()
    return net, build_criterion(KITTI_LOSS), make_optimizer(net.parameters(), lr=1e-3, capturable=True)


def test_graphed_step_matches_eager():
    assert torch.cuda.is_available()
    import ogc_amd  # noqa: F401
    from ogc_amd.graph_step import GraphedTrainStep
    from ogc_amd.train_step import train_step
    from ogc_amd.utils.synthetic import make_scene_batch
    npoint, steps = 1024, 6
    batches = [make_scene_batch(2, npoint, 10, seed=77 + i, outdoor=True, aug=True, device="cuda") for i in range(3)]

    net, crit, opt = _build(npoint)
    eager, pre = [], None
    for i in range(steps):
        p = train_step(net, crit, opt, batches[i % 3], 1000, True, sync=False, prefetched=pre,
                       next_batch=batches[(i + 1) % 3])
        pre = p.prefetched
        losses, stepped = p.result()
        assert stepped
        eager.append(losses)
    w_eager = [p.detach().clone() for p in net.parameters()]

    net, crit, opt = _build(npoint)
    w0 = [p.detach().clone() for p in net.parameters()]
    gs = GraphedTrainStep(net, crit, opt, batches[0], 1000, True)
    # building the graph (one eager warm-up step, undone) leaves the model and the optimizer as they were
    for a, b in zip(w0, net.parameters()):
        assert torch.equal(a, b.detach())
    assert all(float(st["step"]) == 0 for st in opt.state.values())
    for i in range(steps):
        losses, stepped = gs.step(batches[(i + 1) % 3]).result()
        assert stepped
        for k in ("sum", "dynamic", "smooth", "invariance"):
            assert abs(losses[k] - eager[i][k]) <= LOSS_TOL * max(1.0, abs(eager[i][k])), (i, k, losses[k], eager[i][k])
    assert all(float(st["step"]) == steps for st in opt.state.values())
    # six Adam steps of 1e-3 move a weight by at most 6e-3; the two runs must agree far better than that on average
    num = sum(float((a - b.detach()).abs().sum()) for a, b in zip(w_eager, net.parameters()))
    den = sum(float((a - b).abs().sum()) for a, b in zip(w_eager, w0))
    assert num <= WEIGHT_TOL * den, (num, den)


@pytest.mark.gpu
def test_graphed_slot_branch_is_reentrant_safe():
    """The slot branch runs as a pair of HIP graphs on static buffers (MaskFormer3DBase._slots): same gradients as the eager
    branch, and a SECOND forward pass before the first one's backward pass (two clouds through the net, one loss) must not
    disturb the first."""
    import ogc_amd  # noqa: F401
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    torch.manual_seed(5)
    net = MaskFormer3D(n_slot=6, n_point=1024, use_xyz=True, n_transformer_layer=1, transformer_embed_dim=64,
                       transformer_input_pos_enc=False).cuda().train()
    pcs = [torch.rand(2, 1024, 3, device="cuda") * torch.tensor([30.0, 3.0, 40.0], device="cuda") for _ in range(2)]
    probes = [torch.randn(2, 1024, 6, device="cuda") for _ in range(2)]

    def grads(graphed, both):
        net.graph_slot_branch = graphed
        for entry in net.__dict__.get("_subgraphs", {}).values():
            entry[1] = None
        net.zero_grad(set_to_none=True)
        outs = [net(p, p) for p in (pcs if both else pcs[:1])]
        sum((o * q).sum() for o, q in zip(outs, probes)).backward()
        return [o.detach().clone() for o in outs], [p.grad.clone() for p in net.parameters()]

    try:
        ref_o, ref_g = grads(False, True)
        for _ in range(2):                       # twice: the second round replays the graphs made by the first
            got_o, got_g = grads(True, True)
            for a, b in zip(got_o + got_g, ref_o + ref_g):
                # (not bit for bit: the weight-gradient kernels add their partial sums with float atomics)
                assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 2e-5, float((a - b).abs().max())
        made = [e for k, e in net.__dict__.get("_subgraphs", {}).items() if k[0] == "slots"]
        assert made and made[0][0] is not None, "the slot branch was not captured"
    finally:
        net.graph_slot_branch = True


@pytest.mark.gpu
def test_accumulators_are_zeroed_on_every_replay():
    """An entry point that zeroes its output and then adds into it with atomics (ogc_conv1x1_wgrad), captured in a HIP graph and
    replayed with the output poisoned in between: exact zeros wherever the operands are zero, on EVERY replay.  (With
    hipMemsetAsync in the library a memset node of this stack zeroed correctly on the first replay only — tools/memset_probe.py;
    the library now zero-fills with a kernel, csrc/ogc_common.h.)"""
    import ogc_amd  # noqa: F401
    from ogc_amd import pointnet2_cuda as nat
    B, cin, cout, hw = 2, 128, 128, 64
    x = torch.zeros(B, cin, hw, device="cuda")
    x[:, ::2] = torch.randn(B, cin // 2, hw, device="cuda")          # every other input channel is zero
    dy = torch.randn(B, cout, hw, device="cuda")
    ref = torch.bmm(dy, x.transpose(1, 2)).sum(0)
    warm = torch.empty(cout, cin, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, warm)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        dw = torch.empty(cout, cin, device="cuda")
        nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, dw)
    for replay in range(4):
        dw.fill_(1e30)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert bool((dw[:, 1::2] == 0).all()), (replay, float(dw[:, 1::2].abs().max()))
        assert float((dw - ref).norm() / ref.norm()) < 1e-5


@pytest.mark.timeout(900)
def test_graphed_step_with_the_rccl_all_reduce_inside():
    """One process, backend nccl (= RCCL), FlatDataParallel with the collective forced on: the whole step INCLUDING the flat
    gradient all-reduce is captured into one HIP graph (tools/graph_rccl.py) and replayed; six steps give the eager step's
    losses.  (More than one rank needs more than one GPU: the data-parallel graph is exercised at world size 1 here, the
    arithmetic of the collective at world size 2 over gloo in tests/test_ddp_cpu.py.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           "29537", os.path.join(root, "tools", "graph_rccl.py"), "1024"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "GRAPH_RCCL_OK" in out.stdout, out.stdout[-1500:]
