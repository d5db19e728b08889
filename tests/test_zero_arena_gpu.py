"""utils/zero_arena.py: one zero fill per training step, against every operator filling its own buffers.

The region is a fresh allocation per step, so what escapes a step (a weight gradient kept as ``param.grad``) is an ordinary
tensor: it must survive the following steps untouched, with the arena on exactly as with it off; and a step that raises in
its backward pass must leave the arena closed and the next step unaffected."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=10):
    from ogc_amd.models.segnet_sapien import MaskFormer3D
    from ogc_amd.train_step import SAPIEN_LOSS, build_criterion, make_optimizer
    from ogc_amd.utils.synthetic import make_scene_batch
    torch.manual_seed(seed)
    net = MaskFormer3D(n_slot=8, n_point=512, transformer_embed_dim=128).cuda()
    crit = build_criterion(SAPIEN_LOSS)
    opt = make_optimizer(net.parameters(), lr=1e-3)
    batches = [make_scene_batch(2, 512, 8, seed=s, outdoor=False, aug=True, device="cuda") for s in (1, 2, 3, 4)]
    return net, crit, opt, batches


def _run(arena_on, steps=4, fail_at=None, subgraphs=True):
    """`steps` training steps of the C1 shapes; returns per step (loss dict, gradients held WITHOUT a copy, their clones at the
    time) and the final parameters.  With sub-graphs on, the gradients of the slot branch (MF_head, object_mlp: a forward /
    backward pair of HIP graphs, utils/subgraph.py) are left out of what is held: they live in the graphs' static buffers until
    the branch's next replay, as with torch.cuda.make_graphed_callables — nothing the arena changes either way."""
    from ogc_amd import train_step as ts
    from ogc_amd.utils import subgraph as sg
    from ogc_amd.utils import zero_arena as za
    was, was_sg = za.ENABLED, sg.ENABLED
    za.ENABLED, sg.ENABLED = arena_on, subgraphs
    try:
        net, crit, opt, batches = _setup()
        held, out = [], []
        for i in range(steps):
            if fail_at == i:
                # a backward pass that raises: the step must surface it (an operator error is a fault, not a skipped step) and
                # leave no arena open behind it
                real = torch.Tensor.backward

                def boom(self, *a, **k):
                    from ogc_amd._lib import OgcOpsError
                    raise OgcOpsError("forced failure in backward()")
                torch.Tensor.backward = boom
                try:
                    with pytest.raises(Exception):
                        ts.train_step(net, crit, opt, batches[i], 10, True)
                finally:
                    torch.Tensor.backward = real
                assert za._active is None
                continue
            losses, stepped = ts.train_step(net, crit, opt, batches[i], 10, True)
            assert stepped
            grads = [p.grad for n, p in net.named_parameters() if p.grad is not None       # the tensors autograd kept
                     and not (subgraphs and n.startswith(("MF_head.", "object_mlp.")))]
            held.append((grads, [g.clone() for g in grads]))
            out.append(losses)
        torch.cuda.synchronize()
        return out, held, [p.detach().clone() for p in net.parameters()]
    finally:
        za.ENABLED, sg.ENABLED = was, was_sg


def test_held_gradients_survive_later_steps():
    """param.grad of step i, kept by the caller across optimizer.zero_grad(set_to_none=True), is still what it was after the
    steps that follow (the persistent region of round 4 zeroed it at the next step's begin)."""
    for subgraphs in (True, False):         # (False: every parameter's gradient is an ordinary tensor, the slot branch's too)
        _, held, _ = _run(True, subgraphs=subgraphs)
        assert len(held) == 4 and len(held[0][0]) >= (40 if subgraphs else 88)
        for step, (kept, clones) in enumerate(held):
            for g, c in zip(kept, clones):
                assert torch.equal(g, c), "gradient of step %d was overwritten by a later step" % step


def test_arena_on_equals_arena_off():
    on_losses, on_held, on_params = _run(True)
    off_losses, off_held, off_params = _run(False)
    # (atomic accumulation orders differ from run to run, and Adam turns a last-bit difference of a near-zero gradient into a
    # full-size update: the first step is compared to fp32 rounding of the sums, the later ones loosely)
    assert len(on_losses) == len(off_losses) == 4
    for step, (a, b) in enumerate(zip(on_losses, off_losses)):
        assert a.keys() == b.keys()
        for k in a:
            assert a[k] == pytest.approx(b[k], rel=1e-5 if step == 0 else 2e-2, abs=1e-6 if step == 0 else 1e-3), (step, k)
    for x, y in zip(on_held[0][1], off_held[0][1]):      # first step: same parameters on both sides
        torch.testing.assert_close(x, y, rtol=2e-3, atol=1e-6)
    assert all(torch.isfinite(p).all() for p in on_params + off_params)


def test_step_that_raises_in_backward_leaves_the_arena_closed():
    from ogc_amd.utils import zero_arena as za
    losses, held, params = _run(True, steps=4, fail_at=1)
    assert za._active is None and len(losses) == 3
    for kept, clones in held:
        for g, c in zip(kept, clones):
            assert torch.equal(g, c)
    assert all(torch.isfinite(p).all() for p in params)
