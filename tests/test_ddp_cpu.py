"""Multi-process data-parallel test on CPU (gloo, world_size 2): the N>1 path of bench.py / the training driver.

Each rank runs `train_step` on its own shard with DDP gradient averaging; the averaged gradients and the updated
weights must equal a single process that steps on the concatenated batch (GroupNorm has no cross-sample
statistics, so the only coupling between shards is the gradient mean — SURVEY.md §8e).  The native operators are
replaced by the CPU oracle inside the worker processes (test infrastructure only)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup_cpu_ops():
    sys.path.insert(0, ROOT)
    import ogc_amd.pointnet2.pointnet2 as api
    from oracle import oracle as orc
    api._native = orc.Pointnet2CudaCPU()
    orc.set_threads(2)
    torch.set_num_threads(2)


def _make(npoint=256):
    from ogc_amd.models.segnet_sapien import MaskFormer3D
    from ogc_amd.train_step import SAPIEN_LOSS, build_criterion
    torch.manual_seed(10)
    net = MaskFormer3D(n_slot=4, n_point=npoint, transformer_embed_dim=32)
    return net, build_criterion(SAPIEN_LOSS)


def _loss_and_grads(model, crit, batch):
    pcs, segms, flows, _ = batch
    b, t, n = segms.size()
    flat = pcs.view(b * t, n, -1).contiguous()
    masks = model(flat, flat).view(b, t, n, -1)
    loss, _ = crit([pcs[:, i].contiguous() for i in range(t)], [masks[:, i].contiguous() for i in range(t)],
                   [flows[:, i].contiguous() for i in range(t)], step_w=True, it=10, aug_transform=False)
    return loss


def _worker(rank, world, port, out_dir, kind):
    _setup_cpu_ops()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ogc_amd.utils.synthetic import make_scene_batch
    net, crit = _make()
    if kind == "flat":   # one flat all-reduce after backward (ogc_amd/utils/dist_util.py): what bench.py and the drivers use
        from ogc_amd.utils.dist_util import FlatDataParallel
        if rank == 1:     # construction must broadcast rank 0's weights
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(0.5)
        ddp = FlatDataParallel(net)
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(net)
    full = make_scene_batch(2 * world, 256, 4, seed=3, outdoor=False, aug=False)
    shard = tuple(x[rank * 2:(rank + 1) * 2].contiguous() for x in full)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.zero_grad()
    _loss_and_grads(ddp, crit, shard).backward()
    if kind == "flat":
        ddp.average_gradients()
    opt.step()
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, "ddp%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["flat", "ddp"])
def test_ddp_two_ranks_equal_single_process(tmp_path, kind):
    world, port = 2, 29000 + (os.getpid() + (7 if kind == "flat" else 0)) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    _setup_cpu_ops()
    from ogc_amd.utils.synthetic import make_scene_batch
    net, crit = _make()
    full = make_scene_batch(2 * world, 256, 4, seed=3, outdoor=False, aug=False)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.zero_grad()
    # every loss term is a mean over the batch dimension, so the 2-rank mean of shard losses == the full-batch loss
    _loss_and_grads(net, crit, full).backward()
    opt.step()
    ddp_state = torch.load(os.path.join(str(tmp_path), "ddp0.pt"))
    other = torch.load(os.path.join(str(tmp_path), "ddp1.pt"))
    for k, v in ddp_state.items():
        assert torch.equal(v, other[k]), k     # the replicas hold identical weights after the step
    for k, v in net.state_dict().items():
        torch.testing.assert_close(ddp_state[k], v, rtol=2e-4, atol=2e-6, msg=lambda m: "%s: %s" % (k, m))


def _worker_unused(rank, world, port, out_dir):
    """Rank 1 never touches `b` (its gradient stays None there); rank 0 never touches `c`."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ogc_amd.utils.dist_util import FlatDataParallel

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.ones(3))
            self.b = torch.nn.Parameter(torch.ones(2, 2))
            self.c = torch.nn.Parameter(torch.ones(5))

        def forward(self, use_b):
            return (self.a * 2).sum() + ((self.b * 3).sum() if use_b else (self.c * 7).sum())

    net = Net()
    dp = FlatDataParallel(net)
    dp(rank == 0).backward()
    dp.average_gradients()
    torch.save({k: p.grad.clone() for k, p in net.named_parameters()}, os.path.join(out_dir, "g%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_data_parallel_with_rank_dependent_unused_parameters(tmp_path):
    """the flat buffer keeps the parameter order on every rank, so a parameter whose gradient is None on one
    rank averages with the other rank's gradient of the SAME parameter."""
    world, port = 2, 31000 + os.getpid() % 2000
    mp.spawn(_worker_unused, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        g = torch.load(os.path.join(str(tmp_path), "g%d.pt" % r))
        assert torch.equal(g["a"], torch.full((3,), 2.0))
        assert torch.equal(g["b"], torch.full((2, 2), 1.5))
        assert torch.equal(g["c"], torch.full((5,), 3.5))


def test_core_blocks_of_the_ranks_of_a_host():
    """utils/dist_util.core_block: the ranks of a host keep to disjoint contiguous blocks of the allowed CPUs (one hot launch
    thread per rank); fewer CPUs than ranks still gives every rank one."""
    from ogc_amd.utils.dist_util import core_block
    allowed = set(range(4, 100))          # a cpuset that does not start at 0
    blocks = [core_block(allowed, r, 8) for r in range(8)]
    assert all(len(b) == 12 for b in blocks) and len(set().union(*blocks)) == 96
    assert blocks[0] == set(range(4, 16)) and blocks[7] == set(range(88, 100))
    assert [core_block({0, 1, 2}, r, 8) for r in range(8)] == [{0}, {1}, {2}, {0}, {1}, {2}, {0}, {1}]
    assert core_block({5}, 0, 1) == {5}


def test_ranks_are_not_pinned_to_blocks_of_fewer_than_four_cpus(monkeypatch):
    """pin_rank_to_cores leaves the affinity alone when the host cannot give every rank four CPUs (a rank keeps a launch thread,
    autograd's backward thread and a runtime helper busy)."""
    import os
    from ogc_amd.utils import dist_util
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    calls = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: calls.append(set(cpus)))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.delenv("OGC_PIN_CORES", raising=False)
    assert dist_util.pin_rank_to_cores() is None and calls == []          # 16 CPUs, 8 ranks: two each
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)))
    assert dist_util.pin_rank_to_cores() == set(range(24, 32)) and calls == [set(range(24, 32))]
