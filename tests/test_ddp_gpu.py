"""Two data-parallel ranks on ONE MI355X (gloo process group, both ranks on cuda:0): the production `train_step` — HIP
operators, fused Adam with the on-device NaN rule, next-batch geometry prefetch, asynchronous scalars — under
DistributedDataParallel.  RCCL refuses two ranks on one device, so the collective runs over gloo here; everything else
is the N > 1 path of bench.py / train_seg.py.  The two ranks must end with identical weights, equal to a single process
that steps on the concatenated batches."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NPOINT, STEPS = 2048, 2


def _make():
    sys.path.insert(0, ROOT)
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer
    torch.manual_seed(10)
    net = MaskFormer3D(n_slot=6, n_point=NPOINT, transformer_embed_dim=64).cuda()
    return net, build_criterion(KITTI_LOSS), make_optimizer(net.parameters(), lr=1e-3)


def _batches(rank_slice):
    from ogc_amd.utils.synthetic import make_scene_batch
    out = []
    for s in range(STEPS):
        full = make_scene_batch(4, NPOINT, 6, seed=50 + s, outdoor=True, aug=True, device="cuda")
        out.append(tuple(x[rank_slice].contiguous() for x in full))
    return out


def _run(model, crit, opt, batches):
    from ogc_amd.train_step import train_step
    pre, results = None, []
    for i, batch in enumerate(batches):
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        pending = train_step(model, crit, opt, batch, 1000 + i, True, sync=False, prefetched=pre, next_batch=nxt)
        pre = pending.prefetched
        results.append(pending)
    return [p.result() for p in results]


def _worker(rank, world, port, out_dir, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, crit, opt = _make()
    if kind == "flat":   # what bench.py and the drivers use: one flat all-reduce per step (utils/dist_util.py)
        from ogc_amd.utils.dist_util import FlatDataParallel
        ddp = FlatDataParallel(net)
    else:
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0])
    results = _run(ddp, crit, opt, _batches(slice(rank * 2, rank * 2 + 2)))
    assert all(stepped for _, stepped in results)
    torch.save({"state": {k: v.cpu() for k, v in net.state_dict().items()}, "losses": [r[0] for r in results]},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["flat", "ddp"])
def test_ddp_two_ranks_on_one_gpu(tmp_path, kind):
    world, port = 2, 31000 + (os.getpid() + (7 if kind == "flat" else 0)) % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(2))
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), k     # replicas never diverge
    net, crit, opt = _make()
    single = _run(net, crit, opt, _batches(slice(0, 4)))
    # every loss term is a mean over the batch, so the mean of the two shard losses is the full-batch loss — at step 0
    # exactly the same function, at step 1 after one averaged update.  (Weights are not compared one by one: parameters
    # whose gradient is analytically zero, e.g. the key bias of an attention layer, receive pure rounding noise, which
    # Adam turns into +-lr steps of arbitrary sign.)
    for step in range(STEPS):
        for key in ("dynamic", "smooth", "invariance", "sum"):
            sharded = 0.5 * (a["losses"][step][key] + b["losses"][step][key])
            assert abs(sharded - single[step][0][key]) <= 2e-3 * abs(single[step][0][key]) + 1e-6, (step, key)
