"""What does the DistributedDataParallel wrapper cost at one process?  Same C4 step with and without the wrapper
(RCCL initialised either way): GPU time per step and the host's launch time per step.  Development tool.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/ddp_cost.py"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
noinit = bool(os.environ.get("NOINIT"))
if not noinit:
    if os.environ.get("LAZY"):
        dist.init_process_group(os.environ.get("BACKEND", "nccl"))
    else:
        dist.init_process_group("nccl", device_id=dev)
import ogc_amd  # noqa: F401
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = v == "1" if v in ("0", "1") else int(v)
print("ddp kwargs", kw)
from ogc_amd.utils.dist_util import FlatDataParallel, always_reduce
always_reduce(True)
for wrap in ((False, False) if noinit else (False, True, "flat", False, True, "flat")):
    torch.manual_seed(10)
    net = MaskFormer3D(n_slot=10, n_point=8192, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                       transformer_input_pos_enc=False).to(dev)
    model = net
    if wrap == "flat":
        model = FlatDataParallel(net)
    elif wrap:
        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local], bucket_cap_mb=25,
                                                          gradient_as_bucket_view=True, **kw)
    crit = build_criterion(KITTI_LOSS)
    opt = make_optimizer(net.parameters(), lr=1e-3, weight_decay=0.0)
    batch = make_scene_batch(4, 8192, 10, seed=1234, outdoor=True, aug=True, device=dev)
    pre = None
    for _ in range(4):
        pre = train_step(model, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pre = train_step(model, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-6s  step %.2f ms   host launch %.2f ms/step" % ("flat" if wrap == "flat" else "DDP" if wrap else "plain", (t2 - t0) / 20 * 1e3, (t1 - t0) / 20 * 1e3))
    del model, net, opt
if not noinit:
    dist.destroy_process_group()
