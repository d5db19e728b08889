"""Training-step throughput of any of the segmentation configs (config/*_unsup_synthetic.yaml) on one GPU — the same
loop as bench.py (which is pinned to C4), for the other rows of BASELINE.json's config list (development tool).
    python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml [steps]"""
import os, sys, time
import torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.train_seg import build_segnet
from ogc_amd.train_step import build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

cfg = yaml.safe_load(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prec = os.environ.get("PRECISION", cfg.get("matmul_precision", "fp32"))
from ogc_amd.pointnet2 import pointnet2 as _api
_api._native.set_matmul_precision(prec)
torch.manual_seed(cfg["random_seed"])
net = build_segnet(cfg).cuda()
if os.environ.get("OVERLAP_HEAD") is not None:
    net.overlap_head = os.environ["OVERLAP_HEAD"] == "1"
single = cfg["dataset"] == "waymo"
crit = build_criterion(cfg["loss"], single_frame=single)
graph = os.environ.get("GRAPH") == "1"   # the step replayed as one HIP graph (ogc_amd/graph_step.py)
opt = make_optimizer(net.parameters(), lr=cfg["lr"], capturable=graph)
outdoor = cfg["dataset"] in ("kittisf", "waymo")
batch = make_scene_batch(cfg["batch_size"], cfg["segnet"]["n_point"], cfg["segnet"]["n_slot"], seed=1, outdoor=outdoor, aug=True, device="cuda")
if single:
    batch = tuple(x[:, ::2].contiguous() for x in batch)
views = batch[1].shape[1]
pre, pend = None, None
gs = None
if graph:
    from ogc_amd.graph_step import GraphedTrainStep
    gs = GraphedTrainStep(net, crit, opt, batch, 10 ** 6, True)
for i in range(5 + steps):
    if i == 5:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    if gs is not None:
        pend = gs.step(batch)
        continue
    pend = train_step(net, crit, opt, batch, 10 ** 6, True, sync=False, prefetched=pre, next_batch=batch)
    pre = pend.prefetched
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
clouds = cfg["batch_size"] * views
print("%s [%s%s]: batch %d x %d views x %d points: %.2f ms/step = %.0f clouds/s, peak HBM %.2f GiB, loss %s" %
      (cfg["dataset"], prec, ", HIP graph" if graph else "", cfg["batch_size"], views, cfg["segnet"]["n_point"], ms, clouds / ms * 1e3,
       torch.cuda.max_memory_allocated() / 2 ** 30, {k: round(v, 4) for k, v in pend.result()[0].items()}))
