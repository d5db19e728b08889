"""The C4 training step eagerly (with the one-batch-ahead geometry prefetch) and as ONE HIP graph (ogc_amd/graph_step.py):
ms per step of both, and the losses of the first steps side by side."""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PHASE = os.environ.get("PHASE", "both")   # eager | graph | both  (tools/graph_gaps.py traces one at a time)

import ogc_amd  # noqa: F401
from ogc_amd.graph_step import GraphedTrainStep
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

npoint = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)


def build():
    torch.manual_seed(10)
    net = MaskFormer3D(n_slot=10, n_point=npoint, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                       transformer_input_pos_enc=False).to(dev)
    return net, build_criterion(KITTI_LOSS), make_optimizer(net.parameters(), lr=1e-3, capturable=True)


batches = [make_scene_batch(4, npoint, 10, seed=1234 + i, outdoor=True, aug=True, device=dev) for i in range(3)]

net, crit, opt = build()
pre = None
eager_losses = []
for i in range(25 if PHASE != "graph" else 0):
    p = train_step(net, crit, opt, batches[i % 3], 1000, True, sync=False, prefetched=pre, next_batch=batches[(i + 1) % 3])
    pre = p.prefetched
    if i < 6:
        eager_losses.append(p.result()[0]["sum"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(25, 45 if PHASE != "graph" else 25):
    pre = train_step(net, crit, opt, batches[i % 3], 1000, True, sync=False, prefetched=pre,
                     next_batch=batches[(i + 1) % 3]).prefetched
torch.cuda.synchronize()
print("eager : %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
if PHASE == "eager":
    sys.exit(0)

net, crit, opt = build()
gs = GraphedTrainStep(net, crit, opt, batches[0], 1000, True)
graph_losses = []
for i in range(25):
    p = gs.step(batches[(i + 1) % 3])
    if i < 6:
        graph_losses.append(p.result()[0]["sum"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(25, 45):
    gs.step(batches[(i + 1) % 3])
torch.cuda.synchronize()
print("graph : %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
print("losses eager:", ["%.5f" % v for v in eager_losses])
print("losses graph:", ["%.5f" % v for v in graph_losses])
