"""The C4-shaped training step (small clouds) as ONE HIP graph WITH the flat gradient all-reduce over RCCL inside the capture:
one process, backend nccl, FlatDataParallel with the collective forced on.  Prints eager and replayed losses of six steps.
   python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/graph_rccl.py [npoint]"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa
from ogc_amd.graph_step import GraphedTrainStep
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils import dist_util
from ogc_amd.utils.synthetic import make_scene_batch

npoint = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dist_util.always_reduce(True)
dev = torch.device("cuda", 0)


def build():
    torch.manual_seed(10)
    net = MaskFormer3D(n_slot=10, n_point=npoint, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                       transformer_input_pos_enc=False).to(dev)
    return dist_util.FlatDataParallel(net), build_criterion(KITTI_LOSS), make_optimizer(net.parameters(), lr=1e-3, capturable=True)


batches = [make_scene_batch(2, npoint, 10, seed=77 + i, outdoor=True, aug=True, device=dev) for i in range(3)]
net, crit, opt = build()
eager, pre = [], None
for i in range(6):
    p = train_step(net, crit, opt, batches[i % 3], 1000, True, sync=False, prefetched=pre, next_batch=batches[(i + 1) % 3])
    pre = p.prefetched
    eager.append(p.result()[0]["sum"])
net, crit, opt = build()
gs = GraphedTrainStep(net, crit, opt, batches[0], 1000, True)
graph = [gs.step(batches[(i + 1) % 3]).result()[0]["sum"] for i in range(6)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6, 26):
    gs.step(batches[(i + 1) % 3])
torch.cuda.synchronize()
print("graph+rccl: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
print("losses eager:", ["%.5f" % v for v in eager])
print("losses graph:", ["%.5f" % v for v in graph])
worst = max(abs(a - b) / max(1.0, abs(a)) for a, b in zip(eager, graph))
print("GRAPH_RCCL_OK" if worst <= 3e-4 else "GRAPH_RCCL_MISMATCH", worst)
dist.destroy_process_group()
