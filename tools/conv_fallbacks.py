"""Which convolutions of a C4 training step still go to torch's conv (MIOpen) instead of this repo's kernels / rocBLAS?
Prints one line per distinct (op, input shape, weight shape) with its call count for ONE step."""
import collections
import torch
import torch.nn.functional as F

import ogc_amd  # noqa: F401
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

dev = torch.device("cuda", 0)
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                   transformer_input_pos_enc=False).to(dev)
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3, weight_decay=0.0)
batch = make_scene_batch(4, 8192, 10, seed=1234, outdoor=True, aug=True, device=dev)
pre = None
for _ in range(2):
    pre = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
torch.cuda.synchronize()

seen = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if "convolution" in name:
            shapes = tuple(tuple(a.shape) + (("contig" if a.is_contiguous() else "strided"),) for a in args[:3]
                           if isinstance(a, torch.Tensor))
            seen[(name, shapes)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch)
torch.cuda.synchronize()
for (name, shapes), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d x %-40s %s" % (n, name, shapes))
