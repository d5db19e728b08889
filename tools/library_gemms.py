"""Which matrix products of a C4 training step still go to the vendor library (rocBLAS / hipBLASLt through aten::mm, bmm,
addmm, baddbmm, linear) instead of this repo's MFMA kernels?  One line per distinct (op, operand shapes / strides) with its
call count for ONE step and its time on the idle GPU (median of 20 back-to-back calls on fresh tensors of those shapes)."""
import collections
import sys

import torch

import ogc_amd  # noqa: F401
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch
from torch.utils._python_dispatch import TorchDispatchMode

dev = torch.device("cuda", 0)
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                   transformer_input_pos_enc=False).to(dev)
# (the slot branch runs as a HIP graph in training: its products would not pass the dispatcher — list them eagerly)
net.graph_slot_branch = "--graphed" in sys.argv
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3, weight_decay=0.0)
batch = make_scene_batch(4, 8192, 10, seed=1234, outdoor=True, aug=True, device=dev)
pre = None
for _ in range(2):
    pre = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
torch.cuda.synchronize()

GEMM_OPS = ("aten.mm.", "aten.bmm.", "aten.addmm.", "aten.baddbmm.", "aten.linear.", "aten.matmul.", "aten._scaled_mm")
seen = collections.Counter()
examples = {}


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(name.startswith(op) for op in GEMM_OPS):
            key = (name, tuple((tuple(a.shape), tuple(a.stride())) for a in args if isinstance(a, torch.Tensor)))
            seen[key] += 1
            examples.setdefault(key, (func, args, kwargs))
        return func(*args, **(kwargs or {}))


with Log():
    train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch)
torch.cuda.synchronize()


def clock(func, args, kwargs):
    fresh = [torch.randn(a.shape, device=dev).as_strided(a.shape, a.stride()) if isinstance(a, torch.Tensor) and not a.is_contiguous()
             and max(a.stride()) * max(a.shape) <= a.numel() * 4 else (torch.randn_like(a) if isinstance(a, torch.Tensor) else a)
             for a in args]
    for _ in range(3):
        func(*fresh, **(kwargs or {}))
    times = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        func(*fresh, **(kwargs or {}))
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    times.sort()
    return times[len(times) // 2]


rows = []
for key, n in seen.items():
    func, args, kwargs = examples[key]
    try:
        us = clock(func, args, kwargs)
    except Exception as err:  # a view the fresh tensors cannot reproduce: counted, not timed
        us = float("nan")
    flops = 0
    shapes = [s for s, _ in key[1]]
    if "bmm" in key[0] and len(shapes) >= 2:
        flops = 2 * shapes[-2][0] * shapes[-2][1] * shapes[-2][2] * shapes[-1][2]
    elif len(shapes) >= 2 and len(shapes[-1]) == 2 and len(shapes[-2]) == 2:
        flops = 2 * shapes[-2][0] * shapes[-2][1] * shapes[-1][1]
    rows.append((n * us, n, us, flops, key))
rows.sort(key=lambda r: -(r[0] if r[0] == r[0] else 0))
total = sum(r[0] for r in rows if r[0] == r[0])
print("library matrix products of one C4 step (slot branch %s): %d calls, %.0f us on the idle GPU (event-timed, launch floor ~5 us each included)"
      % ("graphed: not listed" if net.graph_slot_branch else "eager", sum(r[1] for r in rows), total))
for tot, n, us, flops, (name, ops) in rows:
    print("%3d x %8.1f us = %8.1f us  %6.1f TF  %-18s %s" % (n, us, tot, flops / max(us, 1e-9) / 1e6, name.replace("aten.", ""),
                                                           "  ".join("%s/%s" % (list(s), list(st)) for s, st in ops)))
