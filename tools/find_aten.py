"""Which Python lines of a training step run framework (aten) operators on device tensors?  (development tool)
A TorchDispatchMode around one steady-state C4 step: every aten call that touches a HIP tensor is recorded with the innermost
ogc_amd frame.  (The backward pass runs on autograd's device thread, which the mode also sees when it is entered there: ops
recorded without an ogc_amd frame are autograd's own — gradient accumulation, the derivative formulas of framework operators.)"""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, PrefetchedGeometry, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to("cuda")
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device="cuda")
pre = PrefetchedGeometry(net, crit, batch, True)
for i in range(4):
    pend = train_step(net, crit, opt, batch, 4000 + i, True, sync=False, prefetched=pre, next_batch=batch)
    pre = pend.prefetched
torch.cuda.synchronize()
main = torch.cuda.current_stream()
VIEWS = {"view", "_unsafe_view", "reshape", "transpose", "permute", "slice", "select", "unsqueeze", "squeeze", "expand", "t", "detach",
         "alias", "as_strided", "unbind", "split", "narrow", "empty", "empty_like", "empty_strided", "new_empty", "unsafe_split",
         "split_with_sizes", "_reshape_alias", "lift_fresh", "is_same_size", "sym_size", "stride", "size", "numel", "view_as", "chunk",
         "record_stream", "new_empty_strided", "unsafe_chunk", "diagonal", "unfold", "movedim", "flatten", "unflatten", "result_type"}
seen = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name not in VIEWS:
            ts = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            if isinstance(out, torch.Tensor):
                ts.append(out)
            if any(t.is_cuda for t in ts):
                where = "(no ogc_amd frame)"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "ogc_amd/" in fr.filename:
                        where = "%s:%d" % (fr.filename.split("ogc_amd/")[-1], fr.lineno)
                        break
                shape = tuple(out.shape) if isinstance(out, torch.Tensor) else ()
                s = "main" if torch.cuda.current_stream() == main else "side"
                seen[(s, where, name, shape)] += 1
        return out


with Log():
    pend = train_step(net, crit, opt, batch, 4010, True, sync=False, prefetched=pre, next_batch=batch)
torch.cuda.synchronize()
tot = collections.Counter()
for (s, where, name, shape), n in sorted(seen.items()):
    print("%-4s %-44s %-28s %2d  %s" % (s, where, name, n, shape))
    tot[s] += n
print("total", dict(tot))
