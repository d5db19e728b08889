"""Which Python lines of a training step make a COPY out of `.contiguous()` / `.clone()` / `torch.cat`?  (development tool)
torch.profiler's stacks come back empty on this build, so the three calls are wrapped and the caller's frame is recorded when a
new tensor is produced.  Prints call site, shape and stream for one steady-state C4 step."""
import collections, os, sys, traceback
import torch
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

seen = collections.Counter()
main = torch.cuda.current_stream()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "ogc_amd" in fr.filename and "find_copies" not in fr.filename:
            return "%s:%d" % (fr.filename.split("ogc_amd/")[-1], fr.lineno)
    return "?"


def note(kind, t):
    s = "main" if torch.cuda.current_stream() == main else "side"
    seen[(kind, site(), tuple(t.shape), s)] += 1


_contig, _clone, _cat, _stack = torch.Tensor.contiguous, torch.Tensor.clone, torch.cat, torch.stack


def contiguous(self, *a, **k):
    if self.is_cuda and not self.is_contiguous():
        note("contiguous", self)
    return _contig(self, *a, **k)


def clone(self, *a, **k):
    if self.is_cuda:
        note("clone", self)
    return _clone(self, *a, **k)


def cat(ts, *a, **k):
    out = _cat(ts, *a, **k)
    if out.is_cuda:
        note("cat", out)
    return out


def stack(ts, *a, **k):
    out = _stack(ts, *a, **k)
    if out.is_cuda:
        note("stack", out)
    return out


torch.Tensor.contiguous, torch.Tensor.clone, torch.cat, torch.stack = contiguous, clone, cat, stack
pend = train_step(net, crit, opt, batch, 4010, True, sync=False, prefetched=pre, next_batch=batch)
torch.cuda.synchronize()
torch.Tensor.contiguous, torch.Tensor.clone, torch.cat, torch.stack = _contig, _clone, _cat, _stack
for (kind, where, shape, s), n in sorted(seen.items(), key=lambda kv: (kv[0][3], kv[0][0], kv[0][1])):
    print("%-10s %-4s %2d  %-48s %s" % (kind, s, n, where, shape))
print("total", sum(seen.values()))
