"""When do the two branches of the segmentation net's backward pass finish?  Identity autograd nodes record HIP events
(on the stream their branch runs on) at: start of backward, start / end of the slot branch's backward, end of the
feature-propagation stack's backward.  Un-traced C4 steps.  Development tool."""
import os, sys
import torch
from torch.autograd import Function
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: F401
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

EVENTS = []


class _Mark(Function):
    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        EVENTS.append((ctx.tag, ev))
        return g, None


torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to("cuda")
net.overlap_head = not os.environ.get("NO_OVERLAP")
orig_slots = net._slots
FWD = {}


def _ev(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    FWD[tag] = e


def slots_probe(cf, cp):
    _ev("fork")
    out = _Mark.apply(orig_slots(_Mark.apply(cf, "slot branch end"), cp), "slot branch start")
    _ev("slot branch end (fwd)")
    return out


net._slots = slots_probe
fp_first = net.FP_modules[0]
orig_fp0 = fp_first.forward


def fp0_forward(*a, **k):
    out = orig_fp0(*a, **k)
    _ev("FP stack end (fwd)")
    return out


fp_first.forward = fp0_forward
fp_last = net.FP_modules[-1]
orig_fp = fp_last.forward


def fp_forward(unknown, known, unknow_feats, known_feats, **kw):
    return orig_fp(unknown, known, unknow_feats, _Mark.apply(known_feats, "FP stack end"), **kw)


fp_last.forward = fp_forward
orig_forward = net.forward
net.forward = lambda *a, **k: _Mark.apply(orig_forward(*a, **k), "backward start")
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device="cuda")
pre = None
for i in range(12):
    EVENTS.clear()
    pend = train_step(net, crit, opt, batch, 4000 + i, True, sync=False, prefetched=pre, next_batch=batch)
    pre = pend.prefetched
    end = torch.cuda.Event(enable_timing=True)
    end.record()
torch.cuda.synchronize()
ev = dict(EVENTS)
t0 = ev["backward start"]
for tag in ("slot branch start", "slot branch end", "FP stack end"):
    print("%-20s +%.3f ms" % (tag, t0.elapsed_time(ev[tag])))
print("%-20s +%.3f ms" % ("step end", t0.elapsed_time(end)))
for tag in ("slot branch end (fwd)", "FP stack end (fwd)"):
    print("%-22s +%.3f ms after the fork" % (tag, FWD["fork"].elapsed_time(FWD[tag])))
