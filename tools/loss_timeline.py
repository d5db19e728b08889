"""GPU time / host issue time / kernel launches of each loss term, forward and backward (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.train_step import KITTI_LOSS, build_criterion
from ogc_amd.utils.synthetic import make_scene_batch

dev = "cuda"
torch.manual_seed(0)
crit = build_criterion(KITTI_LOSS)
pcs, segms, flows, _ = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device=dev)
t = 4
pcs_l = [pcs[:, i].contiguous() for i in range(t)]
flows_l = [flows[:, i].contiguous() for i in range(t)]
logits = torch.randn(4, t, 8192, 10, device=dev, requires_grad=True)


def masks():
    m = logits.softmax(-1)
    return [m[:, i].contiguous() for i in range(t)]


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); cpu = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, cpu


def count_kernels(fn):
    from torch.profiler import profile, ProfilerActivity
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    return sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)


geo = crit.plan_geometry(pcs_l, True)
terms = {
    "dynamic": lambda m: sum(crit.dynamic_loss.forward_views(pcs_l, m, flows_l)),
    "smooth": lambda m: sum(crit.smooth_loss.forward_views(pcs_l, m, geo)),
    "invariance": lambda m: sum(crit.invariance_loss.forward_pairs([(m[0], m[2]), (m[1], m[3])])),
    "entropy(mon)": lambda m: sum(crit.entropy_loss(x) for x in m),
    "rank(mon)": lambda m: sum(crit.rank_loss(x) for x in m),
    "softmax+views only": lambda m: sum(x.sum() for x in m),
}
print("%-20s %9s %9s %8s | %9s %9s %8s" % ("term", "fwd gpu", "fwd cpu", "kernels", "f+b gpu", "f+b cpu", "kernels"))
for name, fn in terms.items():
    mon = "(mon)" in name
    def fwd():
        with torch.no_grad():
            return fn(masks())
    def both():
        logits.grad = None
        fn(masks()).backward()
    g0, c0 = timed(fwd)
    k0 = count_kernels(fwd)
    if mon:
        print("%-20s %9.3f %9.3f %8d |" % (name, g0, c0, k0))
        continue
    g1, c1 = timed(both)
    k1 = count_kernels(both)
    print("%-20s %9.3f %9.3f %8d | %9.3f %9.3f %8d" % (name, g0, c0, k0, g1, c1, k1))
