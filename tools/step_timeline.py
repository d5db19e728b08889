"""Where does a training step spend its time, and who is waiting for whom?  (development tool)

Replays the body of train_step with an event and a host timestamp at every phase boundary and prints, per boundary,
when the launch thread passed it and when the GPU did: `lead` = GPU time - host time is how far ahead of the GPU the
launch thread is there (near zero: the GPU is waiting for launches).

Measured (round 1, C4): the launch thread needs ~12.3 ms per step and runs 40-60 ms (two to three steps) ahead of the
GPU until the queue depth stops it, so the step is GPU-bound; the multi-millisecond idle stretches a rocprofv3 kernel
trace of the same command shows on the main queue (tools/gaps.py) come from the tracer's per-launch cost on the
launch thread, not from the program.
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, PrefetchedGeometry, build_criterion, make_optimizer
from ogc_amd.utils.synthetic import make_scene_batch

dev = "cuda"
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to(dev)
crit = build_criterion(KITTI_LOSS)
nb = int(os.environ.get("B", "4"))
batch = make_scene_batch(nb, 8192, 10, seed=1234, aug=True, device=dev)
pcs, segms, flows, _ = batch
b, t, n = segms.size()
opt = make_optimizer(net.parameters(), lr=1e-3)
state = {"pre": None}
names = ["start", "forward", "loss", "backward", "-", "optimizer"]


def step(record=None):
    marks = []

    def mark():
        if os.environ.get("MARK"):
            torch.cuda._sleep(2000)  # shows up in a kernel trace as a marker between the phases (tools/phase_busy.py)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((ev, time.perf_counter()))

    net.train(); opt.zero_grad(set_to_none=True)
    pre = state["pre"] or PrefetchedGeometry(net, crit, batch, True)
    mark()
    flat, pcs_l, flows_l = pre.flat, pre.pcs_l, pre.flows_l
    masks = net(flat, flat, geometry=pre.model).view(b, t, n, -1)
    masks_l = [masks[:, i].contiguous() for i in range(t)]
    state["pre"] = PrefetchedGeometry(net, crit, batch, True)
    mark()
    loss, ld = crit(pcs_l, masks_l, flows_l, step_w=True, it=4000, aug_transform=True, geometry=pre.loss, sync=False)
    mark()
    loss.backward()
    mark()
    mark()
    from ogc_amd.train_step import _adam_kernel_step
    if _adam_kernel_step(opt) is None:  # (the first step: torch builds the optimizer's state)
        grads = [p.grad for p in net.parameters() if p.grad is not None]
        bad = torch.isnan(torch.stack(torch._foreach_norm(grads)).sum())
        opt.grad_scale = None; opt.found_inf = bad.float().reshape(())
        opt.step()
        del opt.grad_scale, opt.found_inf
    mark()
    if record is not None:
        record.append(marks)


for _ in range(4):
    step()
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True); base.record(); t_base = time.perf_counter()
torch.cuda.synchronize()
rec = []
n_steps = 8
for _ in range(n_steps):
    step(rec)
torch.cuda.synchronize()
t_total = (time.perf_counter() - t_base) * 1e3
print("B=%d: %.2f ms/step over %d steps from an idle start" % (nb, t_total / n_steps, n_steps))
print("%4s %-10s %10s %10s %8s" % ("step", "boundary", "host_ms", "gpu_ms", "lead"))
for i, marks in enumerate(rec):
    if i < n_steps - 3:
        continue
    for nme, (ev, th) in zip(names, marks):
        g = base.elapsed_time(ev)
        h = (th - t_base) * 1e3
        print("%4d %-10s %10.2f %10.2f %8.2f" % (i, nme, h, g, g - h))
last, prev = rec[-1], rec[-2]
print("per phase (last step): host issue / GPU span")
for k in range(1, len(names)):
    print("  %-10s host %6.2f ms   gpu %6.2f ms" % (names[k], (last[k][1] - last[k - 1][1]) * 1e3,
                                                   last[k - 1][0].elapsed_time(last[k][0])))
print("  step-to-step: host %.2f ms, gpu %.2f ms" % ((last[0][1] - prev[0][1]) * 1e3, prev[0][0].elapsed_time(last[0][0])))
