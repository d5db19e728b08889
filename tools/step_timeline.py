"""Where does a training step spend its time?  CPU issue time vs GPU time per phase (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion
from ogc_amd.utils.synthetic import make_scene_batch
from ogc_amd.utils.streams import launch_on_side, side_stream

dev = "cuda"
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to(dev)
crit = build_criterion(KITTI_LOSS)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device=dev)
pcs, segms, flows, _ = batch
b, t, n = segms.size()


from ogc_amd.train_step import PrefetchedGeometry, make_optimizer
opt = make_optimizer(net.parameters(), lr=1e-3)
state = {"pre": None}


def step(record=None):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    cpu = [time.perf_counter()]
    net.train(); opt.zero_grad(set_to_none=True)
    pre = state["pre"] or PrefetchedGeometry(net, crit, batch, True)
    ev[0].record()
    masks = net(pre.flat, pre.flat, geometry=pre.model).view(b, t, n, -1)
    ev[1].record(); cpu.append(time.perf_counter())
    masks_l = [masks[:, i].contiguous() for i in range(t)]
    state["pre"] = PrefetchedGeometry(net, crit, batch, True)
    loss, ld = crit(pre.pcs_l, masks_l, pre.flows_l, step_w=True, it=4000, aug_transform=True, geometry=pre.loss, sync=False)
    ev[2].record(); cpu.append(time.perf_counter())
    loss.backward()
    ev[3].record(); cpu.append(time.perf_counter())
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    bad = torch.isnan(torch.stack(torch._foreach_norm(grads)).sum())
    opt.grad_scale = None; opt.found_inf = bad.float().reshape(())
    ev[4].record(); cpu.append(time.perf_counter())
    opt.step()
    del opt.grad_scale, opt.found_inf
    ev[5].record(); cpu.append(time.perf_counter())
    cpu.append(time.perf_counter())
    if record is not None:
        record.append((ev, [1e3 * (cpu[i + 1] - cpu[i]) for i in range(6)]))


for _ in range(3):
    step()
rec = []
for _ in range(5):
    step(rec)
names = ["forward", "loss", "backward", "nan-check", "optimizer", "final-sync"]
torch.cuda.synchronize()
g = [sum(r[0][i].elapsed_time(r[0][i + 1]) for r in rec) / len(rec) for i in range(5)]
gap = sum(rec[j][0][5].elapsed_time(rec[j + 1][0][0]) for j in range(len(rec) - 1)) / (len(rec) - 1)
print("between steps (opt end -> next forward start): %.2f ms" % gap)
c = [sum(r[1][i] for r in rec) / len(rec) for i in range(6)]
print("phase        gpu_ms(event)  cpu_ms(issue)")
for i, nme in enumerate(names):
    print("%-12s %10s %12.2f" % (nme, ("%.2f" % g[i]) if i < 5 else "-", c[i]))
print("total gpu %.2f  total cpu %.2f" % (sum(g), sum(c)))
