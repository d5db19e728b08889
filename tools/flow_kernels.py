"""Kernels of one FlowStep3D forward (C3: one 8192-point pair, iters = 5): name, launches, total and mean time.  (development tool)"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.flownet_kitti import FlowStep3D
from ogc_amd.utils.synthetic import make_scene_batch
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(0)
N = 8192
net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).to("cuda").eval()
pcs = make_scene_batch(1, N, 10, seed=1, aug=False, device="cuda")[0]
pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
with torch.no_grad():
    for _ in range(3):
        net(pc1, pc2, pc1, pc2, iters=5)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net(pc1, pc2, pc1, pc2, iters=5)
        torch.cuda.synchronize()
cnt, tot = collections.Counter(), collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        cnt[ev.name[:90]] += 1
        tot[ev.name[:90]] += ev.device_time if hasattr(ev, "device_time") else ev.cuda_time
print("%d kernels, %.2f ms of kernel time" % (sum(cnt.values()), sum(tot.values()) / 1e3))
for name, t in tot.most_common(70):
    print("%4d  %8.1f us  %6.2f us each  %s" % (cnt[name], t, t / cnt[name], name))
