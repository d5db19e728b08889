"""Time of the MaskFormer head (2 decoder layers, 10 slots x 512 points, batch 16) forward + backward in isolation, and
its launch count (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to("cuda")
head = net.MF_head
feats = torch.randn(16, 512, 256, device="cuda", requires_grad=True)
pos = torch.randn(16, 512, 3, device="cuda")


def step():
    out = head(feats, pos)
    out.square().mean().backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("MaskFormer head fwd+bwd: %.3f ms per call (host-paced if the launch thread is the limit)" % ((time.perf_counter() - t0) / 50 * 1e3))
g = torch.cuda.CUDAGraph()
for p in head.parameters():
    p.grad = None
feats.grad = None
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print("same, replayed from a captured graph (GPU-side cost): %.3f ms per call" % ((time.perf_counter() - t0) / 50 * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
n = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print("kernels + copies per call:", n)
import collections
c = collections.Counter(e.name[:70] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
for k, v in c.most_common(40):
    print("%4d  %s" % (v, k))
