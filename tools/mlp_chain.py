"""A chain of big kernels only: the SA1 shared MLP (6->32->32->64 on 16 x 2048 x 64 positions) forward + backward,
repeated.  Prints the wall time per repetition; run it again under `rocprofv3 --kernel-trace --stats` and compare with
the sum of the kernel durations to see what a kernel boundary costs between large kernels (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.utils.nn_util import SharedMLP
torch.manual_seed(0)
bn = {"class": "GroupNorm", "num_groups": 4}
mlp = SharedMLP([6, 32, 32, 64], bn=bn).cuda()
x = torch.randn(16, 6, 2048, 64, device="cuda", requires_grad=True)


def rep():
    y = mlp.forward_maxpool(x)
    y.sum().backward()


for _ in range(5):
    rep()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    rep()
torch.cuda.synchronize()
print("wall: %.3f ms per repetition" % ((time.perf_counter() - t0) / n * 1e3))
