"""The slot branch's linear layer (160 x 128 -> 128) on an idle GPU: F.linear vs the single-launch kernels, forward and
forward + backward, microseconds per call from HIP events over 200 calls.  Development tool."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: F401
from ogc_amd.fused import small_linear

rows, ni, no = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (160, 128, 128))]
x = torch.randn(rows, ni, device="cuda", requires_grad=True)
lin = torch.nn.Linear(ni, no).cuda()
g = torch.randn(rows, no, device="cuda")


def timeit(fn, bwd, reps=200):
    for _ in range(10):
        y = fn(x, lin.weight, lin.bias)
        if bwd:
            y.backward(g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = fn(x, lin.weight, lin.bias)
        if bwd:
            x.grad = lin.weight.grad = lin.bias.grad = None
            y.backward(g)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in (("F.linear", F.linear), ("small_linear", small_linear)):
    print("%-13s forward %6.1f us   forward+backward %6.1f us" % (name, timeit(fn, False), timeit(fn, True)))
