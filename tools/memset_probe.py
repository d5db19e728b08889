"""hipMemsetAsync inside a HIP graph on this stack: the weight-gradient entry point (which zeroes its output, then adds with
atomics) captured and replayed with the output poisoned in between — exact zeros must come back where the operands are zero.
With hipMemsetAsync in the library the second replay returned ~3e-41 there (the first was right; eager calls always are); with
the zero-fill kernel of csrc/ogc_common.h it is exact.     PYTHONPATH=. python tools/memset_probe.py"""
import torch, ogc_amd
from ogc_amd import pointnet2_cuda as nat
B, cin, cout, hw = 2, 128, 128, 64
x = torch.zeros(B, cin, hw, device="cuda"); x[:, ::2] = torch.randn(B, cin // 2, hw, device="cuda")  # every other input channel is zero
dy = torch.randn(B, cout, hw, device="cuda")
dw = torch.empty(cout, cin, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, dw)   # warm-up
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    dw2 = torch.empty(cout, cin, device="cuda")
    nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, dw2)
    t3 = torch.empty(1000, device="cuda"); t3.zero_()
for r in range(3):
    dw2.fill_(1e30); t3.fill_(7.0)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("replay", r, "odd columns exact zero:", bool((dw2[:, 1::2] == 0).all()), "max |odd|", float(dw2[:, 1::2].abs().max()), "| torch zero_ in graph:", float(t3.abs().max()))
for r in range(4):
    dw.fill_(1e30); torch.cuda.synchronize()
    nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, dw); torch.cuda.synchronize()
    print("eager", r, "odd columns exact zero:", bool((dw[:, 1::2] == 0).all()), "max |odd|", float(dw[:, 1::2].abs().max()))
# bigger shape like C4
B, cin, cout, hw = 16, 128, 128, 32768
x = torch.zeros(B, cin, hw, device="cuda"); x[:, ::2] = torch.randn(B, cin // 2, hw, device="cuda")
dy = torch.randn(B, cout, hw, device="cuda"); dw = torch.empty(cout, cin, device="cuda")
for r in range(3):
    dw.fill_(1e30); torch.cuda.synchronize()
    nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, dw); torch.cuda.synchronize()
    print("eager big", r, "odd columns exact zero:", bool((dw[:, 1::2] == 0).all()), "max |odd|", float(dw[:, 1::2].abs().max()))
