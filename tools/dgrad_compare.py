"""Input gradient of a 1x1 convolution at the C4 layer shapes: the MFMA kernel (transpose_a = 1) against
torch.matmul(w^T, grad) = rocBLAS (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.pointnet2 import pointnet2 as api
nat = api._native


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B, cin, cout, hw in [(16, 32, 32, 131072), (16, 32, 64, 131072), (16, 64, 64, 65536), (16, 64, 128, 65536), (16, 99, 64, 65536),
                         (16, 128, 128, 32768), (16, 131, 128, 32768), (16, 128, 256, 32768), (16, 64, 64, 8192), (16, 128, 128, 1024)]:
    w = torch.randn(cout, cin, device="cuda")
    g = torch.randn(B, cout, hw, device="cuda")
    x = torch.randn(B, cin, hw, device="cuda")
    dx = torch.empty(B, cin, hw, device="cuda")
    y = torch.empty(B, cout, hw, device="cuda")
    mine = timeit(lambda: nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, g, dx)) if cout <= 160 else float("nan")
    wt = w.t().contiguous()
    vend = timeit(lambda: torch.matmul(wt, g, out=dx))
    vend2 = timeit(lambda: torch.matmul(w.t(), g))
    fm = timeit(lambda: nat.conv1x1_gemm_wrapper(B, cout, cin, hw, 0, w, x, y)) if cin <= 160 else float("nan")
    fv = timeit(lambda: torch.matmul(w, x, out=y))
    print("%4d->%-4d hw=%-6d dgrad: mfma %.3f ms  matmul(out=) %.3f ms  matmul %.3f ms | fwd: mfma %.3f ms  matmul(out=) %.3f ms" %
          (cin, cout, hw, mine, vend, vend2, fm, fv))
