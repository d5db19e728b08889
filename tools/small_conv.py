"""1x1 convolution at small position counts (FlowStep3D's B = 1 layers): this repo's MFMA kernel against the vendor path
torch takes (development tool)."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.pointnet2 import pointnet2 as api
nat = api._native


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n // 20):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (n // 20 * 20) * 1e6
    except Exception:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6


for cin, cout in [(128, 128), (64, 64), (131, 128), (256, 128), (64, 128)]:
    for hw in [int(a) for a in os.environ.get("HW", "256,512,1024,2048,4096,8192,16384").split(",")]:
        x = torch.randn(1, cin, hw, device="cuda")
        w = torch.randn(cout, cin, device="cuda")
        y = torch.empty(1, cout, hw, device="cuda")
        w3 = w.view(cout, cin, 1)
        mine = timeit(lambda: nat.conv1x1_gemm_wrapper(1, cout, cin, hw, 0, w, x, y)) if cin <= 160 else float("nan")
        vend = timeit(lambda: F.conv1d(x, w3))
        mm = timeit(lambda: torch.matmul(w, x[0]))
        print("%4d->%-4d hw=%-6d  mfma kernel %7.1f us   F.conv1d %7.1f us   torch.matmul %7.1f us" % (cin, cout, hw, mine, vend, mm))
