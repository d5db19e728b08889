"""Weight gradient of a 1x1 convolution at the C4 layer shapes: the MFMA kernel against rocBLAS formulations (development tool)."""
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
                         (16, 128, 128, 32768), (16, 131, 128, 32768), (16, 128, 256, 32768), (16, 384, 128, 1024), (16, 128, 128, 1024), (16, 224, 64, 2048), (16, 64, 64, 2048), (16, 67, 64, 8192), (16, 64, 64, 8192), (16, 64, 64, 16384)]:
    g = torch.randn(B, cout, hw, device="cuda")
    x = torch.randn(B, cin, hw, device="cuda")
    dw = torch.empty(cout, cin, device="cuda")
    mine = timeit(lambda: nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, g, dw))
    bmm = timeit(lambda: torch.bmm(g, x.transpose(1, 2)).sum(0))
    # one GEMM over the whole batch: (cout, B*hw) x (B*hw, cin) needs channel-major tensors -> permute copies; time them too
    def flat():
        gp = g.permute(1, 0, 2).reshape(cout, B * hw)
        xp = x.permute(1, 0, 2).reshape(cin, B * hw)
        return gp @ xp.t()
    fl = timeit(flat)
    ein = timeit(lambda: torch.einsum("bmp,bkp->mk", g, x))
    print("%4d->%-4d hw=%-6d wgrad: mfma %.3f ms   bmm+sum %.3f ms   permute+gemm %.3f ms   einsum %.3f ms" % (cin, cout, hw, mine, bmm, fl, ein))
