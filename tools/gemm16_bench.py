"""Times the forward kernels of the 16-bit path at C2's layer shapes (development tool):
    python tools/gemm16_bench.py            # persistent kernel (conv1x1_h.hip);  OGC_GEMM16=0: the tile kernel"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.pointnet2 import pointnet2 as api
nat = api._native
nat.set_matmul_precision("bf16")
BF = torch.bfloat16
shapes = [("SA1 64->64 stats", 32, 64, 64, 2048, 64, False), ("SA1 64->64 pool", 32, 64, 64, 2048, 64, True),
          ("SA1 64->128 pool", 32, 64, 128, 2048, 64, True), ("SA2 128->128 stats", 32, 128, 128, 1024, 64, False),
          ("SA2 128->256 pool", 32, 128, 256, 1024, 64, True), ("SA2 dgrad 128<-128", 32, 128, 128, 1024, 64, None)]
for name, B, cin, cout, P, S, pool in shapes:
    hw = P * S
    x = torch.randn(B, cin, hw, device="cuda").to(BF)
    w = torch.randn(cout, cin, device="cuda") / cin ** 0.5
    pa, pb = torch.rand(B * cin, device="cuda") + 0.5, torch.randn(B * cin, device="cuda")
    y = torch.empty(B, cout, hw, device="cuda", dtype=BF)
    st = torch.zeros(nat.conv1x1_gn_slots() * B * 4 * 2, dtype=torch.float64, device="cuda")
    gamma = torch.randn(cout, device="cuda")
    yext = torch.empty(B, cout, P, device="cuda")
    aext = torch.empty(B, cout, P, dtype=torch.int32, device="cuda")
    def run():
        if pool is None:
            nat.conv1x1_gemm_wrapper(B, cout, cin, hw, 1, w.t().contiguous() if False else w, x, y)
        elif pool:
            nat.conv1x1_gemm_affine_pool_wrapper(B, cout, cin, hw, 1, 4, S, w, x, pa, pb, gamma, y, st, yext, aext)
        else:
            nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, 4, w, x, pa, pb, y, st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    gb = B * hw * (cin + cout) * 2 / 1e9
    print("%-22s %.3f ms  %.2f GB  %.2f TB/s" % (name, ms, gb, gb / ms))
