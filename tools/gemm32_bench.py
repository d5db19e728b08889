"""Times the narrow fp32 forward layers of C4 (development tool): persistent kernel (default) / tile kernel (OGC_GEMM32=0)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.pointnet2 import pointnet2 as api
nat = api._native
shapes = [("SA1 32->32 stats", 16, 32, 32, 2048, 64, False), ("SA1 32->32 pool", 16, 32, 32, 2048, 64, True),
          ("SA1 32->64 pool", 16, 32, 64, 2048, 64, True), ("SA2 64->64 stats", 16, 64, 64, 1024, 64, False),
          ("SA2 64->128 pool", 16, 64, 128, 1024, 64, True)]
for name, B, cin, cout, P, S, pool in shapes:
    hw = P * S
    x = torch.randn(B, cin, hw, device="cuda")
    w = torch.randn(cout, cin, device="cuda") / cin ** 0.5
    pa, pb = torch.rand(B * cin, device="cuda") + 0.5, torch.randn(B * cin, device="cuda")
    y = torch.empty(B, cout, hw, device="cuda")
    st = torch.zeros(nat.conv1x1_gn_slots() * B * 4 * 2, dtype=torch.float64, device="cuda")
    gamma = torch.randn(cout, device="cuda")
    yext = torch.empty(B, cout, P, device="cuda")
    aext = torch.empty(B, cout, P, dtype=torch.int32, device="cuda")
    def run():
        if pool:
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
    gb = B * hw * (cin + cout) * 4 / 1e9
    print("%-22s %.3f ms  %.2f GB  %.2f TB/s" % (name, ms, gb, gb / ms))
