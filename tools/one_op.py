"""Run ONE operator a few times (for rocprofv3 --pmc / --kernel-trace runs).  python tools/one_op.py ball|knn|fps|..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd import pointnet2_cuda as nat
op = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 16; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = torch.Generator().manual_seed(1234)
pc = ((torch.rand(B, 8192, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda().contiguous()
for _ in range(reps):
    if op == "ball":
        idx = torch.zeros(B, 8192, 64, dtype=torch.int32, device="cuda")
        nat.ball_query_wrapper(B, 8192, 8192, 2.0, 64, pc, pc, idx)
    elif op == "knn":
        d2 = torch.empty(B, 8192, 32, device="cuda"); idx = torch.empty(B, 8192, 32, dtype=torch.int32, device="cuda")
        nat.knn_wrapper(B, 8192, 8192, 32, pc, pc, d2, idx)
    elif op == "knnc":   # C4 smoothness term: k = 32, clamp at 1 m
        d2 = torch.empty(B, 8192, 32, device="cuda"); idx = torch.empty(B, 8192, 32, dtype=torch.int32, device="cuda")
        nat.knn_clamped_wrapper(B, 8192, 8192, 32, 1.0, pc, pc, d2, idx)
    elif op == "fps":
        idx = torch.empty(B, 2048, dtype=torch.int32, device="cuda"); temp = torch.full((B, 8192), 1e10, device="cuda")
        nat.furthest_point_sampling_wrapper(B, 8192, 2048, pc, temp, idx)
if op == "conv":  # 128 -> 128 on (B, 128, 512*64): the MFMA-bound layer shape of SA3
    x = torch.randn(B, 128, 32768, device="cuda"); w = torch.randn(128, 128, device="cuda")
    y = torch.empty(B, 128, 32768, device="cuda"); dw = torch.empty(128, 128, device="cuda")
    for _ in range(reps):
        nat.conv1x1_gemm_wrapper(B, 128, 128, 32768, 0, w, x, y)
        nat.conv1x1_gemm_wrapper(B, 128, 128, 32768, 1, w, y, x)
        nat.conv1x1_wgrad_wrapper(B, 128, 128, 32768, x, y, dw)
torch.cuda.synchronize()
