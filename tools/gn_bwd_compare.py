"""Backward of conv(relu(GroupNorm(y_prev))) at the C4 layer shapes: separate passes vs the moment-matrix path, per kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd import pointnet2_cuda as nat


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


B = 16
for cin, cout, hw in [(32, 32, 131072), (32, 64, 131072), (64, 64, 65536), (64, 128, 65536), (128, 128, 32768), (128, 256, 32768)]:
    G = 4
    y_prev = torch.randn(B, cin, hw, device="cuda")
    gy = torch.randn(B, cout, hw, device="cuda")
    w = torch.randn(cout, cin, device="cuda") / cin ** 0.5
    a, bb = torch.rand(B, cin, device="cuda") + 0.5, torch.randn(B, cin, device="cuda")
    mean, rstd = torch.randn(B * G, device="cuda"), torch.rand(B * G, device="cuda") + 0.5
    gamma, beta = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    dw = torch.empty(cout, cin, device="cuda"); gz = torch.empty_like(y_prev); gp = torch.empty_like(y_prev)
    gw, gb = torch.empty(cin, device="cuda"), torch.empty(cin, device="cuda")
    ws = nat.group_norm_ws(B, cin, G, True, "cuda")
    t_w = timeit(lambda: nat.conv1x1_wgrad_affine_wrapper(B, cin, cout, hw, 1, y_prev, a, bb, gy, dw))
    if cout <= 160:
        t_d = timeit(lambda: nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, gy, gz))
    else:
        t_d = float("nan")
    t_dr = timeit(lambda: torch.matmul(w.t(), gy, out=gz))
    t_g = timeit(lambda: nat.group_norm_bwd_wrapper(B, cin, hw, G, 1, y_prev, gamma, beta, mean, rstd, gz, gp, gw, gb, ws))
    line = "%3d->%3d hw=%6d | old: wgrad %.3f dgrad(mine %.3f / rocblas %.3f) gn_bwd %.3f = %.3f" % (
        cin, cout, hw, t_w, t_d, t_dr, t_g, t_w + min(t_d if t_d == t_d else 9, t_dr) + t_g)
    if cout <= 160:
        mom = torch.empty(B, 2, cout, cin, device="cuda"); coef = torch.empty(B, cin, 3, device="cuda")
        ggb = torch.empty(2, cin, device="cuda"); gw, gb = ggb[0], ggb[1]
        t_m = timeit(lambda: nat.conv1x1_wgrad_moments_wrapper(B, cin, cout, hw, 1, y_prev, a, bb, gy, mom))
        t_c = timeit(lambda: nat.gn_moments_combine_wrapper(B, cin, cout, hw, G, mom, w, a, bb, mean, rstd, gamma, dw, coef, gw, gb))
        t_a = timeit(lambda: nat.conv1x1_dgrad_adjoint_wrapper(B, cin, cout, hw, 1, w, gy, y_prev, a, bb, coef, gp))
        line += " | new: moments %.3f combine %.3f dgrad_adj %.3f = %.3f" % (t_m, t_c, t_a, t_m + t_c + t_a)
    print(line)
