"""FlowStep3D correlation layer (C3 level-2 shape: B=1, 2048 points, k=16, 131->128->128->128) in inference: ms per call,
fused chain kernel vs the layer-by-layer path; and the FlowStep3D forward (8192-point pair, 5 iterations)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd import fused
from ogc_amd.utils.flowstep3d_util import FlowEmbedding


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


torch.manual_seed(0)
for B in (1, 4):
    fe = FlowEmbedding(radius=1.5, nsample=16, in_channel=64, mlp=[128, 128, 128]).cuda().eval()
    pc = (torch.rand(B, 2048, 3, device="cuda") - 0.5) * torch.tensor([60.0, 4.0, 80.0], device="cuda")
    p1 = pc.transpose(1, 2).contiguous(); p2 = (pc + 0.05 * torch.randn_like(pc)).transpose(1, 2).contiguous()
    f1, f2 = torch.randn(B, 64, 2048, device="cuda"), torch.randn(B, 64, 2048, device="cuda")
    with torch.no_grad():
        t_f = timeit(lambda: fe(p1, p2, f1, f2))
        avail = fused.mlp_chain_pool_available
        fused.mlp_chain_pool_available = lambda *a: False
        t_u = timeit(lambda: fe(p1, p2, f1, f2))
        fused.mlp_chain_pool_available = avail
        x = torch.randn(B, 131, 2048, 16, device="cuda")
        t_k = timeit(lambda: fused.mlp_chain_pool(x, fe.mlp_convs, fe.mlp_bns))
    flops = 2.0 * B * 2048 * 16 * (131 * 128 + 2 * 128 * 128)
    print("B=%d FlowEmbedding: fused %.3f ms, layer-by-layer %.3f ms; chain kernel alone %.3f ms = %.1f TFLOP/s" % (B, t_f, t_u, t_k, flops / t_k / 1e9))
if len(sys.argv) > 1:
    from ogc_amd.models.flownet_kitti import FlowStep3D
    net = FlowStep3D(npoint=8192, loc_flow_nn=16, loc_flow_rad=1.5).cuda().eval()
    pc1 = (torch.rand(1, 8192, 3, device="cuda") - 0.5) * torch.tensor([60.0, 4.0, 80.0], device="cuda")
    pc2 = pc1 + 0.05 * torch.randn_like(pc1)
    with torch.no_grad():
        t = timeit(lambda: net(pc1, pc2, pc1, pc2, iters=5), iters=10, warm=3)
    print("FlowStep3D forward (8192-point pair, 5 iterations, B=1): %.2f ms" % t)
