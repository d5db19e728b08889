"""FlowStep3D (config C3: flownet_kitti, 8192-point pairs, iters=5) forward / training-step timing (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.flownet_kitti import FlowStep3D
from ogc_amd.losses.flow_loss_unsup import ChamferLoss, SmoothLoss, UnsupervisedFlowStep3DLoss
from ogc_amd.utils.synthetic import make_scene_batch

dev = "cuda"
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).to(dev)
pcs, _, flows, _ = make_scene_batch(B, N, 10, seed=1, aug=False, device=dev)
pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


net.eval()
with torch.no_grad():
    print("forward eval iters=5: %.2f ms" % timed(lambda: net(pc1, pc2, pc1, pc2, iters=5)))
    # the launch thread's share: time until the call returns (nothing waits for the GPU inside the forward)
    import time as _t
    torch.cuda.synchronize(); t0 = _t.perf_counter()
    for _ in range(5):
        net(pc1, pc2, pc1, pc2, iters=5)
    t_issue = (_t.perf_counter() - t0) / 5 * 1e3
    torch.cuda.synchronize()
    print("forward eval iters=5: launch thread %.2f ms per forward" % t_issue)
    # the forward as one replayed HIP graph (round 4: 8.24 ms against 7.74 ms eager over ~880 nodes; re-read with half the nodes)
    if os.environ.get("GRAPH", "1") != "0":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                net(pc1, pc2, pc1, pc2, iters=5)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = net(pc1, pc2, pc1, pc2, iters=5)
        ref = net(pc1, pc2, pc1, pc2, iters=5)
        g.replay()
        torch.cuda.synchronize()
        print("graph replay == eager:", all(torch.equal(a, b) for a, b in zip(outs, ref)))
        print("forward eval iters=5 as a replayed HIP graph: %.2f ms" % timed(g.replay, reps=10))
net.train()
crit = UnsupervisedFlowStep3DLoss(ChamferLoss(2), SmoothLoss(3., 1., {'k': 4, 'radius': 0.5, 'loss_norm': 1},
                                                              {'k': 8, 'radius': 1.0, 'loss_norm': 1}),
                                  weights=[0.75, 0.25], iters_w=[0.8, 0.2, 0.4, 0.6])
from ogc_amd.train_step import flow_train_step, make_optimizer
opt = make_optimizer(net.parameters(), lr=1e-3)
batch = (pcs, None, flows, None)

if B > 1:
    state = {}
    def step():
        state["p"] = flow_train_step(net, crit, opt, batch, 4, sync=False)
    print("train step iters=4 (B=%d): %.2f ms" % (B, timed(step)))
    print(state["p"].result())

# the correlation layer alone (config C3: 2048 points of each cloud at level 2, 64-d features, 16 neighbours)
from ogc_amd.utils.flowstep3d_util import FlowEmbedding
corr = net.local_corr_layer
g = torch.Generator().manual_seed(3)
p1 = ((torch.rand(B, 3, 2048, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0]).view(1, 3, 1)).to(dev)
p2 = (p1 + 0.1 * torch.randn(B, 3, 2048, generator=g).to(dev)).contiguous()
f1, f2 = torch.randn(B, 64, 2048, generator=g).to(dev), torch.randn(B, 64, 2048, generator=g).to(dev)
net.eval()
with torch.no_grad():
    ms = timed(lambda: corr(p1, p2, f1, f2), reps=20, warm=5)
flops = 2.0 * B * 2048 * 16 * (131 * 128 + 128 * 128 + 128 * 128)
print("correlation layer fwd (eval, B=%d, 2048 pts, k=16): %.3f ms  -> %.2f TFLOP/s of the 3 GEMMs" % (B, ms, flops / ms / 1e9))
