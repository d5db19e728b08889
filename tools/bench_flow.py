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
net.train()
crit = UnsupervisedFlowStep3DLoss(ChamferLoss(2), SmoothLoss(3., 1., {'k': 4, 'radius': 0.5, 'loss_norm': 1},
                                                              {'k': 8, 'radius': 1.0, 'loss_norm': 1}),
                                  weights=[0.75, 0.25], iters_w=[0.8, 0.2, 0.4, 0.6])
opt = torch.optim.Adam(net.parameters(), lr=1e-3)


def step():
    opt.zero_grad(set_to_none=True)
    preds = net(pc1, pc2, pc1, pc2, iters=4)
    loss, _ = crit(pc1, pc2, preds)
    loss.backward()
    opt.step()


if B > 1:
    print("train step iters=4 (B=%d): %.2f ms" % (B, timed(step)))
