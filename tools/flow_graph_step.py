"""FlowStep3D training step (C3 training shape) eagerly and as ONE replayed HIP graph: ms per step, launch-thread ms per step."""
import os, sys, time, gc
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa
from ogc_amd.losses.flow_loss_unsup import ChamferLoss, SmoothLoss, UnsupervisedFlowStep3DLoss
from ogc_amd.models.flownet_kitti import FlowStep3D
from ogc_amd.train_step import flow_train_step, make_optimizer, prepare_adam_kernel
from ogc_amd.utils.synthetic import make_scene_batch

dev = torch.device("cuda", 0)
B, N, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 8192, 4
torch.manual_seed(10)
net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).to(dev)
crit = UnsupervisedFlowStep3DLoss(ChamferLoss(2), SmoothLoss(3., 1., {'k': 4, 'radius': 0.5, 'loss_norm': 1}, {'k': 8, 'radius': 1.0, 'loss_norm': 1}),
                                  weights=[0.75, 0.25], iters_w=[0.8, 0.2, 0.4, 0.6])
opt = make_optimizer(net.parameters(), lr=1e-3, capturable=True)
pcs, _, flows, _ = make_scene_batch(B, N, 10, seed=1, aug=False, device=dev)
batch = (pcs, None, flows, None)


def timed(fn, n=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_issue = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_issue


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        p = flow_train_step(net, crit, opt, batch, iters, sync=False)
    print("eager : %.2f ms/step, launch thread %.2f ms/step" % timed(lambda: flow_train_step(net, crit, opt, batch, iters, sync=False)))
    print("   losses", p.result()[0]["sum"])
    prepare_adam_kernel(opt)
    torch.cuda.synchronize()
    opt.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    gc.disable()
    try:
        with torch.cuda.graph(g, stream=s):
            pend = flow_train_step(net, crit, opt, batch, iters, sync=False)
    finally:
        gc.enable()
    for _ in range(2):
        g.replay()
    print("graph : %.2f ms/step, launch thread %.2f ms/step" % timed(g.replay))
    print("   losses", pend.refresh().result()[0]["sum"])
