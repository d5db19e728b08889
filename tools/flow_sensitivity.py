"""How much does FlowStep3D's second prediction move when the input coordinates move by one ulp?  (training mode, GPU; and the same
on the CPU oracle's operators)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests/golden')
import detgen, driver_cases as dc
from oracle import oracle as orc
import ogc_amd.pointnet2.pointnet2 as api
from ogc_amd.models.flownet_sapien import FlowStep3D
cfg = dc.FLOW_CFG
ds = dc.FlowPairs(True)
pcs = torch.from_numpy(np.stack([ds[0][0], ds[1][0]]))
g = torch.Generator().manual_seed(0)
def run(dev, x, train=True):
    net = detgen.fill_module(FlowStep3D(**cfg["flownet"]), 32).to(dev)
    net.train() if train else net.eval()
    p = x.to(dev)
    with torch.no_grad():
        return [o.cpu().double() for o in net(p[:, 0].contiguous(), p[:, 1].contiguous(), p[:, 0].contiguous(), p[:, 1].contiguous(), iters=2)]
for dev in ("cuda", "cpu"):
    if dev == "cpu":
        orc.build(); api._native = orc.Pointnet2CudaCPU()
    for train in (True, False):
        base = run(dev, pcs, train)
        for trial in range(3):
            noise = (torch.randint(0, 3, pcs.shape, generator=g).float() - 1.0)
            pert = torch.nextafter(pcs, pcs + noise)   # +-1 ulp or unchanged
            out = run(dev, pert, train)
            print(dev, "train" if train else "eval ", "trial", trial, " ".join("flow%d %.2e" % (i, float((out[i] - base[i]).norm() / base[i].norm())) for i in range(2)))
