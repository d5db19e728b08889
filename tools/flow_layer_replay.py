"""Replays single FlowStep3D layers (training mode) on the CPU oracle's operators with the inputs they saw in a GPU forward pass:
separates 'the layer computes something else' from 'its input differed'."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests/golden')
import detgen, driver_cases as dc
from oracle import oracle as orc
import ogc_amd.pointnet2.pointnet2 as api
from ogc_amd.models.flownet_sapien import FlowStep3D
cfg = dc.FLOW_CFG
ds = dc.FlowPairs(True)
pcs = torch.from_numpy(np.stack([ds[0][0], ds[1][0]]))
WATCH = ["local_corr_layer", "encoder_loc.sa1", "encoder_loc.sa2", "gru.convz", "flow_conv1", "flow_conv2", "flow_regressor.sa1", "h0_net.sa2", "h0_net.sa1"]
net = detgen.fill_module(FlowStep3D(**cfg["flownet"]), 32).cuda().train()
rec = []
for name, m in net.named_modules():
    if name in WATCH:
        def hook(mod, inp, out, name=name):
            rec.append((name, [x.detach().cpu() if torch.is_tensor(x) else x for x in inp],
                        [o.detach().cpu() for o in (out if isinstance(out, tuple) else (out,)) if torch.is_tensor(o)]))
        m.register_forward_hook(hook)
p = pcs.cuda()
with torch.no_grad():
    net(p[:, 0].contiguous(), p[:, 1].contiguous(), p[:, 0].contiguous(), p[:, 1].contiguous(), iters=2)
orc.build(); api._native = orc.Pointnet2CudaCPU()
cpu = detgen.fill_module(FlowStep3D(**cfg["flownet"]), 32).train()
mods = dict(cpu.named_modules())
for name, inp, out in rec:
    with torch.no_grad():
        o = mods[name](*inp)
    o = [x for x in (o if isinstance(o, tuple) else (o,)) if torch.is_tensor(x)]
    d = [float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)) if a.is_floating_point() else float((a != b).sum()) for a, b in zip(out, o)]
    print("%-22s in %s -> %s" % (name, [tuple(x.shape) if torch.is_tensor(x) else x for x in inp], " ".join("%.2e" % x for x in d)))
