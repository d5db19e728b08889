"""FlowStep3D in TRAINING mode (BatchNorm batch statistics), HIP operators against the same layers on the CPU oracle's operators:
per module call, in execution order, the relative L2 difference of the output — finds the first layer that disagrees."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests/golden')
import detgen, driver_cases as dc
from oracle import oracle as orc
import ogc_amd.pointnet2.pointnet2 as api
from ogc_amd.models.flownet_sapien import FlowStep3D
cfg = dc.FLOW_CFG
ds = dc.FlowPairs(True)
b = [ds[0], ds[1]]
pcs = torch.from_numpy(np.stack([x[0] for x in b]))


def run(dev, train):
    net = detgen.fill_module(FlowStep3D(**cfg["flownet"]), 32).to(dev)
    net.train() if train else net.eval()
    log = []

    def hook(name):
        def fn(mod, inp, out):
            t = out
            while isinstance(t, (tuple, list)):
                t = t[-1] if name.endswith(tuple("0123456789")) or True else t[0]
            if torch.is_tensor(t) and t.is_floating_point():
                log.append((name, t.detach().cpu().double()))
        return fn
    for name, m in net.named_modules():
        if name:
            m.register_forward_hook(hook(name))
    p = pcs.to(dev)
    pc1, pc2 = p[:, 0].contiguous(), p[:, 1].contiguous()
    with torch.no_grad():
        out = net(pc1, pc2, pc1, pc2, iters=2)
    return [o.cpu().double() for o in out], log


gpu_native = api._native
g, glog = run("cuda", True)
orc.build(); api._native = orc.Pointnet2CudaCPU()
c, clog = run("cpu", True)
api._native = gpu_native
print("calls", len(glog), len(clog))
from collections import defaultdict
def keyed(log):
    seen, out = defaultdict(int), {}
    for n, t in log:
        out[(n, seen[n])] = t
        seen[n] += 1
    return out
gk, ck = keyed(glog), keyed(clog)
order = [k for k in keyed(clog) if k in gk]
shown = 0
for k in order:
    a, b_ = gk[k], ck[k]
    if a.shape != b_.shape:
        print("SHAPE", k, tuple(a.shape), tuple(b_.shape)); continue
    d = float((a - b_).norm() / b_.norm().clamp_min(1e-30))
    if d > 2e-5 and shown < 14:
        print("%-44s call %d %s rel %.2e" % (k[0], k[1], tuple(a.shape), d)); shown += 1
for i in range(2):
    print("flow", i, "rel %.2e" % float((g[i] - c[i]).norm() / c[i].norm()))
