"""Which Python lines launch the small aten kernels of a FlowStep3D forward (C3: one 8192-point pair, iters = 5)?
One forward under a TorchDispatchMode; per aten op, the ogc_amd source lines that call it most.  (development tool)"""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.flownet_kitti import FlowStep3D
from ogc_amd.utils.synthetic import make_scene_batch
from torch.utils._python_dispatch import TorchDispatchMode

torch.manual_seed(0)
N = 8192
net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).to("cuda").eval()
pcs = make_scene_batch(1, N, 10, seed=1, aug=False, device="cuda")[0]
pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
SKIP = {"view", "permute", "transpose", "expand", "slice", "select", "detach", "unsqueeze", "squeeze", "_unsafe_view", "t", "alias",
        "as_strided", "reshape", "empty", "empty_like", "empty_strided", "unbind", "split", "narrow", "_reshape_alias", "new_empty"}
by = collections.defaultdict(collections.Counter)


class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in SKIP:
            site = "<none>"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "ogc_amd" in fr.filename and "tools/" not in fr.filename:
                    site = "%s:%d" % (fr.filename.split("ogc_amd/", 1)[-1], fr.lineno)
                    break
            shape = ""
            for a in args:
                if isinstance(a, torch.Tensor):
                    shape = str(list(a.shape))
                    break
            by[name][(site, shape)] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    for _ in range(3):
        net(pc1, pc2, pc1, pc2, iters=5)
    torch.cuda.synchronize()
    with Watch():
        net(pc1, pc2, pc1, pc2, iters=5)
    torch.cuda.synchronize()
total = 0
for name, c in sorted(by.items(), key=lambda kv: -sum(kv[1].values())):
    total += sum(c.values())
    print("== %s: %d calls in the forward" % (name, sum(c.values())))
    for (site, shape), n in c.most_common(14):
        print("   %3d  %-60s %s" % (n, site, shape))
print("total aten ops that launch or allocate:", total)
