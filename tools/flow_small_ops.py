"""Which Python lines launch the small copy / fill / add / cat kernels of a FlowStep3D forward (C3: one 8192-point pair, iters = 5)?
One forward under torch.profiler with stacks; per aten op of interest, the ogc_amd source lines that call it most.  (development tool)"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.flownet_kitti import FlowStep3D
from ogc_amd.utils.synthetic import make_scene_batch
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(0)
N = 8192
net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).to("cuda").eval()
pcs = make_scene_batch(1, N, 10, seed=1, aug=False, device="cuda")[0]
pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
with torch.no_grad():
    for _ in range(3):
        net(pc1, pc2, pc1, pc2, iters=5)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        net(pc1, pc2, pc1, pc2, iters=5)
        torch.cuda.synchronize()
want = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else {"aten::copy_", "aten::zero_", "aten::fill_", "aten::add", "aten::add_", "aten::mul",
                                                              "aten::cat", "aten::sub", "aten::div", "aten::max", "aten::sigmoid", "aten::tanh",
                                                              "aten::where", "aten::sqrt", "aten::sum", "aten::rsub", "aten::clamp", "aten::exp",
                                                              "aten::bmm", "aten::matmul", "aten::mm", "aten::addmm", "aten::gt", "aten::le"}
by = {w: collections.Counter() for w in want}
for ev in prof.events():
    if ev.name in want:
        site = "<no python frame>"
        for fr in ev.stack:
            if "ogc_amd" in fr:
                site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-110:]
                break
        shape = str(ev.input_shapes[0]) if ev.input_shapes else ""
        by[ev.name][(site, shape)] += 1
for name, c in sorted(by.items(), key=lambda kv: -sum(kv[1].values())):
    if not c:
        continue
    print("== %s: %d calls in the forward" % (name, sum(c.values())))
    for (site, shape), n in c.most_common(12):
        print("   %3d  %-100s %s" % (n, site[-100:], shape))
