"""Which Python lines launch the small copy / fill / add kernels of a training step?  (development tool)
One steady-state step under torch.profiler with stacks; prints, per aten op of interest, the ogc_amd source lines that
call it most often."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, PrefetchedGeometry, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch
from torch.profiler import profile, ProfilerActivity

torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to("cuda")
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device="cuda")
pre = PrefetchedGeometry(net, crit, batch, True)
for i in range(4):
    pend = train_step(net, crit, opt, batch, 4000 + i, True, sync=False, prefetched=pre, next_batch=batch)
    pre = pend.prefetched
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    pend = train_step(net, crit, opt, batch, 4010, True, sync=False, prefetched=pre, next_batch=batch)
    torch.cuda.synchronize()
want = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else {"aten::copy_", "aten::zero_", "aten::fill_", "aten::add", "aten::add_", "aten::mul", "aten::cat"}
by = {w: collections.Counter() for w in want}
for ev in prof.events():
    if ev.name in want:
        site = "<autograd engine / no python frame>"
        for fr in ev.stack:
            if "ogc_amd" in fr or "bench.py" in fr:
                site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-110:]
                break
        shape = str(ev.input_shapes[0]) if ev.input_shapes else ""
        by[ev.name][(site, shape)] += 1
for name, c in by.items():
    print("== %s: %d calls in the step" % (name, sum(c.values())))
    for (site, shape), n in c.most_common(14):
        print("   %3d  %-100s %s" % (n, site[-100:], shape))
