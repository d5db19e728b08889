"""ogc_ball_query on the bench's scenes with the four-lanes-per-centre kernel and with the general kernel alone (OGC_BQ_CELLS=0),
in one Python process: per view set, then again after a few training steps in the same process (where bench.py measures)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import ogc_amd
from ogc_amd.pointnet2.pointnet2 import ball_query
from ogc_amd.utils.synthetic import make_scene_batch
def t(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
pcs = make_scene_batch(4, 8192, 10, seed=1, aug=True, device="cuda")[0]
pc = torch.cat([pcs[:, v] for v in range(4)]).contiguous()
plain = make_scene_batch(16, 8192, 10, seed=1, aug=False, device="cuda")[0][:, 0].contiguous()
for name, x in (("views", pc), ("plain", plain), ("view0", pcs[:, 0].repeat(4, 1, 1).contiguous()), ("view1", pcs[:, 1].repeat(4, 1, 1).contiguous()), ("view2", pcs[:, 2].repeat(4, 1, 1).contiguous()), ("view3", pcs[:, 3].repeat(4, 1, 1).contiguous())):
    for mode in ("0", "1"):
        os.environ["OGC_BQ_CELLS"] = mode
        print(name, "cells" if mode == "1" else "general", "%.1f us" % t(lambda: ball_query(2.0, 64, x, x)))

# the same after a few training steps in this process (bench.py measures there)
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
net = MaskFormer3D(n_slot=10, n_point=8192, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                   transformer_input_pos_enc=False).cuda()
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3, weight_decay=0.0)
batch = make_scene_batch(4, 8192, 10, seed=1234, outdoor=True, aug=True, device="cuda")
pre = None
for _ in range(5):
    pre = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
torch.cuda.synchronize()
for mode in ("0", "1", "0", "1"):
    os.environ["OGC_BQ_CELLS"] = mode
    print("after training steps:", "cells" if mode == "1" else "general", "%.1f us" % t(lambda: ball_query(2.0, 64, pc, pc)))
import ogc_amd.utils.streams as st
print("side streams:", len(getattr(st, "_side", {})))
