"""List every host<->device synchronisation a training step makes (development tool)."""
import os, sys, warnings, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

dev = "cuda"
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to(dev)
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device=dev)
for _ in range(2):
    train_step(net, crit, opt, batch, 1000, True)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    pending = train_step(net, crit, opt, batch, 1000, True, sync=False)
torch.cuda.set_sync_debug_mode("default")
print(pending.result())
print("syncs in one step:", len(w))
for x in w:
    print(" ", x.filename.replace(os.getcwd() + "/", ""), x.lineno, str(x.message)[:80])
