"""Is the training step bound by the host's launch rate or by the GPU?  Times the issue loop against the total."""
import os, sys, time
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
pre = None
for _ in range(3):
    pre = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
marks = []
for _ in range(K):
    p = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch)
    pre = p.prefetched
    marks.append(time.perf_counter())
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("issue loop %.2f ms/step, total %.2f ms/step; per-step issue: %s" %
      (t_issue / K * 1e3, t_all / K * 1e3, " ".join("%.1f" % ((marks[i] - (marks[i - 1] if i else t0)) * 1e3) for i in range(K))))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    p = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch)
    pre = p.prefetched
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
