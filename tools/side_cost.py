"""What do the side-stream geometry plans cost the step?  (development tool)
A: the production loop (geometry of the next batch queued on side streams during every step);
B: the same steps re-using one plan (no FPS / kNN / ball query / transposed lists recomputed at all).
The difference is what the 'hidden' side work still takes from the main stream through CU and bandwidth sharing."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, PrefetchedGeometry, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

dev = "cuda"
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to(dev)
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device=dev)


def run(reuse, steps=20, warm=5):
    pre = PrefetchedGeometry(net, crit, batch, True)
    pend = None
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        pend = train_step(net, crit, opt, batch, 4000 + i, True, sync=False, prefetched=pre,
                          next_batch=None if reuse else batch)
        if not reuse:
            pre = pend.prefetched
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    pend.result()
    return ms


for rep in range(2):
    print("A production loop: %.2f ms/step    B plans re-used: %.2f ms/step" % (run(False), run(True)))

# which plan costs what: freeze one of the two (its result is handed out again, nothing is launched for it)
fixed = PrefetchedGeometry(net, crit, batch, True)
torch.cuda.synchronize()
orig_model, orig_loss = net.plan_geometry_async, crit.plan_geometry
net.plan_geometry_async = lambda pc, after=None: fixed.model
print("C network plan frozen (loss plan recomputed): %.2f ms/step" % run(False))
net.plan_geometry_async = orig_model
crit.plan_geometry = lambda pcs, aug=False: fixed.loss.get()
print("D loss plan frozen (network plan recomputed): %.2f ms/step" % run(False))
crit.plan_geometry = orig_loss
