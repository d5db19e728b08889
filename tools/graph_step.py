"""Experiment: the whole C4 training step (forward, OGC loss, backward, fused Adam) captured in ONE HIP graph.
How much of the 12.5 ms step is the launch thread?  Prints eager ms/step and graph-replay ms/step."""
import sys
import time
import torch

import ogc_amd  # noqa: F401
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, train_step
from ogc_amd.utils.synthetic import make_scene_batch

npoint = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=npoint, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                   transformer_input_pos_enc=False).to(dev)
crit = build_criterion(KITTI_LOSS)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
batch = make_scene_batch(4, npoint, 10, seed=1234, outdoor=True, aug=True, device=dev)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5):
        train_step(net, crit, opt, batch, 1000, True, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        train_step(net, crit, opt, batch, 1000, True, sync=False)
    torch.cuda.synchronize()
    print("eager (no prefetch): %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
torch.cuda.current_stream().wait_stream(s)

g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
from ogc_amd.utils import streams as _streams
with torch.cuda.graph(g, stream=s):
    pending = train_step(net, crit, opt, batch, 1000, True, sync=False)
    for key, st in _streams._side.items():  # join every side stream the step forked
        try:
            s.wait_stream(st)
        except Exception as e:  # noqa: BLE001
            print("join", key, "failed:", e)
torch.cuda.synchronize()
print("captured")
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph replay: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
