"""CPU time (not wall time: the launch thread blocks when the GPU's queue is full) a C4 step costs the process and its main thread:
    PYTHONPATH=. python tools/host_cpu_time.py"""
import time

import torch

import ogc_amd  # noqa: F401
import ogc_amd.train_step as ts
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.utils.synthetic import make_scene_batch

if __name__ == "__main__":
    torch.manual_seed(10)
    net = MaskFormer3D(n_slot=10, n_point=8192, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                       transformer_input_pos_enc=False).cuda()
    crit = ts.build_criterion(ts.KITTI_LOSS)
    opt = ts.make_optimizer(net.parameters(), lr=1e-3)
    batch = make_scene_batch(4, 8192, 10, seed=1234, outdoor=True, aug=True, device="cuda")
    pre = None
    for _ in range(5):
        pre = ts.train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
    torch.cuda.synchronize()
    N = 30
    w0, p0, t0 = time.perf_counter(), time.process_time(), time.thread_time()
    for _ in range(N):
        pre = ts.train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
    t1, p1 = time.thread_time(), time.process_time()
    torch.cuda.synchronize()
    w1 = time.perf_counter()
    print("per step: wall %.2f ms | main thread CPU %.2f ms | whole process CPU %.2f ms (main + autograd + runtime threads)"
          % ((w1 - w0) / N * 1e3, (t1 - t0) / N * 1e3, (p1 - p0) / N * 1e3))
