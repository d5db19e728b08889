"""ogc_knn_clamped of a cloud in itself, knn_cells_kernel + deferred pass against knn_grid_kernel alone (OGC_KNN_CELLS=0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: F401
from ogc_amd import pointnet2_cuda as nat
from ogc_amd.utils.synthetic import make_scene_batch


def t(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


pcs = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device="cuda")[0]
pc = torch.cat([pcs[:, v] for v in range(4)]).contiguous()
for k, r in ((32, 1.0), (16, 1.5), (32, 2.0), (32, -1.0)):
    d = torch.empty(16, 8192, k, device="cuda"); i = torch.empty(16, 8192, k, dtype=torch.int32, device="cuda")
    for rep in range(2):
        for mode in ("0", "1"):
            os.environ["OGC_KNN_CELLS"] = mode
            print("k=%d r=%.1f %s %.1f us" % (k, r, "cells+deferred" if mode == "1" else "general      ",
                                              t(lambda: nat.knn_clamped_wrapper(16, 8192, 8192, k, r, pc, pc, d, i))))
