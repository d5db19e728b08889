"""Host time of the pieces of a C4 step's optimizer phase, measured INSIDE running steps (no synchronisation: launches queue behind
the backward pass as they do in training):   PYTHONPATH=. python tools/opt_host_cost.py"""
import time

import torch

import ogc_amd  # noqa: F401
import ogc_amd.train_step as ts
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.utils.synthetic import make_scene_batch

acc = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return wrapper


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
    torch._foreach_norm = timed("_foreach_norm", torch._foreach_norm)
    torch.stack = timed("stack", torch.stack)
    torch.isnan = timed("isnan", torch.isnan)
    ts._step_with_flag = timed("_step_with_flag (Adam)", ts._step_with_flag)
    ts._fused_adam_step = timed("  _fused_adam_step", ts._fused_adam_step)
    torch._foreach_add_ = timed("    _foreach_add_", torch._foreach_add_)
    torch._foreach_sub_ = timed("    _foreach_sub_", torch._foreach_sub_)
    torch._fused_adam_ = timed("    _fused_adam_", torch._fused_adam_)
    opt.zero_grad = timed("zero_grad", opt.zero_grad)
    ts.PendingStep.__init__ = timed("PendingStep()", ts.PendingStep.__init__)
    N = 20
    t0 = time.perf_counter()
    for _ in range(N):
        pre = ts.train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
    host = (time.perf_counter() - t0) / N * 1e3
    torch.cuda.synchronize()
    print("host issue per step: %.2f ms" % host)
    for k, v in acc.items():
        print("%-28s %.3f ms/step" % (k, v / N * 1e3))
