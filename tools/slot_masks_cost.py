"""Mask read-out at the C4 shape (16 x 64 x 8192 features, 10 slots): op sequence vs the fused kernels, forward +
backward, milliseconds per call from HIP events.  Development tool."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: F401
from ogc_amd.fused import slot_masks

B, D, N, K = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 64, 8192, 10))]
feats = torch.randn(B, D, N, device="cuda", requires_grad=True)
slots = torch.randn(B, D, K, device="cuda", requires_grad=True)
w = torch.randn(B, N, K, device="cuda")


def ops():
    logits = torch.einsum('bdn,bdk->bnk', F.normalize(feats, dim=1), F.normalize(slots, dim=1)) / 0.05
    return logits.softmax(dim=-1)


def timeit(fn, bwd, reps=50):
    for _ in range(5):
        m = fn()
        if bwd:
            m.backward(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        m = fn()
        if bwd:
            feats.grad = slots.grad = None
            m.backward(w)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, fn in (("op sequence", ops), ("fused", lambda: slot_masks(feats, slots, 0.05))):
    f = timeit(fn, False)
    fb = timeit(fn, True)
    print("%-12s forward %.3f ms   forward+backward %.3f ms" % (name, f, fb))
