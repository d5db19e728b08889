"""What HBM rate do plain streaming kernels reach on this GPU?  (context for the roofline fractions: peak is 8 TB/s)
copy = read + write of the same size, fill = write only, sum = read only; sizes around the tensors of the C4 step."""
import torch

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x))
    print("copy  %5d MB -> %5d MB: %7.1f us  %5.2f TB/s (read + write)" % (mb, mb, dt * 1e6, 2 * n * 4 / dt / 1e12))
    dt = t(lambda: y.fill_(1.0))
    print("fill  %5d MB          : %7.1f us  %5.2f TB/s" % (mb, dt * 1e6, n * 4 / dt / 1e12))
    dt = t(lambda: x.sum())
    print("sum   %5d MB          : %7.1f us  %5.2f TB/s" % (mb, dt * 1e6, n * 4 / dt / 1e12))
    dt = t(lambda: torch.add(x, 1.0, out=y))
    print("add   %5d MB          : %7.1f us  %5.2f TB/s (read + write)" % (mb, dt * 1e6, 2 * n * 4 / dt / 1e12))
