"""ogc_conv1x1_gemm_any (csrc/gemm_chunk.hip) and ogc_conv1x1_wgrad against the vendor library on the products of a C4 step that
went to it (tools/library_gemms.py): max relative error against a float64 product, and microseconds per call (20 calls
between two events, idle GPU)."""
import torch

import ogc_amd  # noqa: F401
from ogc_amd import pointnet2_cuda as nat

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def clock(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


print("forward / input-gradient products  OUT[b] = A . IN[b]   (B = 16)")
for (M, K, hw, tr, what) in [(128, 384, 1024, 0, "FP3 conv1 fwd"), (128, 128, 1024, 0, "FP3 conv2 fwd"), (64, 224, 2048, 0, "FP2 conv1 fwd"),
                             (64, 64, 2048, 0, "FP2 conv2 fwd"), (384, 128, 1024, 1, "FP3 conv1 dgrad"), (128, 128, 1024, 1, "FP3 conv2 dgrad"),
                             (224, 64, 2048, 1, "FP2 conv1 dgrad"), (64, 64, 2048, 1, "FP2 conv2 dgrad"), (67, 64, 8192, 1, "FP1 conv1 dgrad"),
                             (64, 64, 8192, 1, "FP1 conv2 dgrad"), (128, 256, 32768, 1, "SA3 conv3 dgrad"), (96, 64, 2048, 1, "SA2 point-wise dgrad"),
                             (128, 128, 1024, 1, "SA3 point-wise dgrad"), (32, 3, 8192, 0, "SA1 point-wise fwd"), (64, 96, 2048, 0, "SA2 point-wise fwd"),
                             (128, 64, 65536, 1, "SA2 conv3 dgrad (64 <- 128)"), (64, 64, 65536, 1, "SA2 conv2 dgrad"), (256, 128, 32768, 0, "SA3 conv3 fwd")]:
    B = 16
    w = torch.randn((K, M) if tr else (M, K), device=dev) / K ** 0.5
    x = torch.randn(B, K, hw, device=dev)
    out = torch.empty(B, M, hw, device=dev)
    A = w.t() if tr else w
    nat.conv1x1_gemm_any_wrapper(B, M, K, hw, tr, w, x, out)
    ref = torch.matmul(A.double(), x[:2].double())
    err = float((out[:2].double() - ref).abs().max() / ref.abs().max())
    lib = torch.empty_like(out)
    Ab = A.unsqueeze(0).expand(B, -1, -1)
    t_lib = clock(lambda: torch.bmm(Ab, x, out=lib))
    t_own = clock(lambda: nat.conv1x1_gemm_any_wrapper(B, M, K, hw, tr, w, x, out))
    err_lib = float((lib[:2].double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * B * M * K * hw
    print("%-28s M=%-4d K=%-4d hw=%-6d  own %8.1f us (%5.1f TF, err %.1e)   library %8.1f us (%5.1f TF, err %.1e)"
          % (what, M, K, hw, t_own, fl / t_own / 1e6, err, t_lib, fl / t_lib / 1e6, err_lib))

print("weight gradients  dW = sum_b DY[b] . X[b]^T   (B = 16)")
for (co, ci, hw, what) in [(128, 384, 1024, "FP3 conv1"), (128, 128, 1024, "FP3 conv2"), (64, 224, 2048, "FP2 conv1"), (64, 64, 2048, "FP2 conv2"),
                           (64, 67, 8192, "FP1 conv1"), (64, 64, 8192, "FP1 conv2"), (32, 3, 8192, "SA1 point-wise"), (64, 96, 2048, "SA2 point-wise"),
                           (128, 128, 1024, "SA3 point-wise")]:
    B = 16
    x = torch.randn(B, ci, hw, device=dev)
    dy = torch.randn(B, co, hw, device=dev)
    dw = torch.zeros(co, ci, device=dev)
    nat.conv1x1_wgrad_wrapper(B, ci, co, hw, x, dy, dw)
    ref = torch.einsum("bop,bip->oi", dy.double(), x.double())
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    t_own = clock(lambda: nat.conv1x1_wgrad_wrapper(B, ci, co, hw, x, dy, dw))
    t_lib = clock(lambda: torch.bmm(dy, x.transpose(1, 2)).sum(0))
    print("%-28s cout=%-4d cin=%-4d hw=%-6d  own %8.1f us (err %.1e)   library bmm + sum %8.1f us" % (what, co, ci, hw, t_own, err, t_lib))
