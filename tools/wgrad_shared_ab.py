"""Weight gradient of wide 1x1 convolutions: the 128 x 128 LDS-shared workgroup tile (conv1x1_wgrad_shared_kernel) against the
64 x 64 register tiles (OGC_WGRAD_SHARED=0), plain / with the previous layer's norm folded in / pooled form, at the C4 shapes;
error against a float64 product.   python tools/wgrad_shared_ab.py        (spawns one process per mode)"""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(16, 128, 128, 32768), (16, 128, 256, 32768), (16, 131, 128, 32768), (16, 256, 128, 32768), (16, 384, 128, 1024),
          (4, 128, 256, 65536), (16, 160, 192, 4096)]


def child():
    import ogc_amd  # noqa: F401
    from ogc_amd import pointnet2_cuda as nat

    def t(fn, n=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    for B, cin, cout, hw in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(cin + cout)
        x = torch.randn(B, cin, hw, device="cuda", generator=g)
        dy = torch.randn(B, cout, hw, device="cuda", generator=g)
        pa = torch.rand(B, cin, device="cuda", generator=g) + 0.5
        pb = torch.randn(B, cin, device="cuda", generator=g) * 0.1
        dw = torch.empty(cout, cin, device="cuda")
        ms = t(lambda: nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, dy, dw))
        ref = torch.einsum("bmp,bkp->mk", dy[:2].double(), x[:2].double())
        nat.conv1x1_wgrad_wrapper(2, cin, cout, hw, x[:2].contiguous(), dy[:2].contiguous(), dw)
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        ms2 = t(lambda: nat.conv1x1_wgrad_affine_wrapper(B, cin, cout, hw, 1, x, pa, pb, dy, dw))
        xr = torch.relu(x[:2] * pa[:2, :, None] + pb[:2, :, None]).double()
        nat.conv1x1_wgrad_affine_wrapper(2, cin, cout, hw, 1, x[:2].contiguous(), pa[:2].contiguous(), pb[:2].contiguous(), dy[:2].contiguous(), dw)
        ref2 = torch.einsum("bmp,bkp->mk", dy[:2].double(), xr)
        err2 = float((dw.double() - ref2).abs().max() / ref2.abs().max())
        tf = 2.0 * B * cin * cout * hw / 1e9
        print("%4d -> %-4d hw=%-6d B=%-3d plain %.3f ms (%5.1f TF, err %.1e)   affine+relu %.3f ms (%5.1f TF, err %.1e)" %
              (cin, cout, hw, B, ms, tf / ms, err, ms2, tf / ms2, err2), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for mode in ("0", "1"):
            print("---- OGC_WGRAD_SHARED=%s" % mode, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, OGC_WGRAD_SHARED=mode), check=False)
