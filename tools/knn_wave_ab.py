"""Plain k-NN (ogc_knn, k <= 32): knn_wave_kernel + deferred knn_grid_kernel against knn_grid_kernel alone (OGC_KNN_WAVE=0), and the
share of rows the wave kernel leaves (OGC_KNN_WAVE_ONLY=1), on a uniform slab cloud and on the bench's scene clouds.
    python tools/knn_wave_ab.py      (one process per mode)"""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import ogc_amd  # noqa: F401
    from ogc_amd import pointnet2_cuda as nat
    from ogc_amd.utils.synthetic import make_scene_batch
    g = torch.Generator().manual_seed(1234)
    clouds = {"uniform 16x8192": ((torch.rand(16, 8192, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda().contiguous(),
              "scenes 16x8192": torch.cat([make_scene_batch(4, 8192, 10, seed=1234, outdoor=True, aug=True, device="cuda")[0][:, v]
                                           for v in range(4)]).contiguous(),
              "cube 16x8192": (torch.rand(16, 8192, 3, generator=g) - 0.5).cuda().contiguous(),
              "uniform 1x8192": ((torch.rand(1, 8192, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda().contiguous(),
              "uniform 8x16384": ((torch.rand(8, 16384, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda().contiguous()}
    for name, pc in clouds.items():
        B, N, _ = pc.shape
        for k in (32, 16):
            d2 = torch.empty(B, N, k, device="cuda"); idx = torch.zeros(B, N, k, dtype=torch.int32, device="cuda")
            fn = lambda: nat.knn_wrapper(B, N, N, k, pc, pc, d2, idx)
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record()
            for _ in range(20):
                fn()
            b.record(); torch.cuda.synchronize()
            left = float((idx[:, :, 0] == -1).float().mean())
            print("%-18s k=%-3d %.3f ms   rows left %.1f %%" % (name, k, a.elapsed_time(b) / 20, 100 * left), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for env in ({"OGC_KNN_WAVE": "0"}, {"OGC_KNN_WAVE": "1"}, {"OGC_KNN_WAVE": "1", "OGC_KNN_WAVE_ONLY": "1"}):
            print("----", env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), check=False)
