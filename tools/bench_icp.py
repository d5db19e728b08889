"""Object-aware ICP at the C4 refinement shape (B=4, N=8192, K=10; oa_icp.py:175 uses 20 iterations in round 1):
fused soft-NN kernel vs the reference's op sequence on torch (development tool)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
import ogc_amd.pointnet2.pointnet2 as api
from ogc_amd.oa_icp import object_aware_icp
from ogc_amd.utils.synthetic import make_scene_batch

B, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 8192, 10
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pcs, segms, flows, _ = make_scene_batch(B, N, K, seed=5, aug=False, device="cuda")
pc1, pc2, flow = pcs[:, 0].contiguous(), pcs[:, 1].contiguous(), flows[:, 0].contiguous()
eye = torch.eye(K, device="cuda")
mask1 = (4 * eye[segms[:, 0].long().cuda() % K] + torch.randn(B, N, K, device="cuda")).softmax(-1)
mask2 = (4 * eye[segms[:, 1].long().cuda() % K] + torch.randn(B, N, K, device="cuda")).softmax(-1)
noisy = flow + 0.05 * torch.randn_like(flow)


def run():
    with torch.no_grad():
        return object_aware_icp(pc1, pc2, noisy, mask1, mask2, icp_iter=iters, temperature=0.01)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30, out


ms_f, mem_f, out_f = timed(run)
native = api._native


class NoSoftNN:
    def __getattr__(self, name):
        if name == "soft_nn_target_wrapper":
            raise AttributeError(name)
        return getattr(native, name)


api._native = NoSoftNN()
ms_t, mem_t, out_t = timed(run)
api._native = native
epe = lambda f: (f - flow).norm(dim=-1).mean().item()
print("OA-ICP B=%d N=%d K=%d iters=%d: fused %.1f ms (peak %.2f GiB) | torch op sequence %.1f ms (peak %.2f GiB)" %
      (B, N, K, iters, ms_f, mem_f, ms_t, mem_t))
print("EPE noisy %.4f -> fused %.4f / torch %.4f; max |fused - torch| = %.3e" %
      (epe(noisy), epe(out_f), epe(out_t), (out_f - out_t).abs().max().item()))
