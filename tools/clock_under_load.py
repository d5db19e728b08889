"""Shader clock with the GPU idle and while the C4 training step runs (development tool; needs tools/libclock_sampler.so:
hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libclock_sampler.so tools/clock_sampler.hip)."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, PrefetchedGeometry, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libclock_sampler.so"))
lib.clock_sample.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
side = torch.cuda.Stream()
ITERS = 40000


def sample(n):
    bufs = [torch.zeros(3, dtype=torch.int64, device="cuda") for _ in range(n)]
    return bufs


def launch(buf):
    lib.clock_sample(buf.data_ptr(), ITERS, side.cuda_stream)


def report(tag, bufs):
    torch.cuda.synchronize()
    vals = [b.tolist() for b in bufs]
    mhz = [c / w * 100.0 for c, w, _ in vals if w > 0]
    us = [w / 100.0 for c, w, _ in vals]
    print("%-28s clock %.0f..%.0f MHz (mean %.0f), sampler %.0f..%.0f us for %d dependent multiplies" %
          (tag, min(mhz), max(mhz), sum(mhz) / len(mhz), min(us), max(us), ITERS))


torch.zeros(1, device="cuda")
bufs = sample(10)
for b in bufs:
    launch(b); time.sleep(0.01)
report("idle GPU:", bufs)

torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, transformer_embed_dim=128).to("cuda")
crit = build_criterion(KITTI_LOSS)
opt = make_optimizer(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, aug=True, device="cuda")
pre, pend = PrefetchedGeometry(net, crit, batch, True), None
steps = 150
bufs = sample(steps // 5)
for i in range(steps):
    if i == 20:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    pend = train_step(net, crit, opt, batch, 4000 + i, True, sync=False, prefetched=pre, next_batch=batch)
    pre = pend.prefetched
    if i % 5 == 0:
        launch(bufs[i // 5])
torch.cuda.synchronize()
print("training: %.2f ms/step" % ((time.perf_counter() - t0) / (steps - 20) * 1e3))
report("under the training step:", bufs[6:])
