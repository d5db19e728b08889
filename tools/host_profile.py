"""Where does the launch thread spend its ~11.5 ms per C4 step?  cProfile over a few steps (sync-free, so host time = issue time)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.getcwd())
import ogc_amd
from ogc_amd.models.segnet_kitti import MaskFormer3D
from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
from ogc_amd.utils.synthetic import make_scene_batch
torch.manual_seed(10)
net = MaskFormer3D(n_slot=10, n_point=8192, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128, transformer_input_pos_enc=False).cuda()
crit = build_criterion(KITTI_LOSS); opt = make_optimizer(net.parameters(), lr=1e-3)
batch = make_scene_batch(4, 8192, 10, seed=1234, outdoor=True, aug=True, device="cuda")
pre = None
for _ in range(5):
    pre = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
N = 10
for _ in range(N):
    pre = train_step(net, crit, opt, batch, 1000, True, sync=False, prefetched=pre, next_batch=batch).prefetched
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(sys.argv[1] if len(sys.argv) > 1 else "cumulative")
import io
buf = io.StringIO(); st.stream = buf; st.print_stats(70)
out = buf.getvalue()
for line in out.splitlines():
    if "/root" in line or "ogc_amd" in line or "{" in line or "ncalls" in line or "function calls" in line:
        print(line[:190])
