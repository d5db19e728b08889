"""FlowStep3D forward (eval, B = 1, 8192-point pair, 5 iterations) a few times — for kernel traces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: F401
from ogc_amd.models.flownet_kitti import FlowStep3D
from ogc_amd.utils.synthetic import make_scene_batch
torch.manual_seed(0)
net = FlowStep3D(npoint=8192, loc_flow_nn=16, loc_flow_rad=1.5).cuda().eval()
pcs, _, flows, _ = make_scene_batch(1, 8192, 10, seed=1, aug=False, device="cuda")
pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        net(pc1, pc2, pc1, pc2, iters=5)
torch.cuda.synchronize()
