import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd
from ogc_amd.pointnet2 import pointnet2 as P
from ogc_amd.models import _flownet as F
from ogc_amd.utils import flowstep3d_util as U
from ogc_amd.models.flownet_kitti import FlowStep3D
from ogc_amd.utils.synthetic import make_scene_batch
N = 8192
net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).to("cuda").eval()
pcs, _, flows, _ = make_scene_batch(1, N, 10, seed=1, aug=False, device="cuda")
pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
orig_chain = P.furthest_point_sample_chain
orig_fps = P.furthest_point_sample
def chain(xyz, npoint, parent_ties=None):
    out, ties = orig_chain(xyz, npoint, parent_ties)
    print("chain", tuple(xyz.shape), npoint, None if parent_ties is None else parent_ties.tolist(), "->", ties.tolist())
    return out, ties
def fps(xyz, npoint):
    print("fps", tuple(xyz.shape), npoint)
    return orig_fps(xyz, npoint)
F.furthest_point_sample_chain = chain
U.furthest_point_sample = fps
P.furthest_point_sample_chain = chain
with torch.no_grad():
    net(pc1, pc2, pc1, pc2, iters=5)
torch.cuda.synchronize()
