"""Thin callers of the operators used by data preparation / evaluation (reference: utils/data_util.py:8-38)."""
import torch

from ..pointnet2.pointnet2 import furthest_point_sample, three_interpolate, three_nn


def fps_downsample(pc, n_sample_point=1024, device="cuda"):
    """numpy (N, 3) -> numpy (n_sample_point,) FPS indices. Reference: data_util.py:8-19."""
    pc_t = torch.from_numpy(pc).float().unsqueeze(0).to(device).contiguous()
    return furthest_point_sample(pc_t, n_sample_point).cpu().numpy()[0]


def upsample_feat(pc, pc_fps, feat_fps):
    """3-NN inverse-distance upsampling of per-point features: pc (B, N, 3), pc_fps (B, N', 3),
    feat_fps (B, N', C) -> (B, N, C). Reference: data_util.py:22-38."""
    dist, nn_idx = three_nn(pc.contiguous(), pc_fps.contiguous())
    dist_recip = 1.0 / (dist + 1e-8)
    weight = dist_recip / dist_recip.sum(dim=2, keepdim=True)
    feat = three_interpolate(feat_fps.transpose(1, 2).contiguous(), nn_idx, weight)
    return feat.transpose(1, 2)
