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


def compress_label_id(segm):
    """Object ids -> consecutive ids 0, 1, 2, ... in sorted order of the original ids. Reference: data_util.py:41-49."""
    import numpy as np
    return np.unique(segm, return_inverse=True)[1]


def segm_to_mask(segm, max_n_object=None):
    """(N,) labels -> (N, max_n_object) one-hot float32. Reference: data_util.py:52-61."""
    import numpy as np
    object_ids, inverse = np.unique(segm, return_inverse=True)
    if max_n_object is None:
        max_n_object = object_ids.shape[0]
    return np.eye(max_n_object, dtype=np.float32)[inverse]


def augment_transform(pcs, flows, aug_transform_args, n_view=2, rng=None):
    """`n_view` random similarity transforms of a frame pair: P' = s * R P + t, F' = s * R F (per-axis scale s, Euler
    'zyx' rotation in degrees, shift), optionally followed by a separate rigid motion of frame 2 (`aug_pc2`, used when
    training the flow network).  pcs, flows (2, N, 3) numpy -> (2 * n_view, N, 3) each, ordered
    [view0 frame1, view0 frame2, view1 frame1, ...].  Reference: data_util.py:140-195; random numbers are drawn in the
    same order (rotation, scale, shift[, rotation2, shift2] per view) from `rng` (default: numpy's global state), so a
    seeded run reproduces the reference's augmentations."""
    import numpy as np
    from scipy.spatial.transform import Rotation
    assert pcs.shape[0] == flows.shape[0] == 2, 'Inconsistent number of frames!'
    rng = np.random if rng is None else rng
    out_pcs, out_flows = [], []
    for _ in range(n_view):
        limit = np.array(aug_transform_args['degree_range'])
        rot = Rotation.from_euler('zyx', rng.uniform(-limit, limit), degrees=True).as_matrix()
        scale = rng.uniform(aug_transform_args['scale_low'], aug_transform_args['scale_high'], 3)
        limit = np.array(aug_transform_args['shift_range'])
        shift = rng.uniform(-limit, limit)
        frames = [scale * (p @ rot.T) + shift for p in pcs]
        moves = [scale * (f @ rot.T) for f in flows]
        if 'aug_pc2' in aug_transform_args:
            extra = aug_transform_args['aug_pc2']
            limit = np.array(extra['degree_range'])
            rot2 = Rotation.from_euler('zyx', rng.uniform(-limit, limit), degrees=True).as_matrix()
            limit = np.array(extra['shift_range'])
            shift2 = rng.uniform(-limit, limit)
            # frame 2 moves rigidly; frame 1 stays, so both flows change
            target_of_2 = frames[1] + moves[1]
            target_of_1 = frames[0] + moves[0]
            frames[1] = frames[1] @ rot2.T + shift2
            moves[1] = target_of_2 - frames[1]
            moves[0] = target_of_1 @ rot2.T + shift2 - frames[0]
        out_pcs.extend(frames)
        out_flows.extend(moves)
    return np.stack(out_pcs, 0), np.stack(out_flows, 0)
