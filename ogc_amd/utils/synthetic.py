"""Seeded synthetic scenes with the reference datasets' sample contract (no datasets travel with the repo).

A sample is the 4-tuple the reference loaders yield (datasets/dataset_kittisf.py:119-122):
``pcs (t, N, 3)``, ``segms (t, N)``, ``flows (t, N, 3)``, ``valids`` — here batched as (B, t, N, ...).
Frame 2 is frame 1 moved by per-object rigid motions (+ noise), re-ordered by a random permutation, so the
flows are exact rigid flows per object, which is what the OGC losses assume.  With ``aug=True`` two more views
are appended: the same two frames under a random similarity transform (the role of ``augment_transform``,
utils/data_util.py:140-195), giving t = 4 as in training with ``aug_transform_epoch`` reached.

Scales: outdoor clouds ``(rand - 0.5) * [60, 4, 80]`` m; object-scale clouds in the unit cube.
"""
import math

import torch


def _rot_y(angle):
    c, s = torch.cos(angle), torch.sin(angle)
    R = torch.zeros(*angle.shape, 3, 3)
    R[..., 0, 0], R[..., 0, 2], R[..., 1, 1], R[..., 2, 0], R[..., 2, 2] = c, s, 1.0, -s, c
    return R


def make_scene_batch(B, N, K, seed=1234, outdoor=True, aug=False, device="cpu", flows=None):
    """`flows` (B, 2, N, 3), when given, replaces the ground-truth flows of the two frames BEFORE the augmented views
    are derived — predicted flows are loaded first and then augmented, as datasets/dataset_kittisf.py:91-117 does."""
    g = torch.Generator().manual_seed(seed)
    scale = torch.tensor([60.0, 4.0, 80.0]) if outdoor else torch.ones(3)
    max_shift, noise = (0.5, 0.01) if outdoor else (0.05, 0.002)
    pc1 = (torch.rand(B, N, 3, generator=g) - 0.5) * scale
    centres = pc1[:, torch.randperm(N, generator=g)[:K]]                              # (B, K, 3)
    segm1 = torch.cdist(pc1, centres).argmin(-1)                                      # (B, N)
    ang = (torch.rand(B, K, generator=g) - 0.5) * 2 * math.radians(5.0)
    R = _rot_y(ang)                                                                   # (B, K, 3, 3)
    shift = (torch.rand(B, K, 3, generator=g) - 0.5) * 2 * max_shift
    Rn = R.gather(1, segm1[:, :, None, None].expand(-1, -1, 3, 3))
    tn = shift.gather(1, segm1[:, :, None].expand(-1, -1, 3))
    cn = centres.gather(1, segm1[:, :, None].expand(-1, -1, 3))
    moved = torch.einsum("bnij,bnj->bni", Rn, pc1 - cn) + cn + tn
    flow1 = moved - pc1
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    pc2 = (moved + torch.randn(B, N, 3, generator=g) * noise).gather(1, perm[:, :, None].expand(-1, -1, 3))
    segm2 = segm1.gather(1, perm)
    flow2 = (-flow1).gather(1, perm[:, :, None].expand(-1, -1, 3))                    # backward flow of frame 2

    if flows is not None:
        flow1, flow2 = flows[:, 0].to(pc1), flows[:, 1].to(pc1)
    pcs, segms, flows = [pc1, pc2], [segm1, segm2], [flow1, flow2]
    if aug:
        s = 0.95 + 0.1 * torch.rand(B, 1, 1, generator=g)
        Ra = _rot_y((torch.rand(B, generator=g) - 0.5) * 2 * math.pi)
        ta = (torch.rand(B, 1, 3, generator=g) - 0.5) * 2 * torch.tensor([1.0, 0.1, 1.0]) * (1.0 if outdoor else 0.05)
        for pc, fl, sg in [(pc1, flow1, segm1), (pc2, flow2, segm2)]:
            pcs.append(s * torch.einsum("bij,bnj->bni", Ra, pc) + ta)
            flows.append(s * torch.einsum("bij,bnj->bni", Ra, fl))
            segms.append(sg)
    pcs = torch.stack(pcs, 1).contiguous().to(device)
    flows = torch.stack(flows, 1).contiguous().to(device)
    segms = torch.stack(segms, 1).contiguous().to(device)
    valids = torch.ones_like(segms, dtype=torch.bool)
    return pcs, segms, flows, valids


def make_sequence(T, N, K, seed=1234, outdoor=True, device="cpu"):
    """One scene observed over T frames — the unit multi-frame voting works on (vote.py:95-131).
    Returns pc (T, N, 3), segm (T, N), flows (T-1, 2, N, 3): flows[t, 0] is the forward flow of frame t (to t+1),
    flows[t, 1] the backward flow of frame t+1 (to t); every step moves each object rigidly, adds noise and re-orders
    the points, as make_scene_batch does for a pair."""
    g = torch.Generator().manual_seed(seed)
    scale = torch.tensor([60.0, 4.0, 80.0]) if outdoor else torch.ones(3)
    max_shift, noise = (0.5, 0.01) if outdoor else (0.05, 0.002)
    pc = (torch.rand(N, 3, generator=g) - 0.5) * scale
    centres = pc[torch.randperm(N, generator=g)[:K]]
    segm = torch.cdist(pc, centres).argmin(-1)
    pcs, segms, flows = [pc], [segm], []
    for _ in range(T - 1):
        R = _rot_y((torch.rand(K, generator=g) - 0.5) * 2 * math.radians(5.0))
        shift = (torch.rand(K, 3, generator=g) - 0.5) * 2 * max_shift
        moved = torch.einsum("nij,nj->ni", R[segm], pc - centres[segm]) + centres[segm] + shift[segm]
        fwd = moved - pc
        perm = torch.randperm(N, generator=g)
        nxt = (moved + torch.randn(N, 3, generator=g) * noise)[perm]
        flows.append(torch.stack([fwd, (-fwd)[perm]]))
        centres = centres + shift
        pc, segm = nxt, segm[perm]
        pcs.append(pc)
        segms.append(segm)
    return torch.stack(pcs).to(device), torch.stack(segms).to(device), torch.stack(flows).to(device)
