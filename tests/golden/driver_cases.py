"""Configurations and deterministic in-memory data sets shared by the driver-fixture generator (make_driver_golden.py, which
feeds them to the REFERENCE's trainers) and by the tests that replay the same runs with this repo's trainers.
Everything is integer-hash data (detgen); nothing here comes from the reference."""
import os

import numpy as np
import torch

import detgen

N_SEG, K_SEG = 128, 4
SEG_CFG = {
    "dataset": "sapien", "random_seed": 10, "aug_transform_epoch": 1, "ignore_npoint_thresh": 0,
    "epochs": 3, "batch_size": 2, "lr": 1.0e-3, "lr_decay": 0.5, "lr_clip": 2.0e-4, "bn_momentum": 0.9, "bn_decay": 0.5,
    "weight_decay": 1.0e-4, "decay_step": 4,
    "segnet": {"n_slot": K_SEG, "n_point": N_SEG, "use_xyz": True, "n_transformer_layer": 1, "transformer_embed_dim": 32,
               "transformer_input_pos_enc": False},
    # loss terms switch on at it * b >= start_step: smooth from the second iteration, invariance from the fourth
    "loss": {"weights": [10.0, 0.1, 0.1], "start_steps": [0, 2, 6], "dynamic_loss_params": {"loss_norm": 2},
             "smooth_loss_params": {"w_knn": 3.0, "w_ball_q": 1.0, "knn_loss_params": {"k": 4, "radius": 0.1, "loss_norm": 1},
                                    "ball_q_loss_params": {"k": 8, "radius": 0.2, "loss_norm": 1}},
             "invariance_loss_params": {"loss_norm": 2}},
}

# The Waymo trainer (train_seg_waymo.py:20-242: its own Trainer, every other view kept — `[:, ::2]`, :59 — and the one-frame loss
# of :244-334) with segnet_kitti, as its main() builds it (:357-366).  512 points: the third set-abstraction level groups 64
# neighbours out of 64 points.  Two epochs: without, then with the augmented twin of the frame.
N_WAYMO, K_WAYMO = 512, 4
WAYMO_CFG = {
    "dataset": "waymo", "random_seed": 10, "aug_transform_epoch": 1, "ignore_npoint_thresh": 0,
    # (a small learning rate, as for the flow trainer: at 1e-3 the reference's own run moves by 1e-3 .. 1e-2 in its loss terms under
    # one-ulp changes of the coordinates — Adam turns rounding noise into steps of size lr and the IoU matching of near-uniform
    # masks jumps on them)
    "epochs": 2, "batch_size": 2, "lr": 1.0e-5, "lr_decay": 0.5, "lr_clip": 2.0e-6, "bn_momentum": 0.9, "bn_decay": 0.5,
    "weight_decay": 1.0e-4, "decay_step": 4,
    "segnet": {"n_slot": K_WAYMO, "n_point": N_WAYMO, "use_xyz": True, "n_transformer_layer": 1, "transformer_embed_dim": 32,
               "transformer_input_pos_enc": False},
    "loss": {"weights": [10.0, 0.1, 0.1], "start_steps": [0, 2, 6], "dynamic_loss_params": {"loss_norm": 2},
             "smooth_loss_params": {"w_knn": 3.0, "w_ball_q": 1.0, "knn_loss_params": {"k": 4, "radius": 0.1, "loss_norm": 1},
                                    "ball_q_loss_params": {"k": 8, "radius": 0.2, "loss_norm": 1}},
             "invariance_loss_params": {"loss_norm": 2}},
}

N_FLOW = 256
# (a small learning rate: Adam turns the rounding noise of this net's analytically zero gradients into steps of size lr, and
# its warped-cloud neighbour searches jump on such steps — at lr = 1e-3 two CPUs differ by 1 % in the losses after ONE step)
FLOW_CFG = {
    "dataset": "sapien", "random_seed": 10, "model_iters": 2, "epochs": 2, "batch_size": 2, "lr": 1.0e-5, "lr_decay": 0.5,
    "lr_clip": 1.0e-7, "bn_momentum": 0.9, "bn_decay": 0.5, "weight_decay": 0.0, "decay_step": 4,
    "flownet": {"npoint": N_FLOW, "use_instance_norm": False, "loc_flow_nn": 8, "loc_flow_rad": 0.1, "k_decay_fact": 1.0},
    "loss": {"weights": [0.75, 0.25], "iters_w": [0.5, 0.3], "chamfer_loss_params": {"loss_norm": 2},
             "smooth_loss_params": {"w_knn": 3.0, "w_ball_q": 1.0, "knn_loss_params": {"k": 4, "radius": 0.05, "loss_norm": 1},
                                    "ball_q_loss_params": {"k": 8, "radius": 0.1, "loss_norm": 1}}},
}

N_ICP, K_ICP = 512, 5
ICP_CFG = {
    "dataset": "kittisf", "save_path": None, "data": {"root": None, "decentralize": True},
    "segnet": {"n_slot": K_ICP, "n_point": N_ICP, "use_xyz": True, "n_transformer_layer": 1, "transformer_embed_dim": 32,
               "transformer_input_pos_enc": False},
}


def _similarity(seed):
    """A fixed similarity transform (rotation about y, scale, shift) for the augmented views."""
    a = float(detgen.uniform((1,), seed, -1.0, 1.0)[0])
    s = 1.0 + 0.05 * float(detgen.uniform((1,), seed + 1, -1.0, 1.0)[0])
    t = detgen.uniform((3,), seed + 2, -0.05, 0.05).astype(np.float64)
    c, sn = np.cos(a), np.sin(a)
    R = np.array([[c, 0.0, sn], [0.0, 1.0, 0.0], [-sn, 0.0, c]])
    return s, R, t


class SegScenes(torch.utils.data.Dataset):
    """Six (train) or two (val) scenes with the sample contract of the reference's data sets (datasets/dataset_sapien.py:
    item = pcs (t, N, 3) f32, segms (t, N) i32, flows (t, N, 3) f32, valids (t, N) f32; t = 2, or 4 with `aug_transform`:
    two transformed copies of the frame pair, datasets/dataset_kittisf.py:113-117).  The SECOND time scene 5 is drawn (the
    second epoch) one of its flow vectors is NaN: that step's gradients are NaN and both trainers must leave the weights and
    the optimiser state alone (train_seg.py:81-83)."""

    def __init__(self, train=True, npoint=N_SEG, nslot=K_SEG, seed=0, ulp=0):
        self.n = 6 if train else 2
        self.seed = (700 if train else 900) + seed
        self.npoint, self.nslot, self.ulp = npoint, nslot, ulp   # ulp: coordinates moved by one unit in the last place (see FlowPairs)
        self.aug_transform = False
        self.calls = [0] * self.n
        self.poison = train

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.calls[i] += 1
        pc, flow, mask = detgen.rigid_scene(1, self.npoint, self.nslot, self.seed + 10 * i, scale=(1.0, 1.0, 1.0))
        if self.ulp:
            step = np.sign(detgen.uniform(pc[0].shape, self.seed + 10 * i + 7 + self.ulp)).astype(np.float32)
            pc[0] = np.nextafter(pc[0], pc[0] + step)
        pc1, f1 = pc[0].astype(np.float64), flow[0].astype(np.float64)
        perm = np.argsort(detgen.uniform((self.npoint,), self.seed + 10 * i + 5))
        pc2 = (pc1 + f1)[perm]
        f2 = -f1[perm]
        segm = mask[0].argmax(-1)
        pcs, flows, segms = np.stack([pc1, pc2]), np.stack([f1, f2]), np.stack([segm, segm[perm]])
        if self.poison and i == 5 and self.calls[i] == 2:
            flows[0, 7, 1] = np.nan
        if self.aug_transform:
            views_p, views_f = [], []
            for v in range(2):
                s, R, t = _similarity(self.seed + 10 * i + 6 + 3 * v)
                views_p.append(s * pcs @ R.T + t)
                views_f.append(s * flows @ R.T)
            pcs, flows = np.concatenate(views_p), np.concatenate(views_f)
            segms = np.concatenate([segms, segms])
        valids = np.ones_like(segms, dtype=np.float32)
        return pcs.astype(np.float32), segms.astype(np.int32), flows.astype(np.float32), valids


class FlowPairs(torch.utils.data.Dataset):
    """Four (train) or two (val) frame pairs for the FlowStep3D trainers (train_flow.py:62-76 reads pcs[:, 0], pcs[:, 1] and
    flows[:, 0]).  `seed`: the generator tries seeds until the reference's run is stable under one-ulp changes of the
    coordinates (a neighbour of a warped point sitting exactly on a decision boundary moves the second prediction by 4e-3 on
    every implementation, the reference's included) and stores the seed it took in the fixture; `ulp` applies such a change."""

    def __init__(self, train=True, seed=1100, ulp=0):
        self.n = 4 if train else 2
        self.seed = seed if train else seed + 200
        self.ulp = ulp

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        pc, flow, mask = detgen.rigid_scene(1, N_FLOW, 4, self.seed + 10 * i, scale=(1.0, 1.0, 1.0))
        perm = np.argsort(detgen.uniform((N_FLOW,), self.seed + 10 * i + 5))
        pc2 = (pc[0] + flow[0])[perm]
        segm = mask[0].argmax(-1).astype(np.int32)
        if self.ulp:
            step = np.sign(detgen.uniform(pc[0].shape, self.seed + 10 * i + 7 + self.ulp)).astype(np.float32)
            pc[0] = np.nextafter(pc[0], pc[0] + step)
        return (np.stack([pc[0], pc2]).astype(np.float32), np.stack([segm, segm[perm]]),
                np.stack([flow[0], -flow[0][perm]]).astype(np.float32), np.ones((2, N_FLOW), np.float32))


# ---- on-disk data roots in the reference's layouts (written by this repo's code, read by the reference's data sets) -----------
KITTI_IDS = ["000000", "000001", "000002"]
OGCDR_IDS = ["00000000", "00000001"]


def kitti_scene(i):
    pc, flow, mask = detgen.rigid_scene(1, N_ICP, K_ICP, 1500 + 10 * i, scale=(30.0, 3.0, 40.0), max_shift=0.4, noise=0.01)
    perm = np.argsort(detgen.uniform((N_ICP,), 1505 + 10 * i))
    pc1, f1 = pc[0], flow[0]
    pc2, f2 = (pc1 + f1)[perm], -f1[perm]
    segm = mask[0].argmax(-1).astype(np.int64)
    return pc1, pc2, segm, segm[perm], f1, f2


def write_kitti_root(root):
    """<root>/data/<id>/{pc1,pc2,segm1,segm2,flow1,flow2}.npy + <root>/train.txt (datasets/dataset_kittisf.py:36-79)."""
    for i, sid in enumerate(KITTI_IDS):
        d = os.path.join(root, "data", sid)
        os.makedirs(d, exist_ok=True)
        pc1, pc2, s1, s2, f1, f2 = kitti_scene(i)
        for name, arr in (("pc1", pc1), ("pc2", pc2), ("segm1", s1), ("segm2", s2), ("flow1", f1), ("flow2", f2)):
            np.save(os.path.join(d, name + ".npy"), arr)
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.write("\n".join(KITTI_IDS) + "\n")


def kitti_predicted_flows():
    """(n_scene * 2, N, 3): for every scene the flows of its two frames, adjacent (the order oa_icp.py's loader yields)."""
    return np.stack([detgen.uniform((N_ICP, 3), 1700 + 2 * i + v, -0.3, 0.3) for i in range(len(KITTI_IDS)) for v in range(2)])


def write_kitti_input_flows(root):
    """Round-1 input of the refinement: <root>/flow_preds/flowstep3d/<id>/flow{1,2}.npy = the true flow plus noise."""
    for i, sid in enumerate(KITTI_IDS):
        d = os.path.join(root, "flow_preds", "flowstep3d", sid)
        os.makedirs(d, exist_ok=True)
        _, _, _, _, f1, f2 = kitti_scene(i)
        np.save(os.path.join(d, "flow1.npy"), (f1 + detgen.uniform((N_ICP, 3), 1800 + 2 * i, -0.05, 0.05)).astype(np.float32))
        np.save(os.path.join(d, "flow2.npy"), (f2 + detgen.uniform((N_ICP, 3), 1801 + 2 * i, -0.05, 0.05)).astype(np.float32))


def write_ogcdr_root(root, n_frame=4, n=64):
    """<root>/data/<id>/{pc,segm,pose}_%02d.npy + <root>/data/train.lst (datasets/dataset_ogcdr.py:54-57,82-93)."""
    os.makedirs(os.path.join(root, "data"), exist_ok=True)
    for i, sid in enumerate(OGCDR_IDS):
        d = os.path.join(root, "data", sid)
        os.makedirs(d, exist_ok=True)
        for v in range(n_frame):
            np.save(os.path.join(d, "pc_%02d.npy" % v), detgen.uniform((n, 3), 1900 + 10 * i + v, -0.5, 0.5))
            np.save(os.path.join(d, "segm_%02d.npy" % v), (detgen.uniform((n,), 1950 + 10 * i + v, 0.0, 2.99)).astype(np.int64))
            pose = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
            pose[:, :3, 3] = detgen.uniform((2, 3), 1980 + 10 * i + v, -0.1, 0.1)
            np.save(os.path.join(d, "pose_%02d.npy" % v), pose)
    with open(os.path.join(root, "data", "train.lst"), "w") as f:
        f.write("\n".join(OGCDR_IDS) + "\n")


def ogcdr_predicted_flows(n=64):
    """(n_scene * 6, n, 3): six ordered frame pairs per scene, adjacent."""
    return np.stack([detgen.uniform((n, 3), 2100 + 6 * i + p, -0.1, 0.1) for i in range(len(OGCDR_IDS)) for p in range(6)])
