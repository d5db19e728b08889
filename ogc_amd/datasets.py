"""Readers / writers of the two on-disk layouts the drivers exchange data in — scenes as the reference's preparation scripts
leave them, predicted flows as its refinement script leaves them — with the interface of the reference's data sets (same class
names, constructor arguments and sample contract), so that `train_seg`, `oa_icp_round` and the reference's own tools can work on
one directory tree:

  KITTISceneFlowDataset      <root>/data/<id>/{pc,segm,flow}{1,2}.npy, a split file listing the ids
                             (datasets/dataset_kittisf.py:9-137); predicted flows <root>/flow_preds/<name>/<id>/flow{1,2}.npy
  OGCDynamicRoomDataset      <root>/data/<id>/{pc,segm,pose}_%02d.npy, <root>/data/<split>.lst
                             (datasets/dataset_ogcdr.py:30-157); predicted flows <root>/flow_preds/<name>/<id>.npy (P, N, 3) with
                             <root>/flow_preds/<name>.json {"view_sel": [...]} naming the P ordered frame pairs
  SapienDataset              <root>/data/%06d.npz {pc (V, N, 3), segm (V, N), trans: {"cam": (V, 4, 4), part id: (V, 4, 4)}},
                             <root>/meta.json {split: [ids]} (datasets/dataset_sapien.py:22-170); predicted flows in the
                             OGC-DR layout under %06d.npy.  NOT pinned against the reference's class: it imports pyquaternion
                             (utils/sapien_util.py), which this image lacks — tests/test_sapien_reader.py checks the flows against
                             the rigid motions applied directly

A sample is (pcs (t, N, 3) f32, segms (t, N) i32, flows (t, N, 3) f32, valids (t, N) f32), t = 2 frames, or 4 with
`aug_transform` (two random similarity transforms of the pair, utils/data_util.py:140-195).  The one-hot label variant of the
supervised baselines (`onehot_label`) is out of scope.  tests/golden/flow_store.npz pins files and samples against
the reference's classes.
"""
import os

import numpy as np
from torch.utils.data import Dataset

from .utils import flow_store
from .utils.data_util import augment_transform, compress_label_id


def _finish(pcs, segms, flows, decentralize, aug_transform, aug_transform_args):
    """Common tail of both readers: centring, label compression, augmentation, dtypes (dataset_kittisf.py:95-122)."""
    pcs, segms, flows = np.stack(pcs, 0), np.stack(segms, 0), np.stack(flows, 0)
    if decentralize:
        pcs = pcs - pcs.mean(1).mean(0)
    segms = compress_label_id(np.reshape(segms, -1)).reshape(2, -1)
    valids = np.ones_like(segms, dtype=np.float32)
    if aug_transform:
        pcs, flows = augment_transform(pcs, flows, aug_transform_args)
        segms = np.concatenate((segms, segms), 0)
        valids = np.concatenate((valids, valids), 0)
    return pcs.astype(np.float32), segms.astype(np.int32), flows.astype(np.float32), valids.astype(np.float32)


class KITTISceneFlowDataset(Dataset):
    def __init__(self, data_root, mapping_path, downsampled=False, view_sels=[[0, 1]], predflow_path=None, decentralize=False,
                 aug_transform=False, aug_transform_args=None):
        self.data_root = os.path.join(data_root, "data" if downsampled else "processed")
        with open(mapping_path, "r") as f:
            self.data_ids = f.read().strip().split("\n")
        self.view_sels = [list(v) for v in view_sels]
        self.predflow_path = os.path.join(data_root, "flow_preds", predflow_path) if predflow_path is not None else None
        self.downsampled, self.decentralize = downsampled, decentralize
        self.aug_transform, self.aug_transform_args = aug_transform, aug_transform_args

    def __len__(self):
        return len(self.data_ids) * len(self.view_sels)

    def _load_data(self, idx, view_sel):
        d = os.path.join(self.data_root, self.data_ids[idx])
        a, b = view_sel
        pc1, pc2 = np.load(os.path.join(d, "pc%d.npy" % (a + 1))), np.load(os.path.join(d, "pc%d.npy" % (b + 1)))
        if self.downsampled:
            segms = [np.load(os.path.join(d, "segm%d.npy" % (v + 1))) for v in (a, b)]
            flows = [np.load(os.path.join(d, "flow%d.npy" % (v + 1))) for v in (a, b)]
        else:  # the full-resolution scans: one labelling, points in correspondence (dataset_kittisf.py:73-76)
            segm = np.load(os.path.join(d, "segm.npy"))
            segms, flows = [segm, segm], [pc2 - pc1, pc1 - pc2]
        return [pc1, pc2], segms, flows

    def _load_predflow(self, idx, view_sel):
        stored = flow_store.load_pair(self.predflow_path, self.data_ids[idx], view_sel)
        if stored is None:
            raise FileNotFoundError("no predicted flows for scene %s under %s" % (self.data_ids[idx], self.predflow_path))
        return stored

    def __getitem__(self, sid):
        idx, view_sel = sid // len(self.view_sels), self.view_sels[sid % len(self.view_sels)]
        pcs, segms, flows = self._load_data(idx, view_sel)
        if self.predflow_path is not None:
            flows = self._load_predflow(idx, view_sel)
        return _finish(pcs, segms, flows, self.decentralize, self.aug_transform, self.aug_transform_args)

    def _save_predflow(self, flow_pred, save_root, batch_size, n_frame=1, offset=0):
        """flow_pred (B, N, 3): sample `offset * batch_size + i` of a loader over (scene, frame) pairs -> <save_root>/<id>/flow<k>.npy
        (dataset_kittisf.py:125-137)."""
        flow_pred = flow_pred.detach().cpu().numpy() if hasattr(flow_pred, "detach") else np.asarray(flow_pred)
        for i in range(flow_pred.shape[0]):
            idx, k = divmod(offset * batch_size + i, n_frame)
            d = os.path.join(save_root, self.data_ids[idx])
            os.makedirs(d, exist_ok=True)
            np.save(os.path.join(d, "flow%d.npy" % (k + 1)), flow_pred[i])


def compute_flow(pc1, segm1, pose1, pose2):
    """Per-point flow of frame 1 from the objects' pose change; object ids start at 1, id 0 (background) stays put
    (dataset_ogcdr.py:10-27)."""
    flow = np.zeros_like(pc1)
    for k in range(pose1.shape[0]):
        rel = pose2[k] @ np.linalg.inv(pose1[k])
        sel = segm1 == (k + 1)
        flow[sel] = pc1[sel] @ rel[:3, :3].T + rel[:3, 3] - pc1[sel]
    return flow


class OGCDynamicRoomDataset(Dataset):
    def __init__(self, data_root, split="train", view_sels=[[0, 1]], predflow_path=None, decentralize=False, aug_transform=False,
                 aug_transform_args=None):
        self.data_root = os.path.join(data_root, "data")
        self.split = split
        with open(os.path.join(self.data_root, split + ".lst"), "r") as f:
            self.data_ids = f.read().strip().split("\n")
        self.view_sels = [list(v) for v in view_sels]
        self.predflow_path, self.pf_view_sels = None, None
        if predflow_path is not None:
            self.predflow_path = os.path.join(data_root, "flow_preds", predflow_path)
            self.pf_view_sels = flow_store.read_meta(self.predflow_path)
            if self.pf_view_sels is None:
                raise FileNotFoundError(self.predflow_path + ".json")
            if any(sel not in self.pf_view_sels for sel in self.view_sels):
                raise ValueError("Flow predictions cannot cover specified view selections!")
        self.decentralize = decentralize
        self.aug_transform, self.aug_transform_args = aug_transform, aug_transform_args

    def __len__(self):
        return len(self.data_ids) * len(self.view_sels)

    def _load_data(self, idx, view_sel):
        d = os.path.join(self.data_root, self.data_ids[idx])
        return tuple([np.load(os.path.join(d, "%s_%02d.npy" % (what, v))) for v in view_sel] for what in ("pc", "segm", "pose"))

    def __getitem__(self, sid):
        idx, view_sel = sid // len(self.view_sels), self.view_sels[sid % len(self.view_sels)]
        pcs, segms, poses = self._load_data(idx, view_sel)
        if self.predflow_path is not None:
            flows = flow_store.load_pair(self.predflow_path, self.data_ids[idx], view_sel, self.pf_view_sels)
        else:
            flows = [compute_flow(pcs[0], segms[0], poses[0], poses[1]), compute_flow(pcs[1], segms[1], poses[1], poses[0])]
        return _finish(pcs, segms, flows, self.decentralize, self.aug_transform, self.aug_transform_args)

    def _save_predflow(self, flow_pred, save_root, batch_size, n_frame=1, offset=0):
        """flow_pred (B, N, 3), the n_frame ordered pairs of a scene adjacent -> <save_root>/<id>.npy (n_frame, N, 3)
        (dataset_ogcdr.py:147-157)."""
        flow_pred = flow_pred.detach().cpu().numpy() if hasattr(flow_pred, "detach") else np.asarray(flow_pred)
        for i in range(flow_pred.shape[0] // n_frame):
            idx = offset * batch_size // n_frame + i
            np.save(os.path.join(save_root, self.data_ids[idx] + ".npy"), flow_pred[i * n_frame:(i + 1) * n_frame])


def _rigid_inverse(m):
    """Inverse of a 4x4 rigid transform as (R^T, -R^T t) — what the reference's Isometry.inv() computes on (quaternion, t)
    (utils/sapien_util.py:49-51), not a general matrix inverse."""
    out = np.eye(4, dtype=np.float64)
    out[:3, :3] = m[:3, :3].T
    out[:3, 3] = -(m[:3, :3].T @ m[:3, 3])
    return out


def compute_part_flow(base_pc, base_segms, base_cam, base_motions, dest_cam, dest_motions):
    """Flow of a SAPIEN frame from its parts' motions: part k (label k + 1) moves by
    dest_cam^-1 . dest_motion_k . base_motion_k^-1 . base_cam (all 4x4, camera-to-world and part-to-world);
    points of no part keep whatever np.empty_like holds in the reference — here 0 (datasets/dataset_sapien.py:12-20)."""
    final_pc = base_pc.astype(np.float64).copy()
    for k in range(len(base_motions)):
        sel = np.where(base_segms == (k + 1))[0]
        rel = _rigid_inverse(dest_cam) @ dest_motions[k] @ _rigid_inverse(base_motions[k]) @ base_cam
        final_pc[sel] = base_pc[sel] @ rel[:3, :3].T + rel[:3, 3]
    return (final_pc - base_pc).astype(base_pc.dtype)


class SapienDataset(Dataset):
    """Reference: datasets/dataset_sapien.py:22-170 (the one-hot label variant of the supervised baselines is out of scope)."""

    def __init__(self, data_root, split="train", view_sels=[[0, 1]], predflow_path=None, decentralize=False, aug_transform=False,
                 aug_transform_args=None):
        import json
        self.data_root = os.path.join(data_root, "data")
        with open(os.path.join(data_root, "meta.json")) as f:
            self.meta = json.load(f)
        self.split, self.data_ids = split, self.meta[split]
        self.view_sels = [list(v) for v in view_sels]
        self.predflow_path, self.pf_view_sels = None, None
        if predflow_path is not None:
            self.predflow_path = os.path.join(data_root, "flow_preds", predflow_path)
            self.pf_view_sels = flow_store.read_meta(self.predflow_path)
            if self.pf_view_sels is None:
                raise FileNotFoundError(self.predflow_path + ".json")
            if any(sel not in self.pf_view_sels for sel in self.view_sels):
                raise ValueError("Flow predictions cannot cover specified view selections!")
        self.decentralize = decentralize
        self.aug_transform, self.aug_transform_args = aug_transform, aug_transform_args

    def __len__(self):
        return len(self.data_ids) * len(self.view_sels)

    def _name(self, idx):
        return "%06d" % self.data_ids[idx]

    def _load_data(self, idx):
        data = np.load(os.path.join(self.data_root, self._name(idx) + ".npz"), allow_pickle=True)
        return data["pc"].astype(np.float32), data["segm"], data["trans"].item()

    def __getitem__(self, sid):
        idx, view_sel = sid // len(self.view_sels), self.view_sels[sid % len(self.view_sels)]
        pcs, segms, trans = self._load_data(idx)
        n_parts = len(trans) - 1
        a, b = view_sel
        pcs, segms = pcs[view_sel], segms[view_sel]
        if self.predflow_path is not None:
            flows = flow_store.load_pair(self.predflow_path, self._name(idx), view_sel, self.pf_view_sels)
            if flows is None:
                raise FileNotFoundError(os.path.join(self.predflow_path, self._name(idx) + ".npy"))
        else:
            def motions(v):
                return [np.asarray(trans[t][v], dtype=np.float64) for t in range(1, n_parts + 1)]
            cam = lambda v: np.asarray(trans["cam"][v], dtype=np.float64)
            flows = [compute_part_flow(pcs[0], segms[0], cam(a), motions(a), cam(b), motions(b)),
                     compute_part_flow(pcs[1], segms[1], cam(b), motions(b), cam(a), motions(a))]
        return _finish(list(pcs), list(segms), flows, self.decentralize, self.aug_transform, self.aug_transform_args)

    def _save_predflow(self, flow_pred, save_root, batch_size, n_frame=1, offset=0):
        """flow_pred (B, N, 3), the n_frame ordered pairs of a scene adjacent -> <save_root>/%06d.npy (n_frame, N, 3)
        (dataset_sapien.py:140-151)."""
        flow_pred = flow_pred.detach().cpu().numpy() if hasattr(flow_pred, "detach") else np.asarray(flow_pred)
        for i in range(flow_pred.shape[0] // n_frame):
            idx = offset * batch_size // n_frame + i
            np.save(os.path.join(save_root, self._name(idx) + ".npy"), flow_pred[i * n_frame:(i + 1) * n_frame])
