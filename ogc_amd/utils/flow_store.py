"""Predicted scene flows on disk, in the two layouts the reference's datasets write and read, so that refinement
rounds chain with training exactly as in the reference (train_seg -> oa_icp --save -> train_seg --round 2):

  pair layout (KITTI-SF; datasets/dataset_kittisf.py:99-104 reads, :125-137 writes):
      <dir>/<scene id>/flow1.npy   (N, 3) flow of frame 1 (to frame 2)
      <dir>/<scene id>/flow2.npy   (N, 3) flow of frame 2 (to frame 1)
  sequence layout (SAPIEN / OGC-DR, 4 frames per scene; datasets/dataset_ogcdr.py:58-66,95-112 reads, :147-157 writes,
  oa_icp.py:187-191 writes the meta file):
      <dir>/<scene id>.npy         (P, N, 3) one flow per ordered frame pair
      <dir>.json                   {"view_sel": [[a, b], ...]}  the P ordered pairs, in file order

`<dir>` is `<data root>/flow_preds/<name>[_R<round>]`.
"""
import json
import os

import numpy as np

SEQUENCE_PAIRS = [[0, 1], [1, 0], [1, 2], [2, 1], [2, 3], [3, 2]]   # oa_icp.py:148, train_flow.py:246
TRAIN_PAIRS = [[0, 1], [1, 2], [2, 3]]                              # train_seg.py:295


def write_meta(directory, view_sels):
    with open(directory + ".json", "w") as f:
        json.dump({"view_sel": [list(map(int, v)) for v in view_sels]}, f)


def read_meta(directory):
    """The ordered frame pairs a sequence-layout directory holds, or None for the pair layout."""
    path = directory + ".json"
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["view_sel"]


def save_pair(directory, scene_id, flow1, flow2):
    scene_dir = os.path.join(directory, scene_id)
    os.makedirs(scene_dir, exist_ok=True)
    np.save(os.path.join(scene_dir, "flow1.npy"), np.asarray(flow1, np.float32))
    np.save(os.path.join(scene_dir, "flow2.npy"), np.asarray(flow2, np.float32))


def save_sequence(directory, scene_id, flows):
    """flows (P, N, 3) in the order of the directory's meta file."""
    os.makedirs(directory, exist_ok=True)
    np.save(os.path.join(directory, scene_id + ".npy"), np.asarray(flows, np.float32))


def load_pair(directory, scene_id, view_sel=(0, 1), meta=None):
    """[flow a->b on frame a, flow b->a on frame b] for view_sel = (a, b), or None when the scene has no file.
    With `meta` (read_meta) the sequence layout is read, and a pair the files do not cover is an error, as in the
    reference (datasets/dataset_ogcdr.py:63-65)."""
    a, b = int(view_sel[0]), int(view_sel[1])
    if meta is None:
        paths = [os.path.join(directory, scene_id, "flow%d.npy" % v) for v in (1, 2)]
        if not all(os.path.exists(p) for p in paths):
            return None
        flows = [np.load(p) for p in paths]
        return flows if (a, b) == (0, 1) else flows[::-1]
    if [a, b] not in meta or [b, a] not in meta:
        raise ValueError("Flow predictions cannot cover the specified view selections!")
    path = os.path.join(directory, scene_id + ".npy")
    if not os.path.exists(path):
        return None
    stored = np.load(path)
    return [stored[meta.index([a, b])], stored[meta.index([b, a])]]
