"""SapienDataset (ogc_amd/datasets.py) on a synthetic tree in the reference's SAPIEN layout (datasets/dataset_sapien.py:22-170).

The reference's class cannot be imported in this image (utils/sapien_util.py needs pyquaternion), so there is no fixture made by it:
the flows are checked against the articulated motion that generated the frames, the sample contract against the OGC-DR reader's
(same tail), the predicted-flow files through a write / read round trip in the layout the reference's writer uses (:140-151)."""
import json
import os

import numpy as np
import pytest


def _rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def _rigid(rng):
    m = np.eye(4)
    m[:3, :3] = _rot(rng.normal(size=3), rng.uniform(0, np.pi))
    m[:3, 3] = rng.normal(size=3)
    return m


def _make_tree(root, n_scene=3, n_view=4, n_point=256, n_part=3, seed=0):
    """Objects of n_part rigid parts seen from n_view cameras: part k's canonical points moved by its part-to-world motion of the
    view and expressed in that view's camera frame — so the flow a -> b of a point is cam_b^-1 M_b M_a^-1 cam_a p - p."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "data"))
    ids = list(range(10, 10 + n_scene))
    for sid in ids:
        canon = rng.normal(size=(n_point, 3))
        label = rng.integers(1, n_part + 1, size=n_point)
        cams = np.stack([_rigid(rng) for _ in range(n_view)])
        motions = {k: np.stack([_rigid(rng) for _ in range(n_view)]) for k in range(1, n_part + 1)}
        pc = np.empty((n_view, n_point, 3))
        for v in range(n_view):
            for k in range(1, n_part + 1):
                sel = label == k
                world = canon[sel] @ motions[k][v][:3, :3].T + motions[k][v][:3, 3]
                inv = np.linalg.inv(cams[v])
                pc[v, sel] = world @ inv[:3, :3].T + inv[:3, 3]
        trans = dict(motions)
        trans["cam"] = cams
        np.savez(os.path.join(root, "data", "%06d.npz" % sid), pc=pc, segm=np.stack([label] * n_view), trans=np.array(trans, dtype=object))
    with open(os.path.join(root, "meta.json"), "w") as f:
        json.dump({"train": ids[:-1], "val": ids[-1:], "test": ids}, f)
    return ids


def test_flows_are_the_parts_motions(tmp_path):
    from ogc_amd.datasets import SapienDataset
    root = str(tmp_path / "mbs-shapepart")
    ids = _make_tree(root)
    view_sels = [[0, 1], [1, 2], [2, 3]]
    ds = SapienDataset(data_root=root, split="train", view_sels=view_sels)
    assert len(ds) == (len(ids) - 1) * len(view_sels)
    for sid in range(len(ds)):
        pcs, segms, flows, valids = ds[sid]
        assert pcs.shape == (2, 256, 3) and pcs.dtype == np.float32 and flows.shape == (2, 256, 3) and flows.dtype == np.float32
        assert segms.shape == (2, 256) and segms.dtype == np.int32 and valids.dtype == np.float32 and valids.min() == 1.0
        # frames hold the same canonical points in the same order: the flow carries frame a onto frame b and back
        np.testing.assert_allclose(pcs[0] + flows[0], pcs[1], atol=2e-5)
        np.testing.assert_allclose(pcs[1] + flows[1], pcs[0], atol=2e-5)
        assert segms.min() == 0 and segms.max() == 2          # labels compressed to 0 .. n_part-1 (utils/data_util.py)
    # centring and augmentation: the tail shared with the other readers
    aug = {"scale_low": 0.95, "scale_high": 1.05, "degree_range": [0, 180, 0], "shift_range": [0, 0, 0]}
    ds2 = SapienDataset(data_root=root, split="val", view_sels=view_sels, decentralize=True, aug_transform=True, aug_transform_args=aug)
    pcs, segms, flows, valids = ds2[0]
    assert pcs.shape == (4, 256, 3) and segms.shape == (4, 256) and flows.shape == (4, 256, 3) and valids.shape == (4, 256)
    assert abs(pcs[:2].mean(1).mean(0)).max() < 1e-5
    np.testing.assert_array_equal(segms[:2], segms[2:])


def test_predicted_flows_round_trip_in_the_reference_layout(tmp_path):
    import torch
    from ogc_amd.datasets import SapienDataset
    from ogc_amd.utils import flow_store
    root = str(tmp_path / "mbs-sapien")
    ids = _make_tree(root, seed=1)
    pairs = flow_store.SEQUENCE_PAIRS
    ds = SapienDataset(data_root=root, split="test", view_sels=pairs)
    out = os.path.join(root, "flow_preds", "flowstep3d_R1")
    os.makedirs(out)
    flow_store.write_meta(out, pairs)
    n_frame, batch = len(pairs), 2 * len(pairs)
    pred = np.stack([ds[i][2][0] for i in range(len(ds))]) + 0.25            # flow of the first frame of every ordered pair
    for offset, start in enumerate(range(0, len(ds), batch)):                # as oa_icp.py:222-229 drives the writer
        ds._save_predflow(torch.from_numpy(pred[start:start + batch]), out, batch, n_frame=n_frame, offset=offset)
    assert sorted(os.listdir(out)) == ["%06d.npy" % i for i in ids]
    assert np.load(os.path.join(out, "%06d.npy" % ids[0])).shape == (n_frame, 256, 3)
    rd = SapienDataset(data_root=root, split="test", view_sels=flow_store.TRAIN_PAIRS, predflow_path="flowstep3d_R1")
    for sid in range(len(rd)):
        scene, (a, b) = sid // 3, flow_store.TRAIN_PAIRS[sid % 3]
        flows = rd[sid][2]
        np.testing.assert_array_equal(flows[0], pred[scene * n_frame + pairs.index([a, b])])
        np.testing.assert_array_equal(flows[1], pred[scene * n_frame + pairs.index([b, a])])
    with pytest.raises(ValueError):
        SapienDataset(data_root=root, split="test", view_sels=[[0, 2]], predflow_path="flowstep3d_R1")
