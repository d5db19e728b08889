"""CPU test of the training driver: YAML schema, schedules, augmentation switch, checkpoint format."""
import os

import torch
import yaml

CFG = {
    "dataset": "sapien", "save_path": None, "random_seed": 10,
    "data": {"root": "/nonexistent", "aug_transform_args": {"scale_low": 0.95, "scale_high": 1.05, "degree_range": [0, 180, 0],
                                                          "shift_range": [0, 0, 0]}, "decentralize": False},
    "aug_transform_epoch": 1, "predflow_path": "flowstep3d", "ignore_npoint_thresh": 0,
    "epochs": 2, "batch_size": 2, "lr": 1.0e-3, "lr_decay": 0.7, "lr_clip": 1.0e-5,
    "bn_momentum": 0.9, "bn_decay": 1.0, "weight_decay": 0.0, "decay_step": 4,
    "segnet": {"n_slot": 4, "n_point": 128, "use_xyz": True, "n_transformer_layer": 1, "transformer_embed_dim": 32,
               "transformer_input_pos_enc": False},
    "loss": {"weights": [10.0, 0.1, 0.1], "start_steps": [0, 0, 0], "dynamic_loss_params": {"loss_norm": 2},
             "smooth_loss_params": {"w_knn": 3.0, "w_ball_q": 1.0,
                                    "knn_loss_params": {"k": 4, "radius": 0.1, "loss_norm": 1},
                                    "ball_q_loss_params": {"k": 8, "radius": 0.2, "loss_norm": 1}},
             "invariance_loss_params": {"loss_norm": 2}},
}


def test_schedules_follow_the_reference_formulas():
    from ogc_amd.train_seg import norm_momentum, schedule_factor
    cfg = dict(CFG, decay_step=200000, lr_decay=0.7, lr=1e-3, lr_clip=1e-5, bn_momentum=0.9, bn_decay=0.5)
    assert schedule_factor(cfg, 0) == 1.0
    assert schedule_factor(cfg, 199999) == 1.0
    assert abs(schedule_factor(cfg, 400000) - 0.49) < 1e-12            # train_seg.py:230-234
    assert schedule_factor(cfg, 10 ** 9) == 1e-5 / 1e-3                 # clipped
    assert norm_momentum(cfg, 200000) == 0.45 and norm_momentum(cfg, 10 ** 9) == 1e-2
    assert norm_momentum(dict(cfg, decay_step=-1), 12345) == 0.9


def test_driver_runs_and_writes_reference_style_checkpoints(tmp_path, monkeypatch, oracle, capsys):
    import ogc_amd.pointnet2.pointnet2 as api
    monkeypatch.setattr(api, "_native", oracle.Pointnet2CudaCPU())
    from ogc_amd import train_seg
    cfg = dict(CFG, save_path=str(tmp_path / "ckpt" / "seg"))
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    os.makedirs(tmp_path / "ckpt", exist_ok=True)
    train_seg.main([str(path), "--round", "1", "--synthetic", "4", "--device", "cpu"])
    exp = cfg["save_path"] + "_R1"
    for name in ("current.pth.tar", "best.pth.tar"):
        state = torch.load(os.path.join(exp, name))
        assert list(state.keys()) == ["model_state"]                    # utils/pytorch_util.py:84-90
        assert "SA_modules.0.mlps.0.layer0.conv.weight" in state["model_state"]
        assert "MF_head.query.weight" in state["model_state"]
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 2
    import json
    e1, e2 = json.loads(lines[0]), json.loads(lines[1])
    assert e1["aug"] is False and e2["aug"] is True                      # views switch on after aug_transform_epoch
    assert e1["train"]["invariance"] == 0 and e2["train"]["invariance"] > 0
    assert e2["lr"] < e1["lr"] or e2["lr"] == e1["lr"] * 0.7 or e2["lr"] <= 1e-3


def test_flow_driver_runs_and_writes_checkpoints(tmp_path, monkeypatch, oracle, capsys):
    """train_flow on the CPU oracle's operators: the reference's config schema (config/flow/sapien/sapien_unsup.yaml),
    its loss_dict keys (losses/flow_loss_unsup.py:128-138) and checkpoint format."""
    import json
    import ogc_amd.pointnet2.pointnet2 as api
    monkeypatch.setattr(api, "_native", oracle.Pointnet2CudaCPU())
    from ogc_amd import train_flow
    cfg = {
        "dataset": "sapien", "save_path": str(tmp_path / "ckpt" / "flow"), "random_seed": 10,
        "flownet": {"npoint": 256, "use_instance_norm": False, "loc_flow_nn": 8, "loc_flow_rad": 0.1, "k_decay_fact": 1.0},
        "model_iters": 2, "epochs": 2, "batch_size": 2, "lr": 1e-3, "lr_decay": 0.5, "lr_clip": 1e-5,
        "bn_momentum": 0.9, "bn_decay": 0.5, "weight_decay": 0.0, "decay_step": 4,
        "loss": {"weights": [0.75, 0.25], "iters_w": [0.5, 0.3], "chamfer_loss_params": {"loss_norm": 2},
                 "smooth_loss_params": {"w_knn": 3.0, "w_ball_q": 1.0,
                                        "knn_loss_params": {"k": 4, "radius": 0.05, "loss_norm": 1},
                                        "ball_q_loss_params": {"k": 8, "radius": 0.1, "loss_norm": 1}}},
    }
    path = tmp_path / "flow.yaml"
    path.write_text(yaml.safe_dump(cfg))
    train_flow.main([str(path), "--synthetic", "4", "--device", "cpu"])
    for name in ("current.pth.tar", "best.pth.tar"):
        state = torch.load(os.path.join(cfg["save_path"], name))
        assert list(state.keys()) == ["model_state"]
        assert "encoder_loc.sa1.mlp_convs.0.weight" in state["model_state"]
        assert "global_corr_layer.epsilon" in state["model_state"]
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 2
    assert set(lines[0]["train"]) == {"chamfer_loss_#0", "smooth_loss_#0", "chamfer_loss_#1", "smooth_loss_#1", "sum",
                                      "epe3d_#0", "epe3d_#1"}          # loss_dict | epe_dict (train_flow.py:75-76)
    assert lines[1]["lr"] == 0.5e-3                                          # lr_curve: 4 samples seen -> one decay
    assert all(v == v for v in lines[1]["train"].values())


def test_train_refine_train_loop(tmp_path, monkeypatch, oracle, capsys):
    """train_seg round 1 -> oa_icp_round (refined flows on disk in the reference's layout) -> train_seg round 2 reading
    them: the loop of the reference's README (train_seg.py / oa_icp.py --save / train_seg.py --round 2)."""
    import json
    import numpy as np
    import ogc_amd.pointnet2.pointnet2 as api
    monkeypatch.setattr(api, "_native", oracle.Pointnet2CudaCPU())
    from ogc_amd import oa_icp_round, train_seg
    cfg = dict(CFG, save_path=str(tmp_path / "ckpt" / "seg"), epochs=1)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    root = str(tmp_path / "data")
    train_seg.main([str(path), "--round", "1", "--synthetic", "2", "--device", "cpu"])
    rep = oa_icp_round.main([str(path), "--round", "1", "--synthetic", "2", "--device", "cpu", "--save", "--flow-root", root,
                             "--test_batch_size", "4"])
    assert rep["icp_iter"] == 20 and rep["pairs"] == 4
    assert set(rep["metrics"]) == {"input", "kabsch", "oa_icp"} and set(rep["metrics"]["oa_icp"]) == {"EPE", "AccS", "AccR", "Outlier"}
    out = os.path.join(root, "flow_preds", "flowstep3d_R1")
    for scene in ("000000", "000001"):
        for v in (1, 2):
            f = np.load(os.path.join(out, scene, "flow%d.npy" % v))
            assert f.shape == (cfg["segnet"]["n_point"], 3) and f.dtype == np.float32 and np.isfinite(f).all()
    # round 2 needs the round-2 checkpoint directory to start from scratch weights: train it reading the refined flows
    train_seg.main([str(path), "--round", "2", "--synthetic", "2", "--device", "cpu", "--flow-root", root])
    assert os.path.exists(os.path.join(cfg["save_path"] + "_R2", "best.pth.tar"))
    ds = train_seg.SyntheticScenes(2, cfg["segnet"]["n_point"], cfg["segnet"]["n_slot"], False, seed=1000,
                                   predflow_dir=out)
    np.testing.assert_array_equal(ds[1][2][0].numpy(), np.load(os.path.join(out, "000001", "flow1.npy")))


def test_vote_driver_evaluates_a_trained_checkpoint(tmp_path, monkeypatch, oracle, capsys):
    """train_seg writes best.pth.tar; `python -m ogc_amd.vote` loads it, runs the network on 4-frame synthetic
    sequences and reports the reference's metrics for the raw and the voted masks (vote.py:284-352)."""
    import math
    import ogc_amd.pointnet2.pointnet2 as api
    monkeypatch.setattr(api, "_native", oracle.Pointnet2CudaCPU())
    from ogc_amd import train_seg, vote
    cfg = dict(CFG, save_path=str(tmp_path / "ckpt" / "seg"), epochs=1)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    os.makedirs(tmp_path / "ckpt", exist_ok=True)
    train_seg.main([str(path), "--round", "1", "--synthetic", "2", "--device", "cpu"])
    res = vote.main([str(path), "--round", "1", "--n_sequence", "2", "--device", "cpu"])
    assert set(res) == {"raw", "voted"}
    for key in ("raw", "voted"):
        assert set(res[key]) == {"AP", "PQ", "F1", "Pre", "Rec", "mIoU", "RI"}
        assert all(math.isfinite(v) and 0.0 <= v <= 1.0 for v in res[key].values()), res[key]
    out = capsys.readouterr().out
    assert "Loaded weights from" in out and "voted" in out


def test_sequence_layout_train_refine_train_loop(tmp_path, monkeypatch, oracle, capsys):
    """The SAPIEN / OGC-DR flavour of the loop: 4-frame scenes sampled as the reference's frame pairs, refined flows
    stored as `<dir>/<scene>.npy` (6, N, 3) + `<dir>.json` {'view_sel'} (oa_icp.py:187-191, dataset_ogcdr.py:147-157)
    and read back by round-2 training through the same meta file."""
    import json
    import numpy as np
    import pytest
    import ogc_amd.pointnet2.pointnet2 as api
    monkeypatch.setattr(api, "_native", oracle.Pointnet2CudaCPU())
    from ogc_amd import oa_icp_round, train_seg
    from ogc_amd.utils import flow_store
    cfg = dict(CFG, save_path=str(tmp_path / "ckpt" / "seg"), epochs=2)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    root = str(tmp_path / "data")
    N = cfg["segnet"]["n_point"]
    train_seg.main([str(path), "--round", "1", "--synthetic", "2", "--device", "cpu", "--frames", "4"])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert lines[0]["it"] == 3 and lines[1]["it"] == 6            # 2 scenes x 3 pairs / batch 2 per epoch
    assert lines[1]["aug"] is True and lines[1]["train"]["invariance"] > 0
    rep = oa_icp_round.main([str(path), "--round", "1", "--synthetic", "2", "--device", "cpu", "--save", "--flow-root", root,
                             "--test_batch_size", "6", "--frames", "4"])
    assert rep["pairs"] == 12
    out = os.path.join(root, "flow_preds", "flowstep3d_R1")
    assert flow_store.read_meta(out) == flow_store.SEQUENCE_PAIRS
    stored = np.load(os.path.join(out, "000001.npy"))
    assert stored.shape == (6, N, 3) and stored.dtype == np.float32 and np.isfinite(stored).all()
    # the refined flow is a flow of the right pair: closer to the ground truth than to its reverse
    ds_gt = train_seg.SyntheticSequenceScenes(2, N, cfg["segnet"]["n_slot"], flow_store.TRAIN_PAIRS, seed=1000)
    ds = train_seg.SyntheticSequenceScenes(2, N, cfg["segnet"]["n_slot"], flow_store.TRAIN_PAIRS, seed=1000, predflow_dir=out)
    assert len(ds) == 6
    pcs, segms, flows, valids = ds[4]                                  # scene 1, pair [1, 2]
    assert pcs.shape == (2, N, 3) and flows.shape == (2, N, 3)
    np.testing.assert_array_equal(flows[0].numpy(), stored[2])         # [1, 2] is entry 2 of the meta order
    np.testing.assert_array_equal(flows[1].numpy(), stored[3])         # [2, 1] is entry 3
    assert (flows - ds_gt[4][2]).norm(dim=-1).mean() < 0.05
    with pytest.raises(ValueError, match="cannot cover"):
        flow_store.load_pair(out, "000001", (0, 2), flow_store.read_meta(out))
    train_seg.main([str(path), "--round", "2", "--synthetic", "2", "--device", "cpu", "--flow-root", root, "--frames", "4"])
    assert os.path.exists(os.path.join(cfg["save_path"] + "_R2", "best.pth.tar"))


def test_stored_flows_are_loaded_before_augmentation(tmp_path):
    """with augmentation on (aug_transform_epoch: 0 in the KITTI config) the dataset must hand out the
    augmented PREDICTED flows (datasets/dataset_kittisf.py:91-117 loads them first), never the synthetic ground truth."""
    import numpy as np
    from ogc_amd import train_seg
    from ogc_amd.utils.data_util import augment_transform
    N, K = 64, 3
    store = tmp_path / "flow_preds" / "flowstep3d"
    rng = np.random.RandomState(5)
    stored = rng.randn(2, N, 3).astype(np.float32)
    os.makedirs(store / "000001")
    for v in (1, 2):
        np.save(store / "000001" / ("flow%d.npy" % v), stored[v - 1])
    aug_args = CFG["data"]["aug_transform_args"]
    # the reference's recipe: two similarity transforms of the pair
    ds = train_seg.SyntheticScenes(2, N, K, False, seed=1000, predflow_dir=str(store), aug_transform_args=aug_args)
    plain = ds[1]
    np.testing.assert_array_equal(plain[2].numpy(), stored)
    ds.aug_transform = True
    pcs, segms, flows, valids = ds[1]
    want_p, want_f = augment_transform(plain[0].numpy().astype(np.float64), stored.astype(np.float64), aug_args,
                                       rng=np.random.RandomState(1000 + 1))
    assert flows.shape == (4, N, 3)
    np.testing.assert_array_equal(flows.numpy(), want_f.astype(np.float32))
    np.testing.assert_array_equal(pcs.numpy(), want_p.astype(np.float32))
    # the built-in recipe (no aug_transform_args): views 2, 3 are a similarity transform s*R of the STORED flows
    ds2 = train_seg.SyntheticScenes(2, N, K, False, seed=1000, predflow_dir=str(store))
    ds2.aug_transform = True
    pcs, segms, flows, valids = ds2[1]
    np.testing.assert_array_equal(flows[:2].numpy(), stored)
    ratio = flows[2].norm(dim=-1) / torch.from_numpy(stored[0]).norm(dim=-1)
    assert 0.95 <= float(ratio.min()) and float(ratio.max()) <= 1.05 and float(ratio.max() - ratio.min()) < 1e-5
    gt = train_seg.SyntheticScenes(2, N, K, False, seed=1000)
    gt.aug_transform = True
    assert not torch.allclose(gt[1][2][2], flows[2])
    # a scene without stored flows keeps the ground truth
    torch.testing.assert_close(ds2[0][2], gt[0][2])


def test_all_ranks_train_on_the_scenes_the_refinement_round_wrote(monkeypatch):
    """the flow store is keyed by scene index, so every rank's training set and oa_icp_round must mean the
    same scene by the same index (one seed, DistributedSampler shards)."""
    import inspect
    from ogc_amd import oa_icp_round, train_flow, train_seg
    assert train_seg.TRAIN_SEED == 1000
    src = inspect.getsource(train_seg.main) + inspect.getsource(train_flow.main)
    assert "rank + 1" not in src
    assert "seed=1000" in inspect.getsource(oa_icp_round.main)


def test_device_only_shortcuts_stay_out_of_the_way_on_cpu():
    """The two-launch Adam step (ogc_adam_step) and the sub-graphs of utils/subgraph.py are for HIP tensors in training: with
    CPU parameters the optimizer's own step is taken and bodies run eagerly."""
    import torch
    from ogc_amd.train_step import _adam_kernel_step, make_optimizer
    from ogc_amd.utils import subgraph
    lin = torch.nn.Linear(3, 2)
    opt = make_optimizer(lin.parameters(), 1e-3)
    lin(torch.randn(4, 3)).sum().backward()
    assert _adam_kernel_step(opt) is None
    x = torch.randn(4, 3, requires_grad=True)
    assert not subgraph.allowed(lin.train(), x)
    assert torch.equal(subgraph.run(lin, "lin", lambda t: lin(t), (x,)), lin(x))


def test_backward_errors_are_classified_by_type_and_solver_name():
    """The reference skips a step on ANY RuntimeError of backward() (train_seg.py:75-78); here only the numerical accidents
    that handler exists for are skipped — a failed decomposition — and they are recognised by exception type or by the
    solver's name as a whole word, not by substrings such as "inf" (which matches "info") or "nan"."""
    import warnings

    import torch

    from ogc_amd import train_step as ts
    from ogc_amd._lib import OgcOpsError
    skipped = ["linalg.svd: The algorithm failed to converge because the input matrix is ill-conditioned",
               "hipsolver error: HIPSOLVER_STATUS_INTERNAL_ERROR, when calling `hipsolverDnSsyevd(...)`",
               "cusolver error: CUSOLVER_STATUS_EXECUTION_FAILED", "torch.linalg.eigh: the matrix is singular",
               # anomaly mode names the autograd node, solver glued to "Backward<n>" 
               "Function 'SvdBackward0' returned nan values in its 0th output.",
               "Function 'LinalgEighBackward0' returned nan values in its 0th output.",
               "Function 'LinalgSvdBackward0' returned nan values in its 1th output."]
    surfaced = ["some info about an inference shape mismatch", "The size of tensor a (3) must match the size of tensor b (4)",
                "nanoseconds elapsed", "HIP error: invalid device function", "HIP out of memory"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for text in skipped:
            assert ts._must_surface(RuntimeError(text)) is False, text
        for text in surfaced:
            assert ts._must_surface(RuntimeError(text)) is True, text
        assert ts._must_surface(torch.linalg.LinAlgError("whatever it says")) is False
        assert ts._must_surface(OgcOpsError("ogc_knn failed")) is True
