"""The drivers on the MI355X with the HIP operators (SURVEY §8 f1 / f2 / e): train_seg.main, train_flow.main and the
train -> OA-ICP refinement -> train loop through the on-disk flow store, at the real config shapes (C4: 8192-point
KITTI-style scenes) for a few steps; and bench.py under a one-process torch.distributed launcher so that the RCCL code
path (process-group init, FlatDataParallel, the flat all-reduce, barrier + MAX timing) is executed on the GPU box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

import golden_cases as gc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _hip():
    assert torch.cuda.is_available()
    import ogc_amd  # noqa: F401


def _cfg(name, tmp_path, **over):
    with open(os.path.join(ROOT, "config", name)) as f:
        cfg = yaml.safe_load(f)
    cfg.update(over)
    cfg["save_path"] = str(tmp_path / "ckpt" / "run")
    os.makedirs(tmp_path / "ckpt", exist_ok=True)
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    return cfg, str(path)


def _json_lines(capsys):
    return [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]


def test_train_seg_main_c4(tmp_path, capsys):
    """config/kittisf_unsup_synthetic.yaml (C4: batch 4 x 4 views x 8192 points, augmentation on from epoch 1) for three
    steps: finite losses, the reference's loss_dict keys, checkpoints holding exactly the reference's state_dict keys."""
    from ogc_amd import train_seg
    cfg, path = _cfg("kittisf_unsup_synthetic.yaml", tmp_path, epochs=1)
    train_seg.main([path, "--round", "1", "--synthetic", "12", "--max-iters", "3"])
    line = _json_lines(capsys)[-1]
    assert line["it"] == 3 and line["aug"] is True
    assert set(line["train"]) == {"dynamic", "smooth", "invariance", "entropy", "rank", "sum"}
    assert all(np.isfinite(v) for v in line["train"].values()) and np.isfinite(line["val_loss"])
    assert set(line["val"]) == {"AP", "PQ", "F1", "Pre", "Rec"}
    want = [str(k) for k in gc.load("model_segnet_kitti")["state_keys"]]
    for name in ("current.pth.tar", "best.pth.tar"):
        state = torch.load(os.path.join(cfg["save_path"] + "_R1", name))
        assert list(state) == ["model_state"] and sorted(state["model_state"]) == want
    # the weights moved
    init = torch.load(os.path.join(cfg["save_path"] + "_R1", "best.pth.tar"))["model_state"]
    cur = torch.load(os.path.join(cfg["save_path"] + "_R1", "current.pth.tar"))["model_state"]
    assert any(not torch.equal(init[k], cur[k]) for k in init) or line["val_loss"] < 1e10


def test_train_seg_main_hip_graph(tmp_path, capsys):
    """The same driver with --hip-graph (the step replayed as one HIP graph, ogc_amd/graph_step.py) against the eager run:
    two epochs of three steps — the second one starts with a batch the graph did not plan — and the same epoch losses."""
    from ogc_amd import train_seg
    lines = {}
    for mode, extra in (("eager", []), ("graph", ["--hip-graph"])):
        cfg, path = _cfg("kittisf_unsup_synthetic.yaml", tmp_path, epochs=2, save_path=str(tmp_path / mode))
        train_seg.main([path, "--round", "1", "--synthetic", "12"] + extra)
        lines[mode] = _json_lines(capsys)[-2:]
    for e, g in zip(lines["eager"], lines["graph"]):
        assert e["it"] == g["it"] and e["epoch"] == g["epoch"]
        for k in ("dynamic", "smooth", "invariance", "sum"):
            assert abs(e["train"][k] - g["train"][k]) <= 5e-3 * max(1.0, abs(e["train"][k])), (k, e["train"], g["train"])
        assert abs(e["val_loss"] - g["val_loss"]) <= 2e-2 * max(1.0, abs(e["val_loss"]))


def test_train_flow_main_c3(tmp_path, capsys):
    """config/kittisf_flow_synthetic.yaml (C3: FlowStep3D on 8192-point pairs) for two steps."""
    from ogc_amd import train_flow
    cfg, path = _cfg("kittisf_flow_synthetic.yaml", tmp_path, epochs=1, batch_size=1, model_iters=2)
    cfg["loss"]["iters_w"] = cfg["loss"]["iters_w"][:2]
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    train_flow.main([path, "--synthetic", "4", "--max-iters", "2"])
    line = _json_lines(capsys)[-1]
    assert line["it"] == 2
    assert {"chamfer_loss_#0", "smooth_loss_#0", "chamfer_loss_#1", "smooth_loss_#1", "sum"} <= set(line["train"])
    assert all(np.isfinite(v) for v in line["train"].values()) and np.isfinite(line["val_loss"])
    want = [str(k) for k in gc.load("model_flownet_kitti")["state_keys"]]
    for name in ("current.pth.tar", "best.pth.tar"):
        state = torch.load(os.path.join(cfg["save_path"], name))
        assert sorted(state["model_state"]) == want


def test_train_refine_train_round_trip(tmp_path, capsys):
    """Round 1 training -> oa_icp_round (HIP soft-NN, weighted Kabsch; flows written in the KITTI layout) -> round 2
    training that reads them back through --flow-root, with augmentation on (the stored flows are what gets augmented)."""
    from ogc_amd import oa_icp_round, train_seg
    cfg, path = _cfg("kittisf_unsup_synthetic.yaml", tmp_path, epochs=1)
    cfg["segnet"]["n_point"] = 2048
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    root = str(tmp_path / "data")
    train_seg.main([path, "--round", "1", "--synthetic", "4", "--max-iters", "1"])
    rep = oa_icp_round.main([path, "--round", "1", "--synthetic", "4", "--save", "--flow-root", root, "--test_batch_size", "4"])
    assert rep["icp_iter"] == 20 and rep["pairs"] == 8
    m = rep["metrics"]
    assert all(np.isfinite(v) for k in m for v in m[k].values())
    # (no quality claim: the masks come from a network trained for ONE step)
    out = os.path.join(root, "flow_preds", "flowstep3d_R1")
    stored = np.stack([np.load(os.path.join(out, "000002", "flow%d.npy" % v)) for v in (1, 2)])
    assert stored.shape == (2, 2048, 3) and stored.dtype == np.float32 and np.isfinite(stored).all()
    ds = train_seg.SyntheticScenes(4, 2048, cfg["segnet"]["n_slot"], True, seed=train_seg.TRAIN_SEED, predflow_dir=out)
    ds.aug_transform = True
    np.testing.assert_array_equal(ds[2][2][:2].numpy(), stored)
    capsys.readouterr()
    train_seg.main([path, "--round", "2", "--synthetic", "4", "--max-iters", "1", "--flow-root", root])
    line = _json_lines(capsys)[-1]
    assert all(np.isfinite(v) for v in line["train"].values())
    assert os.path.exists(os.path.join(cfg["save_path"] + "_R2", "best.pth.tar"))


def _bench(args, launcher):
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(29500 + os.getpid() % 1000)]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_bench_under_one_process_rccl_launcher():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: backend nccl (= RCCL), FlatDataParallel with
    the collective forced on, barrier + MAX-over-ranks timing.  The JSON line must parse, carry the collective's
    payload, and the step must cost what the un-launched step costs (the all-reduce of 2.4 MB is tens of microseconds)."""
    args = ["--gpus", "1", "--steps", "12", "--warmup", "5", "--no-cpu-baseline"]
    plain = _bench(args, launcher=False)
    dist_ = _bench(args, launcher=True)
    for r in (plain, dist_):
        assert r["n_gpus"] == 1 and r["steps"] == 12 and r["unit"] == "point-clouds/s" and r["scaling"] == "weak"
        assert r["config"]["optimizer_stepped"] is True
    assert dist_["collective"]["backend"] == "nccl" and dist_["collective"]["payload_bytes"] == 4 * 597248
    assert dist_["collective"]["avg_ms"] > 0
    # (two processes one after the other on a shared box: 10 % + 0.6 ms of slack for the run-to-run spread of short runs)
    assert dist_["ms_per_step"] <= 1.10 * plain["ms_per_step"] + 0.6, (dist_["ms_per_step"], plain["ms_per_step"])


@pytest.mark.timeout(900)
def test_bench_two_ranks_share_the_gpu_over_gloo():
    """bench.py's multi-rank control flow on the one-GPU box: two ranks (both on cuda:0, gloo instead of RCCL, which refuses two
    ranks on one device) under the driver's launcher line.  The timed steps average gradients across the ranks; after them
    NO rank may run further training steps alone (round 5: rank 0's single-GPU extras used to — a gradient collective the
    other ranks never joined, i.e. a hang on the first real multi-GPU run); exactly one JSON line, whole-job throughput."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OGC_BENCH_SHARE_GPU="1", OGC_BENCH_BACKEND="gloo")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["scaling"] == "weak" and r["config"]["optimizer_stepped"] is True
    assert r["config"]["parallelism"] == "dp2" and r["collective"]["world"] == 2 and r["collective"]["backend"] == "gloo"
    # whole-job value: both ranks' clouds over the slower rank's time
    assert abs(r["value"] - 2 * r["config"]["clouds_per_step_per_gpu"] / (r["ms_per_step"] * 1e-3)) <= 0.01 * r["value"]
    assert "cpu_baseline" not in r and "ms_per_step_with_h2d" not in r      # single-GPU readings: N = 1 only
