"""Generates the DRIVER fixtures by running the reference's own trainers and refinement script (imported / executed from
/root/reference, which exists only in the build container) on the CPU oracle's operators:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_driver_golden.py [seg] [waymo] [flow] [store]
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_driver_golden.py seg64 waymo64 flow64     (float64 truths, separate process)

  train_seg_trace.npz   train_seg.Trainer.train (train_seg.py:19-226) over 3 epochs of a six-scene in-memory data set, with its
                        LambdaLR(lr_curve) and BNMomentumScheduler(bn_curve): per iteration the loss_dict, the learning rate,
                        the norm momentum and a summary of the weights after the step (one NaN-gradient step included: a
                        scene whose flow holds a NaN from the second epoch on); per epoch the validation loss, its loss_dict,
                        PQ / F1 / Pre / Rec and the best-checkpoint decision.  Loss terms are gated by `it * b`
                        (start_steps), augmented views switch on after `aug_transform_epoch`.
  train_seg_waymo_trace.npz  the same for train_seg_waymo.Trainer.train (train_seg_waymo.py:20-242: every other view kept, :59; its own
                        one-frame UnsupervisedOGCLoss, :244-334) with segnet_kitti on 512-point scenes, 2 epochs (the second with
                        the augmented twin and the NaN step).
  train_flow_trace.npz  the same for train_flow.Trainer.train (train_flow.py:33-184) with flownet_sapien.
  flow_store.npz        what the reference's data sets write and read as predicted flows, and what its refinement script
                        oa_icp.py does end to end: KITTISceneFlowDataset._save_predflow / __getitem__ with predflow_path
                        (datasets/dataset_kittisf.py:82-137), OGCDynamicRoomDataset likewise (datasets/dataset_ogcdr.py:
                        95-157) with the meta file of oa_icp.py:187-191, and `oa_icp.py <cfg> --split train --round 1 --save`
                        run as a script on a three-scene KITTI-SF style directory (the file bytes it leaves behind).
Weights come from tests/golden/detgen.py (integer hashing), so the fixtures hold data only.
"""
import io
import json
import os
import runpy
import shutil
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import detgen  # noqa: E402
import driver_cases as dc  # noqa: E402  (configurations and data shared with the tests)
from make_golden import install_shims, save  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class _Writer:  # tensorboardX.SummaryWriter as the trainers use it
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass

    def add_scalars(self, *a, **k):
        pass

    def flush(self):
        pass


def shims():
    install_shims()
    sys.modules["tensorboardX"].SummaryWriter = _Writer
    os.environ.setdefault("MPLBACKEND", "Agg")


def weight_summary(net):
    """Per parameter tensor: L2 norm and the first 16 entries (what the tests compare; the full weights would be 2.7 MB)."""
    norms, heads = [], []
    for _, p in net.named_parameters():
        norms.append(float(p.detach().double().norm()))
        h = p.detach().flatten()[:16].double().numpy()
        heads.append(np.pad(h, (0, 16 - len(h))))
    return np.array(norms), np.stack(heads)


class _AsDouble(torch.utils.data.Dataset):
    """The same samples with float64 coordinates and flows (the float64 truth run)."""

    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __setattr__(self, k, v):
        if k == "aug_transform":   # the trainer switches augmentation on through the data set's attribute
            self.ds.aug_transform = v
        object.__setattr__(self, k, v)

    def __getitem__(self, i):
        pcs, segms, flows, valids = self.ds[i]
        return pcs.astype(np.float64), segms, flows.astype(np.float64), valids.astype(np.float64)


def gen_train_seg_waymo(f64=False):
    """As gen_train_flow: fp32 tries data seeds until the reference's own run does not move (< 1e-4 on every loss term) under
    one-ulp changes of the coordinates — a near-uniform mask on an IoU-matching boundary moves the invariance term by a percent
    on every implementation, the reference's included — and writes that seed's trace; float64: the truth for the stored seed."""
    if f64:
        seed = int(np.load(os.path.join(HERE, "train_seg_waymo_trace.npz"))["data_seed"][0])
        return gen_train_seg(True, True, seed, 0, True)
    for seed in range(3000, 3400, 20):
        base = gen_train_seg(False, True, seed, 0, False)
        dev = 0.0
        for ulp in (1, 2):
            other = gen_train_seg(False, True, seed, ulp, False)
            for key in ("loss", "val_avg"):
                a, b = np.nan_to_num(other[key]), np.nan_to_num(base[key])
                dev = max(dev, float(np.max(np.abs(a - b) / np.abs(b).clip(1e-3))))
        print("seed %d: largest change of a loss under one-ulp perturbations %.2e" % (seed, dev), flush=True)
        if dev < 1e-4:
            return gen_train_seg(False, True, seed, 0, True)
    raise RuntimeError("no stable seed")


def gen_train_seg(f64=False, waymo=False, data_seed=3000, ulp=0, write=True):
    """waymo: train_seg_waymo.Trainer.train (train_seg_waymo.py:20-242) with ITS UnsupervisedOGCLoss (:244-334) and segnet_kitti."""
    from losses.seg_loss_unsup import DynamicLoss, EntropyLoss, InvarianceLoss, RankLoss, SmoothLoss
    from metrics.seg_metric import calculate_PQ_F1
    from utils.pytorch_util import BNMomentumScheduler
    if waymo:
        import train_seg_waymo as ref
        from models.segnet_kitti import MaskFormer3D
        UnsupervisedOGCLoss = ref.UnsupervisedOGCLoss
        cfg = dc.WAYMO_CFG
    else:
        import train_seg as ref
        from losses.seg_loss_unsup import UnsupervisedOGCLoss
        from models.segnet_sapien import MaskFormer3D
        cfg = dc.SEG_CFG
    ref.args = types.SimpleNamespace(**cfg)
    net = detgen.fill_module(MaskFormer3D(**cfg["segnet"]), 33 if waymo else 31)
    if f64:
        net = net.double()
    optimizer = torch.optim.Adam(net.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=ref.lr_curve)
    bnm_scheduler = BNMomentumScheduler(net, bn_lambda=ref.bn_curve)
    L = cfg["loss"]
    criterion = UnsupervisedOGCLoss(DynamicLoss(**L["dynamic_loss_params"]), SmoothLoss(**L["smooth_loss_params"]),
                                    InvarianceLoss(**L["invariance_loss_params"]), EntropyLoss(), RankLoss(),
                                    weights=L["weights"], start_steps=L["start_steps"])
    tmp = tempfile.mkdtemp()
    trainer = ref.Trainer(segnet=net, criterion=criterion, optimizer=optimizer, aug_transform_epoch=cfg["aug_transform_epoch"],
                          ignore_npoint_thresh=cfg["ignore_npoint_thresh"], exp_base=os.path.join(tmp, "seg_R1"),
                          lr_scheduler=lr_scheduler, bnm_scheduler=bnm_scheduler)
    if waymo:
        train_set = dc.SegScenes(True, dc.N_WAYMO, dc.K_WAYMO, seed=data_seed, ulp=ulp)
        val_set = dc.SegScenes(False, dc.N_WAYMO, dc.K_WAYMO, seed=data_seed, ulp=ulp)
    else:
        train_set, val_set = dc.SegScenes(train=True), dc.SegScenes(train=False)
    if f64:
        train_set, val_set = _AsDouble(train_set), _AsDouble(val_set)
    train_loader = torch.utils.data.DataLoader(train_set, batch_size=cfg["batch_size"], shuffle=False)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=cfg["batch_size"], shuffle=False)

    its = []
    inner = trainer._train_it

    def recording_train_it(it, batch, aug_transform=False):
        loss_dict, segm, mask = inner(it, batch, aug_transform=aug_transform)
        norms, heads = weight_summary(net)
        mom = next(m.momentum for m in net.modules() if isinstance(m, torch.nn.GroupNorm))
        its.append(dict(loss=dict(loss_dict), lr=optimizer.param_groups[0]["lr"], momentum=mom, aug=bool(aug_transform),
                        norms=norms, heads=heads))
        return loss_dict, segm, mask

    trainer._train_it = recording_train_it
    epochs = []
    inner_eval = trainer.eval_epoch

    def recording_eval(loader):
        val_loss, val_avg, ap = inner_eval(loader)
        ious, matched = np.concatenate(ap["Pred_IoU"]), np.concatenate(ap["Pred_Matched"])
        pq, f1, pre, rec = calculate_PQ_F1(ious, matched, np.sum(ap["N_GT_Inst"]))
        epochs.append(dict(val_loss=val_loss, val_avg=dict(val_avg), pq=pq, f1=f1, pre=pre, rec=rec))
        return val_loss, val_avg, ap

    trainer.eval_epoch = recording_eval
    best = trainer.train(cfg["epochs"], train_set, train_loader, val_loader)
    names = sorted(its[0]["loss"])
    out = dict(loss_names=np.array(names), loss=np.array([[r["loss"][k] for k in names] for r in its], np.float64),
               lr=np.array([r["lr"] for r in its]), momentum=np.array([r["momentum"] for r in its]),
               aug=np.array([r["aug"] for r in its]), norms=np.stack([r["norms"] for r in its]),
               heads=np.stack([r["heads"] for r in its]), val_loss=np.array([e["val_loss"] for e in epochs]),
               val_names=np.array(sorted(epochs[0]["val_avg"])),
               val_avg=np.array([[e["val_avg"][k] for k in sorted(e["val_avg"])] for e in epochs]),
               val_pq=np.array([[e["pq"], e["f1"], e["pre"], e["rec"]] for e in epochs], np.float64), best=np.array([best]))
    ck = torch.load(os.path.join(tmp, "seg_R1", "best.pth.tar"))
    out["ckpt_keys"] = np.array(sorted(ck["model_state"].keys()))
    out["ckpt_top"] = np.array(sorted(ck.keys()))
    shutil.rmtree(tmp)
    if waymo:
        out["data_seed"] = np.array([data_seed])
    if write:
        save(("train_seg_waymo_trace" if waymo else "train_seg_trace") + ("_f64" if f64 else ""), **out)
    return out


def gen_train_flow(f64=False):
    """fp32: tries data seeds until the reference's run does not move (< 1e-4, training and validation terms) under one-ulp changes of the coordinates, writes the
    trace of that seed; float64: the truth for the seed the fp32 fixture holds."""
    if f64:
        seed = int(np.load(os.path.join(HERE, "train_flow_trace.npz"))["data_seed"][0])
        return _trace_flow(True, seed, 0, write=True)
    for seed in range(1100, 1140):
        base = _trace_flow(False, seed, 0, write=False)
        dev = 0.0
        for ulp in (1, 2):
            other = _trace_flow(False, seed, ulp, write=False)
            for key in ("loss", "val_avg"):
                dev = max(dev, float(np.max(np.abs(other[key] - base[key]) / np.abs(base[key]).clip(1e-30))))
        print("seed %d: largest change of a loss under one-ulp perturbations %.2e" % (seed, dev), flush=True)
        if dev < 1e-4:
            return _trace_flow(False, seed, 0, write=True)
    raise RuntimeError("no stable seed")


def _trace_flow(f64, seed, ulp, write):
    import train_flow as ref
    from losses.flow_loss_unsup import ChamferLoss, SmoothLoss, UnsupervisedFlowStep3DLoss
    from models.flownet_sapien import FlowStep3D
    from utils.pytorch_util import BNMomentumScheduler
    cfg = dc.FLOW_CFG
    ref.args = types.SimpleNamespace(**cfg)
    net = detgen.fill_module(FlowStep3D(**cfg["flownet"]), 32)
    if f64:
        net = net.double()
    optimizer = torch.optim.Adam(net.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=ref.lr_curve)
    bnm_scheduler = BNMomentumScheduler(net, bn_lambda=ref.bn_curve)
    L = cfg["loss"]
    criterion = UnsupervisedFlowStep3DLoss(ChamferLoss(**L["chamfer_loss_params"]), SmoothLoss(**L["smooth_loss_params"]),
                                           weights=L["weights"], iters_w=L["iters_w"])
    tmp = tempfile.mkdtemp()
    trainer = ref.Trainer(flownet=net, model_iters=cfg["model_iters"], criterion=criterion, optimizer=optimizer,
                          exp_base=os.path.join(tmp, "flow"), lr_scheduler=lr_scheduler, bnm_scheduler=bnm_scheduler)
    wrap = _AsDouble if f64 else (lambda ds: ds)
    train_loader = torch.utils.data.DataLoader(wrap(dc.FlowPairs(True, seed, ulp)), batch_size=cfg["batch_size"], shuffle=False)
    val_loader = torch.utils.data.DataLoader(wrap(dc.FlowPairs(False, seed, ulp)), batch_size=cfg["batch_size"], shuffle=False)
    its, epochs = [], []
    inner = trainer._train_it

    def recording_train_it(it, batch):
        loss_dict = inner(it, batch)
        norms, heads = weight_summary(net)
        bn = next(m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
        its.append(dict(loss=dict(loss_dict), lr=optimizer.param_groups[0]["lr"], momentum=bn.momentum, norms=norms, heads=heads,
                        running_mean=bn.running_mean.detach().clone().numpy()))
        return loss_dict

    trainer._train_it = recording_train_it
    inner_eval = trainer.eval_epoch

    def recording_eval(loader):
        val_loss, val_avg = inner_eval(loader)
        epochs.append(dict(val_loss=val_loss, val_avg=dict(val_avg)))
        return val_loss, val_avg

    trainer.eval_epoch = recording_eval
    best = trainer.train(cfg["epochs"], train_loader, val_loader)
    names = sorted(its[0]["loss"])
    out = dict(loss_names=np.array(names), loss=np.array([[float(r["loss"][k]) for k in names] for r in its], np.float64),
               lr=np.array([r["lr"] for r in its]), momentum=np.array([r["momentum"] for r in its]),
               norms=np.stack([r["norms"] for r in its]), heads=np.stack([r["heads"] for r in its]),
               running_mean=np.stack([r["running_mean"] for r in its]),
               val_loss=np.array([e["val_loss"] for e in epochs]), val_names=np.array(sorted(epochs[0]["val_avg"])),
               val_avg=np.array([[float(e["val_avg"][k]) for k in sorted(e["val_avg"])] for e in epochs]), best=np.array([best]))
    out["data_seed"] = np.array([seed])
    shutil.rmtree(tmp)
    if write:
        save("train_flow_trace_f64" if f64 else "train_flow_trace", **out)
    return out


def tree_bytes(root):
    """{relative path: file bytes as uint8} of every file under root."""
    out = {}
    for d, _, files in os.walk(root):
        for f in sorted(files):
            p = os.path.join(d, f)
            out[os.path.relpath(p, root)] = np.frombuffer(open(p, "rb").read(), np.uint8)
    return out


def gen_flow_store():
    # the reference's `datasets` directory has no __init__.py; an installed package of that name would win the import
    pkg = types.ModuleType("datasets")
    pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    from datasets.dataset_kittisf import KITTISceneFlowDataset
    from datasets.dataset_ogcdr import OGCDynamicRoomDataset
    out = {}
    # ---- pair layout (KITTI-SF): the reference's writer, then its reader
    tmp = tempfile.mkdtemp()
    root = os.path.join(tmp, "kittisf")
    dc.write_kitti_root(root)
    mapping = os.path.join(root, "train.txt")
    test_set = KITTISceneFlowDataset(data_root=root, mapping_path=mapping, downsampled=True, view_sels=[[0, 1], [1, 0]])
    save_dir = os.path.join(root, "flow_preds", "flowstep3d_R1")
    os.makedirs(save_dir)
    pred = torch.from_numpy(dc.kitti_predicted_flows())            # (n_scene * 2, N, 3): frame pairs of one scene adjacent
    bs = 4                                                         # oa_icp.py: batches of whole scenes (batch_size % n_frame == 0)
    for i in range(0, pred.shape[0], bs):
        test_set._save_predflow(pred[i:i + bs], save_root=save_dir, batch_size=bs, n_frame=2, offset=i // bs)
    for k, v in tree_bytes(os.path.join(root, "flow_preds")).items():
        out["kitti_files/" + k] = v
    reader = KITTISceneFlowDataset(data_root=root, mapping_path=mapping, downsampled=True, view_sels=[[0, 1]],
                                   predflow_path="flowstep3d_R1")
    for sid in range(len(reader)):
        pcs, segms, flows, valids = reader[sid]
        out["kitti_read/%d/flows" % sid], out["kitti_read/%d/pcs" % sid], out["kitti_read/%d/segms" % sid] = flows, pcs, segms

    # ---- sequence layout (OGC-DR / SAPIEN): writer + meta file as oa_icp.py:187-191, reader
    root2 = os.path.join(tmp, "ogcdr")
    dc.write_ogcdr_root(root2)
    view_sels = [[0, 1], [1, 0], [1, 2], [2, 1], [2, 3], [3, 2]]
    ts = OGCDynamicRoomDataset(data_root=root2, split="train", view_sels=view_sels)
    save_dir2 = os.path.join(root2, "flow_preds", "flowstep3d_R1")
    os.makedirs(save_dir2)
    with open(save_dir2 + ".json", "w") as f:
        json.dump({"view_sel": view_sels}, f)
    pred2 = torch.from_numpy(dc.ogcdr_predicted_flows())           # (n_scene * 6, N, 3)
    bs2 = 12
    for i in range(0, pred2.shape[0], bs2):
        ts._save_predflow(pred2[i:i + bs2], save_root=save_dir2, batch_size=bs2, n_frame=6, offset=i // bs2)
    for k, v in tree_bytes(os.path.join(root2, "flow_preds")).items():
        out["ogcdr_files/" + k] = v
    reader2 = OGCDynamicRoomDataset(data_root=root2, split="train", view_sels=[[0, 1], [1, 2], [2, 3]], predflow_path="flowstep3d_R1")
    for sid in range(len(reader2)):
        pcs, segms, flows, valids = reader2[sid]
        out["ogcdr_read/%d/flows" % sid], out["ogcdr_read/%d/pcs" % sid] = flows, pcs
    # a meta file that does not cover the requested pairs is an error (datasets/dataset_ogcdr.py:63-65)
    try:
        OGCDynamicRoomDataset(data_root=root2, split="train", view_sels=[[0, 2]], predflow_path="flowstep3d_R1")
        out["ogcdr_uncovered_raises"] = np.array([0])
    except ValueError:
        out["ogcdr_uncovered_raises"] = np.array([1])

    # ---- the refinement round as the reference runs it: `oa_icp.py cfg --split train --round 1 --save` on the KITTI-SF root
    from models.segnet_kitti import MaskFormer3D
    cfg = dict(dc.ICP_CFG)
    cfg["data"] = dict(cfg["data"], root=root)
    cfg["save_path"] = os.path.join(tmp, "ckpt", "seg")
    os.makedirs(cfg["save_path"] + "_R1")
    net = detgen.fill_module(MaskFormer3D(**cfg["segnet"]), 33)
    torch.save({"model_state": net.state_dict()}, os.path.join(cfg["save_path"] + "_R1", "best.pth.tar"))
    # round-1 input flows: <root>/flow_preds/flowstep3d/<id>/flow{1,2}.npy
    dc.write_kitti_input_flows(root)
    cfg_path = os.path.join(tmp, "icp.yaml")
    import yaml
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    shutil.rmtree(save_dir)                                         # the script writes flowstep3d_R1 itself
    cwd, argv = os.getcwd(), sys.argv
    os.makedirs(os.path.join(tmp, "run", "data_prepare", "kittisf", "splits"))
    shutil.copy(mapping, os.path.join(tmp, "run", "data_prepare", "kittisf", "splits", "train.txt"))
    os.chdir(os.path.join(tmp, "run"))                              # the script reads its split file relative to the cwd
    sys.argv = ["oa_icp.py", cfg_path, "--split", "train", "--round", "1", "--test_batch_size", "4", "--save"]
    real_loader = torch.utils.data.DataLoader

    def loader(ds, **kw):                                           # no worker processes, no pinned memory on this box
        kw.update(num_workers=0, pin_memory=False)
        return real_loader(ds, **kw)

    torch.utils.data.DataLoader = loader
    buf = io.StringIO()
    stdout = sys.stdout
    try:
        sys.stdout = buf
        runpy.run_path(os.path.join(REF, "oa_icp.py"), run_name="__main__")
    finally:
        sys.stdout = stdout
        torch.utils.data.DataLoader = real_loader
        os.chdir(cwd)
        sys.argv = argv
    for k, v in tree_bytes(os.path.join(root, "flow_preds", "flowstep3d_R1")).items():
        out["icp_files/" + k] = v
    report = [l for l in buf.getvalue().splitlines() if "flow:" in l]
    out["icp_report"] = np.array(report)
    print("\n".join(report))
    save("flow_store", **out)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference tree is only present in the build container"
    which = sys.argv[1:] or ["seg", "flow", "store"]
    orc.build()
    if "seg64" in which or "flow64" in which or "waymo64" in which:
        # the float64 truth of the same runs (make_truth_f64.py's shims: indices decided in fp32, floats in fp64) — in a
        # process of its own: the shims replace the native module and the allocation factories
        from make_truth_f64 import install_shims_f64
        install_shims_f64()
        torch.set_default_dtype(torch.float64)   # the losses create their constants with the default dtype
        _einsum = torch.einsum                   # ... except the permutation matrices (float32 by name, seg_loss_unsup.py:239)
        torch.einsum = lambda eq, *ops: _einsum(eq, *[o.double() if torch.is_tensor(o) and o.is_floating_point() else o for o in ops])
        sys.modules["tensorboardX"].SummaryWriter = _Writer
        if "seg64" in which:
            gen_train_seg(f64=True)
        if "waymo64" in which:
            gen_train_seg_waymo(f64=True)
        if "flow64" in which:
            gen_train_flow(f64=True)
        sys.exit(0)
    shims()
    if "seg" in which:
        gen_train_seg()
    if "waymo" in which:
        gen_train_seg_waymo()
    if "flow" in which:
        gen_train_flow()
    if "store" in which:
        gen_flow_store()
