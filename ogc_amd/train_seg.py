"""Unsupervised OGC segmentation training driver on the MI355X operators — the caller of the hot path
(counterpart of the reference's train_seg.py:19-352, written against this repo's layers).

    python -m ogc_amd.train_seg config.yaml --round 1 [--synthetic N_SCENES] [--max-iters K]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ogc_amd.train_seg config.yaml ...

What is kept from the reference so that its configs, schedules and checkpoints carry over:
  * the YAML schema (config/seg/*/*.yaml): dataset, save_path, random_seed, aug_transform_epoch, epochs, batch_size,
    lr, lr_decay, lr_clip, bn_momentum, bn_decay, weight_decay, decay_step, segnet{...}, loss{...};
  * Adam + LambdaLR(lr_curve) + norm-momentum schedule (train_seg.py:230-245), both driven by samples seen
    (`it * batch_size`; under DDP the GLOBAL batch, so the schedule is independent of the number of GPUs);
  * augmented views (and the invariance loss) switch on after `aug_transform_epoch` epochs (train_seg.py:151-154);
  * loss weights gated by `it * batch` (train_seg.py:70), NaN-gradient skip (train_seg.py:81-83, in train_step);
  * checkpoints {'model_state': state_dict} as current.pth.tar / best.pth.tar in `<save_path>_R<round>`, the initial
    weights saved as both (utils/pytorch_util.py:84-99, train_seg.py:137-140).
Out of scope here: the dataset readers and tensorboard.  Validation reports AP / PQ / F1 / Pre / Rec
(ogc_amd/metrics/seg_metric.py) next to the loss; the per-training-step metrics of the reference's loop are not computed.  Without a dataset the
driver trains on seeded synthetic scenes with the loaders' sample contract (ogc_amd/utils/synthetic.py).
"""
import argparse
import importlib
import json
import os
import time

import torch
import torch.distributed as dist
import yaml

from .train_step import build_criterion, make_optimizer, train_step
from .utils.pytorch_util import (NORM_LAYERS, AverageMeter, BNMomentumScheduler, LambdaLR, checkpoint_state,
                                 save_checkpoint)
from .utils.synthetic import make_scene_batch

SEGNETS = {"sapien": "segnet_sapien", "ogcdr": "segnet_ogcdr", "kittisf": "segnet_kitti", "waymo": "segnet_kitti"}
# ONE training set on every rank (DistributedSampler deals its scenes out): the flow store of oa_icp_round is keyed by
# scene index, so all ranks — and the refinement round — must mean the same scene by the same index
TRAIN_SEED = 1000
# a skipped step (NaN gradients, train_seg.py:81-83) is counted and logged per epoch; this many in a row end the run
MAX_CONSECUTIVE_SKIPS = 50


class SyntheticScenes(torch.utils.data.Dataset):
    """`n_scene` seeded scenes; item = (pcs (t,N,3), segms (t,N), flows (t,N,3), valids (t,N)), t = 2 or 4 views."""

    def __init__(self, n_scene, n_point, n_object, outdoor, seed=0, predflow_dir=None, aug_transform_args=None):
        self.n_scene, self.n_point, self.n_object, self.outdoor, self.seed = n_scene, n_point, n_object, outdoor, seed
        self.aug_transform = False
        # flows predicted for the two frames of scene i, `<predflow_dir>/<i as %06d>/flow{1,2}.npy` (the layout
        # datasets/dataset_kittisf.py:125-137 writes and :99-104 reads); ground-truth flows when absent
        self.predflow_dir = predflow_dir
        # `data.aug_transform_args` of the reference's YAMLs: when given, augmented samples are built the way its
        # datasets build them (datasets/dataset_kittisf.py:113-117): utils.data_util.augment_transform draws TWO
        # similarity transforms of the frame pair -> 4 views, none of them the untouched pair
        self.aug_transform_args = aug_transform_args

    def __len__(self):
        return self.n_scene

    def stored_flows(self, i):
        """The predicted flows of scene i as (2, N, 3), or None when the store has none for it."""
        if self.predflow_dir is None:
            return None
        import numpy as np
        paths = [os.path.join(self.predflow_dir, "%06d" % i, "flow%d.npy" % v) for v in (1, 2)]
        if not all(os.path.exists(p) for p in paths):
            return None
        return torch.stack([torch.from_numpy(np.load(p)).float() for p in paths])

    def __getitem__(self, i):
        # predicted flows first, augmentation second (datasets/dataset_kittisf.py:91-117): the augmented views carry
        # the transformed PREDICTED flows, never the synthetic ground truth
        by_reference_recipe = self.aug_transform and self.aug_transform_args is not None
        stored = self.stored_flows(i)
        pcs, segms, flows, valids = make_scene_batch(1, self.n_point, self.n_object, seed=self.seed + i,
                                                     outdoor=self.outdoor, aug=self.aug_transform and not by_reference_recipe,
                                                     flows=None if stored is None else stored[None])
        pcs, segms, flows, valids = pcs[0], segms[0], flows[0], valids[0]
        if by_reference_recipe:
            import numpy as np
            from .utils.data_util import augment_transform
            a, f = augment_transform(pcs.numpy().astype(np.float64), flows.numpy().astype(np.float64),
                                     self.aug_transform_args, rng=np.random.RandomState(self.seed + i))
            pcs, flows = torch.from_numpy(a.astype(np.float32)), torch.from_numpy(f.astype(np.float32))
            segms, valids = torch.cat([segms, segms]), torch.cat([valids, valids])
        return pcs, segms, flows, valids


class SyntheticSequenceScenes(torch.utils.data.Dataset):
    """Scenes of `n_frame` consecutive frames, sampled as frame pairs the way the reference's SAPIEN / OGC-DR loaders do
    (datasets/dataset_ogcdr.py:77-79,100-112): `len = n_scene * len(view_sels)`, item = the two frames of one pair with
    the flow of each towards the other; predicted flows come from the sequence layout of utils/flow_store.py."""

    def __init__(self, n_scene, n_point, n_object, view_sels, n_frame=4, seed=0, predflow_dir=None, aug_transform_args=None):
        from .utils import flow_store
        self.n_scene, self.n_point, self.n_object, self.n_frame, self.seed = n_scene, n_point, n_object, n_frame, seed
        self.view_sels = [list(v) for v in view_sels]
        self.aug_transform, self.aug_transform_args = False, aug_transform_args
        self.predflow_dir, self.meta = predflow_dir, None
        if predflow_dir is not None:
            self.meta = flow_store.read_meta(predflow_dir)
            if self.meta is None or any(v not in self.meta for v in self.view_sels):
                raise ValueError("Flow predictions cannot cover the specified view selections!")

    def __len__(self):
        return self.n_scene * len(self.view_sels)

    def scene(self, i):
        from .utils.synthetic import make_sequence
        return make_sequence(self.n_frame, self.n_point, self.n_object, seed=self.seed + i, outdoor=False)

    @staticmethod
    def pair_flows(seq_flows, a, b):
        """Ground-truth flows of the ordered pair (a, b) of adjacent frames: [a->b on frame a, b->a on frame b]."""
        assert abs(a - b) == 1, "synthetic sequences carry flows between adjacent frames only"
        lo = min(a, b)
        fwd, bwd = seq_flows[lo, 0], seq_flows[lo, 1]
        return [fwd, bwd] if a < b else [bwd, fwd]

    def __getitem__(self, sid):
        import numpy as np
        from .utils import flow_store
        i, (a, b) = sid // len(self.view_sels), self.view_sels[sid % len(self.view_sels)]
        pc, segm, seq_flows = self.scene(i)
        pcs, segms = torch.stack([pc[a], pc[b]]), torch.stack([segm[a], segm[b]])
        flows = torch.stack(self.pair_flows(seq_flows, a, b))
        if self.predflow_dir is not None:
            stored = flow_store.load_pair(self.predflow_dir, "%06d" % i, (a, b), self.meta)
            if stored is not None:
                flows = torch.stack([torch.from_numpy(np.asarray(f)).float() for f in stored])
        valids = torch.ones_like(segms, dtype=torch.bool)
        if self.aug_transform and self.aug_transform_args is not None:
            from .utils.data_util import augment_transform
            p, f = augment_transform(pcs.numpy().astype(np.float64), flows.numpy().astype(np.float64),
                                     self.aug_transform_args, rng=np.random.RandomState(self.seed + sid))
            pcs, flows = torch.from_numpy(p.astype(np.float32)), torch.from_numpy(f.astype(np.float32))
            segms, valids = torch.cat([segms, segms]), torch.cat([valids, valids])
        return pcs, segms, flows, valids


def schedule_factor(cfg, samples_seen):
    """lr multiplier (train_seg.py:230-234)."""
    return max(cfg["lr_decay"] ** int(samples_seen / cfg["decay_step"]), cfg["lr_clip"] / cfg["lr"])


def norm_momentum(cfg, samples_seen):
    """momentum of the norm layers (train_seg.py:237-245)."""
    if cfg["decay_step"] == -1:
        return cfg["bn_momentum"]
    return max(cfg["bn_momentum"] * cfg["bn_decay"] ** int(samples_seen / cfg["decay_step"]), 1e-2)


class Trainer(object):
    """The reference's Trainer (train_seg.py:19-226) on this repo's step: same constructor arguments, `_train_it(it, batch,
    aug_transform)`, `eval_epoch(loader)` and `train(n_epochs, train_set, train_loader, test_loader)` with the same schedule
    calls, loss gating, NaN rule, augmentation switch, best-checkpoint rule and checkpoint files.  Keyword-only extras (not in
    the reference): the wrapped model of a data-parallel run, the device the batches go to, the one-frame loss of the Waymo
    trainer, HIP-graph replay, a cap on the number of iterations and a per-iteration callback for tests.

    What the step does differently from the reference's is invisible here: the scalars of step i are read while step i + 1 is
    already queued (`_train_it(..., sync=False)`), and the coordinate-only work of the next batch runs a step ahead."""

    def __init__(self, segnet, criterion, optimizer, aug_transform_epoch, ignore_npoint_thresh, exp_base, lr_scheduler=None,
                 bnm_scheduler=None, *, model=None, device=None, world=1, rank=0, single_frame=False, hip_graph=False,
                 max_iters=0, log=print, on_iteration=None, loss_start_steps=(0, 0, 0)):
        self.segnet = segnet
        self.model = model if model is not None else segnet   # what is called: the net, or its data-parallel wrapper
        self.criterion = criterion
        self.optimizer = optimizer
        self.aug_transform_epoch = aug_transform_epoch
        self.ignore_npoint_thresh = ignore_npoint_thresh
        self.lr_scheduler = lr_scheduler
        self.bnm_scheduler = bnm_scheduler
        self.exp_base = exp_base
        self.device = device if device is not None else next(segnet.parameters()).device
        self.world, self.rank, self.single_frame = world, rank, single_frame
        self.hip_graph, self.max_iters, self.log, self.on_iteration = hip_graph, max_iters, log, on_iteration
        self.loss_start_steps = tuple(loss_start_steps)
        if rank == 0:
            os.makedirs(exp_base, exist_ok=True)
        self.checkpoint_name, self.best_name = "current", "best"
        self.cur_epoch = 0
        self._skipped_in_a_row = 0
        self._graphed, self._graphed_key, self._graph_holds = None, None, None

    # ---- one optimisation step (train_seg.py:47-86)
    def _train_it(self, it, batch, aug_transform=False, sync=True, prefetched=None, next_batch=None):
        """lr / norm-momentum schedules stepped with the iteration number, then the step itself (ogc_amd/train_step.py: forward
        over the views flattened into the batch, loss with step_w=True and it * b, backward, NaN rule, Adam).  Returns
        (loss_dict, stepped) — or, with sync=False, the PendingStep whose result() gives that pair later."""
        if self.lr_scheduler is not None:
            self.lr_scheduler.step(it)
        if self.bnm_scheduler is not None:
            self.bnm_scheduler.step(it)
        # `it * world`: train_step multiplies by the local batch size, so the loss gates see the samples of ALL ranks
        return train_step(self.model, self.criterion, self.optimizer, batch, it * self.world, aug_transform, sync=sync,
                          prefetched=prefetched, next_batch=next_batch)

    def _graphed_it(self, it, batch, upcoming, aug_transform):
        """The same step replayed as one HIP graph (graph_step.py); re-captured when anything the capture froze changes:
        learning rate, norm momentum, which loss terms are active, augmentation, the batch shape."""
        if self.lr_scheduler is not None:
            self.lr_scheduler.step(it)
        if self.bnm_scheduler is not None:
            self.bnm_scheduler.step(it)
        mom = next((m.momentum for m in self.segnet.modules() if isinstance(m, NORM_LAYERS)), None)
        b = batch[1].size(0)
        gates = tuple(it * self.world * b >= st for st in self.loss_start_steps)
        key = (self.optimizer.param_groups[0]["lr"], mom, gates, aug_transform, tuple(batch[0].shape))
        if self._graphed is None or key != self._graphed_key:
            from .graph_step import GraphedTrainStep
            if self._graphed is None or key[3:] != self._graphed_key[3:]:
                self._graphed = GraphedTrainStep(self.model, self.criterion, self.optimizer, batch, it * self.world, aug_transform)
            else:
                self._graphed.load(batch)
                self._graphed.recapture(it * self.world)
            self._graphed_key, self._graph_holds = key, batch
        elif self._graph_holds is not batch:
            self._graphed.load(batch)
        pending = self._graphed.step(upcoming if upcoming is not None else batch)
        self._graph_holds = upcoming
        return pending

    # ---- validation (train_seg.py:88-133)
    def eval_epoch(self, d_loader):
        """(validation loss, mean loss_dict, {'Pred_IoU', 'Pred_Matched', 'Confidence', 'N_GT_Inst'} lists).  The loss is the
        reference's number: the sum over the n batches divided by n + 1 (its counter starts at 1.0, train_seg.py:93-94,:114) —
        kept, so that logged curves and the best-checkpoint choice of the two trainers agree number for number."""
        from .metrics.seg_metric import accumulate_eval_results
        self.model.eval()
        eval_meter = AverageMeter()
        total_loss, count = 0.0, 1.0
        ap_eval_meter = {"Pred_IoU": [], "Pred_Matched": [], "Confidence": [], "N_GT_Inst": []}
        with torch.no_grad():
            for pcs, segms, flows, _ in d_loader:
                if self.single_frame:
                    pcs, segms, flows = pcs[:, ::2].contiguous(), segms[:, ::2].contiguous(), flows[:, ::2].contiguous()
                pcs, flows = pcs.to(self.device), flows.to(self.device)
                b, t, n = segms.shape
                flat = pcs.view(b * t, n, -1).contiguous()
                masks = self.model(flat, flat).view(b, t, n, -1)
                loss, loss_dict = self.criterion([pcs[:, i].contiguous() for i in range(t)],
                                                 [masks[:, i].contiguous() for i in range(t)],
                                                 [flows[:, i].contiguous() for i in range(t)], step_w=False)
                total_loss += float(loss)
                count += 1
                eval_meter.append_loss(loss_dict)
                iou, matched, conf, n_gt = accumulate_eval_results(segms[:, 0].to(self.device), masks[:, 0],
                                                                   self.ignore_npoint_thresh)
                ap_eval_meter["Pred_IoU"].append(iou)
                ap_eval_meter["Pred_Matched"].append(matched)
                ap_eval_meter["Confidence"].append(conf)
                ap_eval_meter["N_GT_Inst"].append(n_gt)
        return total_loss / count, eval_meter.get_mean_loss_dict(), ap_eval_meter

    def _save(self, is_best):
        if self.rank == 0:
            save_checkpoint(checkpoint_state(self.segnet), is_best, filename=os.path.join(self.exp_base, self.checkpoint_name),
                            bestname=os.path.join(self.exp_base, self.best_name))

    def _device_batches(self, loader):
        for cpu_batch in loader:
            if self.single_frame:  # Waymo: only backward flow exists, every other view is used (train_seg_waymo.py:59)
                cpu_batch = tuple(x[:, ::2].contiguous() for x in cpu_batch)
            yield tuple(x.to(self.device, non_blocking=True) for x in cpu_batch)

    # ---- the epoch loop (train_seg.py:136-226)
    def train(self, n_epochs, train_set, train_loader, test_loader=None):
        import numpy as np
        from .metrics.seg_metric import calculate_AP, calculate_PQ_F1
        it, best_loss, aug_transform = 0, 1e10, False
        self._save(True)  # the initial weights as current and best (train_seg.py:137-140)
        sampler = getattr(train_loader, "sampler", None)
        for epoch in range(1, n_epochs + 1):
            self.cur_epoch = epoch
            # augmented views (and with them the invariance loss) from the epoch after aug_transform_epoch (train_seg.py:151-154)
            if epoch == self.aug_transform_epoch + 1:
                aug_transform, train_set.aug_transform, best_loss = True, True, 1e10
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            train_meter, t0, in_flight, skipped = AverageMeter(), time.time(), None, 0

            def account(pending, at):
                nonlocal skipped
                if pending is None:
                    return
                loss_dict, stepped = pending.result() if hasattr(pending, "result") else pending
                skipped += 0 if stepped else 1
                self._skipped_in_a_row = 0 if stepped else self._skipped_in_a_row + 1
                if self._skipped_in_a_row >= MAX_CONSECUTIVE_SKIPS:
                    raise RuntimeError("%d optimisation steps in a row were skipped (NaN gradients or a failing backward pass): "
                                       "the weights are not being updated" % self._skipped_in_a_row)
                train_meter.append_loss(loss_dict)
                if self.on_iteration is not None:
                    self.on_iteration(at, loss_dict, stepped)

            self.model.train()
            stream_of_batches = self._device_batches(train_loader)
            batch, pre = next(stream_of_batches, None), None
            while batch is not None:
                upcoming = next(stream_of_batches, None)  # one batch ahead: its geometry is queued during this step
                if self.hip_graph:
                    pending = self._graphed_it(it, batch, upcoming, aug_transform)
                else:
                    pending = self._train_it(it, batch, aug_transform, sync=False, prefetched=pre, next_batch=upcoming)
                    pre = pending.prefetched
                # the scalars of step i are read while step i + 1 is already queued: the host never waits inside a step
                account(in_flight, it - 1)
                in_flight, batch = pending, upcoming
                it += 1
                if self.max_iters and it >= self.max_iters:
                    break
            account(in_flight, it - 1)
            record = {"epoch": epoch, "it": it, "lr": self.optimizer.param_groups[0]["lr"], "aug": aug_transform,
                      "train": {k: round(v, 5) for k, v in train_meter.get_mean_loss_dict().items()}, "skipped_steps": skipped}
            if test_loader is not None:
                val_loss, val_avg, ap = self.eval_epoch(test_loader)
                if self.world > 1:
                    t = torch.tensor([val_loss], device=self.device)
                    dist.all_reduce(t)
                    val_loss = float(t) / self.world
                ious, matched, confs = (np.concatenate(ap[k]) for k in ("Pred_IoU", "Pred_Matched", "Confidence"))
                n_gt = int(np.sum(ap["N_GT_Inst"]))
                pq, f1, pre_, rec = calculate_PQ_F1(ious, matched, n_gt)
                is_best = val_loss < best_loss
                best_loss = min(best_loss, val_loss)
                self._save(is_best)
                record.update(val_loss=round(val_loss, 5), val_terms={k: round(v, 5) for k, v in val_avg.items()}, is_best=is_best,
                              val={"AP": round(float(calculate_AP(matched, confs, n_gt)), 4), "PQ": round(float(pq), 4),
                                   "F1": round(float(f1), 4), "Pre": round(float(pre_), 4), "Rec": round(float(rec), 4)})
            record["sec"] = round(time.time() - t0, 2)
            if self.rank == 0:
                self.log(json.dumps(record))
            if self.max_iters and it >= self.max_iters:
                break
        return best_loss


def build_segnet(cfg):
    """The MaskFormer3D of the config's dataset, built from its `segnet` block (train_seg.py:302-307)."""
    seg = cfg["segnet"]
    MaskFormer3D = importlib.import_module("ogc_amd.models." + SEGNETS[cfg["dataset"]]).MaskFormer3D
    return MaskFormer3D(n_slot=seg["n_slot"], n_point=seg["n_point"], use_xyz=seg["use_xyz"],
                        n_transformer_layer=seg["n_transformer_layer"],
                        transformer_embed_dim=seg["transformer_embed_dim"],
                        transformer_input_pos_enc=seg["transformer_input_pos_enc"])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--round", type=int, default=0)
    ap.add_argument("--synthetic", type=int, default=64, help="number of synthetic training scenes")
    ap.add_argument("--max-iters", type=int, default=0, help="stop after this many optimisation steps (0 = all epochs)")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--ddp", action="store_true",
                    help="average gradients with torch's DistributedDataParallel instead of one flat collective per step")
    ap.add_argument("--hip-graph", action="store_true",
                    help="replay the training step as one HIP graph (ogc_amd/graph_step.py; one GPU): removes the launch "
                         "thread's 10-12 ms per step, which is the bound for clouds of fewer than ~8192 points")
    ap.add_argument("--flow-root", default=None,
                    help="read predicted flows from <flow-root>/flow_preds/<predflow_path>[_R<round-1>] (train_seg.py:277-280)")
    ap.add_argument("--data-root", default=None,
                    help="train on scenes in the reference's directory layout instead of synthetic ones (kittisf, ogcdr: "
                         "ogc_amd/datasets.py); predicted flows are read from <data-root>/flow_preds/<predflow_path>[_R<round-1>]")
    ap.add_argument("--train-mapping", default=None, help="kittisf: split file of the training scenes (default: data.train_mapping)")
    ap.add_argument("--val-mapping", default=None, help="kittisf: split file of the validation scenes (default: data.val_mapping)")
    ap.add_argument("--frames", type=int, default=2,
                    help="frames per synthetic scene: 2 = frame pairs (KITTI-style), 4 = SAPIEN / OGC-DR style sequences "
                         "sampled as the pairs [[0,1],[1,2],[2,3]] (train_seg.py:295), flows in the sequence layout")
    args = ap.parse_args(argv)
    with open(args.config) as f:
        cfg = yaml.safe_load(f)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    distributed = world > 1
    if args.device == "cuda":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device(args.device)
    if distributed:
        dist.init_process_group("nccl" if device.type == "cuda" else "gloo")

    torch.manual_seed(cfg["random_seed"])
    seg = cfg["segnet"]
    if device.type == "cuda" and cfg.get("matmul_precision", "fp32") != "fp32":
        # `matmul_precision: bf16` (not a key of the reference's YAMLs): bf16 operands, fp32 accumulation in the 1x1
        # convolutions — what running the reference under torch.autocast(bfloat16) does to its Conv2d layers
        from .pointnet2 import pointnet2 as _api
        _api._native.set_matmul_precision(cfg["matmul_precision"])
    net = build_segnet(cfg).to(device)
    # one gradient collective per step (utils/dist_util.py); `--ddp` keeps torch's DistributedDataParallel
    if distributed and getattr(args, "ddp", False):
        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[device.index] if device.type == "cuda" else None)
    elif distributed:
        from .utils.dist_util import FlatDataParallel
        model = FlatDataParallel(net)
    else:
        model = net

    outdoor = cfg["dataset"] in ("kittisf", "waymo")
    predflow_dir = None
    if args.flow_root is not None:
        name = cfg.get("predflow_path", "flowstep3d")
        predflow_dir = os.path.join(args.flow_root, "flow_preds", name if args.round <= 1 else "%s_R%d" % (name, args.round - 1))
    aug_args = (cfg.get("data") or {}).get("aug_transform_args") or None
    if args.data_root is not None:
        # scenes (and, from round 2 on, the refined flows of the round before) from a directory tree in the reference's layout
        # (ogc_amd/datasets.py; train_seg.py:270-315): KITTI-SF needs the two split files, OGC-DR has them in the tree
        from .datasets import KITTISceneFlowDataset, OGCDynamicRoomDataset, SapienDataset
        data = cfg.get("data") or {}
        name = cfg.get("predflow_path", "flowstep3d")
        predflow = name if args.round <= 1 else "%s_R%d" % (name, args.round - 1)
        if cfg["dataset"] == "kittisf":
            common = dict(data_root=args.data_root, downsampled=True, view_sels=[[0, 1]], predflow_path=predflow,
                          decentralize=data.get("decentralize", False))
            train_set = KITTISceneFlowDataset(mapping_path=args.train_mapping or data["train_mapping"], aug_transform_args=aug_args,
                                              **common)
            val_set = KITTISceneFlowDataset(mapping_path=args.val_mapping or data["val_mapping"], **common)
        elif cfg["dataset"] == "ogcdr":
            from .utils.flow_store import TRAIN_PAIRS
            common = dict(data_root=args.data_root, view_sels=TRAIN_PAIRS, predflow_path=predflow,
                          decentralize=data.get("decentralize", False))
            train_set = OGCDynamicRoomDataset(split="train", aug_transform_args=aug_args, **common)
            val_set = OGCDynamicRoomDataset(split="val", **common)
        elif cfg["dataset"] == "sapien":
            from .utils.flow_store import TRAIN_PAIRS
            sub = os.path.join(args.data_root, "mbs-shapepart")                                   # train_seg.py:296-297
            common = dict(data_root=sub if os.path.isdir(sub) else args.data_root, view_sels=TRAIN_PAIRS, predflow_path=predflow,
                          decentralize=data.get("decentralize", False))
            train_set = SapienDataset(split="train", aug_transform_args=aug_args, **common)
            val_set = SapienDataset(split="val", **common)
        else:
            raise KeyError("no reader for dataset %r (ogc_amd/datasets.py covers kittisf, ogcdr and sapien)" % cfg["dataset"])
    elif args.frames > 2:
        from .utils.flow_store import TRAIN_PAIRS
        assert not outdoor and args.frames == 4, "sequences are the SAPIEN / OGC-DR sample format (4 frames)"
        train_set = SyntheticSequenceScenes(args.synthetic, seg["n_point"], seg["n_slot"], TRAIN_PAIRS, args.frames,
                                            seed=TRAIN_SEED, predflow_dir=predflow_dir, aug_transform_args=aug_args)
    else:
        train_set = SyntheticScenes(args.synthetic, seg["n_point"], seg["n_slot"], outdoor, seed=TRAIN_SEED,
                                    predflow_dir=predflow_dir, aug_transform_args=aug_args)
    if args.data_root is None:
        val_set = SyntheticScenes(max(args.synthetic // 8, cfg["batch_size"]), seg["n_point"], seg["n_slot"], outdoor, seed=7)
    sampler = torch.utils.data.distributed.DistributedSampler(train_set) if distributed else None
    train_loader = torch.utils.data.DataLoader(train_set, batch_size=cfg["batch_size"], shuffle=sampler is None,
                                               sampler=sampler, drop_last=True)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=cfg["batch_size"], shuffle=False)

    # One replayed HIP graph per step also under data parallelism (round 6): the flat gradient all-reduce of FlatDataParallel is
    # captured with the step (RCCL takes part in a capture: tools/graph_rccl.py, tests/test_graph_step_gpu.py), so every rank's
    # launch thread costs ~1 ms per step — what makes eight ranks on one host launchable.  Not with torch's
    # DistributedDataParallel (--ddp): its per-parameter hooks and bucket streams are not part of this capture.
    use_graph = args.hip_graph and device.type == "cuda" and not (distributed and getattr(args, "ddp", False))
    optimizer = make_optimizer(net.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"], capturable=use_graph)
    # Waymo: only backward flow exists, the trainer keeps every other view and uses the one-frame loss
    # (train_seg_waymo.py:59, :244-334)
    single_frame = cfg["dataset"] == "waymo"
    criterion = build_criterion(cfg["loss"], single_frame=single_frame)
    # schedules driven by samples seen (train_seg.py:230-245); under data parallelism the GLOBAL batch, so that they do not
    # depend on the number of GPUs
    global_batch = cfg["batch_size"] * world
    lr_scheduler = LambdaLR(optimizer, lr_lambda=lambda it: schedule_factor(cfg, it * global_batch))
    bnm_scheduler = BNMomentumScheduler(net, bn_lambda=lambda it: norm_momentum(cfg, it * global_batch))
    trainer = Trainer(net, criterion, optimizer, aug_transform_epoch=cfg["aug_transform_epoch"],
                      ignore_npoint_thresh=cfg.get("ignore_npoint_thresh", 0), exp_base=cfg["save_path"] + "_R%d" % args.round,
                      lr_scheduler=lr_scheduler, bnm_scheduler=bnm_scheduler, model=model, device=device, world=world, rank=rank,
                      single_frame=single_frame, hip_graph=use_graph, max_iters=args.max_iters,
                      log=lambda line: print(line, flush=True), loss_start_steps=cfg["loss"].get("start_steps", [0, 0, 0]))
    best = trainer.train(cfg["epochs"], train_set, train_loader, val_loader)
    if distributed:
        dist.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
