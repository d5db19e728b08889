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
import math
import os
import shutil
import time

import torch
import torch.distributed as dist
import yaml

from .train_step import build_criterion, make_optimizer, train_step
from .utils.synthetic import make_scene_batch

SEGNETS = {"sapien": "segnet_sapien", "ogcdr": "segnet_ogcdr", "kittisf": "segnet_kitti", "waymo": "segnet_kitti"}
# ONE training set on every rank (DistributedSampler deals its scenes out): the flow store of oa_icp_round is keyed by
# scene index, so all ranks — and the refinement round — must mean the same scene by the same index
TRAIN_SEED = 1000
NORM_LAYERS = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d, torch.nn.InstanceNorm1d,
               torch.nn.InstanceNorm2d, torch.nn.InstanceNorm3d, torch.nn.GroupNorm)


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


def save_checkpoint(net, exp_base, is_best):
    state = {"model_state": net.state_dict()}
    cur = os.path.join(exp_base, "current.pth.tar")
    torch.save(state, cur)
    if is_best:
        shutil.copyfile(cur, os.path.join(exp_base, "best.pth.tar"))


def evaluate(model, criterion, loader, device, single_frame=False, ignore_npoint_thresh=0):
    """Mean validation loss and the segmentation metrics of the first frame (train_seg.py:88-133: AP, PQ, F1, Pre, Rec)."""
    import numpy as np
    from .metrics.seg_metric import accumulate_eval_results, calculate_AP, calculate_PQ_F1
    model.eval()
    total, count = 0.0, 0
    ious, matched, confs, n_gt = [], [], [], 0
    with torch.no_grad():
        for pcs, segms, flows, valids in loader:
            if single_frame:
                pcs, segms, flows = pcs[:, ::2].contiguous(), segms[:, ::2].contiguous(), flows[:, ::2].contiguous()
            pcs, flows = pcs.to(device), flows.to(device)
            b, t, n = segms.shape
            flat = pcs.view(b * t, n, -1).contiguous()
            masks = model(flat, flat).view(b, t, n, -1)
            loss, _ = criterion([pcs[:, i].contiguous() for i in range(t)], [masks[:, i].contiguous() for i in range(t)],
                                [flows[:, i].contiguous() for i in range(t)], step_w=False)
            total += float(loss)
            count += 1
            a, m, c, g = accumulate_eval_results(segms[:, 0].to(device), masks[:, 0], ignore_npoint_thresh)
            ious.append(a); matched.append(m); confs.append(c); n_gt += g
    ious, matched, confs = np.concatenate(ious), np.concatenate(matched), np.concatenate(confs)
    pq, f1, pre, rec = calculate_PQ_F1(ious, matched, n_gt)
    metrics = {"AP": calculate_AP(matched, confs, n_gt), "PQ": float(pq), "F1": float(f1), "Pre": float(pre), "Rec": float(rec)}
    return total / max(count, 1), metrics


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
    if args.frames > 2:
        from .utils.flow_store import TRAIN_PAIRS
        assert not outdoor and args.frames == 4, "sequences are the SAPIEN / OGC-DR sample format (4 frames)"
        train_set = SyntheticSequenceScenes(args.synthetic, seg["n_point"], seg["n_slot"], TRAIN_PAIRS, args.frames,
                                            seed=TRAIN_SEED, predflow_dir=predflow_dir, aug_transform_args=aug_args)
    else:
        train_set = SyntheticScenes(args.synthetic, seg["n_point"], seg["n_slot"], outdoor, seed=TRAIN_SEED,
                                    predflow_dir=predflow_dir, aug_transform_args=aug_args)
    val_set = SyntheticScenes(max(args.synthetic // 8, cfg["batch_size"]), seg["n_point"], seg["n_slot"], outdoor, seed=7)
    sampler = torch.utils.data.distributed.DistributedSampler(train_set) if distributed else None
    train_loader = torch.utils.data.DataLoader(train_set, batch_size=cfg["batch_size"], shuffle=sampler is None,
                                               sampler=sampler, drop_last=True)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=cfg["batch_size"], shuffle=False)

    use_graph = args.hip_graph and device.type == "cuda" and not distributed
    optimizer = make_optimizer(net.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"], capturable=use_graph)
    graphed, graphed_key, graph_holds = None, None, None
    # Waymo: only backward flow exists, the trainer keeps every other view and uses the one-frame loss
    # (train_seg_waymo.py:59, :244-334)
    single_frame = cfg["dataset"] == "waymo"
    criterion = build_criterion(cfg["loss"], single_frame=single_frame)
    exp_base = cfg["save_path"] + "_R%d" % args.round
    if rank == 0:
        os.makedirs(exp_base, exist_ok=True)
        save_checkpoint(net, exp_base, True)  # initial weights as current and best

    global_batch = cfg["batch_size"] * world
    it, best = 0, 1e10
    aug = False
    for epoch in range(1, cfg["epochs"] + 1):
        if epoch == cfg["aug_transform_epoch"] + 1:
            aug, train_set.aug_transform, best = True, True, 1e10
        if sampler is not None:
            sampler.set_epoch(epoch)
        sums, t0, in_flight = {}, time.time(), None

        def account(pending):
            if pending is not None:
                for k, v in pending.result()[0].items():
                    if math.isfinite(v):  # the reference's AverageMeter drops NaN values
                        sums[k] = sums.get(k, 0.0) + v

        def device_batches():
            for cpu_batch in train_loader:
                if single_frame:
                    cpu_batch = tuple(x[:, ::2].contiguous() for x in cpu_batch)
                yield tuple(x.to(device, non_blocking=True) for x in cpu_batch)

        stream_of_batches = device_batches()
        batch, pre = next(stream_of_batches, None), None
        while batch is not None:
            upcoming = next(stream_of_batches, None)  # one batch ahead: its geometry is queued during this step
            seen = it * global_batch
            for group in optimizer.param_groups:
                group["lr"] = cfg["lr"] * schedule_factor(cfg, seen)
            mom = norm_momentum(cfg, seen)
            for m in net.modules():
                if isinstance(m, NORM_LAYERS):
                    m.momentum = mom
            # the scalars of step i are read while step i+1 is already queued: the host never waits inside a step
            if use_graph:
                # everything the capture froze: learning rate, norm momentum, which loss terms are active, augmentation
                gates = tuple(it * world * cfg["batch_size"] >= st for st in cfg["loss"].get("start_steps", [0, 0, 0]))
                key = (optimizer.param_groups[0]["lr"], mom, gates, aug, tuple(batch[0].shape))
                if graphed is None or key != graphed_key:
                    from .graph_step import GraphedTrainStep
                    if graphed is None or key[3:] != graphed_key[3:]:
                        graphed = GraphedTrainStep(model, criterion, optimizer, batch, it * world, aug)
                    else:
                        graphed.load(batch)
                        graphed.recapture(it * world)
                    graphed_key, graph_holds = key, batch
                elif graph_holds is not batch:
                    graphed.load(batch)
                pending = graphed.step(upcoming if upcoming is not None else batch)
                account(in_flight)
                in_flight = pending
                graph_holds, batch = upcoming, upcoming
                it += 1
                if args.max_iters and it >= args.max_iters:
                    break
                continue
            pending = train_step(model, criterion, optimizer, batch, it * world, aug, sync=False, prefetched=pre,
                                 next_batch=upcoming)
            pre, batch = pending.prefetched, upcoming
            account(in_flight)
            in_flight = pending
            it += 1
            if args.max_iters and it >= args.max_iters:
                break
        account(in_flight)
        n_it = max(len(train_loader) if not args.max_iters else min(len(train_loader), it), 1)
        val_loss, val_metrics = evaluate(model, criterion, val_loader, device, single_frame,
                                         cfg.get("ignore_npoint_thresh", 0))
        if distributed:
            t = torch.tensor([val_loss], device=device)
            dist.all_reduce(t)
            val_loss = float(t) / world
        if rank == 0:
            is_best = val_loss < best
            best = min(best, val_loss)
            save_checkpoint(net, exp_base, is_best)
            print(json.dumps({"epoch": epoch, "it": it, "lr": optimizer.param_groups[0]["lr"], "aug": aug,
                              "train": {k: round(v / n_it, 5) for k, v in sums.items()},
                              "val_loss": round(val_loss, 5), "val": {k: round(v, 4) for k, v in val_metrics.items()},
                              "sec": round(time.time() - t0, 2)}), flush=True)
        if args.max_iters and it >= args.max_iters:
            break
    if distributed:
        dist.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
