"""Unsupervised FlowStep3D training driver on the MI355X operators (counterpart of the reference's
train_flow.py:33-284, written against this repo's layers).

    python -m ogc_amd.train_flow config.yaml [--synthetic N_PAIRS] [--max-iters K]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ogc_amd.train_flow config.yaml ...

Kept from the reference so that its configs, schedules and checkpoints carry over:
  * the YAML schema (config/flow/*/*.yaml): dataset, save_path, random_seed, flownet{npoint, use_instance_norm,
    loc_flow_nn, loc_flow_rad, k_decay_fact}, model_iters, epochs, batch_size, lr, lr_decay, lr_clip, bn_momentum,
    bn_decay, weight_decay, decay_step, loss{weights, iters_w, chamfer_loss_params, smooth_loss_params};
  * Adam + lr_curve + BatchNorm-momentum schedule (train_flow.py:190-206), driven by samples seen (global batch);
  * the step body (train_flow.py:62-88) incl. the NaN-gradient rule; checkpoints {'model_state': ...} as
    current.pth.tar / best.pth.tar in `save_path`.
Out of scope: dataset readers, EPE metrics, tensorboard.  Without a dataset the driver trains on seeded
synthetic frame pairs with the loaders' sample contract (ogc_amd/utils/synthetic.py).
Under DDP every rank normalises with its own BatchNorm statistics (no SyncBN), as noted in SURVEY §8e.
"""
import argparse
import importlib
import json
import math
import os
import time

import torch
import torch.distributed as dist
import yaml

from .losses.flow_loss_unsup import ChamferLoss, SmoothLoss, UnsupervisedFlowStep3DLoss
from .train_seg import SyntheticScenes, save_checkpoint, schedule_factor, norm_momentum
from .train_step import flow_train_step, make_optimizer

FLOWNETS = {"sapien": "flownet_sapien", "ogcdr": "flownet_ogcdr", "ogcdrsv": "flownet_ogcdr", "kittisf": "flownet_kitti",
            "waymo": "flownet_kitti"}
NORM_LAYERS = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)


def build_flow_criterion(cfg):
    return UnsupervisedFlowStep3DLoss(ChamferLoss(**cfg["chamfer_loss_params"]), SmoothLoss(**cfg["smooth_loss_params"]),
                                      weights=cfg["weights"], iters_w=cfg["iters_w"])


def evaluate(model, criterion, loader, device, model_iters):
    model.eval()
    total, count = 0.0, 0
    with torch.no_grad():
        for pcs, _, _, _ in loader:
            pcs = pcs.to(device)
            pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
            loss, _ = criterion(pc1, pc2, model(pc1, pc2, pc1, pc2, iters=model_iters), sync=False)
            total += float(loss)
            count += 1
    return total / max(count, 1)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--synthetic", type=int, default=64, help="number of synthetic training pairs")
    ap.add_argument("--max-iters", type=int, default=0, help="stop after this many optimisation steps (0 = all epochs)")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--ddp", action="store_true",
                    help="average gradients with torch's DistributedDataParallel instead of one flat collective per step")
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

    torch.manual_seed(cfg.get("random_seed", 10))
    fl = cfg["flownet"]
    FlowStep3D = importlib.import_module("ogc_amd.models." + FLOWNETS[cfg["dataset"]]).FlowStep3D
    k_decay = fl["k_decay_fact"]
    net = FlowStep3D(npoint=fl["npoint"], use_instance_norm=fl["use_instance_norm"], loc_flow_nn=fl["loc_flow_nn"],
                     loc_flow_rad=fl["loc_flow_rad"], k_decay_fact=float(k_decay)).to(device)
    # one gradient collective per step (utils/dist_util.py); `--ddp` keeps torch's DistributedDataParallel
    if distributed and getattr(args, "ddp", False):
        model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[device.index] if device.type == "cuda" else None)
    elif distributed:
        from .utils.dist_util import FlatDataParallel
        model = FlatDataParallel(net)
    else:
        model = net
    model_iters = cfg.get("model_iters", len(cfg["loss"]["iters_w"]))

    outdoor = cfg["dataset"] in ("kittisf", "waymo")
    train_set = SyntheticScenes(args.synthetic, fl["npoint"], 8, outdoor, seed=1000)  # one set on all ranks; the sampler shards it
    val_set = SyntheticScenes(max(args.synthetic // 8, cfg["batch_size"]), fl["npoint"], 8, outdoor, seed=7)
    sampler = torch.utils.data.distributed.DistributedSampler(train_set) if distributed else None
    train_loader = torch.utils.data.DataLoader(train_set, batch_size=cfg["batch_size"], shuffle=sampler is None,
                                               sampler=sampler, drop_last=True)
    val_loader = torch.utils.data.DataLoader(val_set, batch_size=cfg["batch_size"], shuffle=False)

    optimizer = make_optimizer(net.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    criterion = build_flow_criterion(cfg["loss"])
    exp_base = cfg["save_path"]
    if rank == 0:
        os.makedirs(exp_base, exist_ok=True)
        save_checkpoint(net, exp_base, True)  # initial weights as current and best (train_flow.py:126-130)

    global_batch = cfg["batch_size"] * world
    it, best = 0, 1e10
    for epoch in range(1, cfg["epochs"] + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        sums, t0, in_flight = {}, time.time(), None

        def account(pending):
            if pending is not None:
                for k, v in pending.result()[0].items():
                    if math.isfinite(v):  # the reference's AverageMeter drops NaN values
                        sums[k] = sums.get(k, 0.0) + v

        for cpu_batch in train_loader:
            seen = it * global_batch
            for group in optimizer.param_groups:
                group["lr"] = cfg["lr"] * schedule_factor(cfg, seen)
            mom = norm_momentum(cfg, seen)
            for m in net.modules():
                if isinstance(m, NORM_LAYERS):
                    m.momentum = mom
            batch = tuple(x.to(device, non_blocking=True) for x in cpu_batch)
            # the scalars of step i are read while step i+1 is already queued: the host never waits inside a step
            pending = flow_train_step(model, criterion, optimizer, batch, model_iters, sync=False)
            account(in_flight)
            in_flight = pending
            it += 1
            if args.max_iters and it >= args.max_iters:
                break
        account(in_flight)
        n_it = max(len(train_loader) if not args.max_iters else min(len(train_loader), it), 1)
        val_loss = evaluate(model, criterion, val_loader, device, model_iters)
        if distributed:
            t = torch.tensor([val_loss], device=device)
            dist.all_reduce(t)
            val_loss = float(t) / world
        if rank == 0:
            is_best = val_loss < best
            best = min(best, val_loss)
            save_checkpoint(net, exp_base, is_best)
            print(json.dumps({"epoch": epoch, "it": it, "lr": optimizer.param_groups[0]["lr"],
                              "train": {k: round(v / n_it, 5) for k, v in sums.items()},
                              "val_loss": round(val_loss, 5), "sec": round(time.time() - t0, 2)}), flush=True)
        if args.max_iters and it >= args.max_iters:
            break
    if distributed:
        dist.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
