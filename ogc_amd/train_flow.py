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
import os
import time

import torch
import torch.distributed as dist
import yaml

from .losses.flow_loss_unsup import ChamferLoss, SmoothLoss, UnsupervisedFlowStep3DLoss
from .train_seg import MAX_CONSECUTIVE_SKIPS, SyntheticScenes, schedule_factor, norm_momentum
from .train_step import flow_train_step, make_optimizer
from .utils.pytorch_util import AverageMeter, BNMomentumScheduler, LambdaLR, checkpoint_state, save_checkpoint

FLOWNETS = {"sapien": "flownet_sapien", "ogcdr": "flownet_ogcdr", "ogcdrsv": "flownet_ogcdr", "kittisf": "flownet_kitti",
            "waymo": "flownet_kitti"}


def build_flow_criterion(cfg):
    return UnsupervisedFlowStep3DLoss(ChamferLoss(**cfg["chamfer_loss_params"]), SmoothLoss(**cfg["smooth_loss_params"]),
                                      weights=cfg["weights"], iters_w=cfg["iters_w"])


class Trainer(object):
    """The reference's flow Trainer (train_flow.py:33-184) on this repo's step: same constructor arguments,
    `_train_it(it, batch)`, `eval_epoch(loader)` and `train(n_epochs, train_loader, val_loader)`, same schedule calls, NaN rule,
    best-checkpoint rule and checkpoint files.  Keyword-only extras: the data-parallel wrapper, device, rank / world, an
    iteration cap, a log sink and a per-iteration callback for tests."""

    def __init__(self, flownet, model_iters, criterion, optimizer, exp_base, lr_scheduler=None, bnm_scheduler=None, *,
                 model=None, device=None, world=1, rank=0, max_iters=0, log=print, on_iteration=None):
        self.flownet = flownet
        self.model = model if model is not None else flownet
        self.model_iters = model_iters
        self.criterion = criterion
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.bnm_scheduler = bnm_scheduler
        self.exp_base = exp_base
        self.device = device if device is not None else next(flownet.parameters()).device
        self.world, self.rank, self.max_iters, self.log, self.on_iteration = world, rank, max_iters, log, on_iteration
        if rank == 0:
            os.makedirs(exp_base, exist_ok=True)
        self.checkpoint_name, self.best_name = "current", "best"
        self.cur_epoch = 0
        self._skipped_in_a_row = 0

    def _train_it(self, it, batch, sync=True, prefetched=None, next_batch=None):
        """train_flow.py:62-88: schedules stepped with the iteration number, forward over `model_iters` GRU iterations, loss
        (+ the EPE of every iteration against the first frame's flow, monitored), backward, NaN rule, Adam."""
        if self.lr_scheduler is not None:
            self.lr_scheduler.step(it)
        if self.bnm_scheduler is not None:
            self.bnm_scheduler.step(it)
        return flow_train_step(self.model, self.criterion, self.optimizer, batch, self.model_iters, sync=sync, prefetched=prefetched,
                               next_batch=next_batch)

    def eval_epoch(self, val_loader):
        """(validation loss, mean loss_dict incl. the EPE terms).  The loss is the reference's number: the sum over the n
        batches divided by n + 1 (train_flow.py:97-98,:115; see ogc_amd/train_seg.py::Trainer.eval_epoch)."""
        from .metrics.flow_metric import epe_terms
        self.model.eval()
        eval_meter = AverageMeter()
        total_loss, count = 0.0, 1.0
        with torch.no_grad():
            for pcs, _, flows, _ in val_loader:
                pcs = pcs.to(self.device)
                pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
                flow_preds = self.model(pc1, pc2, pc1, pc2, iters=self.model_iters)
                loss, loss_dict = self.criterion(pc1, pc2, flow_preds, extra=epe_terms(flows[:, 0].to(self.device), flow_preds))
                total_loss += float(loss)
                count += 1
                eval_meter.append_loss(loss_dict)
        return total_loss / count, eval_meter.get_mean_loss_dict()

    def _save(self, is_best):
        if self.rank == 0:
            save_checkpoint(checkpoint_state(self.flownet), is_best, filename=os.path.join(self.exp_base, self.checkpoint_name),
                            bestname=os.path.join(self.exp_base, self.best_name))

    def train(self, n_epochs, train_loader, val_loader=None):
        it, best_loss = 0, 1e10
        self._save(True)  # initial weights as current and best (train_flow.py:126-130)
        sampler = getattr(train_loader, "sampler", None)
        for epoch in range(1, n_epochs + 1):
            self.cur_epoch = epoch
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

            # one batch ahead (as the segmentation trainer): the sampling chains of batch i + 1 run on a side stream underneath
            # step i's backward pass (train_step.flow_train_step, next_batch=)
            loader = iter(train_loader)

            def fetch():
                cpu_batch = next(loader, None)
                return None if cpu_batch is None else tuple(x.to(self.device, non_blocking=True) for x in cpu_batch)

            batch, ahead = fetch(), None
            while batch is not None:
                upcoming = fetch()
                on_gpu = batch[0].is_cuda
                # the scalars of step i are read while step i + 1 is already queued: the host never waits inside a step
                pending = self._train_it(it, batch, sync=False, **({"prefetched": ahead, "next_batch": upcoming} if on_gpu else {}))
                ahead = getattr(pending, "prefetched", None)
                account(in_flight, it - 1)
                in_flight = pending
                it += 1
                batch = upcoming
                if self.max_iters and it >= self.max_iters:
                    break
            account(in_flight, it - 1)
            record = {"epoch": epoch, "it": it, "lr": self.optimizer.param_groups[0]["lr"],
                      "train": {k: round(v, 5) for k, v in train_meter.get_mean_loss_dict().items()}, "skipped_steps": skipped}
            if val_loader is not None:
                val_loss, val_avg = self.eval_epoch(val_loader)
                if self.world > 1:
                    t = torch.tensor([val_loss], device=self.device)
                    dist.all_reduce(t)
                    val_loss = float(t) / self.world
                is_best = val_loss < best_loss
                best_loss = min(best_loss, val_loss)
                self._save(is_best)
                record.update(val_loss=round(val_loss, 5), val_terms={k: round(v, 5) for k, v in val_avg.items()}, is_best=is_best)
            record["sec"] = round(time.time() - t0, 2)
            if self.rank == 0:
                self.log(json.dumps(record))
            if self.max_iters and it >= self.max_iters:
                break
        return best_loss


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
    global_batch = cfg["batch_size"] * world   # schedules by samples seen (train_flow.py:187-203), over all ranks
    lr_scheduler = LambdaLR(optimizer, lr_lambda=lambda it: schedule_factor(cfg, it * global_batch))
    bnm_scheduler = BNMomentumScheduler(net, bn_lambda=lambda it: norm_momentum(cfg, it * global_batch))
    trainer = Trainer(net, model_iters, criterion, optimizer, exp_base=cfg["save_path"], lr_scheduler=lr_scheduler,
                      bnm_scheduler=bnm_scheduler, model=model, device=device, world=world, rank=rank, max_iters=args.max_iters,
                      log=lambda line: print(line, flush=True))
    best = trainer.train(cfg["epochs"], train_loader, val_loader)
    if distributed:
        dist.destroy_process_group()
    return best


if __name__ == "__main__":
    main()
