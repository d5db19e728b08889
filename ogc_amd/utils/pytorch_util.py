"""Training-loop helpers with the interface of the reference's utils/pytorch_util.py (same class and function names, same
argument meaning), so that a trainer written against the reference's reads the same here:

    BNMomentumScheduler(model, bn_lambda).step(it)      utils/pytorch_util.py:113-136
    LambdaLR(optimizer, lr_lambda).step(it)             torch.optim.lr_scheduler.LambdaLR as train_seg.py:52 drives it
    AverageMeter / RunningAverageMeter                  utils/pytorch_util.py:9-59
    checkpoint_state / save_checkpoint                  utils/pytorch_util.py:84-99
"""
import math
import shutil
from collections import OrderedDict

import torch
import torch.nn as nn

NORM_LAYERS = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d,
               nn.GroupNorm)


class RunningAverageMeter:
    """Exponential running mean per key; the first value of a key initialises it, NaN values are skipped
    (utils/pytorch_util.py:9-27)."""

    def __init__(self, alpha=1.0):
        self.alpha = alpha
        self.loss_dict = OrderedDict()

    def append_loss(self, losses):
        for name, val in losses.items():
            if val is None:
                continue
            val = float(val)
            if math.isnan(val):
                continue
            if name not in self.loss_dict:
                self.loss_dict[name] = val
            else:
                self.loss_dict[name] = self.alpha * self.loss_dict[name] + (1 - self.alpha) * val

    def get_loss_dict(self):
        return dict(self.loss_dict)


class AverageMeter:
    """Mean per key over the values that are not NaN — each key divides by ITS OWN count (utils/pytorch_util.py:30-59)."""

    def __init__(self):
        self.loss_dict = OrderedDict()

    def append_loss(self, losses):
        for name, val in losses.items():
            if val is None:
                continue
            val = float(val)
            if math.isnan(val):
                continue
            if name not in self.loss_dict:
                self.loss_dict[name] = [val, 1]
            else:
                self.loss_dict[name][0] += val
                self.loss_dict[name][1] += 1

    def get_mean_loss(self):
        total = sum(v for v, _ in self.loss_dict.values())
        count = sum(c for _, c in self.loss_dict.values())
        return total / (count / len(self.loss_dict))

    def get_mean_loss_dict(self):
        return {name: v / c for name, (v, c) in self.loss_dict.items()}


def checkpoint_state(model):
    if isinstance(model, torch.nn.DataParallel) or hasattr(model, "module"):
        model = model.module
    return {"model_state": model.state_dict()}


def save_checkpoint(state, is_best, filename="checkpoint", bestname="model_best"):
    filename = "{}.pth.tar".format(filename)
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, "{}.pth.tar".format(bestname))


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, NORM_LAYERS):
            m.momentum = bn_momentum

    return fn


class BNMomentumScheduler:
    """Sets `momentum` of every norm layer to bn_lambda(epoch) (utils/pytorch_util.py:113-136; the constructor applies
    bn_lambda(last_epoch + 1) once)."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def get_momentum(self):
        return self.lmbd(self.last_epoch)

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))


class LambdaLR:
    """lr = initial lr * lr_lambda(epoch), set by step(epoch) — what torch's LambdaLR does when it is stepped with an explicit
    epoch, as train_seg.py:51-52 / train_flow.py:66-67 step it with the iteration number (that call form is deprecated in torch
    and warns on every step)."""

    def __init__(self, optimizer, lr_lambda, last_epoch=-1):
        self.optimizer = optimizer
        self.lr_lambda = lr_lambda
        self.base_lrs = [group.setdefault("initial_lr", group["lr"]) for group in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step(last_epoch + 1)

    def get_last_lr(self):
        return [group["lr"] for group in self.optimizer.param_groups]

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        for group, base in zip(self.optimizer.param_groups, self.base_lrs):
            group["lr"] = base * self.lr_lambda(epoch)
