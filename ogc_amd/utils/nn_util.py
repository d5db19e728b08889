"""Layer builders for the per-point MLPs (reference: utils/nn_util.py).

What must be preserved is the *naming contract*: checkpoints of the reference address parameters as
``...layer{i}.conv.weight``, ``...layer{i}.normlayer.gn.{weight,bias}`` (GroupNorm) or
``...normlayer.bn.*`` (BatchNorm), ``...fc.weight`` (SURVEY.md §5 "checkpoint / resume").  Every block is
therefore an ``nn.Sequential`` whose children carry exactly those names, in the reference's order
(conv -> norm -> activation, or norm -> activation -> conv when ``preact``).
"""
import numpy as np
import torch
import torch.nn as nn


class GroupNorm(nn.Sequential):
    """``<name>gn`` = nn.GroupNorm(num_groups, in_size), affine initialised to (1, 0). Ref nn_util.py:6-11."""

    def __init__(self, in_size, num_groups, name=""):
        super().__init__()
        gn = nn.GroupNorm(num_groups, in_size)
        nn.init.ones_(gn.weight)
        nn.init.zeros_(gn.bias)
        self.add_module(name + "gn", gn)


class _BatchNormNd(nn.Sequential):
    """``<name>bn`` = BatchNorm{1,2}d(in_size), affine initialised to (1, 0). Ref nn_util.py:14-30."""
    _cls = None

    def __init__(self, in_size, name=""):
        super().__init__()
        bn = self._cls(in_size)
        nn.init.ones_(bn.weight)
        nn.init.zeros_(bn.bias)
        self.add_module(name + "bn", bn)


class BatchNorm1d(_BatchNormNd):
    _cls = nn.BatchNorm1d


class BatchNorm2d(_BatchNormNd):
    _cls = nn.BatchNorm2d


def get_norm_layer(layer_def, dimension, **kwargs):
    """``layer_def`` = {"class": "GroupNorm"|"BatchNorm", **ctor kwargs}; None -> Identity. Ref nn_util.py:33-42."""
    if layer_def is None:
        return nn.Identity()
    spec = dict(kwargs)
    spec.update(layer_def)
    kind = spec.pop("class")
    if kind == "GroupNorm":
        return GroupNorm(**spec)
    if kind == "BatchNorm":
        return (BatchNorm1d, BatchNorm2d)[dimension - 1](**spec)
    raise KeyError(kind)


def _assemble(seq, name, main_name, main, norm, activation, preact):
    """Children in the reference's order (nn_util.py:67-82, 138-152)."""
    tail = []
    if norm is not None:
        tail.append((name + "normlayer", norm))
    if activation is not None:
        tail.append((name + "activation", activation))
    order = tail + [(name + main_name, main)] if preact else [(name + main_name, main)] + tail
    for key, mod in order:
        seq.add_module(key, mod)


class _ConvNd(nn.Sequential):
    """conv (+ norm) (+ activation).  The conv has a bias only when there is no norm layer. Ref nn_util.py:45-82."""
    _conv = None
    _dim = None

    def __init__(self, in_size, out_size, kernel_size, stride, padding, dilation, activation, bn, init, bias,
                 preact, name):
        super().__init__()
        use_bias = bias and (bn is None)
        conv = self._conv(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                          dilation=dilation, bias=use_bias)
        init(conv.weight)
        if use_bias:
            nn.init.zeros_(conv.bias)
        norm = None
        if bn is not None:
            norm = get_norm_layer(bn, self._dim, in_size=in_size if preact else out_size)
        _assemble(self, name, "conv", conv, norm, activation, preact)
        # conv -> GroupNorm -> (ReLU | nothing) is executed as conv + ONE fused GroupNorm/activation op
        self._gn_fuse = (not preact and isinstance(norm, GroupNorm)
                         and (activation is None or type(activation) is nn.ReLU))
        self._names = (name + "conv", name + "normlayer", activation is not None)

    def forward(self, input):
        if self._gn_fuse:
            from ..fused import group_norm_act, pointwise_conv
            conv_name, norm_name, relu = self._names
            gn = getattr(self, norm_name)[0]
            y, stats = pointwise_conv(input, getattr(self, conv_name), gn)
            return group_norm_act(y, gn, relu, stats)
        # other norms (BatchNorm in the FlowStep3D nets) / pre-activation order: the children in sequence, with the
        # bias-free 1x1 convolution on the MFMA kernels where it qualifies
        from ..fused import pointwise_conv
        x = input
        for mod in self:
            x = pointwise_conv(x, mod) if isinstance(mod, (nn.Conv1d, nn.Conv2d)) else mod(x)
        return x

    def forward_maxpool(self, input):
        """forward followed by a max over the last dimension (fused with GroupNorm/ReLU where possible)."""
        if self._gn_fuse:
            from ..fused import group_norm_act_maxpool, pointwise_conv
            conv_name, norm_name, relu = self._names
            gn = getattr(self, norm_name)[0]
            y, stats = pointwise_conv(input, getattr(self, conv_name), gn)
            return group_norm_act_maxpool(y, gn, relu, stats)
        return self.forward(input).max(dim=-1)[0]


class Conv1d(_ConvNd):
    _conv, _dim = nn.Conv1d, 1

    def __init__(self, in_size, out_size, kernel_size=1, stride=1, padding=0, dilation=1,
                 activation=nn.ReLU(inplace=True), bn=None, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name=""):
        super().__init__(in_size, out_size, kernel_size, stride, padding, dilation, activation, bn, init, bias,
                         preact, name)


class Conv2d(_ConvNd):
    _conv, _dim = nn.Conv2d, 2

    def __init__(self, in_size, out_size, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), dilation=(1, 1),
                 activation=nn.ReLU(inplace=True), bn=None, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name=""):
        super().__init__(in_size, out_size, kernel_size, stride, padding, dilation, activation, bn, init, bias,
                         preact, name)


class FC(nn.Sequential):
    """Linear (+ norm) (+ activation); child name ``fc``. Ref nn_util.py:113-152."""

    def __init__(self, in_size, out_size, activation=nn.ReLU(inplace=True), bn=None, init=None, preact=False,
                 name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=bn is None)
        if init is not None:
            init(fc.weight)
        if bn is None:
            nn.init.zeros_(fc.bias)
        norm = None
        if bn is not None:
            norm = get_norm_layer(bn, 1, in_size=in_size if preact else out_size)
        _assemble(self, name, "fc", fc, norm, activation, preact)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d blocks ``layer0 .. layer{n-2}`` over channel sizes ``args``. Ref nn_util.py:155-172."""

    def __init__(self, args, bn=None, activation=nn.ReLU(inplace=True), preact=False, first=False, name=""):
        super().__init__()
        for i, (c_in, c_out) in enumerate(zip(args[:-1], args[1:])):
            bare = first and preact and i == 0  # the very first pre-activation layer has no norm/act
            self.add_module(name + "layer%d" % i,
                            Conv2d(c_in, c_out, bn=None if bare else bn, activation=None if bare else activation,
                                   preact=preact))


def _shared_mlp_run(self, x, pool, first=None):
    """The layers in sequence.  Inside the stack the output of conv -> GroupNorm -> ReLU is only read by the next
    convolution, so on the GPU it is never written: the raw convolution output travels on together with the norm's
    statistics (`pending`) and the norm is applied by the next convolution while loading (fused.norm_act_conv).  The
    last layer's norm / activation (/ max over the neighbourhood, `pool`) is one fused op.
    first: optional callable (conv, gn) -> (raw output, statistics) that evaluates the FIRST layer's convolution (a set-
    abstraction level fuses it with the grouping, fused.grouped_first_layer); `x` is then unused."""
    from ..fused import (act16_middle_ok, group_norm_act, group_norm_act_maxpool, norm_act_conv, norm_act_conv_available,
                         norm_act_conv_pool, norm_act_conv_pool_available, pointwise_conv)
    layers = list(self.children())
    pending = None  # (raw conv output, its GroupNorm statistics or None, the GroupNorm, relu?, neighbourhood extremes)

    def flush(p, last):
        y, stats, gn, relu, extremes = p
        if y.dtype is torch.bfloat16:   # (no materialising GroupNorm of a 16-bit tensor: widen; counted in GATE_MISSES)
            from ..fused import act16_leave
            y, extremes = act16_leave(y), None
        if last and pool:
            return group_norm_act_maxpool(y, gn, relu, stats, extremes)
        return group_norm_act(y, gn, relu, stats)

    for li, layer in enumerate(layers):
        fusable = isinstance(layer, _ConvNd) and layer._gn_fuse
        if not fusable:
            if pending is not None:
                x, pending = flush(pending, False), None
            x = layer(x)
            continue
        conv_name, norm_name, relu = layer._names
        conv, gn = getattr(layer, conv_name), getattr(layer, norm_name)[0]
        extremes = None
        if pending is not None and pending[0].dtype is torch.bfloat16:
            # 16-bit activations continue only through the fused forms that have a 16-bit kernel: a middle layer through
            # norm_act_conv, the pooled tail through the one-node norm_act_conv_pool; anything else widens first
            tail = pool and li == len(layers) - 1
            ok = (norm_act_conv_pool_available(pending[0], pending[2], conv, gn) if tail else
                  (act16_middle_ok(pending[0], pending[2], conv) and li < len(layers) - 1))
            if not ok:
                from ..fused import act16_leave
                pending = (act16_leave(pending[0]),) + tuple(pending[1:4]) + (None,)
        if pending is not None and norm_act_conv_available(pending[0], pending[2], conv):
            if pool and li == len(layers) - 1 and pending[0].dim() == 4 and pending[0].shape[-1] in (16, 32, 64):
                # last layer before the max over the neighbourhood: the convolution also leaves each neighbourhood's
                # extremes, from which the pooled activation follows without reading its output again
                if norm_act_conv_pool_available(pending[0], pending[2], conv, gn):
                    # ... and one autograd node for the whole tail: no dense gradient between the pooling and the convolution
                    return norm_act_conv_pool(pending[0], pending[1], pending[2], pending[3], conv, gn, relu)
                y, stats, extremes = norm_act_conv(pending[0], pending[1], pending[2], pending[3], conv, gn,
                                                   pool=pending[0].shape[-1])
            else:
                y, stats = norm_act_conv(pending[0], pending[1], pending[2], pending[3], conv, gn)
        else:
            if pending is not None:
                x, pending = flush(pending, False), None
            y, stats = first(conv, gn) if (first is not None and li == 0) else pointwise_conv(x, conv, gn)
        pending = (y, stats, gn, relu, extremes)
    if pending is not None:
        return flush(pending, True)
    return x.max(dim=-1)[0] if pool else x


def _shared_mlp_forward(self, x):
    return _shared_mlp_run(self, x, False)


def _shared_mlp_forward_maxpool(self, x, first=None):
    """SharedMLP applied to (B, C, npoint, nsample) followed by the max over nsample
    (reference: utils/pointnet2_util.py:38-42); the last layer's norm/activation/pooling run as one fused op."""
    return _shared_mlp_run(self, x, True, first)


def _shared_mlp_first_layer(self):
    """(conv, GroupNorm) of the first layer when it is a fusable conv -> GroupNorm -> act block, else None."""
    layers = list(self.children())
    if not layers or not (isinstance(layers[0], _ConvNd) and layers[0]._gn_fuse):
        return None
    conv_name, norm_name, _ = layers[0]._names
    return getattr(layers[0], conv_name), getattr(layers[0], norm_name)[0]


SharedMLP.forward = _shared_mlp_forward
SharedMLP.forward_maxpool = _shared_mlp_forward_maxpool
SharedMLP.first_layer = _shared_mlp_first_layer


def knead_leading_dims(n_dim: int, data: torch.Tensor):
    """Merge the first ``n_dim`` dimensions. Ref nn_util.py:175-181."""
    if data is None:
        return None
    shape = list(data.size())
    return data.view(int(np.prod(shape[:n_dim])), *shape[n_dim:])


def break_leading_dim(dim_size: list, data: torch.Tensor):
    """Inverse of knead_leading_dims. Ref nn_util.py:184-189."""
    if data is None:
        return None
    return data.view(*dim_size, *list(data.size())[1:])


class Seq(nn.Sequential):
    """Fluent builder: ``Seq(c).conv1d(...).conv1d(...)``; children are named "0", "1", ... Ref nn_util.py:192-249."""

    def __init__(self, input_channels):
        super().__init__()
        self.count = 0
        self.current_channels = input_channels

    def _push(self, module, out_channels=None):
        self.add_module(str(self.count), module)
        self.count += 1
        if out_channels is not None:
            self.current_channels = out_channels
        return self

    def conv1d(self, out_size, kernel_size=1, stride=1, padding=0, dilation=1, activation=nn.ReLU(inplace=True),
               leaky=False, bn=None, init=nn.init.kaiming_normal_, bias=True, preact=False, name=""):
        if leaky:
            activation = nn.LeakyReLU(0.1, inplace=True)
        return self._push(Conv1d(self.current_channels, out_size, kernel_size=kernel_size, stride=stride,
                                 padding=padding, dilation=dilation, activation=activation, bn=bn, init=init,
                                 bias=bias, preact=preact, name=name), out_size)

    def conv2d(self, out_size, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), dilation=(1, 1),
               activation=nn.ReLU(inplace=True), bn=None, init=nn.init.kaiming_normal_, bias=True, preact=False,
               name=""):
        return self._push(Conv2d(self.current_channels, out_size, kernel_size=kernel_size, stride=stride,
                                 padding=padding, dilation=dilation, activation=activation, bn=bn, init=init,
                                 bias=bias, preact=preact, name=name), out_size)

    def fc(self, out_size, activation=nn.ReLU(inplace=True), bn=None, init=None, preact=False, name=""):
        return self._push(FC(self.current_channels, out_size, activation=activation, bn=bn, init=init,
                             preact=preact, name=name), out_size)

    def dropout(self, p=0.5):
        return self._push(nn.Dropout(p=p))

    def maxpool2d(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False, ceil_mode=False):
        return self._push(nn.MaxPool2d(kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
                                       return_indices=return_indices, ceil_mode=ceil_mode))
