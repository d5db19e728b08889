"""Host-side wrappers of the fused HIP extensions that replace Python-level op sequences of the reference's
layers (no counterpart in its native module; see include/ogc_ops.h "fused extensions")."""
import weakref

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .pointnet2 import pointnet2 as _api



# ---- which shapes do NOT take a fused kernel --------------------------------------------------------------------------------
# Every `*_available` gate below answers "does the fused kernel cover this call?"; a False sends the caller down the plain
# operator sequence — correct, slower, and silent.  GATE_MISSES counts those answers per gate (tests/test_fallbacks_gpu.py runs
# one step of each configuration and pins the set, so a shape that quietly leaves a fused path shows up as a test failure).
import collections as _collections
import functools as _functools

from .utils.zero_arena import zeroed_empty  # buffers an operator zeroes before accumulating: one fill per step

GATE_MISSES = _collections.Counter()


def deterministic():
    """The deterministic-gradient mode (OGC_DETERMINISTIC=1 / _lib.set_deterministic; include/ogc_ops.h: ogc_set_deterministic).
    The library's converted entry points then sum in a fixed order; the host layers below leave the fused forms whose kernels are
    not converted — GroupNorm statistics taken in a convolution's epilogue (fp64 atomics), the moment-matrix backward, the
    grouped first layer — for the plain sequence conv -> GroupNorm (+ ReLU, + max), whose kernels write every partial once
    (csrc/group_norm.hip) or go through the ordered passes of csrc/det.hip.  A test mode."""
    from . import _lib
    return _lib.DETERMINISTIC

# ---- 16-bit activations ------------------------------------------------------------------------------------------------------
# Under `matmul_precision: bf16` the raw convolution outputs inside a set-abstraction MLP, and the gradients with respect to
# them, are STORED as bf16 (csrc/act_io.h; the `_h` entry points of include/ogc_ops.h): those tensors are the only ones of the
# size of an activation, every kernel that touches one is bound by its bytes, and nothing else changes type — statistics fp64,
# coefficients / pooled outputs / parameters / their gradients fp32, coordinates and searches fp32.  The tensor's dtype carries
# the decision from the first layer (grouped_first_layer(act16=True)) through the stack: every later node allocates what it
# was given.  OGC_ACT16=0: fp32 activations with bf16 operands only (rounds 2-4), for A/B runs.
ACT16 = __import__("os").environ.get("OGC_ACT16", "1") != "0"


def act16_wanted(t):
    """Should a set-abstraction MLP fed from tensor `t` keep its activations in 16 bits?"""
    return (ACT16 and t is not None and t.is_cuda and getattr(_api._native, "get_matmul_precision", None) is not None
            and _api._native.get_matmul_precision() == "bf16")


def act16_leave(y):
    """A 16-bit activation handed to a path that has no 16-bit form: widen it (counted: tests pin the set of such exits)."""
    GATE_MISSES["act16_leave"] += 1
    return y.float()


def _is_act(t):
    return t.dtype in (torch.float32, torch.bfloat16)


def _gate(fn):
    @_functools.wraps(fn)
    def counted(*args, **kwargs):
        ok = fn(*args, **kwargs)
        if not ok:
            GATE_MISSES[fn.__name__] += 1
        return ok
    return counted


class _GroupNormAct(Function):
    """y = act(GroupNorm(x)) with act = ReLU or identity — one autograd node, two launches forward, three backward.
    Reference sequence: nn.GroupNorm then nn.ReLU(inplace=True) (utils/nn_util.py:6-11, :45-85)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, stats=None):
        nat = _api._native
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        hw = x.numel() // max(B * C, 1)
        y = torch.empty_like(x)
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if stats is not None:  # first pass already done by the producing convolution
            nat.group_norm_fwd_stats_wrapper(B, C, hw, groups, eps, relu, x, weight.contiguous(),
                                             bias.contiguous(), y, mean, rstd, stats,
                                             stats.numel() // (2 * B * groups))
        else:
            ws = _api._native.group_norm_ws(B, C, groups, False, x.device)
            nat.group_norm_fwd_wrapper(B, C, hw, groups, eps, relu, x, weight.contiguous(),
                                       bias.contiguous(), y, mean, rstd, ws)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.cfg = (groups, relu, hw)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        nat = _api._native
        x, weight, bias, mean, rstd = ctx.saved_tensors
        groups, relu, hw = ctx.cfg
        B, C = x.shape[0], x.shape[1]
        grad_y = grad_y.contiguous()
        grad_x = torch.empty_like(x)
        gw = torch.empty_like(weight)
        gb = torch.empty_like(bias)
        ws = _api._native.group_norm_ws(B, C, groups, True, x.device)
        nat.group_norm_bwd_wrapper(B, C, hw, groups, relu, x, weight.contiguous(), bias.contiguous(),
                                   mean, rstd, grad_y, grad_x, gw, gb, ws)
        return grad_x, gw, gb, None, None, None, None


def group_norm_act(x, gn: torch.nn.GroupNorm, relu: bool, stats=None):
    """GroupNorm followed by an optional ReLU.  HIP-fused on the GPU (fp32); the plain torch composition
    otherwise (that is what the reference runs).  stats: the statistics pass, when the convolution that produced x
    already made it (pointwise_conv(..., gn=...))."""
    if (x.is_cuda and x.dtype == torch.float32 and gn.affine
            and getattr(_api._native, "group_norm_fwd_wrapper", None) is not None):
        return _GroupNormAct.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, stats)
    y = F.group_norm(x, gn.num_groups, gn.weight, gn.bias, gn.eps)
    return F.relu(y) if relu else y


class _GroupNormActMaxPool(Function):
    """out[b,c,p] = max_s act(GroupNorm(x))[b,c,p,s] without materialising the normalised activation.
    Reference sequence: nn.GroupNorm, nn.ReLU, F.max_pool2d over nsample (utils/pointnet2_util.py:38-42)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu, stats=None, extremes=None):
        nat = _api._native
        x = x.contiguous()
        B, C, P, S = x.shape
        out = torch.empty(B, C, P, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, C, P, dtype=torch.int32, device=x.device)
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        if extremes is not None:
            # the convolution that wrote x also left the extremes of every neighbourhood: x is not read again
            yext, aext = extremes
            nat.group_norm_pool_extremes_wrapper(B, C, P, S, groups, eps, relu, yext, aext,
                                                 weight.contiguous(), bias.contiguous(), out, arg,
                                                 mean, rstd, stats, stats.numel() // (2 * B * groups))
        elif stats is not None:
            nat.group_norm_maxpool_fwd_stats_wrapper(B, C, P, S, groups, eps, relu, x, weight.contiguous(),
                                                     bias.contiguous(), out, arg, mean, rstd, stats,
                                                     stats.numel() // (2 * B * groups))
        else:
            ws = _api._native.group_norm_ws(B, C, groups, False, x.device)
            nat.group_norm_maxpool_fwd_wrapper(B, C, P, S, groups, eps, relu, x, weight.contiguous(),
                                               bias.contiguous(), out, arg, mean, rstd, ws)
        # (with the extremes at hand the backward sums need not gather x at the arg-max positions)
        yext = extremes[0] if (extremes is not None and stats is not None and POOL_SUMS_FROM_EXTREMES) else None
        ctx.save_for_backward(x, weight, mean, rstd, out, arg, yext)
        ctx.cfg = (groups, relu)
        ctx.mark_non_differentiable(arg)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nat = _api._native
        x, weight, mean, rstd, out, arg, yext = ctx.saved_tensors
        groups, relu = ctx.cfg
        B, C, P, S = x.shape
        grad_x = torch.empty_like(x)
        gw = torch.empty_like(weight)
        gb = torch.empty_like(weight)
        ws = _api._native.group_norm_ws(B, C, groups, True, x.device)
        if yext is not None and getattr(nat, "group_norm_maxpool_bwd_ext_wrapper", None) is not None:
            nat.group_norm_maxpool_bwd_ext_wrapper(B, C, P, S, groups, relu, x, yext, weight.contiguous(), mean, rstd,
                                                   out, arg, grad_out.contiguous(), grad_x, gw, gb, ws)
        else:
            nat.group_norm_maxpool_bwd_wrapper(B, C, P, S, groups, relu, x, weight.contiguous(), mean, rstd, out,
                                               arg, grad_out.contiguous(), grad_x, gw, gb, ws)
        return grad_x, gw, gb, None, None, None, None, None


def group_norm_act_maxpool(x, gn: torch.nn.GroupNorm, relu: bool, stats=None, extremes=None):
    """max over the last dimension of act(GroupNorm(x)), x (B, C, P, S).  Fused when S is a power of two in
    [4, 256] on the GPU; otherwise group_norm_act followed by a max."""
    S = x.shape[-1]
    if (x.is_cuda and x.dtype == torch.float32 and gn.affine and x.dim() == 4 and 4 <= S <= 256 and S & (S - 1) == 0
            and getattr(_api._native, "group_norm_maxpool_fwd_wrapper", None) is not None):
        return _GroupNormActMaxPool.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, stats,
                                          extremes if stats is not None else None)
    return group_norm_act(x, gn, relu, stats).max(dim=3)[0]


# Neighbourhood extremes also from the streaming kernel (K > 100: SA3's tail at C4).  Bit-identical
# (tests/test_pool_extremes_gpu.py).  With the first epilogue it was a loss (128 -> 256: 0.33 -> 0.56 ms, against 0.14 ms for the
# pooling pass it saves); with the round-3 epilogue (signs through LDS, branch-free, one-instruction DPP maxima) and the backward
# sums taking y at the arg-max from the extremes it is a small gain: step 11.31 -> 11.28 ms (A/B, five rounds on one box).
POOL_EXTREMES_WIDE = True
# The backward sums of a pooled GroupNorm take x at the arg-max positions from those extremes instead of gathering it (a 32-byte
# sector per element: 50 -> ~8 us per tail at C4).
POOL_SUMS_FROM_EXTREMES = True


# Layers of more than 100 input channels whose convolution also produces the next GroupNorm's statistics (and, for a tail, the
# neighbourhood extremes): fp32 operands -> the streaming kernel's epilogues; bf16 operands (round 5) -> the same epilogues on the
# register-tile kernel, which has MFMA time to spare at that precision.  OGC_BF16_WIDE_STATS=1 keeps such layers on the fp32
# streaming kernel under bf16 as well (measured slower at C2: 26.3 against 25.7 ms per step); OGC_BF16_WIDE_STATS=0 restores the
# round-4 behaviour (a statistics pass of its own behind the bf16 product, the dense backward path behind a wide pooled tail).
BF16_WIDE_STATS = __import__("os").environ.get("OGC_BF16_WIDE_STATS", "") != "0"


def _stats_ok(nat, B, cout, cin, hw, affine):
    fn = getattr(nat, "conv1x1_gemm_stats_supported", None)
    return (fn is not None and (BF16_WIDE_STATS or nat.get_matmul_precision() == "fp32") and fn(B, cout, cin, hw, affine))


def _gemm_ok(K, hw):
    return hw % 64 == 0 and K <= 160


# Input gradients of 101 .. 160-channel layers on the streaming MFMA kernel (round 4: transposed weights, two workgroups per CU)
# instead of the vendor GEMM
import os as _os
DGRAD_STREAM = _os.environ.get("OGC_DGRAD_STREAM", "1") != "0"   # (0: the vendor GEMM, for A/B runs)


def _plain_gemm_mine(K, dgrad_shape=None):
    """A PLAIN product (no folded normalisation on the way in, no statistics on the way out) of reduction length K:
    the tile kernel wins up to 32 channels; from 64 up rocBLAS is 15-35 % faster than it at every C4 layer shape
    (tools/dgrad_compare.py: 128 -> 128 input gradient 0.246 vs 0.173 ms), so those go through torch.matmul — unless bf16
    operands were asked for (which only this repo's kernels provide), or the product is an input gradient
    (dgrad_shape = (B, M, hw)) that the streaming kernel takes."""
    if K <= 32 or _api._native.get_matmul_precision() != "fp32":
        return True
    if DGRAD_STREAM and dgrad_shape is not None:
        fn = getattr(_api._native, "conv1x1_gemm_stream_supported", None)
        return fn is not None and fn(dgrad_shape[0], dgrad_shape[1], K, dgrad_shape[2])
    return False


# Below this many positions (B * hw) a 1x1 convolution goes to the vendor library: the MFMA kernel gives a wavefront 64
# positions x all output rows, so a 128 -> 128 layer costs ~28 us however few positions there are (2048 positions: 8
# workgroups on 256 CUs), where rocBLAS / MIOpen split the output rows as well and need 3-8 us (tools/small_conv.py).
# FlowStep3D at B = 1 runs ~65 such layers per forward pass.
_SMALL_CONV_POSITIONS = 49152


def _weight_times_batch(w2d, x3, out=None):
    """w2d (M, K) @ x3 (B, K, P) -> (B, M, P) as ONE strided-batched library product with the weight's batch stride zero.
    (torch.matmul gets there through its broadcasting logic: 44 us of launch-thread time per call against 19 here, the same
    GPU time — a C4 step makes ~16 such calls, and its launch thread is level with the GPU.)"""
    return torch.bmm(w2d.unsqueeze(0).expand(x3.shape[0], -1, -1), x3, out=out)


# The products the register-tile kernels do not take — more than 160 reduction channels, fewer than _SMALL_CONV_POSITIONS
# positions, plain 64-channel input gradients — on ogc_conv1x1_gemm_any (csrc/gemm_chunk.hip) and ogc_conv1x1_wgrad instead of the
# vendor library (round 4; tools/gemm_any_compare.py: level with or ahead of rocBLAS on every such product of the C4 / C2 steps
# except the 256-channel input gradient, 0.31 against 0.28 ms).  OGC_LIBRARY_GEMMS=1: the library routes, for A/B runs.
LIBRARY_GEMMS = _os.environ.get("OGC_LIBRARY_GEMMS", "0") == "1"
LIBRARY_WGRAD = LIBRARY_GEMMS or _os.environ.get("OGC_LIBRARY_WGRAD", "0") == "1"   # (the weight gradients alone)


@_gate
def gemm_any_available(x3):
    return (not LIBRARY_GEMMS and x3.is_cuda and x3.dtype == torch.float32 and x3.shape[2] % 64 == 0 and x3.shape[0] <= 65535
            and getattr(_api._native, "conv1x1_gemm_any_wrapper", None) is not None)


def _product(w2d, x3, transpose, out=None):
    """A . x3[b] for every sample: A = w2d (M, K), or w2d^T with w2d stored (K, M) when `transpose`; x3 (B, K, P) -> (B, M, P)."""
    B, K, P = x3.shape
    M = w2d.shape[1] if transpose else w2d.shape[0]
    if gemm_any_available(x3):
        x3 = x3.contiguous()
        if out is None:
            out = torch.empty(B, M, P, dtype=torch.float32, device=x3.device)
        _api._native.conv1x1_gemm_any_wrapper(B, M, K, P, 1 if transpose else 0, w2d.contiguous(), x3, out)
        return out
    return _weight_times_batch(w2d.t() if transpose else w2d, x3, out=out)


def _weight_grad(g3, x3):
    """sum_b g3[b] . x3[b]^T: g3 (B, cout, P), x3 (B, cin, P) -> (cout, cin)."""
    B, cout, P = g3.shape
    cin = x3.shape[1]
    if (not (LIBRARY_GEMMS or LIBRARY_WGRAD) and g3.is_cuda and P % 16 == 0 and g3.dtype == torch.float32
            and getattr(_api._native, "conv1x1_wgrad_wrapper", None) is not None):
        grad_w = zeroed_empty((cout, cin), torch.float32, g3.device)
        _api._native.conv1x1_wgrad_wrapper(B, cin, cout, P, x3.contiguous(), g3.contiguous(), grad_w)
        return grad_w
    return torch.bmm(g3, x3.transpose(1, 2)).sum(0)


class _PointwiseConv(Function):
    """y = conv(x, w) for a bias-free 1x1 convolution on NCHW fp32 tensors, on the hand-written fp32-MFMA kernels:
    ogc_conv1x1_gemm for the forward and the input gradient (when the shape fits its register tile; the vendor library
    otherwise) and ogc_conv1x1_wgrad for the weight gradient (which MIOpen computes through two full NCHW->NHWC
    transposes)."""

    @staticmethod
    def forward(ctx, x, weight, gn_groups=0):
        # gn_groups > 0: also return the statistics of the GroupNorm (gn_groups groups) that consumes y, or None
        ctx.set_materialize_grads(False)  # no zero tensor (one fill launch per layer) for the statistics output
        ctx.save_for_backward(x, weight)
        nat = _api._native
        B, cin = x.shape[0], x.shape[1]
        cout = weight.shape[0]
        hw = x.numel() // (B * cin)
        stats = None
        ctx.small = gn_groups == 0 and B * hw < _SMALL_CONV_POSITIONS
        if _gemm_ok(cin, hw) and not ctx.small:
            y = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
            if (gn_groups > 0 and gn_groups <= 32 and cout % gn_groups == 0 and (cout // gn_groups) % 4 == 0
                    # (K > 100: only the streaming kernel has registers to spare for the statistics epilogue)
                    and getattr(nat, "conv1x1_gemm_gnstats_wrapper", None) is not None
                    and (cin <= 100 or _stats_ok(nat, B, cout, cin, hw, False))):
                stats = zeroed_empty(nat.conv1x1_gn_slots() * B * gn_groups * 2, torch.float64, x.device)
                nat.conv1x1_gemm_gnstats_wrapper(B, cout, cin, hw, gn_groups, weight.contiguous(), x, y, stats)
            elif _plain_gemm_mine(cin):
                nat.conv1x1_gemm_wrapper(B, cout, cin, hw, 0, weight.contiguous(), x, y)
            else:
                _product(weight.detach().view(cout, cin), x.reshape(B, cin, hw), False, out=y.view(B, cout, hw))
        else:
            # shapes the register-tile kernels do not take (K > 160, few positions): the chunked MFMA kernel; ragged position
            # counts: one batched rocBLAS product.  NOT F.conv1d / F.conv2d: for these shapes MIOpen picks a naive direct-
            # convolution kernel (0.46 ms for the 16 x 128 x 10 object features of the C4 step) or a Winograd kernel with
            # transposes around it
            y = _product(weight.detach().reshape(cout, cin), x.reshape(B, cin, hw), False).view((B, cout) + tuple(x.shape[2:]))
        if gn_groups > 0:
            if stats is not None:
                ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, grad_y, _grad_stats=None):
        if grad_y is None:
            return None, None, None
        x, weight = ctx.saved_tensors
        nd = x.dim() - 2
        grad_y = grad_y.contiguous()
        B, cin = x.shape[0], x.shape[1]
        cout = weight.shape[0]
        hw = x.numel() // (B * cin)
        grad_x = grad_w = None
        if ctx.small:
            g3, x3 = grad_y.reshape(B, cout, hw), x.reshape(B, cin, hw)
            if ctx.needs_input_grad[0]:
                grad_x = _product(weight.detach().reshape(cout, cin), g3, True).view_as(x)
            if ctx.needs_input_grad[1]:
                grad_w = _weight_grad(g3, x3).view_as(weight)
            return grad_x, grad_w, None
        if ctx.needs_input_grad[0]:
            if _gemm_ok(cout, hw) and _plain_gemm_mine(cout, (B, cin, hw)):
                grad_x = torch.empty_like(x)
                _api._native.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, weight.contiguous(), grad_y, grad_x)
            else:
                grad_x = _product(weight.detach().reshape(cout, cin), grad_y.reshape(B, cout, hw), True).view_as(x)
        if ctx.needs_input_grad[1]:
            if (LIBRARY_GEMMS or LIBRARY_WGRAD) and hw <= 16384 and _api._native.get_matmul_precision() == "fp32":
                # (until round 4: one batched rocBLAS product per sample and a sum for the feature-propagation layers; the
                # weight-gradient kernel has since caught up at these sizes: tools/gemm_any_compare.py)
                grad_w = torch.bmm(grad_y.reshape(B, cout, hw), x.reshape(B, cin, hw).transpose(1, 2)).sum(0)
            else:
                grad_w = zeroed_empty((cout, cin), torch.float32, x.device)
                _api._native.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, grad_y, grad_w)
            grad_w = grad_w.view_as(weight)
        return grad_x, grad_w, None


def pointwise_conv(x, conv, gn=None):
    """Apply a Conv1d/Conv2d module; 1x1, stride-1, bias-free convolutions on the GPU go through _PointwiseConv.
    gn: the GroupNorm that follows — then returns (y, stats), stats being that norm's statistics pass when the
    convolution kernel could produce it on the way (None otherwise)."""
    hw = x.numel() // max(x.shape[0] * x.shape[1], 1)
    if (x.is_cuda and x.dtype == torch.float32 and conv.bias is None and conv.groups == 1 and hw % 16 == 0
            and x.is_contiguous() and all(k == 1 for k in conv.kernel_size) and all(v == 1 for v in conv.stride)
            and all(v == 0 for v in conv.padding) and getattr(_api._native, "conv1x1_wgrad_wrapper", None) is not None):
        if gn is None:
            return _PointwiseConv.apply(x, conv.weight)
        if deterministic():   # (no statistics from the epilogue: the norm takes its own, slotted pass)
            return _PointwiseConv.apply(x, conv.weight), None
        if gn.affine:
            return _PointwiseConv.apply(x, conv.weight, gn.num_groups)
        return _PointwiseConv.apply(x, conv.weight), None
    if (x.is_cuda and x.dtype == torch.float32 and conv.groups == 1 and x.dim() >= 3
            and all(k == 1 for k in conv.kernel_size) and all(v == 1 for v in conv.stride)
            and all(v == 0 for v in conv.padding)):
        # every other 1x1 convolution (bias, strided input, ragged position count): a batched rocBLAS product through
        # autograd instead of MIOpen's convolution (see _PointwiseConv.forward)
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        y = torch.matmul(conv.weight.reshape(cout, cin), x.reshape(x.shape[0], cin, -1))
        if conv.bias is not None:
            y = y + conv.bias.view(1, cout, 1)
        y = y.view((x.shape[0], cout) + tuple(x.shape[2:]))
        return (y, None) if gn is not None else y
    return (conv(x), None) if gn is not None else conv(x)


def conv1x1_inference(x, weight2d):
    """weight2d (cout, cin) applied along the channels of x (B, cin, ...) with nothing to differentiate: _PointwiseConv's forward
    (the same kernels) on a weight that is not a module's — e.g. two blocks' weights stacked (flow_glue.stacked_weight)."""
    with torch.no_grad():
        return _PointwiseConv.apply(x.contiguous(), weight2d.view(weight2d.shape[0], weight2d.shape[1], *([1] * (x.dim() - 2))))


class _NeighbourConsistency(Function):
    """per-point mean_j ||m_i - m_idx[i,j]||_p for point-major masks — one launch forward, one backward (a gather over
    the neighbour lists and their transposes).  Reference sequence: grouping_operation, broadcast difference, norm
    over channels, mean over neighbours (losses/seg_loss_unsup.py:123-129, :152-158)."""

    @staticmethod
    def forward(ctx, mask, idx, rev_start, rev_src, rev_mult, p):
        nat = _api._native
        mask = mask.contiguous()
        B, N, C = mask.shape
        k = idx.shape[2]
        out = torch.empty(B, N, dtype=torch.float32, device=mask.device)
        nat.neighbour_consistency_fwd_wrapper(B, N, C, k, p, mask, idx, out)
        ctx.save_for_backward(mask, idx, rev_start, rev_src, rev_mult)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nat = _api._native
        mask, idx, rev_start, rev_src, rev_mult = ctx.saved_tensors
        B, N, C = mask.shape
        grad_mask = torch.empty_like(mask)
        nat.neighbour_consistency_bwd_wrapper(B, N, C, idx.shape[2], ctx.p, mask, idx, rev_start, rev_src, rev_mult,
                                              grad_out.contiguous(), grad_mask)
        return grad_mask, None, None, None, None, None


def reverse_neighbours(idx):
    """Transposed neighbour lists of idx (B, N, k) int32: (rev_start (B, N+1), rev_src (B, N*k), rev_mult (B, N)) as
    ogc_reverse_neighbours defines them.  Coordinates only — belongs to the geometry plan of a step."""
    nat = _api._native
    idx = idx.contiguous()
    B, N, k = idx.shape
    rev_start = zeroed_empty((B, N + 1), torch.int32, idx.device)
    rev_src = torch.empty(B, N * k, dtype=torch.int32, device=idx.device)
    rev_mult = torch.empty(B, N, dtype=torch.int32, device=idx.device)
    ws = torch.empty(B, N, dtype=torch.int32, device=idx.device)
    nat.reverse_neighbours_wrapper(B, N, k, idx, rev_start, rev_src, rev_mult, ws)
    return rev_start, rev_src, rev_mult


def group_reverse(idx, n):
    """Transposed lists of a grouping tensor idx (B, npoint, nsample) int32 into n source points, for the gather form of
    the grouping gradient (ogc_group_reverse): (rev_start, rev_pos).  Coordinates only — part of a step's geometry plan.
    None where the kernels do not apply."""
    nat = _api._native
    if (getattr(nat, "group_reverse_wrapper", None) is None or not idx.is_cuda or idx.dtype != torch.int32
            or n > 16384 or (idx.shape[1] * idx.shape[2]) % 16 != 0 or idx.shape[1] * idx.shape[2] == 0):
        return None
    idx = idx.contiguous()
    B, npoint, nsample = idx.shape
    T = npoint * nsample
    tc = nat.group_reverse_chunk(n, npoint, nsample)
    rev_start = torch.empty(B, (T + tc - 1) // tc, n + 1, dtype=torch.int32, device=idx.device)
    rev_pos = torch.empty(B, T, dtype=torch.int16, device=idx.device)
    heads = torch.empty(B, T // 16, dtype=torch.int16, device=idx.device)
    nat.group_reverse_wrapper(B, n, npoint, nsample, idx, rev_start, rev_pos, heads)
    return rev_start, rev_pos, heads


@_gate
def neighbour_consistency_available(mask, loss_norm, cross_entropy):
    return (mask.is_cuda and mask.dtype == torch.float32 and not cross_entropy and loss_norm in (1, 2)
            and mask.shape[-1] <= 40 and getattr(_api._native, "neighbour_consistency_fwd_wrapper", None) is not None)


def neighbour_consistency(mask, idx, reverse, p):
    """mask (B, N, C) point-major, idx (B, N, k), reverse = reverse_neighbours(idx) -> (B, N)."""
    return _NeighbourConsistency.apply(mask, idx, reverse[0], reverse[1], reverse[2], int(p))


# ---- BatchNorm flavour (FlowStep3D nets) ----------------------------------------------------------------------
def _bn_buffers(bn):
    return (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)


class _BatchNormAct(Function):
    """y = act(BatchNorm(x)), act = ReLU or identity; batch statistics in training (running statistics updated in
    place by the kernel), running statistics in evaluation.  Reference sequence: F.relu(bn(.)) in
    utils/flowstep3d_util.py:57-66, :126-138."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu, stats):
        nat = _api._native
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        hw = x.numel() // max(B * C, 1)
        y = torch.empty_like(x)
        # inference: nothing to save for a backward pass -> the kernel reads the running statistics itself (one launch)
        lean = not training and not any(ctx.needs_input_grad[:3])
        mean = None if lean else torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = None if lean else torch.empty_like(mean)
        ws = None if (stats is not None or not training) else zeroed_empty(2 * C, torch.float64, x.device)
        nat.batch_norm_fwd_wrapper(B, C, hw, eps, relu, training, momentum, x, weight.contiguous(),
                                   bias.contiguous(), running_mean, running_var, y, mean, rstd, ws, stats,
                                   0 if stats is None else stats.numel() // (2 * C))
        if not lean:
            ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.cfg = (relu, training, hw)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        nat = _api._native
        x, weight, bias, mean, rstd = ctx.saved_tensors
        relu, training, hw = ctx.cfg
        B, C = x.shape[0], x.shape[1]
        grad_x = torch.empty_like(x)
        gw = torch.empty_like(weight)
        gb = torch.empty_like(bias)
        ws = zeroed_empty(3 * C, torch.float64, x.device)   # (the library zeroes its first 2 C doubles: skipped inside a step arena)
        nat.batch_norm_bwd_wrapper(B, C, hw, relu, training, x, weight.contiguous(),
                                   bias.contiguous(), mean, rstd, grad_y.contiguous(), grad_x, gw, gb, ws)
        return grad_x, gw, gb, None, None, None, None, None, None, None


class _BatchNormActMaxPool(Function):
    """out[b,c,p] = max_s act(BatchNorm(x))[b,c,p,s] without materialising the normalised activation.
    Reference sequence: F.relu(bn(.)) then .max(dim=-1) (utils/flowstep3d_util.py:64-66, :134-136)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, relu, stats):
        nat = _api._native
        x = x.contiguous()
        B, C, P, S = x.shape
        out = torch.empty(B, C, P, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, C, P, dtype=torch.int32, device=x.device)
        lean = not training and not any(ctx.needs_input_grad[:3])
        mean = None if lean else torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = None if lean else torch.empty_like(mean)
        ws = None if (stats is not None or not training) else zeroed_empty(2 * C, torch.float64, x.device)
        nat.batch_norm_maxpool_fwd_wrapper(B, C, P, S, eps, relu, training, momentum, x, weight.contiguous(),
                                           bias.contiguous(), running_mean, running_var, out, arg, mean, rstd,
                                           ws, stats, 0 if stats is None else stats.numel() // (2 * C))
        if not lean:
            ctx.save_for_backward(x, weight, mean, rstd, out, arg)
        ctx.cfg = (relu, training)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nat = _api._native
        x, weight, mean, rstd, out, arg = ctx.saved_tensors
        relu, training = ctx.cfg
        B, C, P, S = x.shape
        grad_x = torch.empty_like(x)
        gw = torch.empty_like(weight)
        gb = torch.empty_like(weight)
        ws = zeroed_empty(3 * C, torch.float64, x.device)   # (the library zeroes its first 2 C doubles: skipped inside a step arena)
        nat.batch_norm_maxpool_bwd_wrapper(B, C, P, S, relu, training, x, weight.contiguous(), mean, rstd, out,
                                           arg, grad_out.contiguous(), grad_x, gw, gb, ws)
        return grad_x, gw, gb, None, None, None, None, None, None, None


def _bn_fusable(x, bn):
    return (isinstance(bn, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)) and x.is_cuda and x.dtype == torch.float32
            and bn.affine and bn.momentum is not None and (bn.training or bn.track_running_stats)
            and getattr(_api._native, "batch_norm_fwd_wrapper", None) is not None)


def _bn_call(fn, x, bn, relu, stats):
    training = bn.training or not bn.track_running_stats
    rm, rv = _bn_buffers(bn)
    if training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    w, b_ = bn.weight, bn.bias
    if not training and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or b_.requires_grad)):
        x, w, b_ = x.detach(), w.detach(), b_.detach()  # inference: the Functions then keep nothing for a backward pass
    return fn.apply(x, w, b_, rm, rv, training, float(bn.momentum), bn.eps, relu, stats)


def conv_norm_act(x, conv, norm, relu=True, maxpool=False):
    """act(norm(conv(x))) (+ max over the last dimension) for the conv / BatchNorm / ReLU chains of the FlowStep3D
    blocks.  On the GPU: MFMA convolution and the fused BatchNorm kernels; otherwise the reference's op sequence."""
    y = pointwise_conv(x, conv)
    if _bn_fusable(y, norm):
        S = y.shape[-1]
        if maxpool and y.dim() == 4 and 4 <= S <= 256 and S & (S - 1) == 0:
            return _bn_call(_BatchNormActMaxPool, y, norm, relu, None)
        y = _bn_call(_BatchNormAct, y, norm, relu, None)
    else:
        y = norm(y)
        if relu:
            y = F.relu(y)
    return y.max(dim=-1)[0] if maxpool else y


# ---- whole MLP + max-pool in one launch (inference) ------------------------------------------------------------------
_FOLDED = weakref.WeakKeyDictionary()   # first conv module -> (key, transposed folded weights, folded biases)
_FOLD_GENERATION = [0]


def note_training_mode():
    """Called by the FlowStep3D blocks when they enter training mode: everything folded so far is stale from now on.  The tensor
    version counters the cache also checks do not see training — the fused optimizer and the BatchNorm kernels of this library
    write parameters and running statistics through raw pointers (measured: versions unchanged across a step) — so a net that
    was evaluated, trained further and evaluated again would reuse the folded weights of the first evaluation."""
    _FOLD_GENERATION[0] += 1


def _fold_batch_norm(convs, norms):
    """W'_l = diag(a) W_l (transposed, rows padded to a multiple of 4), b_l = beta - mean * a with a = gamma / sqrt(var + eps):
    BatchNorm in evaluation mode folded into the convolution before it.  Cached per block until a parameter or buffer changes
    (torch-level writes: version counters) or any block has been in training mode since (note_training_mode)."""
    tensors = [t for conv, bn in zip(convs, norms) for t in (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    key = (_FOLD_GENERATION[0],) + tuple((t.data_ptr(), t._version) for t in tensors)
    hit = _FOLDED.get(convs[0])
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    wts, biases = [], []
    with torch.no_grad():
        for conv, bn in zip(convs, norms):
            a = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            w = conv.weight.reshape(conv.weight.shape[0], -1) * a[:, None]
            wt = torch.zeros((w.shape[1] + 3) // 4 * 4, w.shape[0], dtype=torch.float32, device=w.device)
            wt[:w.shape[1]] = w.t()
            wts.append(wt.contiguous())
            biases.append((bn.bias - bn.running_mean * a).contiguous())
    _FOLDED[convs[0]] = (key, wts, biases)
    return wts, biases


@_gate
def mlp_chain_pool_available(x, convs, norms):
    """Inference only: every norm a BatchNorm in evaluation mode, nothing to differentiate, and a kernel for the shape."""
    nat = _api._native
    if getattr(nat, "mlp_chain_pool_wrapper", None) is None or len(convs) not in (2, 3) or x.dim() != 4:
        return False
    if not (x.is_cuda and x.dtype == torch.float32) or (torch.is_grad_enabled() and (
            x.requires_grad or any(p.requires_grad for m in list(convs) + list(norms) for p in m.parameters()))):
        return False
    for conv, bn in zip(convs, norms):
        if not isinstance(bn, torch.nn.BatchNorm2d) or bn.training or not bn.track_running_stats or not bn.affine:
            return False
        if conv.bias is not None or conv.groups != 1 or any(k != 1 for k in conv.kernel_size):
            return False
    c = [convs[0].weight.shape[1]] + [conv.weight.shape[0] for conv in convs] + [0]
    return nat.mlp_chain_pool_supported(c[0], c[1], c[2], c[3] if len(convs) > 2 else 0, x.shape[-1])


@_gate
def corr_layer_pool_available(feature1, feature2, idx, convs, norms):
    """FlowEmbedding in inference with a kernel for its shape: grouping, concatenation, MLP and max in one launch."""
    nat = _api._native
    if getattr(nat, "corr_layer_pool_wrapper", None) is None or len(convs) != 3 or idx is None:
        return False
    if not all(t.is_cuda and t.dtype == torch.float32 for t in (feature1, feature2)) or feature1.shape[1] != feature2.shape[1]:
        return False
    if torch.is_grad_enabled() and (feature1.requires_grad or feature2.requires_grad or any(
            p.requires_grad for m in list(convs) + list(norms) for p in m.parameters())):
        return False
    for conv, bn in zip(convs, norms):
        if not isinstance(bn, torch.nn.BatchNorm2d) or bn.training or not bn.track_running_stats or not bn.affine:
            return False
        if conv.bias is not None or conv.groups != 1 or any(k != 1 for k in conv.kernel_size):
            return False
    cf = feature1.shape[1]
    if convs[0].weight.shape[1] != 3 + 2 * cf:
        return False
    return nat.corr_layer_pool_supported(cf, convs[0].weight.shape[0], convs[1].weight.shape[0], convs[2].weight.shape[0],
                                         idx.shape[2])


def corr_layer_pool(pos1, pos2, feature1, feature2, idx, convs, norms):
    """max_s relu(bn(conv(...[pos2[idx] - pos1, feature2[idx], feature1]...))) (B, c3, n1); see corr_layer_pool_available."""
    wts, biases = _fold_batch_norm(convs, norms)
    out = torch.empty(feature1.shape[0], wts[-1].shape[1], feature1.shape[2], dtype=torch.float32, device=feature1.device)
    _api._native.corr_layer_pool_wrapper(pos1.contiguous(), pos2.contiguous(), feature1.contiguous(), feature2.contiguous(),
                                         idx.int().contiguous(), wts, biases, out)
    return out


def mlp_chain_pool(x, convs, norms):
    """max over the last dimension of relu(bn(conv(...relu(bn(conv(x)))...))) in one launch (see mlp_chain_pool_available)."""
    wts, biases = _fold_batch_norm(convs, norms)
    x = x.contiguous()
    out = torch.empty(x.shape[0], wts[-1].shape[1], x.shape[2], dtype=torch.float32, device=x.device)
    _api._native.mlp_chain_pool_wrapper(x, wts, biases, out)
    return out


# ---- dynamic / invariance terms of the OGC loss ---------------------------------------------------------------
class _RigidBlend(Function):
    """per-point || sum_k m_k (R_k p + t_k) - q ||_p with R, t constants (the fit is detached in the reference,
    losses/seg_loss_unsup.py:91); gradient w.r.t. the mask only."""

    @staticmethod
    def forward(ctx, pc, pc2, mask, R, t, p):
        nat = _api._native
        VB, N, K = mask.shape
        out = torch.empty(VB, N, dtype=torch.float32, device=mask.device)
        nat.rigid_blend_wrapper(VB, N, K, p, 0, pc, pc2, mask, R, t, None, out)
        ctx.save_for_backward(pc, pc2, mask, R, t)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nat = _api._native
        pc, pc2, mask, R, t = ctx.saved_tensors
        VB, N, K = mask.shape
        grad_mask = torch.empty_like(mask)
        nat.rigid_blend_wrapper(VB, N, K, ctx.p, 1, pc, pc2, mask, R, t, grad_out.contiguous(), grad_mask)
        return None, None, grad_mask, None, None, None


@_gate
def rigid_residual_available(mask, loss_norm):
    return (mask.is_cuda and mask.dtype == torch.float32 and loss_norm in (1, 2) and mask.shape[-1] <= 32
            and getattr(_api._native, "rigid_blend_wrapper", None) is not None)


def rigid_residual(pc, pc2, mask, loss_norm):
    """DynamicLoss per point: fit one rigid motion per (cloud, slot) to (pc -> pc2) weighted by mask[:, :, slot]
    (detached), blend the moved clouds by the mask, residual norm against pc2.  pc, pc2 (VB, N, 3), mask (VB, N, K)
    -> (VB, N).  Five launches: moments, finalize, Kabsch, translation, blend."""
    nat = _api._native
    pc, pc2, mask = pc.contiguous(), pc2.contiguous(), mask.contiguous()
    VB, N, K = mask.shape
    dev = mask.device
    with torch.no_grad():
        mom = zeroed_empty(VB * K * 16, torch.float64, dev)
        S = torch.empty(VB * K, 3, 3, dtype=torch.float32, device=dev)
        means = torch.empty(VB * K, 6, dtype=torch.float32, device=dev)
        nat.rigid_moments_wrapper(VB, N, K, pc, pc2, mask.detach(), mom, S, means)
        R = torch.empty_like(S)
        valid = torch.empty(VB * K, dtype=torch.int32, device=dev)
        nat.kabsch_rotation_wrapper(VB * K, S, R, valid)
        t = torch.empty(VB * K, 3, dtype=torch.float32, device=dev)
        nat.rigid_translation_wrapper(VB * K, means, valid, R, t)
    return _RigidBlend.apply(pc, pc2, mask, R, t, int(loss_norm))


class _MatchedDistance(Function):
    """(||m1 - m2[:, col12]||_p, ||m2 - m1[:, col21]||_p) per point, the permuted operands detached
    (losses/seg_loss_unsup.py:252-262)."""

    @staticmethod
    def forward(ctx, mask1, mask2, col12, col21, p):
        nat = _api._native
        PB, N, K = mask1.shape
        d12 = torch.empty(PB, N, dtype=torch.float32, device=mask1.device)
        d21 = torch.empty_like(d12)
        nat.matched_distance_wrapper(PB, N, K, p, 0, mask1, mask2, col12, col21, None, None, d12, d21)
        ctx.save_for_backward(mask1, mask2, col12, col21)
        ctx.p = p
        return d12, d21

    @staticmethod
    def backward(ctx, g12, g21):
        nat = _api._native
        mask1, mask2, col12, col21 = ctx.saved_tensors
        PB, N, K = mask1.shape
        gm1, gm2 = torch.empty_like(mask1), torch.empty_like(mask2)
        nat.matched_distance_wrapper(PB, N, K, ctx.p, 1, mask1, mask2, col12, col21, g12.contiguous(), g21.contiguous(),
                                     gm1, gm2)
        return gm1, gm2, None, None, None


@_gate
def matched_distance_available(mask, loss_norm, cross_entropy):
    return (mask.is_cuda and mask.dtype == torch.float32 and not cross_entropy and loss_norm in (1, 2)
            and mask.shape[-1] <= 32 and getattr(_api._native, "matched_distance_wrapper", None) is not None
            and getattr(_api._native, "lsap_maximize_wrapper", None) is not None)


def matched_distances(mask1, mask2, loss_norm):
    """Hungarian-match the hard segmentations of mask1 / mask2 (PB, N, K) by IoU in both directions, then the
    per-point distances to the permuted other mask: (d12, d21), each (PB, N).  Four launches."""
    nat = _api._native
    mask1, mask2 = mask1.contiguous(), mask2.contiguous()
    PB, N, K = mask1.shape
    dev = mask1.device
    with torch.no_grad():
        counts = zeroed_empty(PB * K * K, torch.int32, dev)
        iou = torch.empty(PB, K, K, dtype=torch.float32, device=dev)
        nat.mask_iou_wrapper(PB, N, K, mask1.detach(), mask2.detach(), counts, iou)
        both = torch.stack([iou, iou.transpose(1, 2)]).contiguous()          # (2, PB, K, K)
        cols = torch.empty(2 * PB, K, dtype=torch.int32, device=dev)
        nat.lsap_maximize_wrapper(2 * PB, K, both, cols)
    return _MatchedDistance.apply(mask1, mask2, cols[:PB], cols[PB:], int(loss_norm))


FUSED_GN_BACKWARD_MAX_WIDTH = 64
FUSED_GN_BACKWARD = True   # _NormActConv.backward through the moment matrices (csrc/gn_fused_bwd.hip); False: the separate passes
# 16-bit activations: the moment matrices run on bf16 MFMAs (wgrad_moments16_kernel), so the second accumulator set that made the
# fp32 kernel MFMA-bound at 128 channels costs nothing there, and the moment path COULD reach 128-channel layers (C2: SA2's inner
# layer, the 64 -> 128 tail of SA1's second scale; OGC_ACT16_MOMENT_WIDTH=128).  Measured at C2: 17.0-17.1 ms per step against 16.4
# with the fp32 limits (two tile pairs re-read both tensors, the adjoint kernel holds 128 reduction channels at two wavefronts per
# SIMD) — the limit stays at 64.
ACT16_MOMENT_WIDTH = int(__import__("os").environ.get("OGC_ACT16_MOMENT_WIDTH", "64"))


def _moment_width(t):
    return ACT16_MOMENT_WIDTH if t.dtype is torch.bfloat16 else FUSED_GN_BACKWARD_MAX_WIDTH


def _pool_moment_cout(t):
    return ACT16_MOMENT_WIDTH if t.dtype is torch.bfloat16 else max(FUSED_GN_BACKWARD_MAX_WIDTH, SPARSE_POOL_MAX_COUT)


def _norm_act_conv_forward(y_prev, stats_prev, gn_weight, gn_bias, conv_weight, gn_groups, eps, relu, next_groups, pool,
                           next_gamma):
    """y = conv(act(GroupNorm(y_prev))) with the norm applied on the way in (shared by _NormActConv and _NormActConvPool).
    -> (y, statistics of y or None, neighbourhood extremes or None, mean, rstd, a, bb of the norm of y_prev)."""
    nat = _api._native
    B, cin = y_prev.shape[0], y_prev.shape[1]
    cout = conv_weight.shape[0]
    hw = y_prev.numel() // (B * cin)
    dev = y_prev.device
    mean = torch.empty(B * gn_groups, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    a = torch.empty(B * cin, dtype=torch.float32, device=dev)
    bb = torch.empty_like(a)
    gamma, beta = gn_weight.contiguous(), gn_bias.contiguous()
    if stats_prev is not None:
        nat.group_norm_coeffs_wrapper(B, cin, hw, gn_groups, eps, None, gamma, beta, stats_prev,
                                      stats_prev.numel() // (2 * B * gn_groups), None, mean, rstd, a, bb)
    else:
        ws = nat.group_norm_ws(B, cin, gn_groups, False, dev)
        nat.group_norm_coeffs_wrapper(B, cin, hw, gn_groups, eps, y_prev, gamma, beta, None, 0, ws, mean, rstd, a, bb)
    y = torch.empty((B, cout) + tuple(y_prev.shape[2:]), dtype=y_prev.dtype, device=dev)
    w = conv_weight.contiguous()
    stats = extremes = None
    if (next_groups > 0 and next_groups <= 32 and cout % next_groups == 0 and (cout // next_groups) % 4 == 0
            and (cin <= 100 or _stats_ok(nat, B, cout, cin, hw, True))):
        stats = zeroed_empty(nat.conv1x1_gn_slots() * B * next_groups * 2, torch.float64, dev)
        if (pool and next_gamma is not None and (cin <= 100 or POOL_EXTREMES_WIDE)
                and getattr(nat, "conv1x1_gemm_affine_pool_wrapper", None) is not None):
            # last layer of a set-abstraction MLP: also the extreme of every neighbourhood, for the max-pool
            centres = hw // pool
            yext = torch.empty(B, cout, centres, dtype=torch.float32, device=dev)
            aext = torch.empty(B, cout, centres, dtype=torch.int32, device=dev)
            nat.conv1x1_gemm_affine_pool_wrapper(B, cout, cin, hw, relu, next_groups, pool, w, y_prev, a, bb,
                                                 next_gamma.contiguous(), y, stats, yext, aext)
            extremes = (yext, aext)
        else:
            nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, relu, next_groups, w, y_prev, a, bb, y, stats)
    else:
        nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, relu, 0, w, y_prev, a, bb, y, None)
    return y, stats, extremes, mean, rstd, a, bb


# ---- deferred normalisation inside a SharedMLP ----------------------------------------------------------------
class _NormActConv(Function):
    """y = conv(act(GroupNorm(y_prev))) WITHOUT materialising the normalised activation: the norm of the previous layer
    is applied while the convolution loads its input (ogc_conv1x1_gemm_affine), recomputed the same way by the weight
    gradient (ogc_conv1x1_wgrad_affine); the input gradient continues through the GroupNorm backward kernels.
    Returns (y, statistics of y for the NEXT GroupNorm or None)."""

    @staticmethod
    def forward(ctx, y_prev, stats_prev, gn_weight, gn_bias, conv_weight, gn_groups, eps, relu, next_groups, pool=0,
                next_gamma=None):
        ctx.set_materialize_grads(False)  # no zero tensor (one fill launch per layer) for the statistics output
        y_prev = y_prev.contiguous()
        y, stats, extremes, mean, rstd, a, bb = _norm_act_conv_forward(y_prev, stats_prev, gn_weight, gn_bias, conv_weight,
                                                                       gn_groups, eps, relu, next_groups, pool, next_gamma)
        hw = y_prev.numel() // (y_prev.shape[0] * y_prev.shape[1])
        ctx.save_for_backward(y_prev, gn_weight, gn_bias, conv_weight, mean, rstd, a, bb)
        ctx.cfg = (gn_groups, relu, hw)
        if extremes is not None:
            ctx.mark_non_differentiable(stats, *extremes)
            return (y, stats) + extremes
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, grad_y, _grad_stats=None, *_grad_extremes):
        if grad_y is None:
            return (None,) * 11
        nat = _api._native
        y_prev, gn_weight, gn_bias, conv_weight, mean, rstd, a, bb = ctx.saved_tensors
        gn_groups, relu, hw = ctx.cfg
        B, cin = y_prev.shape[0], y_prev.shape[1]
        cout = conv_weight.shape[0]
        grad_y = grad_y.contiguous()
        w = conv_weight.contiguous()
        grad_w = zeroed_empty((cout, cin), torch.float32, y_prev.device)
        # (up to 64 channels: from 128 on the second accumulator set makes the weight-gradient kernel MFMA-bound and the
        # whole path slower than the separate passes — tools/gn_bwd_compare.py)
        if (FUSED_GN_BACKWARD and getattr(nat, "conv1x1_dgrad_adjoint_wrapper", None) is not None and hw % 64 == 0
                and cout <= _moment_width(y_prev) and cin <= _moment_width(y_prev) and gn_groups <= 32
                and cin % gn_groups == 0):
            # (fp32 operands also under `matmul_precision: bf16`: these layers are HBM-bound, bf16 operands buy nothing here)
            # moment matrices next to the weight gradient -> GroupNorm sums -> adjoint in the input gradient's epilogue:
            # the gradient w.r.t. the normalised activation and both GroupNorm backward passes never touch memory
            dev = y_prev.device
            moments = zeroed_empty((B, 2, cout, cin), torch.float32, dev)
            nat.conv1x1_wgrad_moments_wrapper(B, cin, cout, hw, relu, y_prev, a, bb, grad_y, moments)
            coef = torch.empty(B, cin, 3, dtype=torch.float32, device=dev)
            ggb = zeroed_empty((2, cin), torch.float32, dev)
            gw, gb = ggb[0], ggb[1]
            nat.gn_moments_combine_wrapper(B, cin, cout, hw, gn_groups, moments, w.view(cout, cin), a, bb, mean, rstd,
                                           gn_weight.contiguous(), grad_w, coef, gw, gb)
            grad_prev = torch.empty_like(y_prev)
            nat.conv1x1_dgrad_adjoint_wrapper(B, cin, cout, hw, relu, w.view(cout, cin), grad_y, y_prev, a, bb, coef, grad_prev)
            return grad_prev, None, gw, gb, grad_w.view_as(conv_weight), None, None, None, None, None, None
        nat.conv1x1_wgrad_affine_wrapper(B, cin, cout, hw, relu, y_prev, a, bb, grad_y, grad_w)
        # gradient w.r.t. the (never stored) normalised activation, then through GroupNorm (+ ReLU)
        if y_prev.dtype is torch.bfloat16 or (_gemm_ok(cout, hw) and _plain_gemm_mine(cout, (B, cin, hw))):
            grad_z = torch.empty_like(y_prev)
            nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, grad_y, grad_z)
        else:
            grad_z = _product(w.view(cout, cin), grad_y.reshape(B, cout, hw), True).view_as(y_prev)
        grad_prev = torch.empty_like(y_prev)
        gw, gb = torch.empty_like(gn_weight), torch.empty_like(gn_bias)
        ws = nat.group_norm_ws(B, cin, gn_groups, True, y_prev.device)
        nat.group_norm_bwd_wrapper(B, cin, hw, gn_groups, relu, y_prev, gn_weight.contiguous(),
                                   gn_bias.contiguous(), mean, rstd, grad_z, grad_prev, gw, gb, ws)
        return grad_prev, None, gw, gb, grad_w.view_as(conv_weight), None, None, None, None, None, None


@_gate
def norm_act_conv_available(y_prev, gn, conv):
    """Can conv(act(gn(y_prev))) run with the norm folded into the convolution's operand load?"""
    if deterministic():   # (its epilogue statistics and moment matrices are atomic sums: see deterministic())
        return False
    if not (y_prev.is_cuda and _is_act(y_prev) and gn.affine and conv.bias is None and conv.groups == 1
            and all(k == 1 for k in conv.kernel_size) and all(v == 1 for v in conv.stride)
            and all(v == 0 for v in conv.padding)
            and getattr(_api._native, "conv1x1_gemm_affine_wrapper", None) is not None):
        return False
    hw = y_prev.numel() // max(y_prev.shape[0] * y_prev.shape[1], 1)
    if y_prev.dtype is torch.bfloat16 and _api._native.get_matmul_precision() != "bf16":
        return False   # (the 16-bit forms come with bf16 operands)
    return _gemm_ok(y_prev.shape[1], hw)


def act16_middle_ok(y_prev, gn, conv):
    """May a MIDDLE layer of a shared MLP take a 16-bit y_prev?  (_NormActConv: beyond the moment-matrix widths its input
    gradient is the tile kernel's, which holds <= 160 reduction channels.)"""
    return norm_act_conv_available(y_prev, gn, conv) and conv.weight.shape[0] <= 160


def norm_act_conv(y_prev, stats_prev, gn, relu, conv, next_gn=None, pool=0):
    """(y, stats of y for next_gn or None) = conv(act(gn(y_prev))), see _NormActConv.  pool = nsample (16, 32 or 64) on
    the LAST layer of a set-abstraction MLP, y_prev (B, C, npoint, nsample): a third result, the extremes of y over
    each neighbourhood for group_norm_act_maxpool (None when the kernel does not offer them for this shape)."""
    next_groups = next_gn.num_groups if (next_gn is not None and next_gn.affine) else 0
    res = _NormActConv.apply(y_prev, stats_prev, gn.weight, gn.bias, conv.weight, gn.num_groups, gn.eps, bool(relu),
                             next_groups, int(pool) if (pool in (16, 32, 64) and next_groups) else 0,
                             next_gn.weight if next_groups else None)
    if pool:
        return res[0], res[1], (tuple(res[2:]) if len(res) > 2 else None)
    return res[0], res[1]


# The tail of a set-abstraction MLP — last convolution, its GroupNorm / ReLU, the max over the neighbourhood — as ONE autograd
# node, so that the gradient w.r.t. the convolution's output (the size of the level's widest activation) is never written: it
# is affine in that output except at one arg-max position per neighbourhood, and the weight- / input-gradient kernels rebuild
# it from (c2, c3) per channel and (value, position) per neighbourhood while they load the output (gn_fused_bwd.hip,
# *_pooled).  Bit-identical to the two-node sequence (tests/test_ops_gpu.py::test_pooled_tail_backward); C4: the pass it
# removes is 0.28 ms of a step at SA1's two scales.  False: the two nodes.
SPARSE_POOL_BACKWARD = True
# widest convolution output of such a tail (input width: FUSED_GN_BACKWARD_MAX_WIDTH).  128 (SA2's 64 -> 128 tail through the
# moment matrices with 64 x 32 tiles) is 0.95 against 0.98-1.01 ms in isolation (tools/pool_tail_compare.py) but 11.62 against
# 11.56 ms in the step (tools/step_ab.py SPARSE_POOL_MAX_COUT 3 128 64): left at 64.
SPARSE_POOL_MAX_COUT = 64
# Tails wider than that keep the pooled gradient sparse as well (round 4), through the pooled forms of the PLAIN weight- and input-
# gradient kernels instead of the moment-matrix pair (OGC_WIDE_POOL=0: the dense gradient, for A/B runs).
WIDE_POOL_BACKWARD = _os.environ.get("OGC_WIDE_POOL", "1") != "0"

class _NormActConvPool(Function):
    """out (B, cout, P) = max over the neighbourhood of act2(GroupNorm2(conv(act(GroupNorm(y_prev))))), y_prev (B, cin, P, S)
    the raw output of the previous convolution (reference sequence: utils/pointnet2_util.py:38-42 on the last SharedMLP
    layer, utils/nn_util.py:45-85)."""

    @staticmethod
    def forward(ctx, y_prev, stats_prev, gn_weight, gn_bias, conv_weight, gn_groups, eps, relu, gn2_weight, gn2_bias,
                groups2, eps2, relu2):
        nat = _api._native
        y_prev = y_prev.contiguous()
        B, cin, P, S = y_prev.shape
        cout = conv_weight.shape[0]
        y, stats, extremes, mean, rstd, a, bb = _norm_act_conv_forward(y_prev, stats_prev, gn_weight, gn_bias, conv_weight,
                                                                       gn_groups, eps, relu, groups2, S, gn2_weight)
        dev = y_prev.device
        out = torch.empty(B, cout, P, dtype=torch.float32, device=dev)
        arg = torch.empty(B, cout, P, dtype=torch.int32, device=dev)
        mean2 = torch.empty(B * groups2, dtype=torch.float32, device=dev)
        rstd2 = torch.empty_like(mean2)
        g2, b2 = gn2_weight.contiguous(), gn2_bias.contiguous()
        nat.group_norm_pool_extremes_wrapper(B, cout, P, S, groups2, eps2, relu2, extremes[0], extremes[1], g2, b2, out, arg,
                                             mean2, rstd2, stats, stats.numel() // (2 * B * groups2))
        ctx.save_for_backward(y_prev, gn_weight, gn_bias, conv_weight, mean, rstd, a, bb, y, gn2_weight, mean2, rstd2, out, arg,
                              extremes[0])
        ctx.cfg = (gn_groups, relu, groups2, relu2)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        nat = _api._native
        (y_prev, gn_weight, gn_bias, conv_weight, mean, rstd, a, bb, y, gn2_weight, mean2, rstd2, out, arg,
         yext) = ctx.saved_tensors
        gn_groups, relu, groups2, relu2 = ctx.cfg
        B, cin, P, S = y_prev.shape
        cout, hw, dev = conv_weight.shape[0], P * S, y_prev.device
        # pooled GroupNorm backward in sparse form
        coef2 = torch.empty(B, cout, 2, dtype=torch.float32, device=dev)
        inj = torch.empty(B, cout, P, 2, dtype=torch.float32, device=dev)
        gw2, gb2 = torch.empty_like(gn2_weight), torch.empty_like(gn2_weight)
        ws = nat.group_norm_ws(B, cout, groups2, True, dev)
        nat.group_norm_maxpool_bwd_sparse_wrapper(B, cout, P, S, groups2, relu2, y, gn2_weight.contiguous(), mean2,
                                                  rstd2, out, arg, grad_out.contiguous(), coef2, inj, gw2, gb2, ws,
                                                  yext if POOL_SUMS_FROM_EXTREMES else None)
        w = conv_weight.contiguous().view(cout, cin)
        if cin > _moment_width(y_prev) or cout > _pool_moment_cout(y_prev):
            # wide tails (SA2 64 -> 128, SA3 128 -> 256 at C4): weight and input gradient rebuild g_y from (y, coef2, inj) while
            # they load y (round 4: ogc_conv1x1_wgrad_affine_pooled, ogc_conv1x1_dgrad_pooled) — the pass that wrote the dense g_y
            # and the two that read it become two that read y; the GroupNorm of y_prev keeps its own backward kernels
            grad_w = zeroed_empty((cout, cin), torch.float32, dev)
            nat.conv1x1_wgrad_affine_pooled_wrapper(B, cin, cout, hw, relu, S, y_prev, a, bb, y, coef2, inj, grad_w)
            grad_z = torch.empty_like(y_prev)
            nat.conv1x1_dgrad_pooled_wrapper(B, cin, cout, hw, S, w, y, coef2, inj, grad_z)
            grad_prev = torch.empty_like(y_prev)
            gw, gb = torch.empty_like(gn_weight), torch.empty_like(gn_bias)
            ws = nat.group_norm_ws(B, cin, gn_groups, True, dev)
            nat.group_norm_bwd_wrapper(B, cin, hw, gn_groups, relu, y_prev, gn_weight.contiguous(), gn_bias.contiguous(), mean,
                                       rstd, grad_z, grad_prev, gw, gb, ws)
            return (grad_prev, None, gw, gb, grad_w.view_as(conv_weight), None, None, None, gw2, gb2, None, None, None)
        # the convolution's backward (as _NormActConv.backward's moment-matrix path) on that form
        moments = zeroed_empty((B, 2, cout, cin), torch.float32, dev)
        nat.conv1x1_wgrad_moments_pooled_wrapper(B, cin, cout, hw, relu, S, y_prev, a, bb, y, coef2, inj, moments)
        coef = torch.empty(B, cin, 3, dtype=torch.float32, device=dev)
        ggb = zeroed_empty((2, cin), torch.float32, dev)
        gw, gb = ggb[0], ggb[1]
        grad_w = torch.empty(cout, cin, dtype=torch.float32, device=dev)
        nat.gn_moments_combine_wrapper(B, cin, cout, hw, gn_groups, moments, w, a, bb, mean, rstd,
                                       gn_weight.contiguous(), grad_w, coef, gw, gb)
        grad_prev = torch.empty_like(y_prev)
        nat.conv1x1_dgrad_adjoint_pooled_wrapper(B, cin, cout, hw, relu, S, w, y, coef2, inj, y_prev, a, bb, coef, grad_prev)
        return (grad_prev, None, gw, gb, grad_w.view_as(conv_weight), None, None, None, gw2, gb2, None, None, None)


@_gate
def norm_act_conv_pool_available(y_prev, gn, conv, next_gn):
    """Can the last layer of a set-abstraction MLP and its pooled GroupNorm run as _NormActConvPool?  (The shapes for which
    _NormActConv takes the extremes path forwards and the moment-matrix path backwards.)"""
    nat = _api._native
    if not (SPARSE_POOL_BACKWARD and FUSED_GN_BACKWARD and y_prev.dim() == 4 and y_prev.shape[-1] in (16, 32, 64)
            and next_gn is not None and next_gn.affine and norm_act_conv_available(y_prev, gn, conv)
            and getattr(nat, "conv1x1_dgrad_adjoint_pooled_wrapper", None) is not None
            and getattr(nat, "conv1x1_gemm_affine_pool_wrapper", None) is not None):
        return False
    cin, cout, g, g2 = y_prev.shape[1], conv.weight.shape[0], gn.num_groups, next_gn.num_groups
    if cin > _moment_width(y_prev) or cout > _pool_moment_cout(y_prev):
        # wide tails: the pooled forms of the plain weight / input gradient kernels (fp32 operands, enough position tiles for
        # the chunked kernel, neighbourhood extremes from the forward convolution)
        B, hw = y_prev.shape[0], y_prev.shape[2] * y_prev.shape[3]
        return (WIDE_POOL_BACKWARD and getattr(nat, "conv1x1_dgrad_pooled_wrapper", None) is not None
                and hw % 64 == 0 and B * (hw // 64) >= 1024 and B <= 65535   # (any precision: the pooled forms keep fp32 operands)
                and g <= 32 and cin % g == 0 and g2 <= 32 and cout % g2 == 0 and (cout // g2) % 4 == 0
                and (cin <= 100 or (POOL_EXTREMES_WIDE and _stats_ok(nat, B, cout, cin, hw, True))))
    # LDS of ogc_conv1x1_dgrad_adjoint_pooled: the weight tile + one (scale, offset, injection, arg-max) entry per wave,
    # output channel and neighbourhood of a 64-position tile; the kernel keeps the default 64 KiB
    kq = (cout + 3) // 4
    lds = (64 * 4 * (kq | 1) + 320) * 4 + 4 * cout * (64 // y_prev.shape[-1]) * 16
    return (lds <= 65536 and (y_prev.shape[2] * y_prev.shape[3]) % 64 == 0 and cin <= _moment_width(y_prev)
            and cout <= _pool_moment_cout(y_prev) and g <= 32 and cin % g == 0
            and g2 <= 32 and cout % g2 == 0 and (cout // g2) % 4 == 0)


def norm_act_conv_pool(y_prev, stats_prev, gn, relu, conv, next_gn, next_relu):
    return _NormActConvPool.apply(y_prev, stats_prev, gn.weight, gn.bias, conv.weight, gn.num_groups, gn.eps, bool(relu),
                                  next_gn.weight, next_gn.bias, next_gn.num_groups, next_gn.eps, bool(next_relu))


class _SelfAttentionCore(Function):
    """softmax(Q K^T / sqrt(d)) V with Q, K, V the three column blocks of one packed projection (B, L, 3E) — the core
    of nn.MultiheadAttention's self-attention — as one launch each way (ogc_attention_fwd / _bwd) instead of the head
    split / merge copies, two batched GEMMs, scaling and softmax torch runs.  Returns (B, L, E), heads merged."""

    @staticmethod
    def forward(ctx, qkv, n_head):
        nat = _api._native
        qkv = qkv.contiguous()
        B, L, E3 = qkv.shape
        E = E3 // 3
        out = torch.empty(B, L, E, dtype=torch.float32, device=qkv.device)
        prob = torch.empty(B, n_head, L, L, dtype=torch.float32, device=qkv.device)
        scale = (E // n_head) ** -0.5
        nat.attention_fwd_wrapper(n_head, scale, qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], out, prob)
        ctx.save_for_backward(qkv, out, prob)
        ctx.cfg = (n_head, scale)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        qkv, out, prob = ctx.saved_tensors
        n_head, scale = ctx.cfg
        E = qkv.shape[2] // 3
        grad = torch.empty_like(qkv)
        _api._native.attention_bwd_wrapper(n_head, scale, qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:], out, prob,
                                           grad_out.contiguous(), grad[:, :, :E], grad[:, :, E:2 * E], grad[:, :, 2 * E:])
        return grad, None


class _CrossAttentionCore(Function):
    """As _SelfAttentionCore with the queries (B, Lq, E) projected from one tensor and keys / values the two column
    blocks of a packed projection (B, Lk, 2E) of another."""

    @staticmethod
    def forward(ctx, q, kv, n_head):
        nat = _api._native
        q, kv = q.contiguous(), kv.contiguous()
        B, Lq, E = q.shape
        Lk = kv.shape[1]
        out = torch.empty(B, Lq, E, dtype=torch.float32, device=q.device)
        prob = torch.empty(B, n_head, Lq, Lk, dtype=torch.float32, device=q.device)
        scale = (E // n_head) ** -0.5
        nat.attention_fwd_wrapper(n_head, scale, q, kv[:, :, :E], kv[:, :, E:], out, prob)
        ctx.save_for_backward(q, kv, out, prob)
        ctx.cfg = (n_head, scale)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, kv, out, prob = ctx.saved_tensors
        n_head, scale = ctx.cfg
        E = q.shape[2]
        gq, gkv = torch.empty_like(q), torch.empty_like(kv)
        _api._native.attention_bwd_wrapper(n_head, scale, q, kv[:, :, :E], kv[:, :, E:], out, prob, grad_out.contiguous(),
                                           gq, gkv[:, :, :E], gkv[:, :, E:])
        return gq, gkv, None


class _SlotMasks(Function):
    """softmax_k(normalize(feats, 1)^T normalize(slots, 1) / temperature): the mask read-out of the segmentation nets
    (models/segnet_kitti.py:85-88) as one launch forward and two backward (ogc_slot_masks_fwd / _bwd) instead of the
    two normalisations, the einsum, the scaling and the softmax with their adjoints (~35 launches, nine passes over the
    (B, D, N) features or their gradient)."""

    @staticmethod
    def forward(ctx, feats, slots, temperature):
        feats, slots = feats.contiguous(), slots.contiguous()
        B, D, N = feats.shape
        mask = torch.empty(B, N, slots.shape[2], dtype=torch.float32, device=feats.device)
        _api._native.slot_masks_fwd_wrapper(temperature, feats, slots, mask)
        ctx.save_for_backward(feats, slots, mask)
        ctx.temperature = temperature
        return mask

    @staticmethod
    def backward(ctx, grad_mask):
        feats, slots, mask = ctx.saved_tensors
        gf, gs = torch.empty_like(feats), torch.empty_like(slots)
        _api._native.slot_masks_bwd_wrapper(ctx.temperature, feats, slots, mask, grad_mask.contiguous(), gf, gs)
        return gf, gs, None


@_gate
def slot_masks_available(feats, slots):
    """The fused mask read-out takes fp32 device tensors with at most 32 slots and 256 feature channels."""
    return (getattr(_api._native, "slot_masks_fwd_wrapper", None) is not None and feats.dim() == 3 and slots.dim() == 3
            and feats.shape[1] == slots.shape[1] <= 256 and 0 < slots.shape[2] <= 32 and feats.shape[2] > 0
            and all(t.is_cuda and t.dtype == torch.float32 for t in (feats, slots)))


def slot_masks(feats, slots, temperature=0.05):
    """(B, N, K) soft masks from point features (B, D, N) and slot embeddings (B, D, K)."""
    return _SlotMasks.apply(feats, slots, float(temperature))


@_gate
def attention_core_available(embed_dim, n_head, lq, lk, *tensors):
    """The fused attention core handles heads of 16 or 32 columns and needs lq * lk floats of LDS in its backward."""
    nat = _api._native
    return (getattr(nat, "attention_fwd_wrapper", None) is not None and embed_dim % n_head == 0
            and embed_dim // n_head in (16, 32) and 4 * (lq * lk + 2 * lq * (embed_dim // n_head) + lq + 4) <= 64 * 1024
            and all(t.is_cuda and t.dtype == torch.float32 for t in tensors))


class _SmallLinear(Function):
    """F.linear on a few hundred rows as one launch forward and one backward (ogc_small_linear_fwd / _bwd): the slot
    branch's projections and feed-forward layers are 160 x 128 x 128 products, for which the vendor GEMM costs ~15 us a
    call and autograd's backward is two of them plus a bias reduction."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        weight = weight.contiguous()
        y = torch.empty(x2.shape[0], weight.shape[0], dtype=torch.float32, device=x.device)
        _api._native.small_linear_fwd_wrapper(x2, weight, None if bias is None else bias.contiguous(), y)
        ctx.save_for_backward(x2, weight)
        ctx.shape = x.shape
        ctx.has_bias = bias is not None
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.shape[-1]).contiguous()
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        gb = torch.empty(weight.shape[0], dtype=torch.float32, device=gy.device) \
            if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if gx is not None or gw is not None or gb is not None:
            _api._native.small_linear_bwd_wrapper(x2, weight, gy2, gx, gw, gb)
        return (None if gx is None else gx.view(ctx.shape)), gw, gb


_SMALL_LINEAR_ROWS = 1024


def small_linear(x, weight, bias=None):
    """F.linear(x, weight, bias); on the GPU, fp32 and at most 1024 rows, the single-launch kernels."""
    if (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() >= 2
            and 0 < x.numel() // x.shape[-1] <= _SMALL_LINEAR_ROWS
            and getattr(_api._native, "small_linear_fwd_wrapper", None) is not None):
        return _SmallLinear.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class _ManyRowsLinear(Function):
    """F.linear on a (B, L, E) input with many rows (the key / value projection of the point memory: 16 x 512 rows).
    Forward and input gradient are the library GEMMs F.linear runs; the weight gradient, a (out x B*L) by (B*L x E)
    product with a 128 x 256 result, is split over the batch (torch.bmm, then a sum over B) instead of one GEMM whose
    32 output tiles leave the GPU empty (58 us per call on the slot branch's critical path, ~12 us this way)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gy.matmul(weight)
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(gy.transpose(1, 2), x).sum(0)
        if ctx.needs_input_grad[2]:
            gb = gy.sum(1).sum(0)   # two stages: B x out partial sums first (one long-column reduction is ~4x slower)
        return gx, gw, gb


def many_rows_linear(x, weight, bias):
    """F.linear(x, weight, bias); on the GPU, for (B, L, E) inputs with >= 4096 rows, with the batch-split weight
    gradient of _ManyRowsLinear."""
    if x.dim() == 3 and x.is_cuda and x.shape[0] > 1 and x.shape[0] * x.shape[1] >= 4096 and torch.is_grad_enabled():
        return _ManyRowsLinear.apply(x, weight, bias)
    return small_linear(x, weight, bias)


class _CrossProjections(Function):
    """Query projection of the slots and packed key / value projection of the point memory with the module's ONE
    in_proj_weight / in_proj_bias: (q, kv) = (query W[:E]^T + b[:E], memory W[E:]^T + b[E:]).  As separate F.linear
    calls on slices of the parameters, autograd embeds each slice's gradient in a zero tensor and adds the two (ten
    small launches per layer on the slot branch's backward chain); here the backward writes both halves of one
    gradient tensor in place."""

    @staticmethod
    def forward(ctx, query, memory, weight, bias):
        nat = _api._native
        E = weight.shape[1]
        q2 = query.reshape(-1, E).contiguous()
        q = torch.empty(q2.shape[0], E, dtype=torch.float32, device=query.device)
        nat.small_linear_fwd_wrapper(q2, weight[:E], bias[:E], q)
        kv = F.linear(memory, weight[E:], bias[E:])
        ctx.save_for_backward(q2, memory, weight)
        ctx.qshape = query.shape
        return q.view(query.shape), kv

    @staticmethod
    def backward(ctx, gq, gkv):
        q2, memory, weight = ctx.saved_tensors
        E = weight.shape[1]
        gw = torch.empty_like(weight)
        gb = torch.empty(weight.shape[0], dtype=torch.float32, device=weight.device)
        gquery = torch.empty_like(q2) if ctx.needs_input_grad[0] else None
        _api._native.small_linear_bwd_wrapper(q2, weight[:E], gq.reshape(-1, E).contiguous(), gquery, gw[:E], gb[:E])
        gmem = gkv.matmul(weight[E:]) if ctx.needs_input_grad[1] else None
        torch.sum(torch.bmm(gkv.transpose(1, 2), memory), 0, out=gw[E:])
        torch.sum(gkv.sum(1), 0, out=gb[E:])   # two stages, as in _ManyRowsLinear
        return (None if gquery is None else gquery.view(ctx.qshape)), gmem, gw, gb


def multihead_attention(mha, query, key, value):
    """``mha(query, key, value, need_weights=False)[0]`` for a batch-first nn.MultiheadAttention without masks or
    dropout (the use in utils/transformer_util.py:39-47), with the module's own parameters: the three projections stay
    GEMMs (packed: one for self-attention, one for the queries and one for keys + values when key is value), the core
    between them is the fused kernel.  Falls back to the module itself where that does not apply."""
    E, H = mha.embed_dim, mha.num_heads
    ok = (mha.batch_first and mha._qkv_same_embed_dim and mha.in_proj_bias is not None and mha.dropout == 0.0
          and mha.bias_k is None and not mha.add_zero_attn and query.dim() == 3 and key is value
          and attention_core_available(E, H, query.shape[1], key.shape[1], query, key))
    if not ok:
        return mha(query, key, value, need_weights=False)[0]
    W, b = mha.in_proj_weight, mha.in_proj_bias
    if query is key:
        core = _SelfAttentionCore.apply(small_linear(query, W, b), H)
    else:
        if query.numel() // E <= _SMALL_LINEAR_ROWS and torch.is_grad_enabled():
            q_proj, kv_proj = _CrossProjections.apply(query, key, W, b)
        else:
            q_proj, kv_proj = small_linear(query, W[:E], b[:E]), many_rows_linear(key, W[E:], b[E:])
        core = _CrossAttentionCore.apply(q_proj, kv_proj, H)
    return small_linear(core, mha.out_proj.weight, mha.out_proj.bias)


class _GroupedFirstLayer(Function):
    """conv(cat([xyz[idx] - new_xyz, features[idx]])) — grouping followed by the first (bias-free 1x1) convolution of a
    set-abstraction MLP — without the grouped tensor: the layer is linear, so its feature columns are applied per POINT
    (P = W_f f, a small GEMM) and gathered, the three xyz columns are applied to the relative coordinates, which are
    formed first as in the reference (ogc_group_linear_fwd).  Returns (y (B, M, npoint, nsample), GroupNorm statistics of
    y or None).  Backward from existing operators: dP = scatter-add of dy (group_points_grad), then two small GEMMs for
    d features and d W_f; d W_xyz is a weight gradient with three input channels."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx, weight, gn_groups, rev_start=None, rev_pos=None, rev_heads=None, act16=False):
        nat = _api._native
        ctx.set_materialize_grads(False)
        ctx.rev = (rev_start, rev_pos, rev_heads) if rev_start is not None else None
        B, C, N = features.shape
        npoint, nsample = idx.shape[1], idx.shape[2]
        M = weight.shape[0]
        w = weight.detach().reshape(M, 3 + C)
        wx, wf = w[:, :3].contiguous(), w[:, 3:].contiguous()
        rel = torch.empty(B, 3, npoint, nsample, dtype=torch.float32, device=xyz.device)
        nat.group_concat_wrapper(B, 0, N, npoint, nsample, xyz, new_xyz, None, idx, rel)
        y = torch.empty(B, M, npoint, nsample, dtype=torch.bfloat16 if act16 else torch.float32, device=xyz.device)
        stats = None
        if gn_groups > 0:
            stats = zeroed_empty(nat.conv1x1_gn_slots() * B * gn_groups * 2, torch.float64, xyz.device)
        cg = M // gn_groups if gn_groups > 0 else 64
        direct = GROUP_LINEAR_DIRECT and C <= 4 and getattr(nat, "group_linear_fwd_direct_wrapper", None) is not None
        P = None if direct else _product(wf, features.detach(), False)              # (B, M, N)
        if direct:
            # few feature channels (an encoder's first level): no point-wise product, one k-ascending chain per output
            # (ogc_group_linear_fwd_direct) — the rounding of the reference's convolution over the concatenated channels
            nat.group_linear_fwd_direct_wrapper(B, M, C, N, npoint, nsample, gn_groups, features.detach().contiguous(), idx, rel,
                                                w.contiguous(), y, stats)
        elif (act16 and GROUP_LINEAR_POINT_MAJOR and M % 64 == 0 and cg in (16, 32, 64) and (npoint * nsample) % 2 == 0
                and getattr(nat, "group_linear_fwd_pt_wrapper", None) is not None):
            # 16-bit output: P point-major, so that a position's channels are one contiguous read (csrc/gather_group.hip)
            nat.group_linear_fwd_pt_wrapper(B, M, N, npoint, nsample, gn_groups, P.transpose(1, 2).contiguous(), idx, rel, wx, y, stats)
        else:
            nat.group_linear_fwd_wrapper(B, M, N, npoint, nsample, gn_groups, P, idx, rel, wx, y, stats)
        ctx.save_for_backward(features, idx, rel, weight, wf)   # (wf: the backward pass would make the same copy again)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    def backward(ctx, grad_y, _grad_stats=None):
        if grad_y is None:
            return (None,) * 10
        nat = _api._native
        features, idx, rel, weight, wf = ctx.saved_tensors
        B, C, N = features.shape
        npoint, nsample = idx.shape[1], idx.shape[2]
        M = weight.shape[0]
        grad_y = grad_y.contiguous()
        T = npoint * nsample
        # the gather wins where lists are long and planes many (C4: SA2, SA3: 0.50 -> 0.13 ms, 0.33 -> 0.07); with ~16 entries
        # per point (SA1: 8192 points) the per-chunk list headers cost as much as the data and the atomic kernel stays
        gather = ctx.rev is not None and GROUP_GRAD_GATHER and T >= GROUP_GRAD_GATHER_MIN_FANIN * N and B * M >= 256
        if grad_y.dtype is torch.bfloat16 and not gather:
            grad_y = act16_leave(grad_y)   # (the scatter forms read fp32)
        one_pass = not gather and ctx.needs_input_grad[4] and N <= 16384 and T >= 4096 and T % 16 == 0
        if gather:
            # dP as a gather over the transposed neighbour lists of the geometry plan: no atomics, no zero fill; the three
            # xyz columns of the weight gradient are a three-channel weight gradient (second read of grad_y, one of rel)
            dP = torch.empty(B, M, N, dtype=torch.float32, device=grad_y.device)
            if (ctx.needs_input_grad[4] and GROUP_REV_DWX and getattr(nat, "group_points_grad_rev_dwx_wrapper", None) is not None):
                # ... and the three coordinate columns of the weight gradient from the same pass over grad_y
                dwx = zeroed_empty((M, 3), torch.float32, grad_y.device)
                nat.group_points_grad_rev_dwx_wrapper(B, M, N, npoint, nsample, grad_y, ctx.rev[0], ctx.rev[1], ctx.rev[2], rel, dP, dwx)
                one_pass = True   # (dwx is made)
            else:
                nat.group_points_grad_rev_wrapper(B, M, N, npoint, nsample, grad_y, ctx.rev[0], ctx.rev[1], ctx.rev[2], dP)
        else:
            # dP and the three xyz columns of the weight gradient share one pass over grad_y where the scatter kernel's
            # LDS path applies; otherwise the scatter-add and a three-channel weight gradient
            zeroed = torch.zeros(B * M * N + M * 3, dtype=torch.float32, device=grad_y.device)
            dP, dwx = zeroed[:B * M * N].view(B, M, N), zeroed[B * M * N:].view(M, 3)
            if one_pass:
                nat.group_linear_bwd_wrapper(B, M, N, npoint, nsample, grad_y, idx, rel, dP, dwx)
            else:
                nat.group_points_grad_wrapper(B, M, N, npoint, nsample, grad_y, idx, dP)
        grad_feat = _product(wf.contiguous(), dP, True) if ctx.needs_input_grad[2] else None
        grad_w = None
        if ctx.needs_input_grad[4]:
            if not one_pass:
                dwx = zeroed_empty((M, 3), torch.float32, grad_y.device)
                nat.conv1x1_wgrad_wrapper(B, 3, M, T, rel, grad_y, dwx)
            dwf = _weight_grad(dP, features.detach())
            grad_w = torch.cat([dwx, dwf], 1).view_as(weight)
        return None, None, grad_feat, None, grad_w, None, None, None, None, None


@_gate
def grouped_first_layer_available(xyz, new_xyz, features, idx, conv, gn):
    nat = _api._native
    if deterministic():   # (statistics by atomics in ogc_group_linear_fwd's epilogue)
        return False
    return (features is not None and features.is_cuda and features.dtype == torch.float32 and idx is not None
            and getattr(nat, "group_linear_fwd_wrapper", None) is not None and conv.bias is None
            and conv.weight.shape[1] == 3 + features.shape[1] and (idx.shape[1] * idx.shape[2]) % 16 == 0
            and not xyz.requires_grad and not new_xyz.requires_grad
            and (gn is None or (gn.num_groups <= 32 and conv.weight.shape[0] % gn.num_groups == 0)))


# An encoder's FIRST level with fp32 activations (features = the coordinates, no gradient wanted for them): QueryAndGroup's
# six-channel tensor + the matrix kernel with its statistics epilogue (ogc_conv1x1_gemm_gnstats) + a six-channel weight gradient,
# instead of the grouped first layer — whose backward scatter-adds 32 / 64 channels of dy into a per-point tensor nobody needs
# when the features take no gradient (group_bwd_lds_kernel: 0.28 ms per C4 step).  A/B on one box, round 6: 10.39-10.41 -> 10.29-10.31
# ms per C4 step; same rounding (one chain over the concatenated channels), the gate-flip list stays empty.  OGC_FIRST_LEVEL_PLAIN=0:
# the grouped first layer (its direct form) there as well.  16-bit activations (C2) keep the direct kernel: it stores bf16.
FIRST_LEVEL_PLAIN = _os.environ.get("OGC_FIRST_LEVEL_PLAIN", "1") != "0"


def first_level_plain(features):
    return bool(FIRST_LEVEL_PLAIN and features is not None and features.is_cuda and features.shape[1] <= 4
                and not features.requires_grad and not act16_wanted(features) and not deterministic())
GROUP_LINEAR_DIRECT = _os.environ.get("OGC_GROUP_LINEAR_DIRECT", "1") != "0"   # (0: P[idx] + W_xyz rel at every width, as until round 5)
GROUP_LINEAR_POINT_MAJOR = _os.environ.get("OGC_GROUP_LINEAR_PT", "1") != "0"   # (16-bit first layers: P stored (B, N, M))
# dwx inside the gather-form grouping gradient (ogc_group_points_grad_rev_dwx: one read of grad_y less, but 192 bytes of rel per
# thread and chunk through the L2s).  Measured at the end of round 5: C2 16.6-16.7 ms per step with it against 16.2 without (the
# three-channel weight gradient it replaces is 0.42 ms of bf16 MFMAs), C4 10.73-10.75 either way — off; OGC_GROUP_REV_DWX=1 turns it on.
GROUP_REV_DWX = _os.environ.get("OGC_GROUP_REV_DWX", "0") == "1"
GROUP_GRAD_GATHER = True   # grouping gradient as a gather over transposed lists when the geometry plan has them
GROUP_GRAD_GATHER_MIN_FANIN = 24


def grouped_first_layer(xyz, new_xyz, features, idx, conv, gn, rev=None, act16=False):
    """(conv(QueryAndGroup(...)), statistics for `gn`) — see _GroupedFirstLayer.  rev: group_reverse(idx, N) or None.
    act16: store the output as bf16 (see ACT16; only with statistics, i.e. a GroupNorm behind the layer)."""
    rev = rev if rev is not None else (None, None, None)
    T = idx.shape[1] * idx.shape[2]
    act16 = bool(act16 and gn is not None and gn.affine and T % 64 == 0 and act16_wanted(features))
    return _GroupedFirstLayer.apply(xyz.contiguous(), new_xyz.contiguous(), features.contiguous(), idx.int().contiguous(),
                                    conv.weight, 0 if gn is None else gn.num_groups, rev[0], rev[1], rev[2], act16)
