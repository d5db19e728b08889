"""The element-wise glue of FlowStep3D's refinement loop in inference, one launch per group of framework operators
(``csrc/flow_step.hip``).  At C3 (one pair of 8192-point clouds, iters = 5) the forward is ~450 launches of a few microseconds each and
its time is their number; each function here stands for three to six ``aten`` launches of models/flownet_kitti.py:135-151, :229-250
with the same fp32 arithmetic.  Nothing here is differentiable: `available()` is False whenever autograd would record the
operators (training keeps the reference's operator sequence), on CPU tensors, and when the native module is not the HIP one."""
import numpy as np
import torch

from .pointnet2 import pointnet2 as _api

ENABLED = __import__("os").environ.get("OGC_FLOW_GLUE", "1") != "0"   # tests / A-B measurements switch the module off as a whole


# Groups switched off (OGC_FLOW_GLUE_OFF=a,b), for bisecting: linear_cn, soft_corr, gru, advance, gather_pair, three_nn_w.
OFF = set(filter(None, __import__("os").environ.get("OGC_FLOW_GLUE_OFF", "").split(","))) - {"none"}


def available(*tensors, what=None):
    """Inference on the HIP operators: no tensor of the call is being differentiated.  what: the group's name (OGC_FLOW_GLUE_OFF)."""
    nat = _api._native
    if not ENABLED or what in OFF or getattr(nat, "gru_blend_wrapper", None) is None:
        return False
    for t in tensors:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32) or (torch.is_grad_enabled() and t.requires_grad):
            return False
    return True


def gather_xyz_pair(xyz, idx):
    """xyz (B, 3, N), idx (B, M) int32 -> (xyz[:, :, idx] (B, 3, M), the same as (B, M, 3)): gather_operation and the transposed
    copy every set-abstraction layer makes of its centres (utils/flowstep3d_util.py:110-118)."""
    B, _, N = xyz.shape
    M = idx.shape[1]
    out = torch.empty(B, 3, M, dtype=torch.float32, device=xyz.device)
    out_t = torch.empty(B, M, 3, dtype=torch.float32, device=xyz.device)
    _api._native.gather_xyz_pair_wrapper(B, N, M, xyz, idx, out, out_t)
    return out, out_t


def reciprocal_scale(divisor):
    """What torch multiplies by when a float tensor is divided by a Python scalar on the GPU: 1 / divisor evaluated in double,
    then rounded to fp32 (BinaryDivTrueKernel.cu: `a * inv_b` for a CPU-scalar divisor)."""
    return float(np.float32(1.0 / float(divisor)))


def flow_advance(cur, delta, ref, divisor=1.0, want_delta=False, want_t=False, want_flow=True):
    """cur, delta, ref (B, 3, N): d = delta / divisor; new = cur + d; flow = new - ref.  Returns (d or None, new, new as (B, N, 3) or
    None, flow or None) — flownet_kitti.py:229-231 and :245-250 as one launch."""
    B, _, N = cur.shape
    scale = 1.0 if divisor == 1 else reciprocal_scale(divisor)
    new = torch.empty_like(cur)
    d = torch.empty_like(cur) if want_delta and scale != 1.0 else None
    new_t = torch.empty(B, N, 3, dtype=torch.float32, device=cur.device) if want_t else None
    flow = torch.empty_like(cur) if want_flow else None
    _api._native.flow_advance_wrapper(B, N, scale, cur, delta, ref if want_flow else None, d, new, new_t, flow)
    if want_delta and d is None:
        d = delta
    return d, new, new_t, flow


def linear_cn(x, weight, bias):
    """nn.Linear applied along the channel axis of x (B, Cin, N) -> (B, Cout, N), Cout <= 4: the `fc` between two transposes of
    FlowRegressor / Flow0Regressor (flownet_kitti.py:19, :38)."""
    B, cin, N = x.shape
    cout = weight.shape[0]
    y = torch.empty(B, cout, N, dtype=torch.float32, device=x.device)
    _api._native.linear_cn_wrapper(B, cin, cout, N, x, weight, bias, y)
    return y


def gru_reset(rc, hx, c, rc_channel0=0):
    """rc (B, >= C, N, S): the reset gate's convolution output before its max over the neighbours, in channels [rc_channel0,
    rc_channel0 + C); hx (B, C + Cx, N) = cat([h, x]).  -> cat([sigmoid(max_s rc) * h, x]) (flownet_kitti.py:148-149)."""
    B, ctot, N = hx.shape
    out = torch.empty_like(hx)
    _api._native.gru_reset_wrapper(B, c, ctot - c, N, rc.shape[3], rc, hx, out, rc_channel0)
    return out


def gru_blend(zc, qc, hx, c, zc_channel0=0):
    """zc (B, >= C, N, S) (the gate in channels [zc_channel0, zc_channel0 + C)), qc (B, C, N, S): un-pooled gate / candidate outputs,
    h = hx[:, :C]: (1 - z) * h + z * q with z = sigmoid(max_s zc), q = tanh(max_s qc) (flownet_kitti.py:147, :149-150)."""
    B, ctot, N = hx.shape
    out = torch.empty(B, c, N, dtype=torch.float32, device=hx.device)
    _api._native.gru_blend_wrapper(B, c, N, zc.shape[3], zc, qc, hx, ctot * N, out, zc_channel0)
    return out


import weakref as _weakref

_STACKED = {}   # id(first weight) -> (key, stacked weight, weak references to ALL the weights: identity, not id, decides a hit)


def stacked_weight(*weights):
    """The 1x1-convolution weights of blocks that read the same input, stacked along the output channels (one product instead of
    one per block); rebuilt when any of them was written (version counters) or moved, or when any block has been in training mode
    since (fused.note_training_mode: the fused optimizer writes parameters through raw pointers, which the version counters do not
    see — a net evaluated, trained further and evaluated again would otherwise use the stacked weights of its first evaluation;
    found by the flow trainer replay of tests/test_driver_golden.py at the end of round 5).  A hit also needs the cached weak
    references to BE the weights (tensors cannot key a WeakKeyDictionary: their == is element-wise): an `id()` — or a device
    address — is reused as soon as a net is freed, and a later net of the same shape must not inherit a dead one's entry."""
    from . import fused as _fused
    key = (_fused._FOLD_GENERATION[0],) + tuple((w.data_ptr(), w._version, tuple(w.shape)) for w in weights)
    hit = _STACKED.get(id(weights[0]))
    if hit is None or hit[0] != key or len(hit[2]) != len(weights) or any(r() is not w for r, w in zip(hit[2], weights)):
        with torch.no_grad():
            hit = (key, torch.cat([w.reshape(w.shape[0], -1) for w in weights]).contiguous(),
                   tuple(_weakref.ref(w) for w in weights))
        if id(weights[0]) not in _STACKED:
            # the entry (and the device memory of its stacked copy) goes when the first weight does: a freed network leaves nothing
            _weakref.finalize(weights[0], _STACKED.pop, id(weights[0]), None)
        _STACKED[id(weights[0])] = hit
    return hit[1]


def soft_corr_flow(pc1, pc2, f1, f2, epsilon, support):
    """pc* (B, 3, n), f* (B, C, n) channel-major, epsilon the layer's (1,) parameter: the coarse flow (B, 3, n1) of
    GlobalCorrLayer.forward's first three lines (flownet_kitti.py:53-70) without the (B, n1, n2) matrices."""
    B, C, n1 = f1.shape
    flow = torch.empty(B, 3, n1, dtype=torch.float32, device=pc1.device)
    _api._native.soft_corr_flow_wrapper(B, n1, f2.shape[2], C, support, epsilon, pc1, pc2, f1, f2, flow)
    return flow


def soft_corr_flow_supported(f1):
    return f1.shape[1] % 4 == 0 and f1.shape[1] <= 256


def three_nn_with_weights(unknown_t, known_t, mode=0):
    """unknown_t (B, n, 3), known_t (B, m, 3) -> (idx (B, n, 3) int32, weight (B, n, 3)): three_nn and the normalised inverse
    distances of utils/flowstep3d_util.py:168-170 (mode 0) in two launches instead of six."""
    B, n, _ = unknown_t.shape
    dist2 = torch.empty(B, n, 3, dtype=torch.float32, device=unknown_t.device)
    idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknown_t.device)
    nat = _api._native
    nat.three_nn_wrapper(B, n, known_t.shape[1], unknown_t, known_t, dist2, idx)
    weight = torch.empty_like(dist2)
    nat.three_nn_weights_wrapper(B, n, mode, dist2, weight)
    return idx, weight
