"""Portable deterministic data for the golden fixtures: integer hashing only (no libm, no torch RNG), so the
generator script (run once, in the build container, with the reference importable) and the tests (run
anywhere) produce bit-identical inputs and weights without storing them."""
import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)


def _mix(x):
    x = x & _M32
    x = ((x ^ (x >> np.uint64(16))) * np.uint64(0x7FEB352D)) & _M32
    x = ((x ^ (x >> np.uint64(15))) * np.uint64(0x846CA68B)) & _M32
    return (x ^ (x >> np.uint64(16))) & _M32


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array in [lo, hi): hash(index, seed) / 2^32, exact integer arithmetic."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.uint64)
    h = _mix(i * np.uint64(0x9E3779B1) + np.uint64((seed * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF))
    h = _mix(h + np.uint64(seed & 0xFFFFFFFF))
    u = h.astype(np.float64) / 4294967296.0
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def _name_seed(name):
    s = 0
    for ch in name:
        s = (s * 131 + ord(ch)) & 0x7FFFFFFF
    return s


def fill_module(module, seed=0):
    """Overwrite every parameter/buffer of ``module`` with deterministic values:
    weights ~ U(-a, a) with a = sqrt(3 / fan_in) (variance-preserving), norm scales in [0.5, 1.5], biases /
    means in [-0.1, 0.1], running variances in [0.5, 1.5]."""
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.dtype.is_floating_point:
                continue  # num_batches_tracked
            sd = _name_seed(name) + seed * 7919
            leaf = name.split(".")[-1]
            is_norm = any(tag in name for tag in (".gn.", ".bn.", "mlp_bns.", "norm_"))
            if leaf == "running_var" or (leaf == "weight" and (is_norm or t.dim() == 1)):
                v = uniform(tuple(t.shape), sd, 0.5, 1.5)
            elif leaf in ("bias", "running_mean", "in_proj_bias", "epsilon"):
                v = uniform(tuple(t.shape), sd, -0.1, 0.1)
            else:
                fan_in = int(np.prod(t.shape[1:])) if t.dim() > 1 else int(t.shape[0])
                a = float(np.sqrt(3.0 / max(fan_in, 1)))
                v = uniform(tuple(t.shape), sd, -a, a)
            t.copy_(torch.from_numpy(v))
    return module


def cloud(B, N, seed, scale=(60.0, 4.0, 80.0)):
    return uniform((B, N, 3), seed, -0.5, 0.5) * np.asarray(scale, np.float32)


def rigid_scene(B, N, K, seed, scale=(1.0, 1.0, 1.0), max_shift=0.05, noise=0.002):
    """A cloud, a per-object rigid flow and a soft mask (B,N,K) that roughly follows the objects."""
    pc = cloud(B, N, seed, scale)
    centres = pc[:, :K]                                                     # (B, K, 3)
    assign = ((pc[:, :, None, :] - centres[:, None]) ** 2).sum(-1).argmin(-1)   # (B, N)
    ang = uniform((B, K), seed + 1, -0.08, 0.08)
    shift = uniform((B, K, 3), seed + 2, -max_shift, max_shift)
    c, s = np.cos(ang), np.sin(ang)
    R = np.zeros((B, K, 3, 3), np.float32)
    R[..., 0, 0], R[..., 0, 2], R[..., 1, 1], R[..., 2, 0], R[..., 2, 2] = c, s, 1.0, -s, c
    Rn = np.take_along_axis(R, assign[:, :, None, None].repeat(3, 2).repeat(3, 3), 1)   # (B, N, 3, 3)
    tn = np.take_along_axis(shift, assign[:, :, None].repeat(3, 2), 1)
    cn = np.take_along_axis(centres, assign[:, :, None].repeat(3, 2), 1)
    moved = np.einsum("bnij,bnj->bni", Rn, pc - cn) + cn + tn
    flow = (moved - pc + uniform((B, N, 3), seed + 3, -noise, noise)).astype(np.float32)
    logits = 4.0 * np.eye(K, dtype=np.float32)[assign] + uniform((B, N, K), seed + 4, -1.0, 1.0)
    e = np.exp(logits - logits.max(-1, keepdims=True))
    mask = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return pc.astype(np.float32), flow, mask
