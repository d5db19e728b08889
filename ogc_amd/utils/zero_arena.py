"""One zero fill per training step (csrc/api.hip: ogc_zero_arena_begin / _end).

~40 operators of a step accumulate with atomics into small buffers this package allocates for them — GroupNorm statistics,
weight gradients, moment matrices, counters — and each zeroes its buffer with a launch of its own.  Inside ``with
zero_arena(device):`` those buffers come from ``zeroed_empty`` instead of ``torch.empty``: slices of ONE tensor that was
allocated and zeroed by a single launch when the block was entered; the library then skips the operators' own fills (it
recognises the region).  Outside such a block, on another stream, or when the step needs more than the region zeroed so far
(the first step: the extent is learnt from the step before), ``zeroed_empty`` is ``torch.empty`` and the operator zeroes it as
before.

The region is a FRESH allocation per step (normally the caching allocator hands the same block back: no cost).  A slice is an
ordinary view of it, so a tensor that escapes the step — a weight gradient autograd keeps as ``param.grad``, a statistic saved
by somebody — keeps its step's region alive and is never zeroed or handed out again: gradients may be held or accumulated
across steps exactly as with ``torch.empty`` (tests/test_zero_arena_gpu.py).  The library's side of the contract is per step:
every byte of the region goes to at most one operator between ``begin`` and ``end`` (the bump pointer below).
"""
import torch

from .. import _lib

CAPACITY = 64 << 20      # largest region of a step, bytes
LARGEST = 4 << 20        # larger requests are not worth a place in it (their fill is bandwidth, not launch latency)
import os as _os
ENABLED = _os.environ.get("OGC_ZERO_ARENA", "1") != "0"   # (0: every operator fills its own buffers — A/B runs)
CAPTURE = _os.environ.get("OGC_ZERO_ARENA_CAPTURE", "1") != "0"   # (0: off inside a HIP-graph capture, as until round 5)
PINNED_LIMIT = 1 << 30   # bytes of regions kept alive by escaped slices, in a row, before the arena switches itself off

_ITEMSIZE = {torch.float32: 4, torch.float64: 8, torch.int32: 4, torch.int64: 8, torch.int16: 2, torch.uint8: 1}

_arenas = {}
_active = None


class _Arena:
    def __init__(self, device):
        self.device = device
        self.buf = None      # this step's region (a new tensor per step: escaped slices keep the old one alive)
        self.filled = 0      # bytes zeroed by this step's fill
        self.used = 0        # bump pointer
        self.want = 0        # extent the last step asked for (what the next fill covers)
        self.stream = None
        self.pinned_regions = 0   # retired regions, in a row, that escaped slices kept alive
        self.pinned_bytes = 0
        self.disabled = False

    def _retire(self):
        """The last step's region is dropped here.  Slices of it that somebody still holds (a gradient kept across steps, a saved
        statistic) pin the WHOLE region, not their own bytes: count what may be pinned that way, and when a caller keeps
        collecting such tensors step after step (they pile up, up to CAPACITY each) say so once and stop handing out slices —
        the operators then zero ordinary tensors, as they do without the arena."""
        if self.buf is None:
            return
        try:  # references to the storage: `buf`, the wrapper made on this line, + one per live slice
            held = torch._C._storage_Use_Count(self.buf.untyped_storage()._cdata) - 2
        except Exception:  # (a private call of torch: without it nothing is counted)
            held = 0
        if held > 0:
            self.pinned_regions += 1
            self.pinned_bytes += self.filled
            if self.pinned_bytes > PINNED_LIMIT and not self.disabled:
                self.disabled = True
                import warnings
                warnings.warn("zero arena: %d step regions in a row (%.0f MiB) are still referenced by tensors that escaped their "
                              "step; the arena is switched off for this process (operators zero their own buffers again)"
                              % (self.pinned_regions, self.pinned_bytes / 2 ** 20))
        else:
            self.pinned_regions = self.pinned_bytes = 0   # only an unbroken run of pinned regions is a pile
        self.buf = None

    def begin(self):
        self.stream = torch.cuda.current_stream(self.device).cuda_stream
        self._retire()       # (first: the allocator may hand the same block back when nothing of the last step escaped)
        self.filled = 0 if self.disabled else min((self.want + 255) // 256 * 256, CAPACITY)
        self.used = self.want = 0
        if self.filled:
            # inside a HIP-graph capture this allocation comes from the graph's pool (a static address) and the fill below is a
            # kernel node: every replay starts from a zeroed region, and what escapes (param.grad) lives in graph memory like
            # every other tensor of the capture
            self.buf = torch.empty(self.filled, dtype=torch.uint8, device=self.device)
        _lib.call("ogc_zero_arena_begin", self.buf.data_ptr() if self.filled else 0, self.filled, self.stream)

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * _ITEMSIZE[dtype]
        if nbytes > LARGEST or nbytes == 0 or torch.cuda.current_stream(self.device).cuda_stream != self.stream:
            return None                     # never part of the region
        start = (self.want + 255) // 256 * 256
        self.want = start + nbytes          # (learnt even when this request does not fit yet: the next step's fill is larger)
        if start + nbytes > self.filled:
            return None
        return self.buf[start:start + nbytes].view(dtype).view(shape)


class zero_arena:
    """``with zero_arena(device):`` around the forward, loss and backward pass of ONE training step on the current stream."""

    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        global _active
        self.on = ENABLED and self.device.type == "cuda" and _active is None and (
            CAPTURE or not torch.cuda.is_current_stream_capturing())
        if self.on:
            a = _arenas.get(self.device)
            if a is None:
                a = _arenas[self.device] = _Arena(self.device)
            a.begin()
            _active = a
        return self

    def __exit__(self, *exc):
        global _active
        if self.on:
            _active = None
            _lib.call("ogc_zero_arena_end")


def zeroed_empty(shape, dtype, device):
    """A buffer an operator of this library will zero before accumulating into it: a pre-zeroed slice when a step's region is
    active (the operator then skips its fill), plain ``torch.empty`` otherwise."""
    a = _active
    if a is not None and a.device == torch.device(device):
        t = a.take(tuple(shape) if not isinstance(shape, int) else (shape,), dtype)
        if t is not None:
            return t
    return torch.empty(shape, dtype=dtype, device=device)
