"""Side-stream helpers: overlap the coordinate-only part of a step (FPS, kNN, three-NN, ball query — none of it
depends on learned features) with the dense feature work.  FPS keeps one workgroup per cloud busy (16 of 256 CUs at
B = 16), so running it next to the GEMM / GroupNorm kernels costs nothing.  Results are identical: only the launch
stream changes; ordering is enforced with events and the caching allocator is told about cross-stream use."""
import torch

_side = {}
_namespace = [""]
_inline = [()]


class stream_namespace:
    """Inside this scope side_stream() hands out streams of their own (`<prefix><key>`).  A graph capture runs in one: the
    native operators keep their scratch buffers per stream (ogc_workspace), and kernels replayed from a graph are not ordered
    against eager kernels queued on the stream they were captured from — so captured work and the eager work that runs
    underneath a replay (the next batch's geometry plan) must never share a stream."""

    def __init__(self, prefix, inline=()):
        """inline: keys of side streams that are NOT forked inside this scope — side_stream() answers with the current stream
        for them ("all": for every key).  A graph capture uses it for branches whose kernels cost less than the joins they need:
        the graph executor leaves ~20 us at a dependency between two branches where back-to-back kernels of one branch leave ~2."""
        self.prefix, self.inline = prefix, inline

    def __enter__(self):
        self._prev = (_namespace[0], _inline[0])
        _namespace[0], _inline[0] = self.prefix, self.inline
        return self

    def __exit__(self, *exc):
        _namespace[0], _inline[0] = self._prev


def side_stream(device, key="geometry", priority=0):
    """The process-wide side stream `key` of `device`; `priority` (-1 = high) applies when it is first created."""
    if _inline[0] == "all" or key in _inline[0]:
        return torch.cuda.current_stream(device)
    k = (torch.device(device).index, _namespace[0] + key)
    if k not in _side:
        _side[k] = torch.cuda.Stream(device=device, priority=priority)
    return _side[k]


class Pending:
    """A value being produced on a side stream.  ``get()`` makes the current stream wait for it (once)."""

    def __init__(self, value, event):
        self._value, self._event = value, event

    def get(self):
        if self._event is not None:
            main = torch.cuda.current_stream()
            main.wait_event(self._event)
            for t in _tensors(self._value):
                t.record_stream(main)
            self._event = None
        return self._value


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


def launch_on_side(stream, fn, after=None):
    """Run ``fn()`` on ``stream`` after everything already queued on the current stream — or, with ``after`` (an
    event), after that event only, so the work may overlap whatever the current stream still has queued; returns a
    Pending."""
    if after is None:
        stream.wait_stream(torch.cuda.current_stream())
    else:
        stream.wait_event(after)
    with torch.cuda.stream(stream):
        value = fn()
        event = torch.cuda.Event()
        event.record(stream)
    return Pending(value, event)


class HostScalars:
    """Device scalars on their way to the host without stopping the launch thread: a non-blocking copy into pinned
    memory plus an event.  ``get()`` waits for that copy only (by then the next kernels are already queued)."""

    def __init__(self, values):
        values = values.detach()
        self._dev = None
        if values.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture: no pinned allocation, no event — the values stay in a tensor of the graph's
            # memory pool, rewritten by every replay, and get() reads it (a synchronising copy) when asked
            self._dev, self._host, self._event = values.clone(), None, None
        elif values.is_cuda:
            self._host = torch.empty(values.shape, dtype=values.dtype, pin_memory=True)
            self._host.copy_(values, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
        else:
            self._host, self._event = values, None

    def get(self):
        if self._dev is not None:
            return self._dev.tolist()
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        return self._host.tolist()
