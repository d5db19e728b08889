"""The whole segmentation training step — forward, OGC loss, backward, gradient all-reduce, fused Adam — as ONE HIP graph.

Why: at C4 (16 clouds of 8192 points per GPU) the eager step launches ~600 kernels from Python; the launch thread needs
10-12 ms for that whatever the cloud size (tools/graph_step.py: 2048-point clouds take 10.3 ms eagerly, 5.7 ms replayed),
which is also what the GPU needs for the kernels of an 8192-point step.  A graph replay costs the host one call, so the
step runs at the speed of its kernels, and kernels that are faster than the launch thread (the small ones of the loss and
of the slot branch) stop being hidden behind it.

What is captured is `train_step` itself (ogc_amd/train_step.py) — the same Python, the same kernels, the same side
streams — on STATIC buffers:

    cur   the batch being trained on (the trainer reads one batch ahead and hands the next one to step())
    plan  the coordinate-only work (FPS / kNN / 3-NN of every encoder level) of `cur`, made during the previous replay

and one step is:   replay (train on cur with plan)  ||  (side streams, eager) plan' = geometry of the next batch,
then plan <- plan', cur <- next batch (two fused copies behind the replay).  Everything the step decides stays on the device (the
NaN-gradient rule is the fused optimizer's found_inf flag, the Hungarian matching and the eigenvalues are kernels), so
there is nothing for the host to do between two replays.

Python-side constants are frozen at capture: the loss weights of iteration `it` (piecewise constant in the reference's
schedule, losses/seg_loss_unsup.py `start_step`) — recapture with `recapture(it)` when they change — and the learning
rate unless the optimizer was built with a tensor `lr`.  Reference step: train_seg.py:40-100.
"""
import torch

import os

from .train_step import PrefetchedGeometry, prepare_adam_kernel, train_step
from .utils import streams as _streams


# Side streams of the step that are NOT forked inside the capture (utils/streams.stream_namespace): "all", or a comma-separated
# list of keys ("monitor", "segnet-slots"); OGC_GRAPH_INLINE sets it.  Default: none — measured at C4 (round 6,
# profiles/r06_graph_step.txt): every branch forked 10.74-10.81 ms per replayed step, the monitor terms in line 11.03-11.08,
# monitor + slot branch 11.56, everything on one stream 11.53: the serialised small kernels cost more than the joins they save.
INLINE_BRANCHES = tuple(k for k in os.environ.get("OGC_GRAPH_INLINE", "").split(",") if k)
if INLINE_BRANCHES == ("all",):
    INLINE_BRANCHES = "all"


def _plan_tensors(obj):
    """Every tensor of a geometry plan (nested dicts / lists / Pending / PrefetchedGeometry), in a fixed order."""
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, _streams.Pending):
        yield from _plan_tensors(obj._value)
    elif isinstance(obj, PrefetchedGeometry):
        yield from _plan_tensors([obj.flat, obj.pcs_s, obj.flows_s, obj.model, obj.loss])
    elif isinstance(obj, dict):
        for k in sorted(obj):
            yield from _plan_tensors(obj[k])
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _plan_tensors(v)


def _copy_all(dst, src):
    """dst[i] <- src[i] for lists of tensors of MIXED dtypes: one multi-tensor launch per dtype.  torch._foreach_copy_ on the
    mixed list (int32 index tables next to fp32 coordinates) leaves its fast route and issues one device-to-device copy per
    tensor — 40 blit launches of ~5 us behind every replay of the step, which the next replay waits for (0.2 ms per step in
    the kernel trace, profiles/r06_graph_step.txt)."""
    groups = {}
    for d, v in zip(dst, src):
        if d.dtype != v.dtype or d.shape != v.shape:
            raise RuntimeError("graph step: a plan tensor changed its type or shape (%s %s <- %s %s)"
                               % (d.dtype, tuple(d.shape), v.dtype, tuple(v.shape)))
        g = groups.setdefault(d.dtype, ([], []))
        g[0].append(d)
        g[1].append(v)
    for ds, vs in groups.values():
        torch._foreach_copy_(ds, vs)


def _resolve(obj, wait):
    """Pending -> waited for (wait=True: the current stream joins the producing stream) or marked as done."""
    if isinstance(obj, _streams.Pending):
        if wait:
            obj.get()
        else:
            obj._event = None
    elif isinstance(obj, PrefetchedGeometry):
        _resolve([obj.model, obj.loss], wait)
    elif isinstance(obj, dict):
        for v in obj.values():
            _resolve(v, wait)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _resolve(v, wait)


def _join_side_streams(main):
    """`main` (capturing) waits for every side stream that was forked into the capture."""
    for st in list(_streams._side.values()):
        with torch.cuda.stream(st):
            forked = torch.cuda.is_current_stream_capturing()
        if forked:
            main.wait_stream(st)


def _state_tensors(segnet, optimizer):
    net = segnet.module if hasattr(segnet, "module") else segnet
    ts = [p.data for p in net.parameters()] + [b.data for b in net.buffers()]
    for st in optimizer.state.values():
        ts += [v for v in st.values() if torch.is_tensor(v)]
    return ts


class GraphedTrainStep:
    """step(next_batch) trains on the batch handed over by the PREVIOUS call (the first one: `first_batch`) and returns
    that step's PendingStep; `next_batch` is planned meanwhile and trained on by the next call.

    optimizer: torch.optim.Adam(..., fused=True, capturable=True) (train_step.make_optimizer(capturable=True))."""

    def __init__(self, segnet, criterion, optimizer, first_batch, it, aug_transform):
        assert first_batch[0].is_cuda, "GraphedTrainStep needs the batch on the GPU"
        assert optimizer.defaults.get("capturable"), "build the optimizer with capturable=True (make_optimizer)"
        self.segnet, self.criterion, self.optimizer, self.aug = segnet, criterion, optimizer, aug_transform
        from .utils import subgraph
        subgraph.forbid(segnet)  # the whole step becomes one graph: no graphs of their own for its parts (the eager warm-up
        #                          would otherwise make them, and move those parameters' gradient accumulation to another stream)
        self.cur = tuple(t.clone() if torch.is_tensor(t) else t for t in first_batch)
        self.stream = torch.cuda.Stream()
        self.graph, self.pending, self.plan = None, None, None
        self.recapture(it)

    def _warm_up(self, it):
        """One eager step on the capture stream, undone afterwards: library handles, the workspaces of this stream and of
        the side streams and the optimizer's state tensors must exist before a capture begins (a capture records
        allocations and zero-fills as nodes: the optimizer state would be reset by every replay)."""
        fresh = len(self.optimizer.state) == 0
        before = _state_tensors(self.segnet, self.optimizer)
        saved = [t.clone() for t in before]
        train_step(self.segnet, self.criterion, self.optimizer, self.cur, it, self.aug, sync=False)
        torch._foreach_copy_(before, saved)
        if fresh:  # the state the optimizer would have created on its first step: zeros
            for st in self.optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        # the step's own optimizer kernel (NaN rule + Adam, csrc/adam.hip) uploads its pointer tables from the host: before the
        # capture, now that the state tensors exist (inside it the step would fall back to torch's eleven launches)
        prepare_adam_kernel(self.optimizer)

    def recapture(self, it):
        """(Re)build the graph for the loss weights of iteration `it` (and the optimizer's current hyper-parameters)."""
        self.it = it
        s = self.stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            if self.plan is None:
                self.plan = PrefetchedGeometry(self.segnet, self.criterion, self.cur, self.aug)
            # warm-up and capture on side streams of their OWN (streams.stream_namespace): the eager geometry plans of step()
            # run on the ordinary side streams underneath a replay, and must not meet captured kernels on a stream (and so on
            # a scratch buffer) of theirs
            with _streams.stream_namespace("graph:", inline=INLINE_BRANCHES):
                self._warm_up(it)
                torch.cuda.synchronize()
                _resolve(self.plan, wait=False)
                self.optimizer.zero_grad(set_to_none=True)
                graph = torch.cuda.CUDAGraph()
                import gc
                collecting = gc.isenabled()  # (no collector pass inside the capture: see utils/subgraph._make)
                gc.disable()
                try:
                    with torch.cuda.graph(graph, stream=s):
                        pending = train_step(self.segnet, self.criterion, self.optimizer, self.cur, it, self.aug, sync=False,
                                             prefetched=self.plan)
                        _join_side_streams(torch.cuda.current_stream())
                finally:
                    if collecting:
                        gc.enable()
        torch.cuda.current_stream().wait_stream(s)
        self.graph, self.pending = graph, pending
        self._plan_dst = list(_plan_tensors(self.plan))

    def load(self, batch):
        """Make `batch` the one the next step() trains on (start of an epoch, or after steps that did not go through this
        object): its geometry plan is computed here, eagerly."""
        s = self.stream
        s.wait_stream(torch.cuda.current_stream())
        self.segnet.train()
        with torch.cuda.stream(s):
            fresh = PrefetchedGeometry(self.segnet, self.criterion, batch, self.aug)
            _resolve(fresh, wait=True)
            _copy_all(self._plan_dst + [d for d in self.cur if torch.is_tensor(d)],
                      list(_plan_tensors(fresh)) + [v for d, v in zip(self.cur, batch) if torch.is_tensor(d)])

    def step(self, next_batch):
        s = self.stream
        self.segnet.train()
        s.wait_stream(torch.cuda.current_stream())  # whoever produced next_batch
        with torch.cuda.stream(s):
            # The geometry of the next batch: ~40 eager launches on the side streams, which wait for the point reached on
            # `s` HERE (the end of the previous step) and so run underneath the replay queued right after.  It is NOT a
            # branch of the graph: the graph executor ran that branch after the step instead of underneath it (13.7 ms
            # per step against 12.6 for the eager step with the same overlap).
            upcoming = PrefetchedGeometry(self.segnet, self.criterion, next_batch, self.aug)
            self.graph.replay()
            _resolve(upcoming, wait=True)  # behind the replay: join the side streams, then shift
            src = list(_plan_tensors(upcoming))
            assert len(src) == len(self._plan_dst)
            _copy_all(self._plan_dst + [d for d in self.cur if torch.is_tensor(d)],
                      src + [v for d, v in zip(self.cur, next_batch) if torch.is_tensor(d)])
            return self._pending_of_this_replay()

    def _pending_of_this_replay(self):
        """The scalars of the replay just queued, on their way to the host (pinned copy + event, like an eager step's), so
        that the caller can read step i while step i+1 is already running: the graph's own scalar tensors are rewritten by
        the next replay."""
        import copy
        from .train_step import PendingStep
        losses = self.pending._losses
        if hasattr(losses, "_scalars") and getattr(losses._scalars, "_dev", None) is not None:
            losses = copy.copy(losses)
            losses._scalars = _streams.HostScalars(self.pending._losses._scalars._dev)
            losses._dict = None
        bad = self.pending._bad
        if getattr(bad, "_dev", None) is not None:
            bad = _streams.HostScalars(bad._dev)
        return PendingStep(losses, bad)
